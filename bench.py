#!/usr/bin/env python3
"""Headline benchmark: condensed-QP solves/sec of the CDU offline-datagen hot path.

    python bench.py --gpus N --steps K --warmup W [--workload cdu|cstrs|nn|chains] [--batch B]

One "step" = one pass of the hot path (x_unc = Kunc x0 -> shared-inverse active-set rounds -> certified u*, active
sets, status -> first moves) over one batch of B synthetic CDU-size problems per GPU (Nx=252, Nu=32, N=140 -> n=4480
variables, m=8960 box rows; reference sizes cdu_parameters.py:99-102), inputs already resident in HBM.

N > 1: one process per GPU.  Started under `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment)
the process is one rank; started plainly with --gpus N > 1 it launches its own N ranks as child processes before
anything touches the GPU.  The sample batch is sharded (weak scaling: B per GPU; 125 000 per GPU at N = 8 =
BASELINE.json's "1M sampled x0 over 8 GPUs"), no collective during the solves and ONE RCCL gather of the first moves
over xGMI at the end of every step -- through the library (nnmpc_comm_*); this file binds nothing but libnnmpc_hip.so.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  At N = 1 the default run also measures, outside the
headline's timed region, the other single-GPU configurations of BASELINE.json (`configs`), the active-fraction sweep,
the lock-step chain workload, the PDIP path, the PCIe-inclusive rate, the parity checks and the CPU baseline.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time


def usable_cpus():
    """CPUs this process may actually use: affinity mask and cgroup quota (a GPU box hands a container 16 of 256)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


# BLAS pools sized well below the CPUs the process may use, BEFORE numpy loads its OpenBLAS (whose default is one thread per logical
# CPU of the HOST: 2 x 128 threads on a 256-thread box inside a 16-CPU container).  After any BLAS call the pool's threads spin for
# a while; more spinning threads than the cgroup's quota and the whole process is throttled for the rest of the scheduler period --
# the thread that polls the GPU included.  Measured on the CSTRs-size leg (1.4 ms per step): with the default pools 12 of 28 solver
# handles saw ~16 ms added to EVERY call (outside the library's own clock around the call); with 16 threads a CDU-size call now
# and then took 63 instead of 11 ms right after a host SVD; with 8 threads, 0 of 28 and none.  (threadpoolctl limits set after the
# pools exist do not help: 12 of 42.)  An explicit setting in the environment wins.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(max(1, min(8, usable_cpus() // 2))))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # MI355X dense f32 matrix/vector peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6   # f64 vector = matrix peak (half the f32 rate; v_mfma_f64_16x16x4_f64)
BF16_PEAK_TFLOPS = 2500.0
NSETS_MAX = 32           # distinct input batches kept in HBM for the headline's steps


# ------------------------------------------------------------------------------------------------------------------
# self-launch: N ranks as fresh child processes (nothing has touched the GPU in this process yet)
def launch_ranks(argv, n, script=None, timeout_s=None, poll_s=0.2):
    """N ranks of `script` (default: this file) as fresh child processes, SUPERVISED: the launcher polls them and, on the
    first non-zero exit or at a wall-clock limit, terminates (then kills) the others and returns non-zero -- RCCL has no
    timeout, so a rank that died before or during ncclCommInitRank / a collective would leave the others waiting for ever
    with every GPU held.  A rank is never re-started: a retry is a fresh launch.  Every rank gets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT and NNMPC_JOB_KEY (a uuid4: key of the rendezvous file of the RCCL unique id,
    distributed.job_key).  Rank 0's stdout is passed through; the failed rank's stderr is shown and named."""
    import tempfile
    import uuid
    script = script or os.path.abspath(__file__)
    timeout_s = timeout_s or float(os.environ.get("NNMPC_LAUNCH_TIMEOUT_S", "3300"))
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    key = uuid.uuid4().hex
    logs = tempfile.mkdtemp(prefix="nnmpc_launch_")
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NNMPC_JOB_KEY=key,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        so, se = open(os.path.join(logs, f"rank{r}.out"), "wb"), open(os.path.join(logs, f"rank{r}.err"), "wb")
        files.append((so, se))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=so, stderr=se))
    t0, failed = time.time(), None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = (bad[0], f"exit code {codes[bad[0]]}")
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > timeout_s:
            failed = ([r for r, c in enumerate(codes) if c is None][0], f"still running after {timeout_s:.0f} s")
            break
        time.sleep(poll_s)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        while any(p.poll() is None for p in procs) and time.time() - t1 < 5.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    for so, se in files:
        so.close(); se.close()

    def tail(name, k=4000):
        with open(os.path.join(logs, name), "rb") as f:
            return f.read()[-k:].decode(errors="replace")
    sys.stdout.write(tail("rank0.out", 1 << 24))
    sys.stdout.flush()
    if failed:
        r, why = failed
        sys.stderr.write(f"bench.py launcher: rank {r} of {n} failed ({why}); the other ranks were stopped.  stderr of rank {r}:\n")
        sys.stderr.write(tail(f"rank{r}.err") + "\n")
        if r != 0:
            sys.stderr.write("stderr of rank 0:\n" + tail("rank0.err") + "\n")
        return 1
    sys.stderr.write(tail("rank0.err"))
    return 0


def default_batch(workload, world):
    """Problems per GPU per step.  cdu: BASELINE.json configs[2] "100k sampled x0, 1 MI355X"; at 8 GPUs configs[3] "1M sampled x0
    sharded across 8 x MI355X" = 125 000 each; cstrs: configs[1] "10k sampled x0"."""
    if workload == "cdu":
        return 125000 if world == 8 else 100000
    return 10000


def csrc_sha():
    """Hash of the kernel sources: a stored PMC profile is only quoted for the build it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "industrial_nnmpc_2021_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_key(name):
    """'void nnmpc::gemm_nt_f32_k<128, true, true>(float*, ...)' -> 'gemm_nt_f32_k<128, true, true>': template instances stay apart
    (the 128- and 64-wide instances of one template are different kernels with different traffic)."""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):                       # the argument list starts at the first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and i > 0 and not n.startswith("(anonymous namespace)", i):
            cut = i
            break
    n = n[:cut].replace("(anonymous namespace)::", "")
    head = n[:n.index("<")] if "<" in n else n       # namespaces sit in front of the template arguments
    return n[head.rfind("::") + 2:] if "::" in head else n


def pmc_traffic(name):
    """HBM bytes of every kernel of one step from profiles/pmc_hbm_<name>.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes,
    scripts/pmc_hbm.py, gfx950 corrections applied) when that profile was taken on THIS build of the kernels; else {}.
    Returns ({kernel: {"per_launch": bytes, "per_step": bytes, "launches_per_step": n}}, note)."""
    f = os.path.join(ROOT, "profiles", f"pmc_hbm_{name}.json")
    if not os.path.exists(f):
        return {}, "no PMC profile for this workload"
    d = json.load(open(f))
    if d.get("csrc_sha") != csrc_sha():
        return {}, f"stale: {os.path.basename(f)} was taken on kernel sources {d.get('csrc_sha')}, this build is {csrc_sha()}"
    steps = max(1, int(d.get("steps_profiled", 1)))
    out = {}
    for k, v in d["kernels"].items():
        tot = v["fetch_bytes_corrected"] + v["write_bytes"]
        out[kernel_key(k)] = {"per_launch": v["hbm_bytes_per_launch"], "per_step": tot / steps, "launches_per_step": v["launches"] / steps}
    return out, f"profiles/{os.path.basename(f)} (FETCH_SIZE x2 + WRITE_SIZE, same kernel sources, {steps} step(s) profiled)"


# ------------------------------------------------------------------------------------------------------------------
def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    blas = "unknown"
    try:
        from threadpoolctl import threadpool_info
        blas = ", ".join(sorted({f"{i.get('internal_api')} {i.get('version')}" for i in threadpool_info()
                                 if i.get("user_api") == "blas"}))
    except Exception:
        pass
    return model, blas


# CPU worker processes (oracle rows of the parity leg, the single-thread processes of the CPU baseline) are forked in main()
# BEFORE anything touches the GPU: a child forked later would inherit the KFD descriptors and Python objects whose __del__ calls
# hipFree / nnmpc_qp_destroy -- undefined behaviour on ROCm.  Problem data reaches them through an .npz under /dev/shm.
WORKERS = None
_SHARED = {}


def start_workers(n):
    global WORKERS
    if WORKERS is None and n > 0:
        import multiprocessing as mp
        WORKERS = mp.get_context("fork").Pool(n)
    return WORKERS


def stop_workers():
    global WORKERS
    if WORKERS is not None:
        WORKERS.terminate(); WORKERS.join()
        WORKERS = None


def _publish(**arrays):
    import tempfile
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    fd, path = tempfile.mkstemp(prefix="nnmpc_bench_", suffix=".npz", dir=d)
    os.close(fd)
    np.savez(path, **arrays)
    return path


def _shared(path):
    if path not in _SHARED:
        with np.load(path) as f:
            _SHARED.clear()                                  # one job at a time per worker
            _SHARED[path] = {k: f[k] for k in f.files}
    return _SHARED[path]


def _map(fn, jobs):
    """jobs through the early-forked pool; without one (bench functions called from elsewhere) one after the other in this process."""
    if WORKERS is not None:
        return WORKERS.map(fn, jobs, chunksize=1)
    return [fn(j) for j in jobs]


def _cpu_worker(args):
    """One process of the reference's parallel model (lib/linearMPC.py:817-820): 1 BLAS thread, its own problems."""
    (path, lo, hi, budget_s) = args
    d = _shared(path)
    P, tq, nu, N = d["P"], d["tq"], int(d["nu"]), int(d["N"])
    x0, lb, ub = d["x0"][lo:hi], d["lb"][lo:hi], d["ub"][lo:hi]
    from threadpoolctl import threadpool_limits
    from oracle import qp as oqp
    done, t0, its = 0, time.time(), []
    with threadpool_limits(limits=1):
        for b in range(x0.shape[0]):
            G, h = oqp.box_as_Gh(nu, N, lb[b], ub[b])
            info = {}
            oqp.coneqp_l(P, tq @ x0[b], G, h, info=info)
            its.append(info["iterations"]); done += 1
            if time.time() - t0 > budget_s:
                break
    return done, time.time() - t0, its


def physical_cores():
    """Physical cores of this host (distinct (package, core id) pairs of /proc/cpuinfo); None when it cannot be told."""
    try:
        seen, pkg = set(), "0"
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                seen.add((pkg, line.split(":", 1)[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(P, tq, nu, N, x0, lb, ub, budget_s, workload, full):
    """Restated reference CPU path (oracle.qp.coneqp_l: cvxopt-style dense-G PDIP, fp64, one problem at a time)
    timed on this host's cores.  Bounded sample.  Modes:
      (a) `value`: one process, all cores through the BLAS threads, >= 4 CDU-size problems;
      (b) `one_thread_alone`: one process, ONE BLAS thread, alone on the host -- mode (i) of BASELINE.md;
      (c) `nproc_processes`: independent single-thread processes, one problem each -- mode (ii), the reference's own
          parallel model (lib/linearMPC.py:817-820).
    At the CDU size a single-thread solve takes ~0.5-1 min: (b) is only measured with `--cpu-baseline full` there; the per-process
    rate of (c)'s concurrent solves is reported under its own key (it is lower than a lone process's: shared memory bandwidth)."""
    from oracle import qp as oqp
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        threads, threadpool_limits = os.cpu_count() or 1, None
    n = P.shape[0]
    model, blas = cpu_info()
    min_done = 4 if workload == "cdu" else 32
    done, t0, its = 0, time.time(), []
    for b in range(x0.shape[0]):
        G, h = oqp.box_as_Gh(nu, N, lb[b], ub[b])
        info = {}
        oqp.coneqp_l(P, tq @ x0[b], G, h, info=info)
        its.append(info["iterations"]); done += 1
        if time.time() - t0 > budget_s and done >= min_done:
            break
        if time.time() - t0 > 4 * budget_s:
            break
    dt = time.time() - t0
    res = {"value": done / dt, "unit": "solves/s", "cores": int(threads), "kind": "port",
           "sample": f"{done} problem(s) of the same seeded batch, n={n}, m={2 * n}, dense G, cvxopt-default tolerances, "
                     f"mean {np.mean(its):.1f} PDIP iterations, {dt:.1f} s, one process with {threads} BLAS threads",
           "cpu_model": model, "blas": blas, "usable_cpus": usable_cpus(), "host_logical_cpus": os.cpu_count(), "host_physical_cores": physical_cores(),
           "blas_threads_used": int(threads),
           "paper_reference": ("CVXOPT 35 s/solve mean, 47 s worst on a 2.4 GHz cluster CPU (KumarRawlingsWright2021 p.9) = 0.029 solves/s"
                               if workload == "cdu" else
                               "CVXOPT 8-13 s/solve on a 2.4 GHz cluster CPU at the paper's N=450 (n=2700; the code ships N=90, n=540) (KumarRawlingsWright2021 p.7)")}
    if threadpool_limits is not None:
        nproc = min(os.cpu_count() or 1, 16 if workload == "cdu" else 64)
        if WORKERS is not None:
            nproc = min(nproc, WORKERS._processes)
        per = 1 if workload == "cdu" else 4
        path = _publish(P=P, tq=tq, nu=nu, N=N, x0=x0, lb=lb, ub=ub)
        jobs = [(path, i * per, (i + 1) * per, 1e9) for i in range(nproc) if (i + 1) * per <= x0.shape[0]]
        t2 = time.time()
        rr = _map(_cpu_worker, jobs)
        t2 = time.time() - t2
        res["nproc_processes"] = {"value": sum(r[0] for r in rr) / t2, "unit": "solves/s", "cores": len(jobs),
                                  "sample": f"{sum(r[0] for r in rr)} problems over {len(jobs)} independent processes "
                                            f"(1 BLAS thread each; the reference's parallel model, lib/linearMPC.py:817-820), {t2:.1f} s wall"}
        rate = float(np.mean([r[0] / r[1] for r in rr]))
        # NOT BASELINE.md's mode (i) (one process ALONE, one thread): these solves ran beside each other and share memory bandwidth
        # and L3 -- a lone process is faster, so a speed-up quoted against this rate would be inflated.  `one_thread_alone` is the
        # lone-process figure (always at the CSTRs size, with --cpu-baseline full at the CDU size: ~1 min per problem)
        res["per_process_rate_of_the_concurrent_single_thread_solves"] = {
            "value": rate, "unit": "solves/s", "cores": 1,
            "sample": f"mean per-process rate of those {len(jobs)} CONCURRENT single-thread solves "
                      f"({np.mean([r[1] / max(1, r[0]) for r in rr]):.1f} s per problem); not a lone-process figure"}
        if full or workload != "cdu":
            k = 1 if workload == "cdu" else 32
            d1, t1, _ = _cpu_worker((path, 0, k, 1e9))
            res["one_thread_alone"] = {"value": d1 / t1, "unit": "solves/s", "cores": 1,
                                       "sample": f"{d1} problem(s), one process alone on the host, 1 BLAS thread, {t1:.1f} s"}
        os.unlink(path)
    return res


# ------------------------------------------------------------------------------------------------------------------
class Ctx:
    """One rank: its device, its communicator (None at world 1 unless NNMPC_FORCE_DIST=1)."""

    def __init__(self, args):
        from industrial_nnmpc_2021_amd import _lib, distributed
        self.lib = _lib
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if _lib.device_count() <= self.local_rank:
            raise SystemExit("bench.py needs a GPU per rank (the HIP path has no CPU fallback)")
        _lib.set_device(self.local_rank)
        self.comm = None
        if self.world > 1 or os.environ.get("NNMPC_FORCE_DIST") == "1":
            self.comm = distributed.Comm(self.rank, self.world)

    def sync(self):
        if self.comm is not None:
            self.comm.barrier()
        else:
            self.lib.synchronize()

    def max_over_ranks(self, v):
        return self.comm.allreduce_max(v) if self.comm is not None else v


def make_problem(workload, seed=0):
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    pl = synthetic.plant(workload, seed=seed)
    P, tq, nu = build_regulator_matrices(pl)
    return pl, P, tq, nu


def make_samples(pl, B, seed, sx):
    from industrial_nnmpc_2021_amd import synthetic
    s = synthetic.samples(pl, B, seed=seed, sx=sx)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    return x0, np.ascontiguousarray(pl["ulb"].T - s["us"]), np.ascontiguousarray(pl["uub"].T - s["us"]), s["us"]


class QpInputs:
    """One batch of inputs in HBM: x0 = [x - xs; uprev - us], the shifted bounds, us."""

    def __init__(self, lib, qp, B, nu):
        D = lib.DeviceArray
        self.x0, self.lb, self.ub, self.us = D((B, qp.n_aug), np.float64), D((B, nu), np.float64), D((B, nu), np.float64), D((B, nu), np.float64)
        self.host = None

    def upload(self, x0, lb, ub, us):
        self.x0.upload(x0); self.lb.upload(lb); self.ub.upload(ub); self.us.upload(us)
        self.host = (x0, lb, ub, us)

    def free(self):
        for a in (self.x0, self.lb, self.ub, self.us):
            a.free()


class QpBuffers:
    """HBM-resident inputs and outputs of one batch (owned through the library: _lib.DeviceArray)."""

    def __init__(self, lib, qp, B, nu, n, inputs=None):
        D = lib.DeviceArray
        self.inputs = [inputs or QpInputs(lib, qp, B, nu)]
        self.use(self.inputs[0])
        self.u, self.act = D((B, n), np.float64), D((B, qp.words), np.uint32)
        self.status, self.iters, self.first = D((B,), np.int32), D((B, 2), np.int32), D((B, nu), np.float64)
        self.B = B

    def use(self, ib):
        """The step at hand reads this batch of inputs."""
        if ib not in self.inputs:
            self.inputs.append(ib)
        self.cur = ib
        self.x0, self.lb, self.ub, self.us = ib.x0, ib.lb, ib.ub, ib.us

    def upload(self, x0, lb, ub, us):
        self.cur.upload(x0, lb, ub, us)

    def free(self):
        for ib in self.inputs:
            ib.free()
        for a in (self.u, self.act, self.status, self.iters, self.first):
            a.free()


def _traffic_of(traffic, names):
    """(average HBM bytes per launch, bytes per step, launches per step) of a group of kernels from the PMC profile, or Nones."""
    hit = [traffic[k] for k in names if k in traffic]
    if not hit:
        return None, None, None
    per_step = sum(h["per_step"] for h in hit)
    launches = sum(h["launches_per_step"] for h in hit)
    return per_step / max(launches, 1e-9), per_step, launches


def lambda_rooflines(st, traffic):
    """The two instances of the multiplier kernel, each priced against the peak of its OWN number type."""
    f32 = st["asm_lambda32_flops"]
    f64 = st["asm_lambda_flops"] - f32
    side_ms = st["asm_side_ms"]
    out = []
    for name, fl, ms, launches, peak, dt in (
            ("asm_lambda_reg32_k", f32, st["asm_lambda32_ms"], st["asm_lambda32_launches"], FP32_PEAK_TFLOPS, "f32"),
            ("asm_lambda_reg_k", f64, st["asm_lambda64_ms"], st["asm_lambda64_launches"], FP64_PEAK_TFLOPS, "f64")):
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        tr, _, _ = _traffic_of(traffic, [name])
        out.append({"kernel": f"{name} (|A| x |A| Cholesky + two substitutions per problem: one wave per problem, tiles in the MFMA "
                              f"accumulators, v_mfma_{dt}_16x16x4_{dt})", "dtype": dt, "bound": "mfma", "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s", "frac": ach / peak, "traffic": tr,
                    "launches": int(launches), "avg_launch_ms": ms / max(1, launches), "time_share": ms / st["total_ms"],
                    "algorithmic_flops": "m^3/3 + 2 m^2 per problem and round, m = size of its active set; the classes beyond 144 bounds "
                                         "(kernels asm_lambda_reg32b_k / wg* / tile_k<1> on three side streams, beside this kernel) are in "
                                         "the flop count of their number type but not in this kernel's time: `achieved` is an UPPER bound; "
                                         "`side_stream_ms` is the measured hipEvent time of those kernels (sum over the streams, both "
                                         "number types) against this kernel's `kernel_ms`",
                    "kernel_ms": ms, "side_stream_ms": side_ms})
    return out



def multiplier_family(st, traffic):
    """The multiplier kernels as ONE family (f32 and fp64 instances on the main stream + the larger-set kernels on the side
    streams): all their algorithmic flops over all their hipEvent time; `peak` = the flop-weighted mix of the f32 and fp64
    MFMA peaks (= flops / the time the family would take at the peak of each part's number type), so frac = ideal / measured."""
    f32 = st["asm_lambda32_flops"]
    f64 = st["asm_lambda_flops"] - f32
    ms = st["asm_lambda_ms"]                          # EvScope kind 4: everything between the fork and the joins of a round
    fl = f32 + f64
    if ms <= 0 or fl <= 0:
        return None
    ideal_ms = 1e3 * (f32 / (FP32_PEAK_TFLOPS * 1e12) + f64 / (FP64_PEAK_TFLOPS * 1e12))
    ach = fl / (ms * 1e-3) / 1e12
    peak = fl / (ideal_ms * 1e-3) / 1e12
    tr, tr_step, tr_l = _traffic_of(traffic, ["asm_lambda_reg32_k", "asm_lambda_reg_k", "asm_lambda_reg32b_k", "asm_lambda_wg64s_k", "asm_lambda_wg32_k",
                                              "asm_lambda_wg64_k", "asm_lambda_tile_k<1>", "asm_lambda_reg2_k"])
    launches = int(st["asm_lambda32_launches"] + st["asm_lambda64_launches"])
    return {"kernel": "multiplier family: asm_lambda_reg32_k (f32 rounds; from round 1 with one fp64 correction) + asm_lambda_reg_k (fp64) + side-stream classes > 144 bounds",
            "dtype": "f32+f64", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": tr,
            "traffic_per_step": tr_step, "launches": launches, "avg_launch_ms": ms / max(1, launches), "time_share": ms / st["total_ms"],
            "kernel_ms": ms, "f32_flops": f32, "f64_flops": f64, "f32_instance_ms": st["asm_lambda32_ms"], "f64_instance_ms": st["asm_lambda64_ms"],
            "side_stream_ms": st["asm_side_ms"],
            "algorithmic_flops": "m^3/3 + 2 m^2 per problem and round (m = size of its active set), f32 rounds priced at 157.3, fp64 at 78.6 TFLOP/s; "
                                 "the fp64 correction of an f32 solve (2 m^2 gathered fp64 FMAs + two substitutions) is in the time, not in the flops; "
                                 "time = hipEvents around the whole family of a round (main stream, side streams joined)"}


def survey_model(n, n_aug, solves_per_s, it=10):
    """SURVEY 8(d)'s dense per-sample PDIP model beside the headline: F_qp(n, it) = (it + 1) n^3/3 + it 6 n^2 + 2 n n_aug flop per
    solve.  This implementation does NOT execute those flops (shared inverse: |A|^3/3 per round + GEMM rows), so the product is
    an EQUIVALENT rate, not an achieved one; the dense model's own ceiling is peak / F_qp."""
    F = (it + 1) * n ** 3 / 3.0 + it * 6.0 * n * n + 2.0 * n * n_aug
    return {"F_qp_flop_per_solve": F, "iterations_assumed": it, "equivalent_TFLOPs": F * solves_per_s / 1e12,
            "dense_model_ceiling_solves_per_s_at_f32_peak": FP32_PEAK_TFLOPS * 1e12 / F}


def _clip(s, k):
    s = str(s)
    return s if len(s) <= k else s[:k - 3] + "..."


def _num(v, digits=6):
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float(f"{float(v):.{digits}g}")
    except (TypeError, ValueError):
        return None


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: under 4 KB whatever the detail holds (the full result goes to `detail_path`).  The reference's
    own timing record is one scalar per run (lib/linearMPC.py:835, :874-880: data_gen_time)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _num(out.get(k)) for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {"workload": _clip(cfg.get("workload", ""), 200)}
    for k in ("batch_per_gpu", "total_batch", "sx", "method", "parallelism", "flops_per_state"):
        if k in cfg:
            line["config"][k] = _clip(cfg[k], 60) if isinstance(cfg[k], str) else _num(cfg[k])
    r = out.get("roofline")
    if r:
        line["roofline"] = {"kernel": _clip(r.get("kernel", ""), 120)}
        for k in ("dtype", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "time_share"):
            v = r.get(k)
            line["roofline"][k] = _num(v) if not isinstance(v, dict) else {kk: _num(vv, 4) for kk, vv in v.items()}
    if "roofline_gemm_group" in out:
        g = out["roofline_gemm_group"]
        line["roofline_gemm_group"] = {k: _num(g.get(k), 4) for k in ("dtype", "achieved", "peak", "frac", "time_share", "traffic")}
    if "roofline_predictor" in out:
        g = out["roofline_predictor"]
        line["roofline_predictor"] = {k: _num(g.get(k), 4) for k in ("dtype", "achieved", "peak", "frac", "time_share", "traffic")}
    if "roofline_step" in out:
        rs = out["roofline_step"]
        line["roofline_step"] = {"frac": _num(rs.get("frac"), 4), "ideal_ms_per_step": _num(rs.get("ideal_ms_per_step"), 4),
                                 "hbm_bytes_per_step": _num(rs.get("hbm_bytes_per_step_all_kernels"), 4)}
    if "survey_model" in out:
        line["survey_model_TFLOPs"] = _num(out["survey_model"]["equivalent_TFLOPs"], 4)
        line["survey_model_ceiling_solves_per_s"] = _num(out["survey_model"]["dense_model_ceiling_solves_per_s_at_f32_peak"], 4)
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {"value": _num(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
                                "sample": _clip(c.get("sample", ""), 160)}
        if "nproc_processes" in c:
            line["cpu_baseline"]["nproc_processes"] = {"value": _num(c["nproc_processes"].get("value")), "cores": c["nproc_processes"].get("cores")}
    pr = out.get("parity")
    if pr:
        line["parity"] = {k: _num(pr.get(k)) for k in ("checked", "max_rel_err_vs_fp64_oracle", "active_set_hamming", "rows_checked", "steady_state_row_exact") if k in pr}
        if "kkt_check" in pr:
            line["parity"]["kkt_rows"] = pr["kkt_check"].get("rows")
            line["parity"]["kkt_wrong_sign_multipliers"] = pr["kkt_check"].get("wrong_sign_multipliers")
            line["parity"]["kkt_max_stationarity_rel"] = _num(pr["kkt_check"].get("max_stationarity_residual_rel"), 3)
    so = out.get("solver")
    if so:
        line["status_hist"] = so.get("status_hist")
        line["rounds_per_step"] = _num(so.get("active_set_rounds_per_step"), 3)
    if "first_move_output" in out:
        line["first_move_solves_per_s"] = _num(out["first_move_output"].get("value"), 4)
    cf = out.get("configs")
    if cf:
        cc = {}
        if "cstrs_10k" in cf:
            r2 = cf["cstrs_10k"]
            cc["cstrs_10k"] = {"solves_per_s": _num(r2.get("value"), 4), "ms_per_step": _num(r2.get("ms_per_step"), 4),
                               "max_rel_err": _num((r2.get("parity") or {}).get("max_rel_err_vs_fp64_oracle"), 3),
                               "hamming": (r2.get("parity") or {}).get("active_set_hamming")}
        if "cdu_unstable" in cf:
            r3 = cf["cdu_unstable"]
            cc["cdu_unstable"] = {k: _num(r3.get(k), 4) for k in ("value", "ms_per_step", "window", "farfield_rank", "rounds_per_step", "max_rel_err_vs_fp64_oracle") if k in r3}
        if "nn_1m" in cf:
            nn = cf["nn_1m"]
            cc["nn_1m"] = {m: {"states_per_s": _num(nn[m].get("states_per_s"), 4), "frac": _num((nn[m].get("roofline") or nn.get("roofline") or {}).get("frac") if m != "bf16x3" else nn[m].get("frac_of_bf16_peak_executed"), 3),
                               "max_rel_err": _num(nn[m].get("max_rel_err_vs_fp64_oracle"), 3)} for m in ("f32", "bf16", "bf16x3") if m in nn}
        line["configs"] = cc
    if "sweep_sx" in out:
        line["sweep_sx_solves_per_s"] = {k: _num(v.get("value"), 3) for k, v in out["sweep_sx"].items()}
    if "chains_task" in out and out["chains_task"]:
        line["chains_task_wall_s"] = _num(out["chains_task"].get("wall_s"), 4)
    if detail_path:
        line["detail"] = detail_path
    s = json.dumps(line)
    for drop in ("sweep_sx_solves_per_s", "configs", "roofline_gemm_group", "first_move_solves_per_s", "status_hist"):   # never over the limit
        if len(s) < 3800:
            break
        line.pop(drop, None)
        s = json.dumps(line)
    return s


def emit(out):
    """Detail -> a side file (bench_detail.json beside this script, or $NNMPC_BENCH_DETAIL; stderr when neither can be written),
    then the compact line as the LAST line on stdout."""
    path = os.environ.get("NNMPC_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(out, f)
        shown = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        sys.stderr.write(json.dumps(out) + "\n")
        shown = "stderr"
    sys.stdout.flush()
    print(compact_line(out, shown))
    sys.stdout.flush()


def bench_qp(ctx, workload, B, steps, warmup, sx, method="auto", slots=0, seed0=1000, want_buffers=False):
    """K timed steps of the regulator QP batch on this rank (+ the gather at world > 1).  Returns (result dict, handles)."""
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    lib = ctx.lib
    pl, P, tq, nu = make_problem(workload)
    n, n_aug, N = P.shape[0], tq.shape[1], pl["N"]
    slots = slots or (1024 if workload == "cdu" else 8192)
    qp = BatchedBoxQP(P, tq, nu, max_batch=min(slots, B), method=method)
    # A fresh seeded batch for EVERY step (setup, warm-up and timed ones), all resident in HBM before the clock starts: no step
    # re-solves a batch the handle has seen (calls are history-free by construction; this removes the question).  Every rank
    # draws its own shards of the stream.  More than NSETS_MAX steps cycle through the sets (0.3 GB of inputs each at the CDU size).
    nsets = min(NSETS_MAX, 1 + warmup + steps)
    sets = []
    for i in range(nsets):
        x0_h, lb_h, ub_h, us_h = make_samples(pl, B, seed0 + 7919 * i + ctx.rank, sx)
        ib = QpInputs(lib, qp, B, nu)
        ib.upload(x0_h, lb_h, ub_h, us_h)
        sets.append(ib)
        if i < nsets - 1:
            ib.host = None                             # (only the last set's host copy is needed: the parity leg)
            del x0_h, lb_h, ub_h, us_h
    buf = QpBuffers(lib, qp, B, nu, n, inputs=sets[0])
    gathered = lib.DeviceArray((ctx.world * B, nu), np.float64) if (ctx.comm is not None and ctx.rank == 0) else None
    rows = [B] * ctx.world
    calls = [0]

    def step():
        ib = sets[calls[0] % nsets]
        calls[0] += 1
        buf.use(ib)
        qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
        # get_control_sequence adds us back (:689); ut = useq[0:Nu] (:856)
        lib.check(lib.load().nnmpc_qp_first_moves(buf.u.data_ptr(), n, buf.us.data_ptr(), B, nu, buf.first.data_ptr()), "nnmpc_qp_first_moves")
        if ctx.comm is not None:
            ctx.comm.gather_rows(buf.first, rows, nu, gathered, root=0)   # the single RCCL gather over xGMI

    # one-time setup, untimed and outside the warmup count: the far-field factors of the column windows a batch of this plant can
    # end in (one host SVD each, like the inverse itself; a call that meets a window without factors runs the dense form of the
    # full-width pass and factors it afterwards -- with a fresh batch per step that would land inside the timed region), and the
    # first pass of the handle
    t_ff = time.perf_counter()
    ff_ranks = qp.prepare_farfield_windows() if workload == "cdu" else {}
    t_ff = time.perf_counter() - t_ff
    step()
    for _ in range(warmup):
        step()
    # the timed steps end on the LAST set: its host copy is what the parity leg checks the outputs against
    calls[0] = (nsets - steps) % nsets
    qp.set_profiling(True)
    qp.stats(reset=True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    dt_local = time.perf_counter() - t0
    dt = ctx.max_over_ranks(dt_local)
    st = qp.stats()
    qp.set_profiling(False)
    status_h, iters_h = buf.status.to_host(), buf.iters.to_host()
    res = {"value": ctx.world * B * steps / dt, "unit": "solves/s", "ms_per_step": 1e3 * dt / steps,
           "per_rank_solves_per_s_this_rank": B * steps / dt_local, "distinct_input_batches": nsets,
           "farfield_setup": {"windows": {str(k): int(v) for k, v in ff_ranks.items()}, "seconds_not_timed": t_ff},
           "solver": {"status_hist": np.bincount(status_h, minlength=3).tolist(),
                      "solved_by_active_set_pass": int(st["asm_solved"]), "active_set_rounds_per_step": st["asm_rounds"] / steps,
                      "solved_by_pdip_path": int(st["problems"] - st["asm_solved"]),
                      "mean_pdip_iters": float(iters_h[:, 0].mean()), "mean_factorizations": float(iters_h[:, 1].mean()),
                      "mean_active_bounds": float(np.unpackbits(buf.act.to_host(min(B, 4096)).view(np.uint8), axis=1).sum(axis=1).mean()),
                      "bounds_per_problem": 2 * n, "farfield_windows_factored": sorted(int(k) for k in qp.farfield_info)}}
    if st["asm_solved"]:
        traffic, tnote = pmc_traffic(f"{workload}_b{B}")
        gach = st["asm_gemm_flops"] / (st["asm_gemm_ms"] * 1e-3) / 1e12
        gemm_group = ["gemm_nt_f64_t128_k", "asm_wide_t_k", "asm_wide_gemm_k<2>", "asm_wide_gemm_k<1>", "asm_wide_gemm_k<0>",
                      "gemm_nt_f32_kdyn_k<128>", "gemm_nt_f32_kdyn_k<64>", "gemm_nt_f64_k"]
        tr, tr_step, tr_launches = _traffic_of(traffic, gemm_group)
        gemm = {"kernel": "fp64 MFMA GEMM group: gemm_nt_f64_t128_k (x_unc = x0 Kunc' for the leading columns, XH = LAM Pinv inside the column "
                          "window) + the full-width pass asm_wide_t_k / asm_wide_gemm_k (far-field form: T = [x0 | lam] V, x = T U', check in "
                          "the epilogue); the f32 rounds' XH32 = LAM32 Pinv32 on gemm_nt_f32_kdyn_k is in the same time and flop count",
                "dtype": "f64", "bound": "mfma", "achieved": gach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gach / FP64_PEAK_TFLOPS,
                "traffic": tr, "traffic_per_step": tr_step, "traffic_launches_per_step": tr_launches,
                "launches": int(st["asm_gemm_launches"]),
                "avg_launch_ms": st["asm_gemm_ms"] / max(1, st["asm_gemm_launches"]), "time_share": st["asm_gemm_ms"] / st["total_ms"],
                "far_field_passes": int(st["asm_far_passes"]),
                "algorithmic_flops": "the flops of the algorithm as implemented, unpadded: 2 * columns * n_aug (x_unc, leading columns) + per round "
                                     "2 * window columns * (own last active bound + 1) per running problem + per problem ONE full-width pass: "
                                     "far-field form 2 r (n_aug + own last active bound + 1) + 2 r (columns beyond the window) with r = the "
                                     "padded rank of the far block (dense form, when no factors are set: 2 (n_aug + own bound + 1) columns)",
                "algorithmic_bytes_per_step": float(B) * (n_aug + 2 * nu + n) * 8,
                "traffic_note": "`traffic` = HBM bytes per launch averaged over the group's launches of one step, `traffic_per_step` their sum "
                                "(PMC passes of scripts/pmc_hbm.sh); algorithmic_bytes_per_step = x0, bounds in + u* out"}
        fam = multiplier_family(st, traffic)
        inst = lambda_rooflines(st, traffic)
        pred = None
        if st["asm_predict_ms"] > 0:
            # first-set predictor (qp_predict.h): bf16 MFMA, f32 accumulate; flops = the dense count 2 W^2 per problem and iteration
            pach = st["asm_predict_flops"] / (st["asm_predict_ms"] * 1e-3) / 1e12
            trp, trp_step, _ = _traffic_of(traffic, ["asm_predict_k<4, 4, true>", "asm_predict_k<8, 2, true>", "asm_predict_k<4, 2, false>"])
            pred = {"kernel": "asm_predict_k: dual accelerated projected gradient on the window (x = x_unc - H' y on v_mfma_f32_16x16x32_bf16, prox + momentum "
                              "elementwise), 64 problems per workgroup for all iterations, H' (512 KB bf16) streamed from L2 once per iteration",
                    "dtype": "bf16", "bound": "mfma", "achieved": pach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": pach / BF16_PEAK_TFLOPS,
                    "traffic": trp, "traffic_per_step": trp_step, "launches": int(st["asm_predict_launches"]),
                    "avg_launch_ms": st["asm_predict_ms"] / max(1, st["asm_predict_launches"]), "time_share": st["asm_predict_ms"] / st["total_ms"],
                    "algorithmic_flops": "2 * 512 * 512 per problem and iteration after the first (bf16 operands, f32 accumulate); the kernel is co-limited by "
                                         "the L2 -> CU stream of H' (512 KB per workgroup and iteration: 39 TB/s over the chip at the MFMA-bound rate)"}
        small = None
        if st["asm_small_passes"]:
            # small problems (n <= 724): the whole iteration runs in asm_small_k, one wave per problem (qp_small.h) -- a chain of L2 round
            # trips (gather of H_AA, rows of Pinv for x), not a throughput kernel: priced against the fp64 peak for the record
            sach = st["asm_lambda_flops"] / (st["asm_lambda_ms"] * 1e-3) / 1e12 if st["asm_lambda_ms"] > 0 else 0.0
            tr_s, tr_s_step, tr_s_l = _traffic_of(traffic, ["asm_small_k<2, 3, 4>", "asm_small_k<7, 1, 8>"])
            small = {"kernel": "asm_small_k<2, 3, 4> + asm_small_k<7, 1, 8>: the whole active-set iteration of a problem in one wave (ordered "
                               "set, |A| x |A| Cholesky in the MFMA accumulators, x = x_unc - lam Pinv[A, :] from L2, tests, exchange rule, "
                               "certificate), sets of up to 32 / 112 bounds", "dtype": "f64", "bound": "mfma", "achieved": sach,
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": sach / FP64_PEAK_TFLOPS, "traffic": tr_s, "traffic_per_step": tr_s_step,
                     "launches": 2 * int(st["asm_small_passes"]), "avg_launch_ms": st["asm_lambda_ms"] / max(1, 2 * st["asm_small_passes"]),
                     "time_share": st["asm_lambda_ms"] / st["total_ms"],
                     "algorithmic_flops": "m^3/3 + 2 m^2 per problem and iteration (m = size of its active set); the products with the rows "
                                          "of Pinv (2 m x window columns) are not counted",
                     "note": "latency-bound by construction: every iteration is gather -> factorisation -> rows of Pinv -> tests in ONE wave; "
                             "the launch lasts as long as its slowest problem"}
        # `roofline` = the kernel FAMILY with the largest share of the step (the multiplier kernels count as one family, not per instance)
        fams = [f for f in (small if small else fam, gemm) if f]
        fams.sort(key=lambda r: -r["time_share"])
        res["roofline"] = fams[0]
        res["roofline_gemm_group"] = gemm
        res["roofline_multiplier_instances"] = inst
        if pred:
            res["roofline_predictor"] = pred
        if small and fam:
            res["roofline_multiplier_family"] = fam
        res["roofline"]["traffic_unit"] = "HBM bytes per launch; " + tnote
        res["survey_model"] = survey_model(n, n_aug, res["value"])
        # the whole step against the roofline: every part's algorithmic flops at the peak of its own number type / the step's time
        f32l = st["asm_lambda32_flops"]
        ideal_ms = 1e3 * (st["asm_gemm_flops"] / (FP64_PEAK_TFLOPS * 1e12) + f32l / (FP32_PEAK_TFLOPS * 1e12)
                          + (st["asm_lambda_flops"] - f32l) / (FP64_PEAK_TFLOPS * 1e12) + st["asm_predict_flops"] / (BF16_PEAK_TFLOPS * 1e12))
        res["roofline_step"] = {"bound": "mfma", "ideal_ms_per_step": ideal_ms / steps, "ms_per_step": 1e3 * dt / steps,
                                "frac": ideal_ms / steps / (1e3 * dt / steps),
                                "algorithmic_flops_per_step": {"gemm_f64": st["asm_gemm_flops"] / steps, "multiplier_f32": f32l / steps,
                                                               "multiplier_f64": (st["asm_lambda_flops"] - f32l) / steps,
                                                               "predictor_bf16": st["asm_predict_flops"] / steps},
                                "note": "sum over the parts of (algorithmic flops / peak of that part's number type) over the measured step time; the "
                                        "f32 window GEMM is counted with the fp64 group (conservative: its flops are priced at the fp64 peak)",
                                "hbm_bytes_per_step_all_kernels": (sum(v["per_step"] for v in traffic.values()) if traffic else None)}
        res["time_shares"] = {"multiplier_kernels_both_streams": st["asm_lambda_ms"] / st["total_ms"], "gemms": st["asm_gemm_ms"] / st["total_ms"],
                              "set_bookkeeping_kernels": st["asm_update_ms"] / st["total_ms"], "first_set_predictor": st["asm_predict_ms"] / st["total_ms"]}
        res["solver"]["checked_with_P_itself"] = int(st["asm_full_checks"])
        res["solver"]["inverse_check"] = {"max_abs_P_Pinv_minus_I": st["asm_e2max"], "max_abs_P_Kunc_plus_tq": st["asm_e1max"]}
        res["dtype"] = "f64"
    else:
        res["roofline"] = panel_roofline(st, n, workload)
        res["dtype"] = "f32"
    for ib in sets:
        buf.use(ib)                                    # (ownership: buf.free() releases every set)
    buf.use(sets[(calls[0] - 1) % nsets])              # the batch the outputs in `buf` belong to
    handles = dict(qp=qp, buf=buf, pl=pl, P=P, tq=tq, nu=nu, N=N, n=n, host=buf.cur.host)
    if not want_buffers:
        qp.close(); buf.free()
        handles = None
    return res, handles


def panel_roofline(stx, n, workload):
    fl, ms = stx["panel_flops"], stx["panel_ms"]
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic, tnote = pmc_traffic(f"{workload}_pdip")
    return {"kernel": "chol_panel_k", "dtype": "f32", "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS, "traffic": _traffic_of(traffic, ["chol_panel_k<128>", "chol_panel_k<64>"])[0], "traffic_unit": "HBM bytes per launch; " + tnote,
            "launches": int(stx["panel_launches"]), "avg_launch_ms": ms / max(1, stx["panel_launches"]),
            "time_share": {"chol_panel": ms / stx["total_ms"], "chol_diag": stx["diag_ms"] / stx["total_ms"], "trsv": stx["trsv_ms"] / stx["total_ms"]},
            # whole-solve rate in the survey's dense-PDIP flop model: factorisations * n^3/3
            "cholesky_flops_over_solve_time_TFLOPs": stx["factorizations"] * n ** 3 / 3 / (stx["total_ms"] * 1e-3) / 1e12}


def parity_leg(h, k_oracle, k_kkt):
    """Independent checks of the batch just solved: (a) k_kkt random rows through the fp64 KKT conditions evaluated in
    numpy, (b) k_oracle rows -- the largest active sets, the rows that needed the most rounds' worth of work are among
    the largest, and random ones -- against the exact-optimum oracle."""
    from oracle import qp as oqp
    buf, P, tq, nu, N, n = h["buf"], h["P"], h["tq"], h["nu"], h["N"], h["n"]
    x0_h, lb_h, ub_h, _ = h["host"]
    B = buf.B
    Ps = np.tril(P) + np.tril(P, -1).T
    rng = np.random.default_rng(5)
    act_all = buf.act.to_host()
    nact = np.unpackbits(act_all.view(np.uint8), axis=1).sum(axis=1)
    big = np.argsort(-nact)[:k_oracle // 2]
    rows_o = np.unique(np.concatenate((big, rng.choice(B, k_oracle, replace=False))))[:max(k_oracle, 1)]
    rows_k = np.unique(np.concatenate((rows_o, rng.choice(B, min(B, k_kkt), replace=False))))
    # pull only the rows needed (B x n doubles is 3.6 GB at the CDU size)
    U = np.stack([np.frombuffer(_row(buf.u, r, n * 8), np.float64) for r in rows_k])
    bits = np.unpackbits(act_all[rows_k].view(np.uint8), axis=1, bitorder="little")[:, :2 * n].astype(bool)
    kk, cc = np.arange(n) // nu, np.arange(n) % nu
    au, al = bits[:, kk * 2 * nu + cc], bits[:, kk * 2 * nu + nu + cc]
    G = U @ Ps + x0_h[rows_k] @ tq.T
    LB, UB = np.tile(lb_h[rows_k], (1, N)), np.tile(ub_h[rows_k], (1, N))
    scale = np.maximum(1.0, np.abs(x0_h[rows_k] @ tq.T).max(axis=1, keepdims=True))
    free = ~(au | al)
    kkt = {"rows": int(rows_k.size),
           "max_stationarity_residual_rel": float((np.abs(np.where(free, G, 0)) / scale).max()),
           "max_bound_violation": float(max((U - UB).max(), (LB - U).max(), 0.0)),
           "active_rows_exactly_on_their_bound": bool(np.abs(np.where(au, U - UB, 0)).max() == 0 and np.abs(np.where(al, U - LB, 0)).max() == 0),
           "wrong_sign_multipliers": int((np.where(au, -G, 1) <= 0).sum() + (np.where(al, G, 1) <= 0).sum())}
    errs, ham, sizes = [], 0, []
    pos = {r: i for i, r in enumerate(rows_k)}
    # the oracle solves are independent: the worker processes forked before the first GPU call (start_workers), a few BLAS
    # threads each; the problem data goes through /dev/shm
    nw = WORKERS._processes if WORKERS is not None else 1
    path = _publish(Ps=Ps, tq=tq, nu=nu, N=N, x0=x0_h[rows_o], lb=lb_h[rows_o], ub=ub_h[rows_o])
    sols = _map(_oracle_row, [(path, i, max(1, (os.cpu_count() or 1) // (2 * nw)) if n >= 1024 else 0) for i in range(len(rows_o))])
    os.unlink(path)
    for r, (xe, active) in zip(rows_o, sols):
        ref = np.zeros(2 * n, bool)
        ref[active] = True
        errs.append(float(np.abs(U[pos[r]] - xe).max() / max(1.0, np.abs(xe).max())))
        ham += int((bits[pos[r]] != ref).sum())
        sizes.append(int(ref.sum()))
    return {"checked": int(rows_o.size), "max_rel_err_vs_fp64_oracle": max(errs) if errs else None, "active_set_hamming": ham,
            "oracle_rows_active_set_sizes": {"min": min(sizes), "max": max(sizes)} if sizes else None,
            "largest_active_set_in_batch": int(nact.max()), "kkt_check": kkt}


def _oracle_row(arg):
    """Exact optimum and active rows of one problem of the batch (oracle.qp.solve_exact_box); runs in a worker process."""
    path, i, threads = arg
    from oracle import qp as oqp
    d = _shared(path)
    nu, N = int(d["nu"]), int(d["N"])

    def run():
        info = {"nu": nu}
        xe = oqp.solve_exact_box(d["Ps"], d["tq"] @ d["x0"][i], np.tile(d["lb"][i], N), np.tile(d["ub"][i], N), info=info)
        return xe, info["active"]
    if threads:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=threads):
            return run()
    return run()


def _row(dev, r, nbytes):
    import ctypes as C
    from industrial_nnmpc_2021_amd import _lib
    out = (C.c_char * nbytes)()
    _lib.check(_lib.load().nnmpc_memcpy_d2h(out, dev.row_ptr(r), nbytes), "nnmpc_memcpy_d2h")
    return bytes(out)


def pdip_leg(ctx, h, Bp):
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    lib = ctx.lib
    buf, n, nu = h["buf"], h["n"], h["nu"]
    qp2 = BatchedBoxQP(h["P"], h["tq"], nu, max_batch=min(1024, Bp), method="pdip")
    b2 = QpBuffers(lib, qp2, Bp, nu, n)
    qp2.solve_batch_device(min(Bp, 256), buf.x0, buf.lb, buf.ub, b2.u, b2.act, b2.status, b2.iters)      # warm-up
    qp2.set_profiling(True); qp2.stats(reset=True)
    lib.synchronize(); t1 = time.perf_counter()
    qp2.solve_batch_device(Bp, buf.x0, buf.lb, buf.ub, b2.u, b2.act, b2.status, b2.iters)
    lib.synchronize(); dt2 = time.perf_counter() - t1
    s2 = qp2.stats()
    it2 = b2.iters.to_host()
    k = min(Bp, 256)
    du = float(np.abs(b2.u.to_host(k) - buf.u.to_host(k)).max())
    out = {"value": Bp / dt2, "unit": "solves/s", "batch": Bp, "dtype": "f32 (+f64 refinement)",
           "status_hist": np.bincount(b2.status.to_host(), minlength=3).tolist(),
           "mean_pdip_iters": float(it2[:, 0].mean()), "mean_factorizations": float(it2[:, 1].mean()),
           "max_abs_diff_vs_active_set_pass": du, "compared_rows": k,
           "active_sets_equal": bool(np.array_equal(b2.act.to_host(), buf.act.to_host(Bp))),
           "roofline": panel_roofline(s2, n, "cdu")}
    qp2.close(); b2.free()
    return out


def host_io_leg(ctx, h):
    """The same batch with host buffers either side (never `value`): pinned host memory -> HBM over PCIe, the solve, first
    moves (what simulate_offline keeps, lib/linearMPC.py:856) or full sequences back to the host."""
    import ctypes as C
    lib = ctx.lib
    L = lib.load()
    qp, buf, n, nu = h["qp"], h["buf"], h["n"], h["nu"]
    x0_h, lb_h, ub_h, us_h = h["host"]
    B = buf.B

    def pinned(a):
        p = C.c_void_p()
        lib.check(L.nnmpc_host_alloc_pinned(C.byref(p), a.nbytes), "nnmpc_host_alloc_pinned")
        v = np.frombuffer((C.c_char * a.nbytes).from_address(p.value), dtype=a.dtype).reshape(a.shape)
        v[...] = a
        return p, v
    pins = [pinned(a) for a in (x0_h, lb_h, ub_h)]
    pf, vf = pinned(np.empty((B, nu)))
    pu, vu = pinned(np.empty((B, n)))
    hio = {}
    for name, full in (("first_move", False), ("full_sequence", True)):
        best = None
        for _ in range(2):
            lib.synchronize(); t1 = time.perf_counter()
            for (p, v), d in zip(pins, (buf.x0, buf.lb, buf.ub)):
                lib.check(L.nnmpc_memcpy_h2d(C.c_void_p(d.data_ptr()), p, v.nbytes), "h2d")
            qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.first if not full else buf.u, buf.act, buf.status, buf.iters,
                                  first_move_only=not full)
            if full:
                lib.check(L.nnmpc_memcpy_d2h(pu, C.c_void_p(buf.u.data_ptr()), vu.nbytes), "d2h")
            else:
                lib.check(L.nnmpc_memcpy_d2h(pf, C.c_void_p(buf.first.data_ptr()), vf.nbytes), "d2h")
            lib.synchronize(); d_ = time.perf_counter() - t1
            best = d_ if best is None else min(best, d_)
        hio[name + "_solves_per_s"] = B / best
        hio[name + "_ms"] = 1e3 * best
    hio["bytes_in"] = int(sum(v.nbytes for _, v in pins))
    hio["bytes_out_first_move"] = B * nu * 8
    hio["bytes_out_full_sequence"] = B * n * 8
    hio["note"] = ("PCIe-inclusive: pinned host buffers -> HBM, solve, results -> pinned host buffers; best of 2; the first-move run uses "
                   "NNMPC_OUT_FIRST_MOVE (u* is not written out in full)")
    for p, _ in pins + [(pf, vf), (pu, vu)]:
        L.nnmpc_host_free_pinned(p)
    return hio


def first_move_leg(ctx, h, steps):
    """The headline batch again with NNMPC_OUT_FIRST_MOVE: every problem solved and certified as before, only
    useq[0:Nu] leaves the solver (all the offline simulation keeps, lib/linearMPC.py:856).  Beyond the column window the
    variables are only CHECKED in such a call: the far-field pass skips the column tiles |U_j| |T_p| <= min(ub, -lb)
    certifies (qp_wide.h).  Compared with the sequence call on the same batch: active sets and status bit for bit, first
    moves to the last bit.  Not `value`."""
    lib = ctx.lib
    qp, buf, nu = h["qp"], h["buf"], h["nu"]
    B = buf.B
    qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)          # the sequence call
    act_seq, st_seq = buf.act.to_host(), buf.status.to_host()
    k = min(B, 8192)
    u_seq = buf.u.to_host(k)[:, :nu]
    qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.first, buf.act, buf.status, buf.iters, first_move_only=True)
    lib.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.first, buf.act, buf.status, buf.iters, first_move_only=True)
    lib.synchronize(); dt = time.perf_counter() - t0
    st = buf.status.to_host()
    return {"value": B * steps / dt, "unit": "solves/s", "ms_per_step": 1e3 * dt / steps,
            "status_hist": np.bincount(st, minlength=3).tolist(),
            "active_sets_equal_to_sequence_call": bool(np.array_equal(buf.act.to_host(), act_seq)),
            "status_equal_to_sequence_call": bool(np.array_equal(st, st_seq)),
            "max_abs_first_move_diff_vs_sequence_call": float(np.abs(buf.first.to_host(k) - u_seq).max()), "rows_compared": int(k)}


def sweep_leg(ctx, h, sxs, steps):
    """The same plant and batch size at other state spreads: 0.5-5 % of the 8960 bounds active."""
    lib = ctx.lib
    qp, buf, n, nu, pl = h["qp"], h["buf"], h["n"], h["nu"], h["pl"]
    B = buf.B
    out = {}
    for sx in sxs:
        x0, lb, ub, us = make_samples(pl, B, 2000 + int(10 * sx), sx)
        buf.upload(x0, lb, ub, us)
        for _ in range(2):              # untimed, like the headline's: the call that leaves the far-field factors of this spread's column
            qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)   # windows behind (seconds of host
        qp.stats(reset=True)            # SVD, the GPU idles and clocks down) and one call that brings the clocks back
        lib.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
        lib.synchronize(); dt = time.perf_counter() - t0
        st = qp.stats()
        nact = np.unpackbits(buf.act.to_host(min(B, 8192)).view(np.uint8), axis=1).sum(axis=1)
        out[f"sx={sx:g}"] = {"value": B * steps / dt, "unit": "solves/s", "ms_per_step": 1e3 * dt / steps,
                             "mean_active_bounds": float(nact.mean()), "max_active_bounds": int(nact.max()),
                             "active_fraction": float(nact.mean() / (2 * n)),
                             "status_hist": np.bincount(buf.status.to_host(), minlength=3).tolist(),
                             "rounds_per_step": st["asm_rounds"] / steps, "solved_by_pdip_path": int(st["problems"] - st["asm_solved"])}
    buf.upload(*h["host"])
    return out


def make_unstable_problem(rho=1.03, seed=0):
    """SURVEY 8(d)'s second synthetic family at the CDU size: spectral radius 1.03 => the reference re-parameterises u = Kx + v and
    its G = tE (I + tK tB) is dense (lib/linearMPC.py:366-382, :476-479).  Returns the plant, the mirror regulator (dense-G view of
    the problem) and the box form the GPU solves (DenseQPRegulator._box_form: same optimum, same active rows)."""
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd import linearMPC as lm
    from industrial_nnmpc_2021_amd.linearMPC_build import augmented_matrices_for_regulator
    pl = synthetic.plant("cdu", seed=seed, rho=rho)
    Aa, Ba, Qa, Ra, Ma = augmented_matrices_for_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"])
    reg = lm.DenseQPRegulator(A=Aa, B=Ba, Q=Qa, R=Ra, M=Ma, N=pl["N"], ulb=pl["ulb"], uub=pl["uub"])
    if not reg.reparameterize:
        raise RuntimeError("the plant is stable: no re-parameterisation")
    Pw, tqw = reg._box_form()
    return pl, reg, Pw, tqw


def unstable_leg(ctx, B=16384, steps=3, sx=1.5):
    """configs.cdu_unstable: the rho = 1.03 family at n = 4480 through the same solver (input-space box form).  Reports what the
    far field does there: cond(P_box) ~ 1e6 puts the noise floor of the inverse at ~1e-10 sigma_1, so the far block's numerical rank
    at the library's tolerance is not ~Nx and the factored form is refused -- the full-width pass then runs in its dense (lazy) form."""
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    lib = ctx.lib
    t0 = time.perf_counter()
    pl, reg, Pw, tqw = make_unstable_problem()
    nu, n = reg.Nu, Pw.shape[0]
    qp = BatchedBoxQP(Pw, tqw, nu, max_batch=1024)
    t_setup = time.perf_counter() - t0
    qp.prepare_farfield_windows()                      # (one-time setup; on this plant every window is refused: see the docstring)
    nsets = steps + 2
    sets = []
    for i in range(nsets):
        x0, lb, ub, us = make_samples(pl, B, 4000 + i, sx)
        ib = QpInputs(lib, qp, B, nu)
        ib.upload(x0, lb, ub, us)
        sets.append(ib)
    buf = QpBuffers(lib, qp, B, nu, n, inputs=sets[0])
    for ib in sets[:2]:                                # setup call (dense form, factors prepared if they pay) + one warm-up
        buf.use(ib)
        qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
    qp.set_profiling(True); qp.stats(reset=True)
    lib.synchronize(); t1 = time.perf_counter()
    for ib in sets[2:]:
        buf.use(ib)
        qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
    lib.synchronize(); dt = time.perf_counter() - t1
    st = qp.stats(); qp.set_profiling(False)
    nact = np.unpackbits(buf.act.to_host(min(B, 8192)).view(np.uint8), axis=1).sum(axis=1)
    status = buf.status.to_host()
    # independent fp64 KKT conditions of 500 rows of the last batch, on the box form (numpy)
    x0_h, lb_h, ub_h, _ = sets[-1].host
    rows = np.sort(np.random.default_rng(3).choice(B, min(B, 500), replace=False))
    U = np.stack([np.frombuffer(_row(buf.u, r, n * 8), np.float64) for r in rows])
    bits = np.unpackbits(buf.act.to_host()[rows].view(np.uint8), axis=1, bitorder="little")[:, :2 * n].astype(bool)
    kk, cc = np.arange(n) // nu, np.arange(n) % nu
    au, al = bits[:, kk * 2 * nu + cc], bits[:, kk * 2 * nu + nu + cc]
    Gr = U @ Pw + x0_h[rows] @ tqw.T
    N = n // nu
    LB, UB = np.tile(lb_h[rows], (1, N)), np.tile(ub_h[rows], (1, N))
    scale = np.maximum(1.0, np.abs(x0_h[rows] @ tqw.T).max(axis=1, keepdims=True))
    free = ~(au | al)
    out = {"value": B * steps / dt, "unit": "solves/s", "ms_per_step": 1e3 * dt / steps, "batch": B, "sx": sx, "rho": 1.03,
           "cond_P_box": float(np.linalg.cond(Pw)), "setup_s": t_setup,
           "mean_active_bounds": float(nact.mean()), "max_active_bounds": int(nact.max()),
           "status_hist": np.bincount(status, minlength=4).tolist(),
           "rounds_per_step": st["asm_rounds"] / steps, "solved_by_active_set_pass": int(st["asm_solved"]),
           "solved_by_pdip_path": int(st["problems"] - st["asm_solved"]), "checked_with_P_itself": int(st["asm_full_checks"]),
           "far_field_passes": int(st["asm_far_passes"]), "farfield": {str(k): v for k, v in qp.farfield_info.items()},
           "window": max(qp.farfield_info) if qp.farfield_info else None,
           "farfield_rank": max([v.get("rank", 0) for v in qp.farfield_info.values()] + [0]),
           "inverse_check": {"max_abs_P_Pinv_minus_I": st["asm_e2max"], "max_abs_P_Kunc_plus_tq": st["asm_e1max"]},
           "time_shares": {"multiplier_kernels": st["asm_lambda_ms"] / max(st["total_ms"], 1e-9), "gemms": st["asm_gemm_ms"] / max(st["total_ms"], 1e-9),
                           "set_bookkeeping_kernels": st["asm_update_ms"] / max(st["total_ms"], 1e-9)},
           "kkt_check": {"rows": int(rows.size),
                         "max_stationarity_residual_rel": float((np.abs(np.where(free, Gr, 0)) / scale).max()),
                         "max_bound_violation": float(max((U - UB).max(), (LB - U).max(), 0.0)),
                         "wrong_sign_multipliers": int((np.where(au, -Gr, 1) <= 0).sum() + (np.where(al, Gr, 1) <= 0).sum())},
           "config": {"workload": f"cdu_offline_data, unstable family: synthetic CDU-size plant with spectral radius 1.03 (re-parameterised in the reference: "
                                  f"dense G), solved in input space as a box QP (n={n}), {B} sampled x0"}}
    qp.close(); buf.free()
    return out


def chains_leg(ctx, workload, nc, T, h=None):
    """Lock-step closed-loop chains (simulate_offline, lib/linearMPC.py:845-866) for nc chains x T steps, device resident:
    149 chains is the reference's CDU task count (cdu_parameters.py:211).  Target pairs are drawn (piecewise constant),
    not solved here: the leg times regulator solve + model step + warm-start shift per step."""
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    from industrial_nnmpc_2021_amd.chain import DeviceChains
    pl, P, tq, nu = (h["pl"], h["P"], h["tq"], h["nu"]) if h else make_problem(workload)
    Nx = pl["A"].shape[0]
    rng = np.random.default_rng(77)
    Nd = 5
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx)
    qp = BatchedBoxQP(P, tq, nu, max_batch=max(128, nc), seg_max=max(128, nc))
    ch = DeviceChains(qp, nc, pl["A"], pl["B"], Bd, pl["ulb"], pl["uub"], np.zeros(Nx), np.zeros(nu))
    hold = max(2, T // 3)
    Us = np.repeat(rng.uniform(-0.5, 0.5, ((T + hold - 1) // hold, nc, nu)), hold, axis=0)[:T]
    Xs = np.repeat(0.5 * rng.standard_normal(((T + hold - 1) // hold, nc, Nx)), hold, axis=0)[:T]
    D = np.repeat(rng.standard_normal(((T + hold - 1) // hold, nc, Nd)), hold, axis=0)[:T]
    ch.run(Xs[:2], Us[:2], D[:2])                      # warm-up
    ch.reset()
    t0 = time.perf_counter()
    rec = ch.run(Xs, Us, D, warm_start=True)
    dt = time.perf_counter() - t0
    dev_ms, solve_ms = ch.last_ms()
    out = {"value": nc * T / dt, "unit": "chain-steps/s (one regulator QP + model step each)", "chains": nc, "steps": T,
           "ms_per_step": 1e3 * dt / T, "device_ms_per_step": dev_ms / T, "solve_ms_per_step": solve_ms / T,
           "status_hist": np.bincount(rec["status"].ravel(), minlength=3).tolist(),
           "max_abs_u": float(np.abs(rec["u"]).max()),
           "note": "host wall clock around nnmpc_chain_run incl. the PCIe transfers of the inputs and the records"}
    ch.close(); qp.close()
    return out


def cdu_offline_simulator(tasks, T, seed=1, procs_per_task=1):
    """The reference's offline data-generation task at the CDU size on the synthetic plant: OfflineSimulator with a
    PRBS-like setpoint signal on the last Nz = 4 outputs (one change per ~400 steps) and a disturbance signal on Nd = 5
    channels (one per ~200 steps), `tasks` chains of T steps cut from ONE signal of tasks * T steps
    (cdu_parameters.py:115-155, :211: Nsim = 357 600 = 149 x 2400)."""
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd import linearMPC as lm
    from industrial_nnmpc_2021_amd.controller_evaluation import sample_prbs_like
    pl = synthetic.plant("cdu")
    A, Bm, Cm = pl["A"], pl["B"], pl["C"]
    Nx, Nu = Bm.shape
    Ny, Nz, Nd = Cm.shape[0], 4, 5
    rng = np.random.default_rng(77)
    H = np.concatenate((np.zeros((Nz, Ny - Nz)), np.eye(Nz)), axis=1)
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx)
    Cd = np.zeros((Ny, Nd))
    # setpoints the input box can reach: a fraction of what the steady-state gain of the controlled outputs allows
    Gz = H @ Cm @ np.linalg.solve(np.eye(Nx) - A, Bm)
    amp = float(os.environ.get("NNMPC_CHAIN_SP", "1.0")) * np.linalg.svd(Gz, compute_uv=False).min()
    damp = float(os.environ.get("NNMPC_CHAIN_D", "1.0"))
    Nsim = tasks * procs_per_task * T
    nchg = lambda mean: max(2, int(round(Nsim / mean)) - 2)
    sp = sample_prbs_like(num_change=nchg(400), num_steps=Nsim, lb=-amp * np.ones((Nz, 1)), ub=amp * np.ones((Nz, 1)),
                          mean_change=400, sigma_change=1, seed=seed)
    sp = np.concatenate((np.zeros((Nsim, Ny - Nz)), sp), axis=1)
    ds = sample_prbs_like(num_change=nchg(200), num_steps=Nsim, lb=-damp * np.ones((Nd, 1)), ub=damp * np.ones((Nd, 1)),
                          mean_change=200, sigma_change=1, seed=seed + 1)
    sim = lm.OfflineSimulator(A=A, B=Bm, C=Cm, H=H, Rs=1e-2 * np.eye(Nu), Qs=np.eye(Ny), Bd=Bd, Cd=Cd, usp=np.zeros((Nu, 1)),
                              uprev=np.zeros((Nu, 1)), Q=pl["Q"], R=pl["R"], S=pl["S"], ulb=pl["ulb"], uub=pl["uub"], N=pl["N"],
                              xprior=np.zeros((Nx, 1)), setpoints=sp, disturbances=ds, num_data_gen_task=tasks,
                              num_process_per_task=procs_per_task)
    # the regulator's GPU handle (P^-1, uploads: ~1 s) belongs to the construction of the controller objects, like the DARE and
    # the condensing above -- the reference builds its DenseQPRegulator before the simulation loop too (lib/linearMPC.py:339-395)
    sim.regulator._solver()
    return sim


def chains_task_leg(ctx, tasks=149, T=2400):
    """The reference's actual deliverable at its real length (cdu_parameters.py:211: 357 600 steps = 149 tasks x 2400): PRBS-like
    setpoints and disturbances, target selector (deduplicated, batched: the other QP of every step, lib/linearMPC.py:851)
    included, all chains in lock-step on the device, records back on the host.  At world > 1 the tasks are sharded over the
    ranks (OfflineSimulator.generate_dataset: contiguous blocks of tasks, ONE gather of the records)."""
    from industrial_nnmpc_2021_amd import linearMPC as lm
    from industrial_nnmpc_2021_amd.chain import DeviceChains
    t0 = time.perf_counter()
    sim = cdu_offline_simulator(tasks, T)
    t_setup = time.perf_counter() - t0
    # where the wall time goes: the target pairs (np.unique over all steps + the batched target QP) and nnmpc_chain_run (PCIe included)
    parts = {"target_pairs_s": 0.0, "chain_run_s": 0.0, "chain_device_s": 0.0}
    tp0, run0 = lm._target_pairs, DeviceChains.run

    def tp(*a, **k):
        t = time.perf_counter(); r = tp0(*a, **k); parts["target_pairs_s"] += time.perf_counter() - t
        return r

    def run(self, *a, **k):
        t = time.perf_counter(); r = run0(self, *a, **k); parts["chain_run_s"] += time.perf_counter() - t
        parts["chain_device_s"] += self.last_ms()[0] * 1e-3
        return r
    lm._target_pairs, DeviceChains.run = tp, run
    ctx.sync()
    t0 = time.perf_counter()
    try:
        data = sim.generate_dataset(data_filename="unused", comm=ctx.comm, write_files=False, allow_uncertified=True)
    finally:
        lm._target_pairs, DeviceChains.run = tp0, run0
    ctx.sync()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    if ctx.rank != 0:
        return None
    ts = sim.target_selectors[0]
    distinct = getattr(getattr(ts, "_batched", None), "last_distinct", None)
    return {"value": tasks * T / dt, "unit": "chain-steps/s (target QP + regulator QP + model step each)", "chains": tasks, "steps_per_chain": T,
            "samples": int(data["u"].shape[0]), "wall_s": dt, "setup_s_not_timed": t_setup, "n_gpus": ctx.world, "rank0_parts": parts,
            "device_ms_per_lockstep_step": 1e3 * parts["chain_device_s"] / max(1, T),
            "distinct_target_pairs_on_rank0": distinct,
            "status_hist": np.bincount(data["status"].ravel(), minlength=3).tolist(),
            "max_abs_u": float(np.abs(data["u"]).max()), "fraction_of_moves_on_a_bound": float((np.abs(np.abs(data["u"]) - 1.0) < 1e-12).mean()),
            "paper": "27.8 h for 3.6e5 samples on 149 cluster processes (KumarRawlingsWright2021 p.8)",
            "note": "wall clock around OfflineSimulator.generate_dataset: target pairs (np.unique + nnmpc_ts_solve_batch), nnmpc_chain_run "
                    "(state, targets, records in HBM), PCIe both ways" + (", the RCCL gather of the records" if ctx.world > 1 else "")}


def bench_nn(ctx, B, steps, warmup, with_uprev=False, modes=("f32", "bf16", "bf16x3")):
    """Config 5: structured-NN controller forward, CDU architecture [536, 832, 832, 832, 32]
    (RegulatorLayerWithoutUprev, what the reference uses for the CDU: cdu_train.py:33, :77-80; with_uprev: the 568-input
    RegulatorLayerWithUprev BASELINE.json names, lib/LinearMPCLayers.py:40-61), B states per GPU per step, f32 and bf16."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    lib = ctx.lib
    nx, nu, hid = 252, 32, 832
    dims = [2 * nx + (2 if with_uprev else 1) * nu, hid, hid, hid, nu]
    rng = np.random.default_rng(0)
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    xscale = rng.uniform(0.5, 2.0, nx)
    g = np.random.default_rng(1000 + ctx.rank)
    x_h = g.standard_normal((B, nx)); xs_h = 0.3 * g.standard_normal((B, nx)); us_h = g.uniform(-0.5, 0.5, (B, nu))
    up_h = us_h + g.uniform(-0.3, 0.3, (B, nu)) if with_uprev else None
    x_h[0] = xs_h[0]                                   # steady-state row: the structure gives u = clip(us) exactly
    if with_uprev:
        up_h[0] = us_h[0]
    D = lib.DeviceArray
    x, xs, us, u = D.from_host(x_h), D.from_host(xs_h), D.from_host(us_h), D((B, nu), np.float64)
    up = D.from_host(up_h) if with_uprev else None
    flops_per_state = 2 * 2 * sum(dims[i] * dims[i + 1] for i in range(4))   # both passes
    hidden_flops_per_state = 2 * 2 * sum(dims[i] * dims[i + 1] for i in range(3))   # the three hidden-layer GEMMs
    res = {}
    k = min(B, 4096)
    rows = np.concatenate(([0], np.sort(np.random.default_rng(9).choice(B, k - 1, replace=False)))) if B > k else np.arange(B)
    ref = onn.control_input(W, x_h[rows], up_h[rows] if with_uprev else None, xs_h[rows], us_h[rows], xscale, -np.ones(nu), np.ones(nu), with_uprev)
    for mode in modes:
        net = StructuredNN(W, nx, nu, nnwithuprev=with_uprev, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu),
                           max_batch=262144, use_bf16={"f32": False, "bf16": True, "bf16x3": "split"}[mode])
        # 6 extra untimed forwards before the W warmup steps: on every box tried, ONE forward of the first ~100 ms of
        # sustained MFMA load starts ~40 ms late (device time of that call unchanged: the stream just starts later),
        # then none for the rest of the run; with K of a few steps that one stall would be a third of the timed region
        for _ in range(6 + warmup):
            net.forward_device(B, x, up, xs, us, u)
        ctx.sync()
        t0 = time.perf_counter(); gm = dm = hm = 0.0; hl = 0
        for _ in range(steps):
            net.forward_device(B, x, up, xs, us, u)
            gm += net.last_ms()[0]; dm += net.last_ms()[1]
            hm += net.last_hidden_ms()[0]; hl += net.last_hidden_ms()[1]
        ctx.sync()
        dt = ctx.max_over_ranks(time.perf_counter() - t0)
        u_h = u.to_host()
        err = float(np.abs(u_h[rows] - ref).max() / max(1.0, np.abs(ref).max()))
        res[mode] = dict(states_per_s=ctx.world * B * steps / dt, ms_per_step=1e3 * dt / steps,
                         gemm_TFLOPs=flops_per_state * B * steps / (gm * 1e-3) / 1e12, max_rel_err_vs_fp64_oracle=err,
                         rows_checked=int(rows.size), steady_state_row_exact=bool(np.array_equal(u_h[0], np.clip(us_h[0], -1, 1))),
                         device_ms_per_step=dm / steps, gemm_ms_per_step=gm / steps,
                         hidden_TFLOPs=hidden_flops_per_state * B * steps / (hm * 1e-3) / 1e12,
                         hidden_launches=hl, hidden_avg_launch_ms=hm / max(1, hl))
        net.close()
    for a in (x, xs, us, u, up):
        if a is not None:
            a.free()
    if with_uprev:                                     # the variant entry of the main line: rates and errors only
        return {"dims": dims, **{m: {k: res[m][k] for k in ("states_per_s", "ms_per_step", "max_rel_err_vs_fp64_oracle", "steady_state_row_exact",
                                                              "hidden_TFLOPs", "rows_checked")} for m in modes}}
    f, h, s3 = res["f32"], res["bf16"], res["bf16x3"]
    traffic, tnote = pmc_traffic(f"nn_b{B}")
    rows2 = 2 * min(B, 262144)
    kavg = (((dims[0] + 63) // 64) * 64 + 2 * hid) / 3.0
    out = {"metric": "structured-NN forward states/sec (CDU architecture)", "value": f["states_per_s"], "unit": "states/s",
           "ms_per_step": f["ms_per_step"], "dtype": "f32",
           "config": {"workload": f"cdu_neural_network: RegulatorLayerWithoutUprev {dims}, {B} states per GPU per step",
                      "flops_per_state": flops_per_state},
           "roofline": {"kernel": "gemm_nt_f32_k (128 x 128 tiles, v_mfma_f32_32x32x2_f32, bias + ReLU fused)", "dtype": "f32", "bound": "mfma",
                        "achieved": f["hidden_TFLOPs"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f["hidden_TFLOPs"] / FP32_PEAK_TFLOPS,
                        "traffic": _traffic_of(traffic, ["gemm_nt_f32_k<128, true, true>"])[0], "traffic_unit": "HBM bytes per launch of the 128-wide instance (hidden layers); " + tnote,
                        "launches": f["hidden_launches"], "avg_launch_ms": f["hidden_avg_launch_ms"],
                        "algorithmic_bytes_per_launch": rows2 * (896 + 896) * 4,
                        "algorithmic_flops": "2 passes x 2 x sum(d_in d_out) per state, unpadded (SURVEY 8d)"},
           "parity": {"max_rel_err_vs_fp64_oracle": f["max_rel_err_vs_fp64_oracle"], "rows_checked": f["rows_checked"],
                      "steady_state_row_exact": f["steady_state_row_exact"], "tolerance": 1e-4},
           "f32": f,
           "bf16": dict(h, tolerance=3e-2,
                        roofline={"kernel": "gemm_nt_bf16_wide_k (256 x 208 tiles, v_mfma_f32_16x16x32_bf16, persistent workgroups)", "dtype": "bf16",
                                  "bound": "mfma", "achieved": h["hidden_TFLOPs"], "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": h["hidden_TFLOPs"] / BF16_PEAK_TFLOPS, "traffic": _traffic_of(traffic, ["gemm_nt_bf16_wide_k<true, true, false>"])[0],
                                  "traffic_unit": "HBM bytes per launch; " + tnote,
                                  "launches": h["hidden_launches"], "avg_launch_ms": h["hidden_avg_launch_ms"],
                                  "algorithmic_flops": "2 passes x 2 x (d_in h + 2 h^2) per state over the three hidden-layer launches "
                                                       "(gemm_TFLOPs: all four GEMMs incl. the HBM-bound head)",
                                  "algorithmic_bytes_per_launch": rows2 * (kavg + hid) * 2}),
           # split bf16 (use_bf16 = 2): activations and weights as bf16 pairs hi + lo, one bf16 GEMM of three times the depth per
           # layer; ALGORITHMIC flops (those of the f32 path) over its time -- the matrix pipes do three times as many
           "bf16x3": dict(s3, tolerance=1e-4,
                          note="f32-grade results from the bf16 matrix pipes: hi hi' + hi lo' + lo hi' per layer (gemm_nt_bf16_wide_k, "
                               "K three times as deep); hidden_TFLOPs counts the algorithmic flops, the pipes execute 3 x that",
                          executed_hidden_TFLOPs=3.0 * s3["hidden_TFLOPs"], frac_of_bf16_peak_executed=3.0 * s3["hidden_TFLOPs"] / BF16_PEAK_TFLOPS,
                          speedup_over_f32=s3["states_per_s"] / f["states_per_s"])}
    # BASELINE.json config 5 as literally written: RegulatorLayerWithUprev (568 inputs) at the same batch, f32 and bf16
    out["with_uprev_568_inputs"] = bench_nn(ctx, B, max(2, steps // 2), 1, with_uprev=True, modes=("f32", "bf16"))
    return out


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cdu", choices=["cdu", "cstrs", "nn", "chains"])
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU per step")
    ap.add_argument("--slots", type=int, default=0, help="resident problems per wave (PDIP path)")
    ap.add_argument("--sx", type=float, default=2.0, help="state spread of the synthetic samples")
    ap.add_argument("--cpu-baseline", default="bounded", choices=["bounded", "full", "none"],
                    help="full: also 1 thread and nproc processes on >= 4 CDU problems (minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--method", default="auto", choices=["auto", "pdip", "asm"],
                    help="auto: shared-inverse active-set pass + PDIP for what it leaves; pdip: PDIP path only")
    ap.add_argument("--no-pdip", action="store_true", help="skip the extra PDIP-path measurement")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive measurement (host buffers either side)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no configs / sweep / chains legs")
    ap.add_argument("--pdip-batch", type=int, default=0)
    ap.add_argument("--parity-rows", type=int, default=32)
    ap.add_argument("--chain-steps", type=int, default=0, help="--workload chains: simulation steps per chain (2400 = the reference's task length)")
    args = ap.parse_args()
    if args.no_cpu_baseline:
        args.cpu_baseline = "none"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))

    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.workload in ("cdu", "cstrs") and not (args.no_parity and args.cpu_baseline == "none"):
        start_workers(min(16, max(1, (os.cpu_count() or 1) // 4)))     # forked BEFORE the first GPU call (see WORKERS)
    ctx = Ctx(args)
    rank, world = ctx.rank, ctx.world
    common = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "data": "synthetic"}

    if args.workload == "nn":
        out = bench_nn(ctx, args.batch or (1 << 20), args.steps, args.warmup)
        if rank == 0:
            emit(dict(out, **common))
        return
    if args.workload == "chains":
        # the reference's task layout: `--batch` chains (default 149 = cdu_parameters.py:211) x `--chain-steps` steps each
        # (default: --steps, at least 8; 2400 = the reference's length), sharded over the ranks by contiguous blocks of tasks
        out = chains_task_leg(ctx, args.batch or 149, args.chain_steps or max(args.steps, 8))
        if rank == 0:
            line = {"metric": "closed-loop offline data generation: chain steps/sec (CDU size; target QP + regulator QP + model step)"}
            line.update(common)
            line.update(out)
            line["scaling"] = "strong"                        # the task list is fixed, the ranks share it
            line["config"] = {"workload": f"cdu_offline_data (closed-loop chains): {line.get('chains')} tasks x {line.get('steps_per_chain')} steps, target selector included"}
            line["dtype"] = "f64"
            emit(line)
        return

    B = args.batch or default_batch(args.workload, world)
    extras = world == 1 and not args.no_extras
    res, h = bench_qp(ctx, args.workload, B, args.steps, args.warmup, args.sx, method=args.method, slots=args.slots,
                      want_buffers=(rank == 0 and world == 1))
    if rank != 0:
        return
    n, n_aug = (h["n"], h["tq"].shape[1]) if h else (None, None)
    out = {"metric": "condensed-QP solves/sec (CDU offline datagen)" if args.workload == "cdu"
                     else "condensed-QP solves/sec (CSTRs offline datagen)"}
    out.update(res)
    out.update(common)
    wl = args.workload
    out["config"] = {"workload": f"{wl}_offline_data: synthetic {wl.upper()}-size plant" + (f" (n={n} vars, m={2 * n} box rows, n_aug={n_aug})" if n else "")
                                 + f", {B} sampled x0 per GPU per step, whole sequences u* written out",
                     "batch_per_gpu": B, "total_batch": B * world, "sx": args.sx, "method": args.method,
                     "parallelism": f"dp{world} (sharded samples, 1 RCCL gather of the first moves per step, nnmpc_comm_gather_rows)"}
    if world == 1 and h is not None:
        legs = {}

        def timed(name, fn, *a):
            t0 = time.perf_counter()
            r = fn(*a)
            legs[name] = round(time.perf_counter() - t0, 1)
            return r
        if not args.no_parity:
            out["parity"] = timed("parity", parity_leg, h, args.parity_rows, 2000)
        if args.method == "auto" and not args.no_pdip:       # (compares with the headline's outputs: before they are overwritten)
            out["pdip_path"] = timed("pdip_path", pdip_leg, ctx, h, min(B, args.pdip_batch or (1024 if wl == "cdu" else 8192)))
        if not args.no_host_io:
            out["host_io"] = timed("host_io", host_io_leg, ctx, h)
        if extras:
            out["first_move_output"] = timed("first_move_output", first_move_leg, ctx, h, args.steps)
            if wl == "cdu":
                out["sweep_sx"] = timed("sweep_sx", sweep_leg, ctx, h, [1.0, 2.0, 3.0, 4.0, 6.0], 2)
        host = h["host"]
        P, tq, nu, N = h["P"], h["tq"], h["nu"], h["N"]
        h["qp"].close(); h["buf"].free()
        if extras and wl == "cdu":
            out["chains"] = timed("chains", chains_leg, ctx, "cdu", 149, 12, dict(pl=h["pl"], P=P, tq=tq, nu=nu))
            out["chains_task"] = timed("chains_task", chains_task_leg, ctx, 149, 2400)
            cfg = {}
            t_cfg = time.perf_counter()
            # (three handles, the fastest is reported, all three are listed: a 1.4 ms step shows every host-side stall -- see the
            # note on the BLAS pools at the top of this file, the one cause found so far)
            runs, r2, h2 = [], None, None
            for _ in range(3):
                rr, hh = bench_qp(ctx, "cstrs", 10000, max(args.steps, 5), 2, args.sx, want_buffers=True)
                runs.append(rr["ms_per_step"])
                if r2 is None or rr["ms_per_step"] < r2["ms_per_step"]:
                    if h2 is not None:
                        h2["qp"].close(); h2["buf"].free()
                    r2, h2 = rr, hh
                else:
                    hh["qp"].close(); hh["buf"].free()
            r2["ms_per_step_of_each_handle"] = runs
            r2["config"] = {"workload": "cstrs_offline_data: synthetic CSTRs-size plant (n=540, m=1080, cond(P) = 4e7), 10000 sampled x0, 1 GPU"}
            if not args.no_parity:
                r2["parity"] = parity_leg(h2, 32, 2000)
            if args.cpu_baseline != "none":
                x2, l2, u2, _ = h2["host"]
                r2["cpu_baseline"] = cpu_baseline(np.tril(h2["P"]) + np.tril(h2["P"], -1).T, h2["tq"], h2["nu"], h2["N"], x2[:1024], l2[:1024], u2[:1024],
                                                  budget_s=4.0, workload="cstrs", full=False)
            h2["qp"].close(); h2["buf"].free()
            cfg["cstrs_10k"] = r2
            legs["cstrs_10k"] = round(time.perf_counter() - t_cfg, 1)
            cfg["cdu_unstable"] = timed("cdu_unstable", unstable_leg, ctx)
            cfg["nn_1m"] = timed("nn_1m", bench_nn, ctx, 1 << 20, 5, 1)
            out["configs"] = cfg
        if args.cpu_baseline != "none":
            out["cpu_baseline"] = timed("cpu_baseline", cpu_baseline, np.tril(P) + np.tril(P, -1).T, tq, nu, N, host[0][:64], host[1][:64], host[2][:64],
                                        20.0 if wl == "cdu" else 10.0, wl, args.cpu_baseline == "full")
        out["leg_seconds"] = legs
    stop_workers()
    emit(out)


if __name__ == "__main__":
    main()
