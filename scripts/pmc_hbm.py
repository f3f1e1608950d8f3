"""HBM bytes per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) + a kernel trace.

usage: pmc_hbm.py <dir with fetch/ and write/ subdirs of rocprofv3 csv output> <out.json> "<command string>"
Corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide streaming reads -> x2 (uncalibrated for 8-byte gathers:
treat the gather kernels' figure as an upper bound of the same order).
"""
import csv, glob, json, sys, collections

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        n = r["Kernel_Name"].split("(")[0]
        acc[n][0] += 1; acc[n][1] += float(r["Counter_Value"])
    t = collections.defaultdict(float)
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            t[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return acc, t

root, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_sha: the profile is only quoted by bench.py for the kernel sources it was taken on)
fe, tf = load(root + "/fetch", "FETCH_SIZE")
wr, tw = load(root + "/write", "WRITE_SIZE")
def _steps(c):
    """solves per kernel list in the trace: --steps + --warmup of the bench command (defaults 3 + 1)"""
    a = c.split()
    g = lambda k, d: int(a[a.index(k) + 1]) if k in a else d
    return g("--steps", 3) + g("--warmup", 1) + 1       # + the untimed setup pass of bench_qp (dense form of the full-width pass)

res = {"command": cmd, "csrc_sha": bench.csrc_sha(), "steps_profiled": _steps(cmd), "correction": "FETCH_SIZE x2 x1024, WRITE_SIZE x1024 (MI355X_MICROARCH.md, HBM)", "kernels": {}}
for n, (calls, v) in sorted(fe.items(), key=lambda kv: -kv[1][1]):
    if "nnmpc" not in n: continue
    fb = v * 2 * 1024; wb = wr.get(n, [0, 0.0])[1] * 1024; ms = tf.get(n, 0.0)
    res["kernels"][n] = {"launches": calls, "fetch_bytes_corrected": fb, "write_bytes": wb, "time_ms_under_pmc": ms,
                         "hbm_bytes_per_launch": (fb + wb) / max(1, calls), "hbm_TBps": (fb + wb) / (ms * 1e-3) / 1e12 if ms else None}
def steady_state(res):
    """HBM bytes of a step once the far-field factors are there: one of the profiled passes is the handle's setup pass (dense form
    of the full-width pass: asm_wide_gemm_k<1>, far_verify_k), the far-field kernels ran in the others only."""
    sp = res["steps_profiled"]
    setup_only = ("asm_wide_gemm_k<1>", "far_verify_k", "gemm_nt_f64_k")
    far_only = ("asm_wide_gemm_k<2>", "asm_wide_t_k", "asm_wide_tnorm_k")
    tot = 0.0
    for n, k in res["kernels"].items():
        b = k["fetch_bytes_corrected"] + k["write_bytes"]
        if any(t in n for t in setup_only): continue
        tot += b / (sp - 1) if any(t in n for t in far_only) else b / sp
    return tot
if any("asm_wide_gemm_k<1>" in n for n in res["kernels"]) and res["steps_profiled"] > 1:
    res["steady_state_bytes_per_step"] = steady_state(res)
json.dump(res, open(out, "w"), indent=1)
for n, k in res["kernels"].items(): print(n, k["launches"], f"{k['hbm_bytes_per_launch'] / 1e6:.1f} MB/launch", k["hbm_TBps"])
