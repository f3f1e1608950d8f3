#!/usr/bin/env python3
"""Headline benchmark: condensed-QP solves/sec of the CDU offline-datagen hot path.

    python bench.py --gpus N --steps K --warmup W [--workload cdu|cstrs] [--batch B]

One "step" = one pass of the hot path (q = tq x0 -> batched PDIP + polish ->
first moves) over one batch of B synthetic CDU-size problems per GPU
(Nx=252, Nu=32, N=140 -> n=4480 variables, m=8960 box rows; reference sizes
cdu_parameters.py:99-102), inputs already resident in HBM.  N > 1: one process
per GPU (torch.distributed, backend nccl = RCCL), the sample batch is sharded
(weak scaling: B per GPU), no collective during the solves and ONE gather of
the first moves over xGMI at the end of every step.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # MI355X dense f32 matrix/vector peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6   # f64 vector = matrix peak (half the f32 rate; v_mfma_f64_16x16x4_f64)


def cpu_baseline(P, tq, nu, N, x0, lb, ub, budget_s=30.0, workload="cdu"):
    """Restated reference CPU path (oracle.qp.coneqp_l: cvxopt-style dense-G PDIP,
    fp64, one problem at a time) timed on this host.  Bounded sample."""
    from oracle import qp as oqp
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    n = P.shape[0]
    done, t0, its = 0, time.time(), []
    for b in range(x0.shape[0]):
        G, h = oqp.box_as_Gh(nu, N, lb[b], ub[b])
        info = {}
        oqp.coneqp_l(P, tq @ x0[b], G, h, info=info)
        its.append(info["iterations"])
        done += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return {"value": done / dt, "unit": "solves/s", "cores": int(threads), "kind": "port",
            "sample": f"{done} problem(s) of the same seeded batch, n={n}, m={2 * n}, dense G, "
                      f"cvxopt-default tolerances, mean {np.mean(its):.1f} PDIP iterations, {dt:.1f} s",
            "paper_reference": ("CVXOPT 35 s/solve mean, 47 s worst on a 2.4 GHz cluster CPU (KumarRawlingsWright2021 p.9) = 0.029 solves/s"
                                if workload == "cdu" else
                                "CVXOPT 8-13 s/solve on a 2.4 GHz cluster CPU at the paper's N=450 (n=2700; the code ships N=90, n=540) (KumarRawlingsWright2021 p.7)")}


def bench_nn(args, torch, dev, rank, world, dist):
    """Config 5: structured-NN controller forward, CDU architecture [536, 832, 832, 832, 32]
    (RegulatorLayerWithoutUprev, cdu_train.py:33, :77-80), B states per GPU per step, f32 and bf16."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    nx, nu, hid = 252, 32, 832
    dims = [2 * nx + nu, hid, hid, hid, nu]
    rng = np.random.default_rng(0)
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    B = args.batch or (1 << 20)
    xscale = rng.uniform(0.5, 2.0, nx)
    g = torch.Generator(device=dev); g.manual_seed(1000 + rank)
    x = torch.randn((B, nx), dtype=torch.float64, device=dev, generator=g)
    xs = 0.3 * torch.randn((B, nx), dtype=torch.float64, device=dev, generator=g)
    us = torch.rand((B, nu), dtype=torch.float64, device=dev, generator=g) - 0.5
    u = torch.empty((B, nu), dtype=torch.float64, device=dev)
    flops_per_state = 2 * 2 * sum(dims[i] * dims[i + 1] for i in range(4))   # both passes
    hidden_flops_per_state = 2 * 2 * sum(dims[i] * dims[i + 1] for i in range(3))   # the three hidden-layer GEMMs
    res = {}
    for mode in ("f32", "bf16"):
        net = StructuredNN(W, nx, nu, nnwithuprev=False, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu),
                           max_batch=262144, use_bf16=(mode == "bf16"))
        # 6 extra untimed forwards before the W warmup steps: on every box tried, ONE forward of the first ~100 ms of
        # sustained MFMA load starts ~40 ms late (device time of that call unchanged: the stream just starts later),
        # then none for the rest of the run; with K of a few steps that one stall would be a third of the timed region
        for _ in range(6 + args.warmup):
            net.forward_device(B, x, None, xs, us, u)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter(); gm = 0.0; dm = 0.0; hm = 0.0; hl = 0
        for _ in range(args.steps):
            net.forward_device(B, x, None, xs, us, u)
            gm += net.last_ms()[0]; dm += net.last_ms()[1]
            hm += net.last_hidden_ms()[0]; hl += net.last_hidden_ms()[1]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        k = 512
        ref = onn.control_input(W, x[:k].cpu().numpy(), None, xs[:k].cpu().numpy(), us[:k].cpu().numpy(), xscale,
                                -np.ones(nu), np.ones(nu), False)
        err = float(np.abs(u[:k].cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
        res[mode] = dict(states_per_s=world * B * args.steps / dt, ms_per_step=1e3 * dt / args.steps,
                         gemm_TFLOPs=flops_per_state * B * args.steps / (gm * 1e-3) / 1e12, max_rel_err_vs_fp64_oracle=err,
                         device_ms_per_step=dm / args.steps, gemm_ms_per_step=gm / args.steps,
                         hidden_TFLOPs=hidden_flops_per_state * B * args.steps / (hm * 1e-3) / 1e12,
                         hidden_launches=hl, hidden_avg_launch_ms=hm / max(1, hl))
        net.close()
    if rank == 0:
        f, h = res["f32"], res["bf16"]
        # HBM bytes per launch of the hidden-layer GEMMs from the PMC passes of profiles/r01i_pmc_nn.json (same batch,
        # 262144 states per launch); algorithmic bytes = rows x (K + N) x element size, averaged over the 3 layers
        pmc = os.path.join(ROOT, "profiles", "r01i_pmc_nn.json")
        tr = {}
        if os.path.exists(pmc) and B >= 262144:
            for k, v in json.load(open(pmc))["kernels"].items():
                tr[k.split("::")[-1].split("<")[0]] = v["hbm_bytes_per_launch"]
        rows = 2 * min(B, 262144)
        kavg = (((dims[0] + 63) // 64) * 64 + 2 * hid) / 3.0
        out = {"metric": "structured-NN forward states/sec (CDU architecture)", "value": f["states_per_s"], "unit": "states/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": f["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"cdu_neural_network: RegulatorLayerWithoutUprev {dims}, {B} states per GPU per step",
                          "flops_per_state": flops_per_state},
               "roofline": {"kernel": "gemm_nt_f32_k (128 x 128 tiles, v_mfma_f32_32x32x2_f32, bias + ReLU fused)", "bound": "mfma",
                            "achieved": f["hidden_TFLOPs"], "peak": FP32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": f["hidden_TFLOPs"] / FP32_PEAK_TFLOPS, "traffic": tr.get("gemm_nt_f32_k"),
                            "launches": f["hidden_launches"], "avg_launch_ms": f["hidden_avg_launch_ms"],
                            "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01i_pmc_nn.json)",
                            "algorithmic_bytes_per_launch": rows * (896 + 896) * 4,
                            "algorithmic_flops": "2 passes x 2 x sum(d_in d_out) per state, unpadded (SURVEY 8d)"},
               "parity": {"max_rel_err_vs_fp64_oracle": f["max_rel_err_vs_fp64_oracle"]},
               "bf16": dict(h, roofline={"kernel": "gemm_nt_bf16_wide_k (256 x 208 tiles, v_mfma_f32_16x16x32_bf16, persistent workgroups)",
                                         "bound": "mfma", "achieved": h["hidden_TFLOPs"], "peak": 2500.0, "unit": "TFLOP/s",
                                         "frac": h["hidden_TFLOPs"] / 2500.0, "traffic": tr.get("gemm_nt_bf16_wide_k"),
                                         "launches": h["hidden_launches"], "avg_launch_ms": h["hidden_avg_launch_ms"],
                                         "algorithmic_flops": "2 passes x 2 x (d_in h + 2 h^2) per state over the three hidden-layer launches (gemm_TFLOPs: all four GEMMs incl. the HBM-bound head)",
                                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01i_pmc_nn.json)",
                                         "algorithmic_bytes_per_launch": rows * (kavg + hid) * 2})}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cdu", choices=["cdu", "cstrs", "nn"])
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU per step")
    ap.add_argument("--slots", type=int, default=0, help="resident problems per wave")
    ap.add_argument("--sx", type=float, default=2.0, help="state spread of the synthetic samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--method", default="auto", choices=["auto", "pdip", "asm"],
                    help="auto: shared-inverse active-set pass + PDIP for what it leaves; pdip: PDIP path only")
    ap.add_argument("--no-pdip", action="store_true", help="skip the extra PDIP-path measurement")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive measurement (host buffers either side)")
    ap.add_argument("--pdip-batch", type=int, default=0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # NNMPC_FORCE_DIST=1: run the process-group code (RCCL init, gather, max-reduce) with a single rank too
    use_dist = world > 1 or os.environ.get("NNMPC_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "nn":
        return bench_nn(args, torch, dev, rank, world, dist)
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP

    B = args.batch or (100000 if args.workload == "cdu" else 131072)   # cdu: BASELINE.json configs[2], "100k sampled x0, 1 MI355X"
    slots = args.slots or (1024 if args.workload == "cdu" else 8192)
    pl = synthetic.plant(args.workload, seed=0)
    P, tq, nu = build_regulator_matrices(pl)
    n, n_aug, N = P.shape[0], tq.shape[1], pl["N"]
    qp = BatchedBoxQP(P, tq, nu, max_batch=min(slots, B), method=args.method)

    # every rank draws its own shard of the seeded sample stream
    s = synthetic.samples(pl, B, seed=1000 + rank, sx=args.sx)
    x0_h = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    lb_h, ub_h = pl["ulb"].T - s["us"], pl["uub"].T - s["us"]
    x0 = torch.from_numpy(x0_h).to(dev)
    lb = torch.from_numpy(np.ascontiguousarray(lb_h)).to(dev)
    ub = torch.from_numpy(np.ascontiguousarray(ub_h)).to(dev)
    us = torch.from_numpy(s["us"]).to(dev)
    u = torch.empty((B, n), dtype=torch.float64, device=dev)
    act = torch.empty((B, qp.words), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    iters = torch.empty((B, 2), dtype=torch.int32, device=dev)
    gathered = [torch.empty((B, nu), dtype=torch.float64, device=dev) for _ in range(world)] if rank == 0 else None

    def step():
        qp.solve_batch_device(B, x0, lb, ub, u, act, status, iters)
        first = (u[:, :nu] + us).contiguous()      # get_control_sequence adds us back (:689); ut = useq[0:Nu] (:856)
        if dist is not None:
            dist.gather(first, gathered, dst=0)     # the single RCCL gather over xGMI
        return first

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    qp.set_profiling(True)
    qp.stats(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = qp.stats()
    status_h = status.cpu().numpy()
    iters_h = iters.cpu().numpy()
    qp.set_profiling(False)

    def panel_roofline(stx):
        fl, ms = stx["panel_flops"], stx["panel_ms"]
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM traffic per launch: bytes per algorithmic flop measured with rocprofv3 PMC passes
        # (FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied; profiles/r01_pmc_chol_panel.json)
        # scaled to this run's flops per launch; null when that profile does not exist.
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_chol_panel.json")
        if args.workload == "cdu" and os.path.exists(pmc) and stx["panel_launches"]:
            traffic = json.load(open(pmc))["hbm_bytes_per_algorithmic_flop"] * fl / stx["panel_launches"]
        return {"kernel": "chol_panel_k", "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC-derived, see profiles/r01_pmc_chol_panel.json)",
                "launches": stx["panel_launches"], "avg_launch_ms": ms / max(1, stx["panel_launches"]),
                "time_share": {"chol_panel": ms / stx["total_ms"], "chol_diag": stx["diag_ms"] / stx["total_ms"],
                               "trsv": stx["trsv_ms"] / stx["total_ms"]},
                # whole-solve rate in the survey's dense-PDIP flop model: factorisations * n^3/3
                "cholesky_flops_over_solve_time_TFLOPs": stx["factorizations"] * n ** 3 / 3 / (stx["total_ms"] * 1e-3) / 1e12}

    if rank == 0:
        out = {
            "metric": "condensed-QP solves/sec (CDU offline datagen)" if args.workload == "cdu"
                      else "condensed-QP solves/sec (CSTRs offline datagen)",
            "value": world * B * args.steps / dt, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if st["asm_solved"] else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}_offline_data: synthetic {args.workload.upper()}-size plant "
                                   f"(n={n} vars, m={2 * n} box rows, n_aug={n_aug}), {B} sampled x0 per GPU per step",
                       "batch_per_gpu": B, "sx": args.sx, "method": args.method,
                       "parallelism": f"dp{world} (sharded samples, 1 RCCL gather/step)"},
            "solver": {"status_hist": np.bincount(status_h, minlength=3).tolist(),
                       "solved_by_active_set_pass": int(st["asm_solved"]), "active_set_rounds_per_step": st["asm_rounds"] / args.steps,
                       "solved_by_pdip_path": int(st["problems"] - st["asm_solved"]),
                       "mean_pdip_iters": float(iters_h[:, 0].mean()), "mean_factorizations": float(iters_h[:, 1].mean())},
        }
        if st["asm_solved"]:
            # shared-inverse active-set pass: two kernels carry the time; the one with the larger
            # hipEvent share is reported as `roofline`, the other as `roofline_secondary`
            # mean HBM bytes per launch of the kernels from the PMC passes of profiles/r01j_pmc_asm.json (scripts/pmc_hbm.py,
            # same command, B = 100000); null at other batch sizes
            pmc_asm, pmc_file = {}, os.path.join(ROOT, "profiles", "r01j_pmc_asm.json")
            if args.workload == "cdu" and B == 100000 and os.path.exists(pmc_file):
                for k, v in json.load(open(pmc_file))["kernels"].items():
                    pmc_asm[k.split("::")[-1].split("<")[0]] = v["hbm_bytes_per_launch"]
            gach = st["asm_gemm_flops"] / (st["asm_gemm_ms"] * 1e-3) / 1e12
            gemm = {"kernel": "gemm_nt_f64_128_k (x_unc = x0 Kunc', XH = LAM Pinv inside the column window, one full-width pass; "
                              "the f32 rounds' XH32 = LAM32 Pinv32 on gemm_nt_f32_kdyn_k is in the same time and flop count)",
                    "bound": "mfma", "achieved": gach, "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": gach / FP64_PEAK_TFLOPS,
                    "traffic": pmc_asm.get("gemm_nt_f64_128_k"),
                    "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01j_pmc_asm.json)",
                    "launches": st["asm_gemm_launches"], "avg_launch_ms": st["asm_gemm_ms"] / max(1, st["asm_gemm_launches"]),
                    "algorithmic_flops": "2 * (columns evaluated) * (own last active bound + 1) per running problem and round (columns = window past the round's last active bound) + x_unc = x0 Kunc' + one full-width pass per problem",
                    "time_share": st["asm_gemm_ms"] / st["total_ms"]}
            lach = st["asm_lambda_flops"] / (st["asm_lambda_ms"] * 1e-3) / 1e12
            lam = {"kernel": "asm_lambda_reg32_k + asm_lambda_reg_k (|A|x|A| Cholesky + solves of the multiplier systems: one "
                             "wave per problem, tiles in the MFMA accumulators; f32 rounds until the set settles, then fp64)",
                   "bound": "mfma", "achieved": lach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": lach / FP64_PEAK_TFLOPS,
                   # the gathered Pinv blocks (algorithmic_GBps below) come from L2 / Infinity Cache, not from HBM
                   "traffic": pmc_asm.get("asm_lambda_reg32_k"),
                   "traffic_fp64_kernel": pmc_asm.get("asm_lambda_reg_k"),
                   "traffic_unit": "HBM bytes per launch of asm_lambda_reg32_k / asm_lambda_reg_k (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01j_pmc_asm.json)",
                   "algorithmic_flops": "m^3/3 + 2 m^2 per problem and round, m = size of its active set (f32 and fp64 rounds "
                                        "alike; priced against the fp64 MFMA peak)",
                   "algorithmic_GBps": st["asm_lambda_bytes"] / (st["asm_lambda_ms"] * 1e-3) / 1e9,
                   "launches": st["asm_rounds"], "avg_launch_ms": st["asm_lambda_ms"] / max(1, st["asm_rounds"]),
                   "time_share": st["asm_lambda_ms"] / st["total_ms"],
                   "note": "latency-bound, not throughput-bound: per problem a chain of |A| dependent pivot steps "
                           "(16 x 16 diagonal tiles on the VALU) between the MFMA tile updates; 1024 problems in flight"}
            first, second = (lam, gemm) if st["asm_lambda_ms"] >= st["asm_gemm_ms"] else (gemm, lam)
            out["roofline"] = first
            out["roofline_secondary"] = second
            out["roofline"]["other_time_share"] = {"set_bookkeeping_kernels": st["asm_update_ms"] / st["total_ms"]}
            out["solver"]["checked_with_P_itself"] = int(st["asm_full_checks"])
            out["solver"]["inverse_check"] = {"max_abs_P_Pinv_minus_I": st["asm_e2max"], "max_abs_P_Kunc_plus_tq": st["asm_e1max"]}
        else:
            out["roofline"] = panel_roofline(st)
        # ---- parity spot check against the fp64 oracle on the first problems of the batch
        if not args.no_parity:
            from oracle import qp as oqp
            k = 2 if args.workload == "cdu" else 8
            u_h = u[:k].cpu().numpy()
            a_h = act[:k].cpu().numpy().view(np.uint32)
            errs, ham = [], 0
            Ps = np.tril(P) + np.tril(P, -1).T
            for b in range(k):
                info = {"nu": nu}
                xe = oqp.solve_exact_box(Ps, tq @ x0_h[b], np.tile(lb_h[b], N), np.tile(ub_h[b], N), info=info)
                rows = np.zeros(2 * n, bool)
                rows[info["active"]] = True
                bits = np.unpackbits(a_h[b].view(np.uint8), bitorder="little")[:2 * n].astype(bool)
                errs.append(float(np.abs(u_h[b] - xe).max() / max(1.0, np.abs(xe).max())))
                ham += int((bits != rows).sum())
            out["parity"] = {"checked": k, "max_rel_err_vs_fp64_oracle": max(errs), "active_set_hamming": ham}
        # ---- the PDIP path (method="pdip") on the head of the same batch, with its own roofline
        if args.method == "auto" and not args.no_pdip:
            Bp = min(B, args.pdip_batch or (2048 if args.workload == "cdu" else 16384))
            qp2 = BatchedBoxQP(P, tq, nu, max_batch=min(slots, Bp), method="pdip")
            u2 = torch.empty((Bp, n), dtype=torch.float64, device=dev)
            act2 = torch.empty((Bp, qp.words), dtype=torch.int32, device=dev)
            st2 = torch.empty((Bp,), dtype=torch.int32, device=dev)
            it2 = torch.empty((Bp, 2), dtype=torch.int32, device=dev)
            qp2.solve_batch_device(min(Bp, 256), x0, lb, ub, u2, act2, st2, it2)      # warm-up
            qp2.set_profiling(True); qp2.stats(reset=True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            qp2.solve_batch_device(Bp, x0, lb, ub, u2, act2, st2, it2)
            torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
            s2 = qp2.stats()
            it2h = it2.cpu().numpy()
            out["pdip_path"] = {"value": Bp / dt2, "unit": "solves/s", "batch": Bp, "dtype": "f32 (+f64 refinement)",
                                "status_hist": np.bincount(st2.cpu().numpy(), minlength=3).tolist(),
                                "mean_pdip_iters": float(it2h[:, 0].mean()), "mean_factorizations": float(it2h[:, 1].mean()),
                                "max_abs_diff_vs_active_set_pass": float((u2 - u[:Bp]).abs().max()),
                                "active_sets_equal": bool(torch.equal(act2, act[:Bp])),
                                "roofline": panel_roofline(s2)}
            qp2.close()
        # ---- the same batch with host buffers either side (never `value`): pinned host tensors -> HBM over PCIe, the
        # solve, first moves (what simulate_offline keeps, lib/linearMPC.py:856) or full sequences back to the host
        if world == 1 and not args.no_host_io:
            qp.set_profiling(False)
            xh, lbh, ubh = (t.cpu().pin_memory() for t in (x0, lb, ub))
            fh = torch.empty((B, nu), dtype=torch.float64).pin_memory()
            uh = torch.empty((B, n), dtype=torch.float64).pin_memory()
            hio = {}
            for name, full in (("first_move", False), ("full_sequence", True)):
                best = None
                for _ in range(2):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    x0.copy_(xh, non_blocking=True); lb.copy_(lbh, non_blocking=True); ub.copy_(ubh, non_blocking=True)
                    torch.cuda.synchronize()
                    qp.solve_batch_device(B, x0, lb, ub, u, act, status, iters)
                    if full:
                        uh.copy_(u, non_blocking=True)
                    else:
                        fh.copy_(u[:, :nu] + us, non_blocking=True)
                    torch.cuda.synchronize(); d = time.perf_counter() - t1
                    best = d if best is None else min(best, d)
                hio[name + "_solves_per_s"] = B / best
                hio[name + "_ms"] = 1e3 * best
            hio["bytes_in"] = int(xh.numel() + lbh.numel() + ubh.numel()) * 8
            hio["bytes_out_first_move"] = B * nu * 8
            hio["bytes_out_full_sequence"] = B * n * 8
            hio["note"] = "PCIe-inclusive: pinned host buffers -> HBM, solve, results -> pinned host buffers; best of 2"
            out["host_io"] = hio
            del uh, fh
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np.tril(P) + np.tril(P, -1).T, tq, nu, N, x0_h, lb_h, ub_h,
                                               budget_s=20.0 if args.workload == "cdu" else 10.0, workload=args.workload)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
