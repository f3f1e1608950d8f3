"""A/B of the first-set predictor (qp_predict.h) on the bench's CDU batch: ms per step, rounds, multiplier-kernel time by
NNMPC_PRED_ITERS (GPU box).  usage: pred_probe.py <B> <sx> <iters,iters,...> [trace]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import bench
    from industrial_nnmpc_2021_amd import _lib
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    B, sx = int(sys.argv[2]), float(sys.argv[3])
    pl, P, tq, nu = bench.make_problem("cdu")
    n = P.shape[0]
    qp = BatchedBoxQP(P, tq, nu, max_batch=1024)
    qp.prepare_farfield_windows()
    sets = []
    for i in range(4):
        ib = bench.QpInputs(_lib, qp, B, nu); ib.upload(*bench.make_samples(pl, B, 1000 + 7919 * i, sx)); sets.append(ib)
    buf = bench.QpBuffers(_lib, qp, B, nu, n, inputs=sets[0])
    for ib in sets[:2]:
        buf.use(ib); qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
    qp.set_profiling(True); qp.stats(reset=True)
    _lib.synchronize(); t0 = time.perf_counter()
    for ib in sets[2:]:
        buf.use(ib); qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
    _lib.synchronize(); dt = (time.perf_counter() - t0) / 2
    st = qp.stats()
    nact = np.unpackbits(buf.act.to_host(4096).view(np.uint8), axis=1).sum(axis=1)
    print(json.dumps({"ms_per_step": 1e3 * dt, "rounds": st["asm_rounds"] / 2, "lambda_ms": st["asm_lambda_ms"] / 2, "l32_ms": st["asm_lambda32_ms"] / 2,
                      "l64_ms": st["asm_lambda64_ms"] / 2, "gemm_ms": st["asm_gemm_ms"] / 2, "update_ms": st["asm_update_ms"] / 2,
                      "predict_ms": st["asm_predict_ms"] / 2, "total_ms": st["total_ms"] / 2, "status": np.bincount(buf.status.to_host(), minlength=3).tolist(),
                      "mean_active": float(nact.mean()), "f32_flops": st["asm_lambda32_flops"] / 2, "f64_flops": (st["asm_lambda_flops"] - st["asm_lambda32_flops"]) / 2}))
    sys.exit(0)
B, sx = sys.argv[1], sys.argv[2]
for it in sys.argv[3].split(","):
    env = dict(os.environ, NNMPC_PRED_ITERS=it)
    if len(sys.argv) > 4:
        env["NNMPC_TRACE_ROUNDS"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B, sx], env=env, capture_output=True, text=True)
    print("iters", it, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "NO OUTPUT", flush=True)
    if len(sys.argv) > 4:
        print("\n".join(r.stderr.strip().splitlines()[-40:]), flush=True)
    elif r.returncode:
        print(r.stderr[-2000:])
