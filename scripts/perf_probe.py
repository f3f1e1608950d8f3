"""Quick throughput / kernel-time probe of the batched QP solver (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP

import os as _os
name = sys.argv[1]; B = int(sys.argv[2]); nb = int(sys.argv[3]); slots = int(sys.argv[4]); sx = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
pl = synthetic.plant(name, 0)
t = time.time(); P, tq, nu = build_regulator_matrices(pl); print("condense s", time.time() - t, P.shape, flush=True)
t = time.time(); qp = BatchedBoxQP(P, tq, nu, nb=nb, max_batch=slots, ipm_tol=float(_os.environ.get("IPM_TOL", "0")), stale_max_changes=int(_os.environ.get("STALE_CHG", "0")), stale_cg_limit=int(_os.environ.get("STALE_CG", "0")), sub_steps=int(_os.environ.get("SUBSTEPS", "0")), method=_os.environ.get("METHOD", "auto")); print("create s", time.time() - t, flush=True)
s = synthetic.samples(pl, B, 1, sx)
x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), 1); lb = pl["ulb"].T - s["us"]; ub = pl["uub"].T - s["us"]
out = qp.solve_batch(x0[:min(B, slots)], lb[:min(B, slots)], ub[:min(B, slots)])  # warmup
qp.set_profiling(True); qp.stats(reset=True)
t = time.time(); out = qp.solve_batch(x0, lb, ub); dt = time.time() - t
st = qp.stats()
n = P.shape[0]
print("B", B, "time", dt, "solves/s", B / dt)
print("status hist", np.bincount(out["status"], minlength=3), "ipm iters mean/max", out["ipm_iters"].mean(), out["ipm_iters"].max(),
      "fact mean/max", out["factorizations"].mean(), out["factorizations"].max(), "active mean", out["active"].sum(1).mean())
print(st)
fl = st["factorizations"] * n ** 3 / 3
print("chol algorithmic TFLOP/s over total:", fl / (st["total_ms"] * 1e-3) / 1e12, " panel-only:", st["panel_flops"] / (st["panel_ms"] * 1e-3) / 1e12,
      "panel share", st["panel_ms"] / st["total_ms"], "diag share", st["diag_ms"] / st["total_ms"], "trsv share", st["trsv_ms"] / st["total_ms"])
T = (n + nb - 1) // nb if nb else 0
NBv = nb or (64 if n <= 1024 else 128); T = (n + NBv - 1) // NBv
bytes_per_solve = 2 * (T * (T + 1) // 2 + T) * NBv * NBv * 4
print("trsv solves/problem", st["trsv_solves"] / B, "trsv TB/s", st["trsv_solves"] * bytes_per_solve / (st["trsv_ms"] * 1e-3) / 1e12)
