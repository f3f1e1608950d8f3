import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
pl = synthetic.plant("cstrs", 0); P, tq, nu = build_regulator_matrices(pl)
qp = BatchedBoxQP(P, tq, nu, nb=64, max_batch=4096)
B = 8192
s = synthetic.samples(pl, B, 1, 2.0)
x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), 1); lb = pl["ulb"].T - s["us"]; ub = pl["uub"].T - s["us"]
out = qp.solve_batch(x0, lb, ub)
bad = np.flatnonzero(out["status"] != 0)
print("bad", bad[:20], "n", bad.size, "facts", out["factorizations"][bad][:20], "ipm", out["ipm_iters"][bad][:20])
np.savez("gpurun_out/cstrs_fail.npz", bad=bad, u=out["u"][bad], act=out["active"][bad], fact=out["factorizations"][bad], x0=x0[bad], lb=lb[bad], ub=ub[bad])
