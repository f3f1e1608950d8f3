import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
pl = synthetic.plant("cdu", 0); P, tq, nu = build_regulator_matrices(pl)
qp = BatchedBoxQP(P, tq, nu, max_batch=128)
B = 4096
s = synthetic.samples(pl, B, 1, 2.0)
x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), 1); lb = pl["ulb"].T - s["us"]; ub = pl["uub"].T - s["us"]
out = qp.solve_batch(x0, lb, ub)
st = qp.active_to_state(out["active"]) != 0      # (B, n)
n = st.shape[1]
stage = np.arange(n) // nu
print("active per stage (mean over problems), first 40 stages:", np.round(st.reshape(B, -1, nu).sum(2).mean(0)[:40], 2))
print("last stage with any active, quantiles:", np.quantile([(np.flatnonzero(r).max() // nu if r.any() else 0) for r in st], [0.5, 0.9, 0.99, 1.0]))
for kb in (16, 64):
    blk = st.reshape(B // 64, 64, n // kb, kb).any(axis=(1, 3))      # (rowblocks, kblocks)
    print("kblock", kb, "fraction of (64-row block, k-block) pairs that are non-zero:", blk.mean())
m = st.sum(1)
print("|A| at the solution: quantiles 1/10/50/90/99/100 %:", np.quantile(m, [0.01, 0.1, 0.5, 0.9, 0.99, 1.0]))
edges = [0, 64, 80, 96, 112, 128, 144, 160, 176, 192, 256, 10000]
print("histogram over", edges, ":", np.histogram(m, edges)[0] / B)
