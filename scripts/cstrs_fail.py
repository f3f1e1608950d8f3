"""Which CSTRs-size samples end with status != 0, and how do the single paths treat them?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
from oracle import qp as oqp
pl = synthetic.plant("cstrs", 0); P, tq, nu = build_regulator_matrices(pl)
n = P.shape[0]
B = 131072
s = synthetic.samples(pl, B, 1, 2.0)
x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), 1); lb = pl["ulb"].T - s["us"]; ub = pl["uub"].T - s["us"]
qp = BatchedBoxQP(P, tq, nu, nb=64, max_batch=4096)
out = qp.solve_batch(x0, lb, ub)
st = qp.stats()
bad = np.array([836, 5564, 7016, 14320, 63988, 126862])   # the samples the all-at-once exchange rule cycled on
print("auto: bad", bad.size, bad[:10], "asm_solved", st["asm_solved"], "full checks", st["asm_full_checks"], "e2max", st["asm_e2max"])
X, L, U = x0[bad], lb[bad], ub[bad]
for kw in (dict(method="asm"), dict(method="asm", asm_f32_rounds=-1), dict(method="asm", asm_f32_rounds=-1, asm_max_rounds=400), dict(method="pdip")):
    q2 = BatchedBoxQP(P, tq, nu, nb=64, max_batch=256, **kw)
    o2 = q2.solve_batch(X, L, U)
    print(kw, "status", o2["status"], "facts", o2["factorizations"], "ipm", o2["ipm_iters"])
Ps = np.tril(P) + np.tril(P, -1).T
for i in range(min(3, bad.size)):
    info = {"nu": nu}
    xe = oqp.solve_exact_box(Ps, tq @ X[i], np.tile(L[i], n // nu), np.tile(U[i], n // nu), info=info)
    print("oracle: active", len(info["active"]), "kkt", info["kkt"], "err of returned u", np.abs(out["u"][bad[i]] - xe).max())
