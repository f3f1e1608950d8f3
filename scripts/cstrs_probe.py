import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
pl = synthetic.plant("cstrs", 0); P, tq, nu = build_regulator_matrices(pl)
B = 131072
for seed in (1, 1000):
    s = synthetic.samples(pl, B, seed, 2.0)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), 1); lb = pl["ulb"].T - s["us"]; ub = pl["uub"].T - s["us"]
    for kw in (dict(), dict(asm_max_rounds=40), dict(asm_max_rounds=1000), dict(asm_f32_rounds=-1)):
        qp = BatchedBoxQP(P, tq, nu, nb=64, max_batch=4096, method="asm", **kw)
        out = qp.solve_batch(x0, lb, ub)
        st = qp.stats()
        print(seed, kw, "unsolved", int((out["status"] != 0).sum()), "rounds", st["asm_rounds"], "full checks", st["asm_full_checks"])
