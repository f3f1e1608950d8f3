import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
pl = synthetic.plant("cdu", 0); P, tq, nu = build_regulator_matrices(pl)
qp = BatchedBoxQP(P, tq, nu, max_batch=1024, method="asm")
for seed in (1, 1000):
    for B in (256, 2048, 14336):
        s = synthetic.samples(pl, B, seed, 2.0)
        x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), 1); lb = pl["ulb"].T - s["us"]; ub = pl["uub"].T - s["us"]
        qp.stats(reset=True)
        out = qp.solve_batch(x0, lb, ub)
        st = qp.stats()
        print("seed", seed, "B", B, "status", np.bincount(out["status"], minlength=3), "asm_solved", st["asm_solved"], "rounds", st["asm_rounds"], flush=True)
