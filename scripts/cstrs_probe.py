"""Iteration counts and set sizes of the CSTRs-size batch (GPU box): which problems make a launch of asm_small_k last.
usage: cstrs_probe.py [B] [sx]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from industrial_nnmpc_2021_amd import _lib
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sx = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
pl, P, tq, nu = bench.make_problem("cstrs")
n = P.shape[0]
qp = BatchedBoxQP(P, tq, nu, max_batch=1024)
for seed in (1000, 8919, 16838):
    ib = bench.QpInputs(_lib, qp, B, nu); ib.upload(*bench.make_samples(pl, B, seed, sx))
    buf = bench.QpBuffers(_lib, qp, B, nu, n, inputs=ib)
    for rep in range(3):
        _lib.synchronize(); t0 = time.perf_counter()
        qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
        _lib.synchronize(); dt = time.perf_counter() - t0
    it = np.asarray(buf.iters.to_host()).reshape(-1)[:B]
    nact = np.unpackbits(buf.act.to_host(B).view(np.uint8), axis=1).sum(axis=1)
    order = np.argsort(-it)[:12]
    print(json.dumps({"seed": seed, "ms": 1e3 * dt, "iters_mean": float(it.mean()), "iters_pct": [int(np.percentile(it, q)) for q in (50, 90, 99, 99.9, 100)],
                      "nact_mean": float(nact.mean()), "nact_max": int(nact.max()),
                      "top": [(int(it[i]), int(nact[i])) for i in order],
                      "n_gt32": int((nact > 32).sum()), "n_gt64": int((nact > 64).sum())}))
