"""Step time of the headline batch with and without the library's hipEvent scopes (profiling on / off)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from industrial_nnmpc_2021_amd import _lib
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP

_lib.set_device(0)
pl, P, tq, nu = bench.make_problem("cdu")
B = 100000
qp = BatchedBoxQP(P, tq, nu, max_batch=1024)
x0, lb, ub, us = bench.make_samples(pl, B, 1000, 2.0)
buf = bench.QpBuffers(_lib, qp, B, nu, P.shape[0])
buf.upload(x0, lb, ub, us)
for prof in (False, True, False, True):
    qp.set_profiling(prof)
    qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, buf.u, buf.act, buf.status, buf.iters)
    _lib.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"profiling {prof}: {1e3 * dt:.3f} ms per step, {B / dt / 1e6:.3f} M solves/s")
