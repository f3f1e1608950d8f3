"""Far-field form of the full-width pass against the dense form: same results (sequence and first-move calls), timing.
   python scripts/ff_check.py [B]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from industrial_nnmpc_2021_amd import _lib
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sx = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
pl, P, tq, nu = bench.make_problem("cdu")
n = P.shape[0]
x0, lb, ub, us = bench.make_samples(pl, B, 1000, sx)
res = {}
for name, ff in (("dense", None), ("far", "auto")):
    qp = BatchedBoxQP(P, tq, nu, max_batch=1024, farfield=ff)
    buf = bench.QpBuffers(_lib, qp, B, nu, n)
    buf.upload(x0, lb, ub, us)
    for fm in (False, True):
        out = buf.first if fm else buf.u
        for _ in range(2):
            qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, out, buf.act, buf.status, buf.iters, first_move_only=fm)
        qp.stats(reset=True)
        _lib.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            qp.solve_batch_device(B, buf.x0, buf.lb, buf.ub, out, buf.act, buf.status, buf.iters, first_move_only=fm)
        _lib.synchronize(); dt = (time.perf_counter() - t0) / 3
        st = qp.stats()
        res[(name, fm)] = dict(u=out.to_host(min(B, 4000)), act=buf.act.to_host(), status=buf.status.to_host(), ms=1e3 * dt, far=st["asm_far_passes"])
        print(name, "first-move" if fm else "sequence", f"{1e3 * dt:.2f} ms  {B / dt / 1e6:.2f} M/s  status", np.bincount(res[(name, fm)]["status"], minlength=3), "far passes", st["asm_far_passes"], flush=True)
    qp.close(); buf.free()
for fm in (False, True):
    a, b = res[("dense", fm)], res[("far", fm)]
    print("first-move" if fm else "sequence", "max |u_dense - u_far|", np.abs(a["u"] - b["u"]).max(), "active sets equal", bool(np.array_equal(a["act"], b["act"])),
          "status equal", bool(np.array_equal(a["status"], b["status"])))
print("first move of sequence == first-move call:", np.abs(res[("far", False)]["u"][:, :nu] - res[("far", True)]["u"]).max())
