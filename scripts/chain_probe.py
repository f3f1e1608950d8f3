"""Closed-loop chains at CDU size: cold vs warm-started solves (factorisations / solve, wall time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd import linearMPC as lm, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "cdu"
nc, T = int(sys.argv[2]), int(sys.argv[3])
pl = synthetic.plant(name, seed=0)
Nx, Nu = pl["B"].shape
rng = np.random.default_rng(2)
t = time.time()
reg = lm.LinearMPCController.setup_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"], pl["N"], pl["ulb"], pl["uub"], max_batch=nc)
print("setup s", time.time() - t, flush=True)
class FixedTarget:
    def __init__(self, xs, us): self.xs, self.us = xs, us
    def solve(self, ysp, d): return self.xs, self.us
ts = [FixedTarget(np.zeros((Nx, 1)), rng.uniform(-.5, .5, (Nu, 1))) for _ in range(nc)]
sp = [np.zeros((T, 1)) for _ in range(nc)]
Nd = 5
Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx)
# PRBS-like disturbance: new level every ~8 steps per chain
ds = [np.repeat(3.0 * rng.standard_normal((T // 8 + 1, Nd)), 8, axis=0)[:T] for _ in range(nc)]
x0 = np.zeros((Nx, 1)); u0 = np.zeros((Nu, 1))
for warm in (False, True):
    t = time.time()
    out = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg, pl["ulb"], pl["uub"], ts, sp, ds, warm_start=warm)
    dt = time.time() - t
    f = out["factorizations"]
    print("warm" if warm else "cold", "time %.2fs" % dt, "solves/s %.1f" % (nc * T / dt), "status ok", (out["status"] == 0).all(),
          "fact/solve mean %.2f (steps>0: %.2f)" % (f.mean(), f[:, 1:].mean()), "sat frac", (np.abs(out["u"]) > 0.999).mean())
