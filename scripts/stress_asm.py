"""Random generic box QPs of many shapes through the default path, checked against the fp64 oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
from oracle import qp as oqp

def spd(n, rng, cond):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.exp(rng.uniform(0.0, np.log(cond), n))
    return (Q * ev) @ Q.T

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nfail = ntot = 0
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    nu = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 16]))
    N = int(rng.integers(2, 90))
    n = nu * N
    if n > 700: N = 700 // nu; n = nu * N
    cond = float(10 ** rng.uniform(0, 6))
    P = spd(n, rng, cond)
    n_aug = int(rng.integers(1, 12))
    tq = rng.standard_normal((n, n_aug)) * np.sqrt(np.diag(P))[:, None] * rng.uniform(0.1, 3.0)
    B = int(rng.choice([1, 3, 17, 64, 130]))
    x0 = rng.standard_normal((B, n_aug)) * rng.uniform(0.2, 3.0)
    if B > 2: x0[1] = 0.0                                   # trivial problem
    lb = -rng.uniform(0.2, 2.0, (B, nu)); ub = rng.uniform(0.2, 2.0, (B, nu))
    method = str(rng.choice(["auto", "asm", "asm"]))
    method = os.environ.get("METHOD", method)
    f32 = int(rng.choice([0, -1]))
    qp = BatchedBoxQP(P, tq, nu, max_batch=128, method=method, asm_f32_rounds=f32, seg_max=int(rng.choice([0, 128])),
                      asm_max_rounds=int(os.environ.get("ASM_MAX_ROUNDS", "0")))
    out = qp.solve_batch(x0, lb, ub)
    st = qp.stats()
    bad = []
    for b in range(B):
        info = {"nu": nu}
        xe = oqp.solve_exact_box(P, tq @ x0[b], np.tile(lb[b], N), np.tile(ub[b], N), info=info)
        act = np.zeros(2 * n, bool); act[info["active"]] = True
        err = np.abs(out["u"][b] - xe).max() / max(1.0, np.abs(xe).max())
        tol = 1e-7 * max(1.0, cond / 1e3)
        if out["status"][b] != 0 or err > tol or not (out["active"][b] == act).all():
            bad.append((b, int(out["status"][b]), float(err), int((out["active"][b] != act).sum()), int(act.sum())))
    ntot += B; nfail += len(bad)
    print(f"case {case}: n={n} nu={nu} n_aug={n_aug} cond={cond:.1e} B={B} {method} f32={f32} asm_solved={st['asm_solved']} "
          f"rounds={st['asm_rounds']} full_checks={st['asm_full_checks']} -> {'OK' if not bad else bad[:4]}", flush=True)
    qp.close()
print("problems", ntot, "failures", nfail, "time", round(time.time() - t0, 1))
