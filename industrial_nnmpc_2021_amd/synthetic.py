"""Seeded synthetic plants and sample batches of the reference's sizes and tuning.

The real plants cannot be rebuilt (CDU_Model.mat is not in the reference repo,
the CSTRs linearisation needs casadi), so benchmarks and tests use random
plants with the reference's dimensions and regulator tuning:
  CDU   Nx=252 Nu=32 Ny=90 N=140  Q=2C'C    R=0.1I S=0     (cdu_parameters.py:99-102)
  CSTRs Nx=12  Nu=6  Ny=12 N=90   Q=1e3C'C  R=0.1I S=0.1I  (cstrs_parameters.py:300-303)
inputs scaled so that uub - ulb = 2 (cdu_parameters.py:35-40).
"""
import numpy as np

SIZES = {
    "cdu": dict(Nx=252, Nu=32, Ny=90, N=140, qw=2.0, rw=0.1, sw=0.0),
    "cstrs": dict(Nx=12, Nu=6, Ny=12, N=90, qw=1e3, rw=0.1, sw=0.1),
    # mid-size stand-in whose padded n is a multiple of 128: the lock-step rounds with the fused full-width pass (far-field form)
    "mid_cdu": dict(Nx=40, Nu=16, Ny=12, N=64, qw=2.0, rw=0.1, sw=0.0),
    # small stand-ins used by the CPU-oracle parity tests
    "mini_cdu": dict(Nx=24, Nu=4, Ny=8, N=20, qw=2.0, rw=0.1, sw=0.0),
    "mini_cstrs": dict(Nx=6, Nu=3, Ny=6, N=25, qw=1e3, rw=0.1, sw=0.1),
}


def plant(name, seed=0, rho=0.97):
    """Random plant (A, B, C) with spectral radius rho and the named tuning."""
    s = SIZES[name]
    rng = np.random.default_rng(seed)
    Nx, Nu, Ny = s["Nx"], s["Nu"], s["Ny"]
    W = rng.standard_normal((Nx, Nx)) / np.sqrt(Nx)
    A = rho * W / np.max(np.abs(np.linalg.eigvals(W)))
    B = rng.standard_normal((Nx, Nu)) / np.sqrt(Nx)
    Cm = rng.standard_normal((Ny, Nx)) / np.sqrt(Nx)
    return dict(A=A, B=B, C=Cm, Q=s["qw"] * Cm.T @ Cm, R=s["rw"] * np.eye(Nu),
                S=s["sw"] * np.eye(Nu), N=s["N"], ulb=-np.ones((Nu, 1)), uub=np.ones((Nu, 1)))


def samples(pl, B, seed=1, sx=1.0):
    """i.i.d. (x, uprev, xs, us) tuples: x - xs ~ sx N(0, I), uprev - us ~ U(-.3,.3),
    us ~ U(-.5,.5), xs = 0 (only x - xs enters the regulator)."""
    rng = np.random.default_rng(seed)
    Nx, Nu = pl["B"].shape
    us = rng.uniform(-0.5, 0.5, (B, Nu))
    xs = np.zeros((B, Nx))
    x = xs + sx * rng.standard_normal((B, Nx))
    uprev = us + rng.uniform(-0.3, 0.3, (B, Nu))
    return dict(x=x, uprev=uprev, xs=xs, us=us)
