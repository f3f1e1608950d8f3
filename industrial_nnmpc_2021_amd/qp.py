"""Batched box-constrained condensed QP on the GPU (host wrapper over the C ABI).

    min 1/2 u'Pu + (tq x0)'u   s.t.  lb <= u_k <= ub,  k = 0..N-1

is what DenseQPRegulator.solve hands to cvxopt.solvers.qp for a stable plant
(reference lib/linearMPC.py:495-512 with G = tE, :476-482).  ``BatchedBoxQP``
solves B of them per call in HIP kernels (see csrc/qp_solver.hip).
"""
import ctypes as C
import numpy as np

from . import _lib


def _ptr(a):
    """Pointer of a numpy array (host) or of anything exposing data_ptr() (device)."""
    if a is None:
        return None, None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr()), _lib.DEVICE
    return a.ctypes.data_as(C.c_void_p), _lib.HOST


def _staircase(U0, tol, block=128, chunk=16):
    """Rotate the basis of the far-field factors so that far column tiles use few coordinates.

    U0 (rows of the far block x r).  The row space of the suffix U0[128 j:] shrinks as j grows -- the closed loop forgets its
    fast modes first.  Going down the blocks, the coordinates (within the ones still in use) whose singular values over ALL the
    remaining rows are below tol are moved behind the ones that stay (SVD of the suffix in the current basis; what is dropped is
    small on every later row by construction, nothing is ever estimated from a weak block).  With the orthonormal G (r' x r) this
    gives  U0 = U1 G,  U1[block j, k_j:] = 0  up to tol per dropping step: a staircase.  Returns (U1 with those entries set to
    exact zeros, k_j rounded up to whole K chunks of the GEMM; G).  The library's GEMM stops a column tile's K loop at k_j
    (nnmpc_qp_set_farfield finds the zeros), and its device check of U V' against M covers the truncation."""
    import scipy.linalg as sla
    nf, r = U0.shape
    nb = -(-nf // block)
    Rs = [None] * nb                                     # R factor of the suffix U0[128 j:], bottom-up (same singular values / row space)
    below = np.zeros((0, r))
    for j in range(nb - 1, -1, -1):
        below = Rs[j] = sla.qr(np.vstack((U0[j * block:(j + 1) * block], below)), mode="r")[0][:r]
    cur = np.eye(r)                                      # rows: the coordinates still in use
    drops, kj = [], np.zeros(nb, int)
    for j in range(nb):
        if cur.shape[0]:
            _, sv, vt = sla.svd(Rs[j] @ cur.T)           # (full: vt is square, the rows beyond the rank span what is dropped)
            k = int((sv > tol).sum())
            drops.append(vt[k:] @ cur)
            cur = vt[:k] @ cur
        else:
            drops.append(np.zeros((0, r)))
        kj[j] = cur.shape[0]
    G = np.vstack([cur] + drops[:0:-1])                  # (what block 0 drops is below tol everywhere: not part of the factors)
    if G.shape[0] == 0:
        G = np.eye(1, r)
    kj = np.minimum(-(-kj // chunk) * chunk, G.shape[0])
    U1 = U0 @ G.T
    for j in range(nb):
        U1[j * block:(j + 1) * block, kj[j]:] = 0.0
    return np.ascontiguousarray(U1), G


class BatchedBoxQP:
    """Owns the device copies of (P, tq) and the solver workspace.

    P: (n, n) condensed Hessian (lower triangle read, as cvxopt does),
    tq: (n, n_aug) so that q = tq @ x0, nu: inputs per stage (bounds are per
    stage and tiled along the horizon like the reference's _get_h).
    """

    def __init__(self, P, tq, nu, *, Kunc="auto", method="auto", farfield="auto", max_batch=1024, nb=0, ipm_tol=0.0,
                 max_rounds=0, max_ipm_iters=0, max_polish_rounds=0, max_refine=0, sub_steps=0, stale_max_changes=0, stale_cg_limit=0, asm_max_active=0,
                 asm_max_rounds=0, asm_f32_rounds=0, seg_max=0, asm_tail_batch=0, asm_predict_iters=0):
        lib = _lib.load()
        P = np.ascontiguousarray(P, dtype=np.float64)
        tq = np.ascontiguousarray(tq, dtype=np.float64)
        n, n_aug = tq.shape
        if P.shape != (n, n) or n % nu:
            raise ValueError("P must be (n, n), tq (n, n_aug), n a multiple of nu")
        meth = {"auto": 0, "pdip": 1, "asm": 2}[method]
        Hinv = None
        if isinstance(Kunc, str) and Kunc == "auto":
            # one-time host setup in fp64 (like the reference's DARE/condensing in
            # DenseQPRegulator.__init__): u_unc = Kunc x0 = -P^-1 tq x0 (warm start), and, for
            # the shared-inverse active-set pass, P^-1 itself
            import scipy.linalg as sla
            Ps = np.tril(P) + np.tril(P, -1).T
            cf = sla.cho_factor(Ps, lower=True)
            Kunc = -sla.cho_solve(cf, tq)
            if meth != 1:
                Hinv = sla.cho_solve(cf, np.eye(n))
        elif meth == 2:
            raise ValueError("method='asm' needs Kunc='auto' (the inverse Hessian is built with it)")
        if Kunc is not None:
            Kunc = np.ascontiguousarray(Kunc, dtype=np.float64)
        self.n, self.n_aug, self.nu = n, n_aug, nu
        self.words = (2 * n + 31) // 32
        opts = _lib.QpOpts(max_batch=max_batch, nb=nb, max_ipm_iters=max_ipm_iters,
                           max_polish_rounds=max_polish_rounds, max_refine=max_refine,
                           max_rounds=max_rounds, sub_steps=sub_steps, stale_max_changes=stale_max_changes,
                           stale_cg_limit=stale_cg_limit, method=meth, asm_max_active=asm_max_active,
                           asm_max_rounds=asm_max_rounds, asm_f32_rounds=asm_f32_rounds, seg_max=seg_max, asm_tail_batch=asm_tail_batch,
                           asm_predict_iters=asm_predict_iters,
                           ipm_tol=ipm_tol, refine_tol=0.0, bound_tol=0.0)
        self._h = C.c_void_p()
        kp = Kunc.ctypes.data_as(C.c_void_p) if Kunc is not None else None
        _lib.check(lib.nnmpc_qp_create(C.byref(self._h), n, nu, n_aug,
                                       P.ctypes.data_as(C.c_void_p), tq.ctypes.data_as(C.c_void_p),
                                       kp, C.byref(opts)), "nnmpc_qp_create")
        self._lib = lib
        self.have_inverse = False
        if Hinv is not None:
            Hinv = np.ascontiguousarray(Hinv, dtype=np.float64)
            rc = lib.nnmpc_qp_set_inverse(self._h, Hinv.ctypes.data_as(C.c_void_p), Kunc.ctypes.data_as(C.c_void_p))
            if rc == _lib.EINVAL and meth == 0:
                # the fp64 inverse of an ill-conditioned P missed the library's |P Pinv - I| check: the active-set
                # pass cannot be certified with it, the PDIP path needs no inverse -- method "auto" goes on without
                import warnings
                warnings.warn("BatchedBoxQP: " + lib.nnmpc_last_error().decode(errors="replace") +
                              " -- continuing with the PDIP path only", RuntimeWarning)
            else:
                _lib.check(rc, "nnmpc_qp_set_inverse")
                self.have_inverse = True
        # far-field form of the full-width pass (include/nnmpc.h: nnmpc_qp_set_farfield): "auto" factors the far block
        # for a window the first time a call's full-width pass had to do without (one-time host setup like the inverse:
        # an fp64 SVD, ~1.5 s at the CDU size); a list of windows factors them now; None / False never
        # (kept for those factorisations: the leading half of the columns of P^-1 and Kunc -- a window beyond n / 2 cannot pay --,
        # 80 of the 160 MB at the CDU size; released by close())
        self._ff_wmax = (n // 2 // 128) * 128
        self._ff_src = (np.ascontiguousarray(Hinv[:, :self._ff_wmax]), Kunc) if (self.have_inverse and farfield and self._ff_wmax > 0) else None
        nb_lib = nb if nb else (128 if n > 1024 else 64)     # the library pads n to its tile size (nnmpc_qp_create)
        self._np = -(-n // nb_lib) * nb_lib
        self._ff_done = set()
        self.farfield_info = {}        # window -> dict(rank, tiles' K extents ...) of the factors handed to the library (0: not usable)
        if self._ff_src is not None and not isinstance(farfield, str):
            for W in farfield:
                self.prepare_farfield(int(W))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nnmpc_qp_destroy(self._h)
            self._h = C.c_void_p()
        self._ff_src = None

    __del__ = close

    def prepare_farfield(self, W, rtol=1e-13):
        """Factor M = [Kunc[W:] | -Hinv[W:, 0:W]] = U [Vx | Vl] (truncated SVD: singular values above rtol times the
        largest; M has numerical rank ~Nx; basis rotated so that U is a staircase, see _staircase) and hand the factors to the
        library, which verifies them on the device.  Returns the rank, or 0 when W is not a window the full-width pass can use."""
        if self._ff_src is None or W in self._ff_done:
            return 0
        self._ff_done.add(W)
        self.farfield_info[W] = dict(rank=0, reason="window not on the 128-column grid")
        Hinv, Kunc = self._ff_src
        if W <= 0 or W % 128 or W >= self.n or self._np % 128 or W > self._ff_wmax:   # (the library's own gate is on the PADDED n)
            return 0
        import scipy.linalg as sla
        M = np.hstack((Kunc[W:], -Hinv[W:, :W]))
        U, s, Vt = sla.svd(M, full_matrices=False, lapack_driver="gesdd")
        r = max(1, int((s > rtol * s[0]).sum()))
        rp, nf, k = -(-r // 128) * 128, self.n - W, -(-self.n_aug // 32) * 32 + W
        self.farfield_info[W] = dict(rank=0, numerical_rank=r, sigma_ratio_at_rank=float(s[min(r, s.size - 1)] / s[0]),
                                     reason="the factored form would not pay: numerical rank too large")
        if rp * (k + nf) > 0.8 * k * nf:             # the factored form would not pay (the library refuses such factors too):
            return 0                                 # a generic Hessian's far block has full rank, the MPC structure makes it ~Nx
        Uf, G = _staircase(U[:, :r] * s[:r], rtol * s[0])
        r = G.shape[0]
        Vt = G @ Vt[:G.shape[1]]
        Vx, Vl = np.ascontiguousarray(Vt[:, :self.n_aug]), np.ascontiguousarray(Vt[:, self.n_aug:])
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._lib.nnmpc_qp_set_farfield(self._h, W, r, p(Uf), p(Vx), p(Vl))
        if rc == _lib.EINVAL:                        # refused (rank too large for the workspace, factors too inaccurate): dense form stays
            import warnings
            self.farfield_info[W] = dict(rank=0, numerical_rank=r, reason=self._lib.nnmpc_last_error().decode(errors="replace"))
            warnings.warn("BatchedBoxQP.prepare_farfield: " + self.farfield_info[W]["reason"], RuntimeWarning)
            return 0
        _lib.check(rc, "nnmpc_qp_set_farfield")
        kj = [int(np.flatnonzero(np.abs(Uf[j:j + 128]).max(axis=0) > 0).max(initial=-1)) + 1 for j in range(0, Uf.shape[0], 128)]
        self.farfield_info[W] = dict(rank=int(r), staircase_mean_k=float(np.mean(kj)), staircase_max_k=int(max(kj)))
        return r

    def prepare_farfield_windows(self, lo=128, hi=None):
        """Factor every window lo, lo + 128, ... <= hi (default: a quarter of the horizon) now -- one-time setup like the inverse,
        ~0.5 s of host SVD and ~10 MB of HBM per window at the CDU size -- so that no later call meets a window without factors
        (the window of a call follows its batch: the last active bound of any of its problems).  Returns {W: rank}."""
        hi = self.n // 4 if hi is None else hi
        return {W: self.prepare_farfield(W) for W in range(lo, hi + 1, 128)}

    def _farfield_auto(self):
        if self._ff_src is None:
            return
        for _ in range(32):                           # every window the call met without factors
            W = C.c_int32(0)
            _lib.check(self._lib.nnmpc_qp_farfield_missing(self._h, C.byref(W)), "nnmpc_qp_farfield_missing")
            if not W.value:
                break
            self.prepare_farfield(W.value)

    def solve_batch(self, x0, lb, ub, guess=None, first_move_only=False):
        """numpy in / numpy out.  x0 (B, n_aug), lb/ub (B, nu) or (nu,); guess: optional (B, n) uint8
        active-set estimate (0 free, 1 upper, 2 lower) that replaces the PDIP phase (warm start).

        Returns dict(u (B, n) -- or (B, nu) with first_move_only, all the reference keeps of a solve in its
        simulation loops (lib/linearMPC.py:856) --, active (B, 2n) bool in the row order of the reference's G,
        status (B,), ipm_iters (B,), factorizations (B,)).
        """
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, self.n_aug)
        B = x0.shape[0]
        lb = np.ascontiguousarray(np.broadcast_to(np.asarray(lb, np.float64).reshape(-1, self.nu), (B, self.nu)))
        ub = np.ascontiguousarray(np.broadcast_to(np.asarray(ub, np.float64).reshape(-1, self.nu), (B, self.nu)))
        u = np.empty((B, self.nu if first_move_only else self.n))
        act = np.zeros((B, self.words), np.uint32)
        status = np.empty(B, np.int32)
        iters = np.empty((B, 2), np.int32)
        gp = None
        if guess is not None:
            guess = np.ascontiguousarray(guess, dtype=np.uint8).reshape(B, self.n)
            gp = guess.ctypes.data_as(C.c_void_p)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.nnmpc_qp_solve_batch_ex(
            self._h, B, p(x0), p(lb), p(ub), gp, p(u), p(act), p(status), p(iters), _lib.HOST,
            _lib.OUT_FIRST_MOVE if first_move_only else _lib.OUT_SEQUENCE), "nnmpc_qp_solve_batch_ex")
        self._farfield_auto()
        bits = np.unpackbits(act.view(np.uint8), axis=1, bitorder="little")[:, :2 * self.n].astype(bool)
        return dict(u=u, active=bits, status=status, ipm_iters=iters[:, 0], factorizations=iters[:, 1])

    def active_to_state(self, active):
        """(B, 2n) bool rows of G -> (B, n) uint8 per-variable state (the `guess` format)."""
        n, nu = self.n, self.nu
        k, c = np.arange(n) // nu, np.arange(n) % nu
        return (active[:, k * 2 * nu + c].astype(np.uint8) + 2 * active[:, k * 2 * nu + nu + c].astype(np.uint8))

    def solve_batch_device(self, B, x0, lb, ub, u, active=None, status=None, iters=None, guess=None,
                           first_move_only=False):
        """HBM-resident buffers (objects with data_ptr(): _lib.DeviceArray, torch CUDA tensors; f64/u32/i32/u8).
        u is (B, n), or (B, nu) with first_move_only."""
        q = lambda a: (_ptr(a)[0] if a is not None else None)
        _lib.check(self._lib.nnmpc_qp_solve_batch_ex(
            self._h, B, q(x0), q(lb), q(ub), q(guess), q(u), q(active), q(status), q(iters), _lib.DEVICE,
            _lib.OUT_FIRST_MOVE if first_move_only else _lib.OUT_SEQUENCE), "nnmpc_qp_solve_batch_ex")
        self._farfield_auto()

    def set_profiling(self, on=True):
        _lib.check(self._lib.nnmpc_qp_set_profiling(self._h, int(on)), "nnmpc_qp_set_profiling")

    def stats(self, reset=False):
        s = _lib.QpStats()
        _lib.check(self._lib.nnmpc_qp_get_stats(self._h, C.byref(s), int(reset)), "nnmpc_qp_get_stats")
        return {k: getattr(s, k) for k, _ in s._fields_}

    def debug_factor_solve(self, dvec, mask, rhs):
        """Kernel-level hook: solve (mask mask' o P + diag(dvec)) sol = rhs per row (f32)."""
        dvec = np.ascontiguousarray(dvec, np.float32)
        mask = np.ascontiguousarray(mask, np.float32)
        rhs = np.ascontiguousarray(rhs, np.float32)
        B = dvec.shape[0]
        sol = np.empty((B, self.n), np.float32)
        _lib.check(self._lib.nnmpc_qp_debug_factor_solve(
            self._h, B, *(a.ctypes.data_as(C.c_void_p) for a in (dvec, mask, rhs, sol))),
            "nnmpc_qp_debug_factor_solve")
        return sol
