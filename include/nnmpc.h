/* nnmpc.h -- C ABI of libnnmpc_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the offline-MPC hot path of
 * pratyushkumar211/industrial_nnmpc_2021.  Every entry point replaces one
 * reference interface (file:line into the reference repo):
 *
 *   nnmpc_qp_create        <- DenseQPRegulator.__init__/_setup_fixed_matrices
 *                             (lib/linearMPC.py:339-395): takes the condensed
 *                             P (n x n) and tq (n x n_aug) the reference builds.
 *   nnmpc_qp_solve_batch   <- DenseQPRegulator.solve -> cvxopt.solvers.qp(P, tq@x0, G, h)
 *                             (lib/linearMPC.py:495-512, :503-504) with the
 *                             box G = blockdiag([I;-I]), h = tile([uub;-ulb])
 *                             (:476-493), for B independent (x0, ulb, uub).
 *   nnmpc_nn_create        <- RegulatorLayerWith/WithoutUprev.__init__
 *                             (lib/LinearMPCLayers.py:22-32, :73-83) /
 *                             NeuralNetworkController regulator_weights, xscale
 *                             (lib/controller_evaluation.py:780-839).
 *   nnmpc_nn_forward       <- RegulatorLayerWith/WithoutUprev.call
 *                             (lib/LinearMPCLayers.py:40-61, :91-112) ==
 *                             NeuralNetworkController._get_control_input
 *                             (lib/controller_evaluation.py:863-892).
 *
 * Conventions: plain pointers and sizes only; all matrices row-major; the
 * caller owns every buffer it passes; a handle owns its device copies and
 * workspace; functions return 0 on success or a negative NNMPC_E* code and
 * never throw; nnmpc_last_error() describes the last failure of the calling
 * thread.  A handle is bound to the HIP device current at create time and is
 * not re-entrant.  `ptr_kind` says where the per-call buffers live:
 * NNMPC_HOST (the library stages them through PCIe) or NNMPC_DEVICE (HBM
 * resident, used in place).
 */
#ifndef NNMPC_H
#define NNMPC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NNMPC_OK 0
#define NNMPC_EINVAL (-1)   /* bad argument */
#define NNMPC_EHIP (-2)     /* HIP runtime error */
#define NNMPC_ENOMEM (-3)
#define NNMPC_ENOTIMPL (-4)

#define NNMPC_HOST 0
#define NNMPC_DEVICE 1

/* per-problem status written by nnmpc_qp_solve_batch */
#define NNMPC_ST_OPTIMAL 0   /* KKT conditions verified in fp64 */
#define NNMPC_ST_MAXITER 1   /* round / polish budget exhausted, not certified */
#define NNMPC_ST_NUMERIC 2   /* non-positive pivot or NaN */

typedef struct nnmpc_qp nnmpc_qp;
typedef struct nnmpc_nn nnmpc_nn;

typedef struct {
  int32_t max_batch;         /* resident problems per wave (rounded up to 128); 0 = 1024 */
  int32_t nb;                /* Cholesky block 64 | 128; 0 = auto */
  int32_t max_ipm_iters;     /* 0 = 40 */
  int32_t max_polish_rounds; /* 0 = 150 (the single-exchange fallback against cycling may take one round per bound) */
  int32_t max_refine;        /* PCG steps per active set; 0 = 60 */
  int32_t max_rounds;        /* lock-step rounds a problem may stay resident; 0 = 1000 */
  int32_t sub_steps;         /* solve sub-steps (PCG steps / KKT check) per round; 0 = 8 */
  int32_t stale_max_changes; /* polish: reuse the previous factor as PCG preconditioner when at most
                                this many bounds changed; 0 = 4, < 0 = always refactor */
  int32_t stale_cg_limit;    /* ... and refactor anyway after this many PCG steps; 0 = 16 */
  int32_t method;            /* 0 = auto: shared-inverse active-set pass (needs nnmpc_qp_set_inverse),
                                PDIP for what it leaves; 1 = PDIP only; 2 = active-set pass only */
  int32_t asm_max_active;    /* active-set pass: largest active set handled (<= 768); 0 = 768 */
  int32_t asm_max_rounds;    /* ... and its budget; 0 = 200 lock-step rounds (all-at-once exchanges settle in ~5) and 50 000
                                iterations per problem in the device tail (the single-exchange fallback against cycling is
                                finite but can take tens of thousands of ~30 us iterations on ill-conditioned Hessians with
                                half of the bounds active); > 0: that many of either; what is left goes to the PDIP path */
  int32_t asm_f32_rounds;    /* 0 = the rounds run in f32 until a problem's set settles, then in fp64 (only fp64
                                results are accepted); < 0 = fp64 from the first round */
  int32_t seg_max;           /* problems per segment (one lock-step pass; ~0.3 MB of workspace each at n = 4480);
                                0 = as many as a quarter of the free HBM holds (at most 2^20) */
  int32_t asm_tail_batch;    /* a call (segment) of at most this many problems is finished on the device from the start
                                (asm_tail_k, one workgroup per problem, no lock-step rounds): the chains of a task, a
                                controller's single QP; 0 = 256 (the most the kernel's slabs hold), < 0 = never */
  int32_t asm_predict_iters; /* iterations of the dual accelerated-projected-gradient predictor that names the FIRST active sets of
                                the rounds (bf16 MFMA, csrc/qp_predict.h): 0 = adaptive (0.3 per bound x_unc violates, 8 .. 64, by workgroup of 64
                                problems), > 0 = that many (<= 64), < 0 = off (first sets = the bounds x_unc violates).
                                Affects only the number of rounds, never a result; used when n >= 512, nu <= 64, no caller's guess */
  float ipm_tol;             /* PDIP exit, objective scaled by 1/median(diag P):
                                |r_d|_inf and mu <= tol*max(1,|q|_inf); 0 = 1e-2 */
  double refine_tol;         /* PCG exit: |step|_inf <= tol*max(1,|x|_inf); 0 = 1e-10 */
  double bound_tol;          /* primal feasibility slack of the KKT check; 0 = 1e-9 */
} nnmpc_qp_opts;

typedef struct {
  int64_t problems;          /* problems solved since the last reset */
  int64_t rounds;            /* lock-step rounds executed */
  int64_t factorizations;    /* per-problem Cholesky factorisations */
  int64_t ipm_iterations;    /* per-problem PDIP iterations */
  int64_t panel_launches;    /* launches of the dominant kernel (chol_panel) */
  double panel_ms;           /* hipEvent time of those launches (profiling on) */
  double diag_ms;            /* same for chol_diag */
  double trsv_ms;            /* same for trsv */
  double total_ms;           /* hipEvent time of whole solve_batch calls */
  double panel_flops;        /* algorithmic flops executed by chol_panel launches */
  int64_t trsv_solves;       /* per-problem triangular solve pairs (L y = r, L'x = y) executed */
  int64_t asm_solved;        /* problems finished (and certified) by the active-set pass */
  int64_t asm_rounds;        /* lock-step rounds of the active-set pass */
  int64_t asm_gemm_launches; /* launches of its dominant kernel (gemm_nt_f64_k: LAM * Hinv) */
  double asm_gemm_ms;        /* hipEvent time of those (profiling on) */
  double asm_gemm_flops;     /* algorithmic flops: 2 * columns evaluated * k_max per problem and round (+ one full-width pass) */
  double asm_lambda_ms;      /* hipEvent time of the multiplier-system kernels (asm_lambda_reg_k, asm_lambda_tile_k) */
  double asm_update_ms;      /* hipEvent time of the set bookkeeping (asm_count_k, asm_bins_*_k, asm_update_k, asm_wide_k) */
  double asm_lambda_flops;   /* algorithmic fp64 flops of the multiplier systems: sum of m^3/3 + 2 m^2 */
  double asm_lambda_bytes;   /* ... and their algorithmic bytes (gathered Pinv block + rhs/result) */
  double asm_e1max;          /* max |P Kunc + tq| of the verified inverse (nnmpc_qp_set_inverse) */
  double asm_e2max;          /* max |P Pinv - I| */
  int64_t asm_full_checks;   /* finished problems the inverse-error bound could not certify: checked with P itself */
  double asm_lambda32_ms;    /* hipEvent time of the f32 instance of the main multiplier kernel (asm_lambda_reg32_k) alone */
  double asm_lambda64_ms;    /* ... and of the fp64 instance (asm_lambda_reg_k) */
  double asm_lambda32_flops; /* the part of asm_lambda_flops solved in f32 rounds (price it against the f32 MFMA peak) */
  int64_t asm_lambda32_launches, asm_lambda64_launches;
  int64_t asm_far_passes;    /* full-width passes that ran in the far-field form (nnmpc_qp_set_farfield) */
  double asm_side_ms;        /* hipEvent time of the multiplier kernels of the larger sets on the three side streams (they run
                                beside asm_lambda_reg32_k / asm_lambda_reg_k; sum over the streams) */
  int64_t asm_small_passes;  /* segments that went through the one-wave-per-problem kernel of small problems (asm_small_k: n <= 724) */
  int64_t asm_predict_launches; /* launches of the first-set predictor (asm_predict_k) */
  double asm_predict_ms;     /* their hipEvent time (profiling on) */
  double asm_predict_flops;  /* bf16 MFMA flops they executed: 2 * 64 * 512 * (columns of Y in use) per workgroup and iteration */
} nnmpc_qp_stats;

const char* nnmpc_last_error(void);

/* P: n x n (only the lower triangle is read, like cvxopt does), tq: n x n_aug,
 * Kunc: n x n_aug warm-start gain (u_unc = Kunc x0 = -P^-1 tq x0) or NULL.
 * n = N * nu (horizon * inputs per stage); bounds are per stage and tiled. */
int nnmpc_qp_create(nnmpc_qp** out, int32_t n, int32_t nu, int32_t n_aug, const double* P,
                    const double* tq, const double* Kunc, const nnmpc_qp_opts* opts);
int nnmpc_qp_destroy(nnmpc_qp* h);

/* x0: B x n_aug, lb/ub: B x nu  ->  u: B x n, active: B x ceil(2n/32) words
 * (bit i = row i of the reference's G: stage k, rows [k*2nu, k*2nu+nu) upper,
 * [k*2nu+nu, (k+1)*2nu) lower), status: B, iters: B x 2 (PDIP iterations,
 * factorisations).  active/status/iters may be NULL. */
int nnmpc_qp_solve_batch(nnmpc_qp* h, int32_t B, const double* x0, const double* lb,
                         const double* ub, double* u, uint32_t* active, int32_t* status,
                         int32_t* iters, int32_t ptr_kind);

/* Same, warm-started: guess (B x n bytes; 0 free, 1 at upper, 2 at lower bound; NULL = cold; a row whose first byte
 * is 255 = no guess for that problem) is the
 * caller's estimate of the active set -- e.g. the previous step's set of a closed-loop chain shifted by
 * one stage (simulate_offline, lib/linearMPC.py:845-866, solves a slowly varying sequence of QPs).
 * The PDIP phase is skipped, the polish starts on the guess; the result is KKT-certified as always. */
int nnmpc_qp_solve_batch_warm(nnmpc_qp* h, int32_t B, const double* x0, const double* lb,
                              const double* ub, const uint8_t* guess, double* u, uint32_t* active,
                              int32_t* status, int32_t* iters, int32_t ptr_kind);

/* Same with a choice of what comes back in u:
 *   NNMPC_OUT_SEQUENCE    u: B x n  -- the whole input sequence DenseQPRegulator.solve returns (lib/linearMPC.py:506-512)
 *   NNMPC_OUT_FIRST_MOVE  u: B x nu -- useq[0:Nu], all that simulate_offline (:856) and control_law (:662-665) keep;
 *                         every problem is still solved and certified over all n variables, only the write-out shrinks
 *                         (3.6 GB -> 26 MB per 100 000 CDU-size problems).
 * guess may be NULL (cold start).  Problems whose inputs hold a NaN / Inf or a bound pair with lb > ub are not solved:
 * status NNMPC_ST_NUMERIC, u = NaN. */
#define NNMPC_OUT_SEQUENCE 0
#define NNMPC_OUT_FIRST_MOVE 1
int nnmpc_qp_solve_batch_ex(nnmpc_qp* h, int32_t B, const double* x0, const double* lb,
                            const double* ub, const uint8_t* guess, double* u, uint32_t* active,
                            int32_t* status, int32_t* iters, int32_t ptr_kind, int32_t out_kind);

/* Enables the shared-inverse active-set pass: Hinv = P^-1 (n x n, fp64), Kunc = -Hinv tq (n x n_aug).
 * All samples share P (reference lib/linearMPC.py:472), so on an active set A the equality-constrained
 * optimum is x = x_unc - Hinv[:,A] lam with lam = (Hinv_AA)^-1 (x_unc,A - b_A): a primal-dual active-set
 * iteration needs no n^3 factorisation.  The inverse is verified once on the device (|P Pinv - I|,
 * |P Kunc + tq|); those bounds certify each result's KKT residual and multiplier signs, results too
 * close to call are re-checked against P itself in fp64, and problems the pass cannot finish go
 * through the PDIP path. */
int nnmpc_qp_set_inverse(nnmpc_qp* h, const double* Hinv, const double* Kunc);

/* Far-field form of the full-width pass.  A problem whose active set lies inside the leading W variables (the column
 * window of the active-set rounds) has, for the variables beyond,
 *     x[W:] = M [x0 ; lam[0:W]],      M = [Kunc[W:] | -Hinv[W:, 0:W]]      ((n - W) x (n_aug + W)),
 * and M has numerical rank ~Nx: beyond the last active bound the optimum of the reference's condensed problem
 * (lib/linearMPC.py:430-474, terminal penalty = the DARE solution, :356) follows the unconstrained recursion, a linear
 * function of the state at the window's end.  With M = U [Vx | Vl] (r columns; e.g. a truncated SVD, U = U_r S_r) the pass
 * costs 2 r (n_aug + W + n - W) instead of 2 (n_aug + W)(n - W) flops per problem.  U: (n - W) x r, Vx: r x n_aug,
 * Vl: r x W, host, row-major.  The library verifies the factors on the device (max |U [Vx | Vl] - M|, refused above 1e-9,
 * and that bound enters every certificate that rests on them).  W must be a multiple of 128 below n; several windows
 * may be set.  Trailing exact zeros in the rows of U shorten the work: a 128-column tile of the pass only multiplies the
 * leading columns of U that any of its rows uses (a basis ordered so that far tiles need few coordinates -- a staircase --
 * pays: the closed loop forgets its fast modes first).  First-move calls (NNMPC_OUT_FIRST_MOVE) additionally skip the
 * 128-column tiles that |x_j| <= |U_j| |T_p| + e_far (|x0|_1 + |lam|_1) <= min_k min(ub_k, -lb_k) - bound_tol certifies feasible
 * (e_far = the verified max |U [Vx | Vl] - M|) -- nothing out there is delivered.  Replacing a window's factors releases the old
 * ones; a refused set leaves nothing behind. */
int nnmpc_qp_set_farfield(nnmpc_qp* h, int32_t W, int32_t r, const double* U, const double* Vx, const double* Vl);
/* *W = a window whose full-width pass had to run in the dense form for want of such factors (0: none left; each window is
 * handed out once per time it is met); the host wrapper factors M for it and calls nnmpc_qp_set_farfield (one-time setup,
 * like the inverse itself).  The library never factors M itself: a caller that binds the C ABI directly -- or drives
 * nnmpc_chain_run -- has to poll this after its calls (at most 16 windows are queued), otherwise its full-width passes stay in
 * the dense form (correct, 2.6 x the flops at the CDU size). */
int nnmpc_qp_farfield_missing(nnmpc_qp* h, int32_t* W);

/* out (B x nu) = u[:, 0:nu] + us for HBM-resident sequences u (B rows of ldu doubles): the absolute first moves, i.e.
 * get_control_sequence's "+ tile(us)" (lib/linearMPC.py:689) restricted to what simulate_offline keeps (:856).
 * us may be NULL.  Device pointers; returns after the kernel has finished. */
int nnmpc_qp_first_moves(const double* u, int64_t ldu, const double* us, int32_t B, int32_t nu, double* out);
int nnmpc_qp_dims(nnmpc_qp* h, int32_t* n, int32_t* nu, int32_t* n_aug);   /* sizes the handle was created with */
int nnmpc_qp_set_profiling(nnmpc_qp* h, int32_t on);
int nnmpc_qp_get_stats(nnmpc_qp* h, nnmpc_qp_stats* out, int32_t reset);

/* Kernel-level test hook: factor K_b = mask_b mask_b' o P + diag(dvec_b) and
 * solve K_b sol_b = rhs_b for b < B <= max_batch (all B x n, host pointers). */
int nnmpc_qp_debug_factor_solve(nnmpc_qp* h, int32_t B, const float* dvec, const float* mask,
                                const float* rhs, float* sol);

/* Structured NN controller.  dims = [d_in, h1, ..., h_{L-1}, nu]; W[l] is
 * dims[l] x dims[l+1] row-major (Keras kernel layout), b[l] has dims[l+1]
 * entries for l < L-1 (the output layer has no bias).  with_uprev selects
 * RegulatorLayerWithUprev (d_in = 2 nx + 2 nu) or WithoutUprev (2 nx + nu).
 * xscale (nx) divides x and xs; NULL = ones.  ulb/uub (nu) clip the output;
 * NULL = no clipping (the Keras layer).
 * use_bf16: 0 = f32 MFMA GEMMs; 1 = bf16 operands with f32 accumulation (~2e-2 relative error on the CDU architecture);
 * 2 = split bf16: activations and weights as pairs hi + lo of bf16 numbers, every layer ONE bf16 GEMM of three times the
 * depth (hi hi' + hi lo' + lo hi'; only lo lo' is dropped): f32-grade results (~1e-5) from the bf16 matrix pipes. */
int nnmpc_nn_create(nnmpc_nn** out, int32_t nlayers, const int32_t* dims,
                    const double* const* W, const double* const* b, int32_t nx, int32_t nu,
                    int32_t with_uprev, const double* xscale, const double* ulb,
                    const double* uub, int32_t use_bf16, int32_t max_batch);
int nnmpc_nn_destroy(nnmpc_nn* h);
/* x, xs: B x nx; uprev (ignored when !with_uprev), us: B x nu; u: B x nu */
int nnmpc_nn_forward(nnmpc_nn* h, int32_t B, const double* x, const double* uprev,
                     const double* xs, const double* us, double* u, int32_t ptr_kind);
int nnmpc_nn_last_ms(nnmpc_nn* h, double* gemm_ms, double* total_ms);
/* hipEvent time of the hidden-layer GEMM launches of the last forward and their number (bench.py: roofline of the
 * dominant kernel from its own launches; the reference only has time.time() pairs around the whole call,
 * lib/controller_evaluation.py:849-860) */
int nnmpc_nn_last_hidden_ms(nnmpc_nn* h, double* hidden_ms, int32_t* launches);

/* ---- Lock-step closed-loop chains, device resident  <-  simulate_offline (lib/linearMPC.py:827-880), one OS process
 * per chain in the reference (:814-825).  All nc chains of a task advance together; per step (loop :845-866):
 *     x0 = [x - xs; uprev - us], bounds ulb - us / uub - us   (get_control_sequence :682-689)
 *     regulator QP for all chains in ONE batched solve, warm-started on the previous step's active set shifted by one
 *     stage; ut = useq[0:Nu] + us (:856, :689);  x+ = A x + B ut + Bd d (:860);  uprev+ = ut
 * State (x, uprev), the target pairs (xs, us) and disturbances d of ALL T steps and the recorded trajectories stay in
 * HBM; the host sees nothing until nnmpc_chain_run returns.  Records are [T][nc][.] like the arrays the reference
 * saves per chain (:868-872) with the chain index in the middle. */
typedef struct nnmpc_chain nnmpc_chain;
/* A: nx x nx, B: nx x nu, Bd: nx x nd (row-major, host); ulb/uub: nu; x0: nx, uprev0: nu (every chain starts there,
 * :808-813).  qp: the regulator handle (n_aug = nx + nu); it is borrowed, not owned. */
int nnmpc_chain_create(nnmpc_chain** out, nnmpc_qp* qp, int32_t nc, int32_t nx, int32_t nu, int32_t nd,
                       const double* A, const double* B, const double* Bd, const double* ulb,
                       const double* uub, const double* x0, const double* uprev0);
int nnmpc_chain_destroy(nnmpc_chain* c);
/* xs: T x nc x nx, us: T x nc x nu, d: T x nc x nd  ->  x_rec: T x nc x nx, uprev_rec, u_rec: T x nc x nu (state and
 * previous input BEFORE the move of step t, and that move), status: T x nc, all host or all device (ptr_kind).
 * The chain state carries over between calls (T steps at a time); nnmpc_chain_reset puts every chain back to (x0, uprev0).
 * warm_start = 0 solves every step cold (same results: every solve is certified). */
int nnmpc_chain_run(nnmpc_chain* c, int32_t T, const double* xs, const double* us, const double* d,
                    double* x_rec, double* uprev_rec, double* u_rec, int32_t* status,
                    int32_t warm_start, int32_t ptr_kind);
int nnmpc_chain_reset(nnmpc_chain* c);
/* hipEvent time of the last nnmpc_chain_run and the part of it spent inside the regulator solves */
int nnmpc_chain_last_ms(nnmpc_chain* c, double* total_ms, double* solve_ms);

/* ---- Steady-state target problems, batched  <-  TargetSelector.solve -> cvxopt.solvers.qp(P, q, G, h, A, b)
 * (lib/linearMPC.py:298-311), the other QP of every simulation step (:851).  With F = [I - A; H C] of full column rank
 * the equalities [I - A, -B; HC, 0][xs; us] = b (:262-266) fix xs = Xb b + Xu us and leave nz = Nz equalities E us = Eb b
 * on the inputs, so each (ysp, dhat) pair is the small QP
 *     min 1/2 us' Pr us + (Qb b + Qy y + q0)' us   s.t.  E us = Eb b,  ulb <= us <= uub          (nu variables)
 * with b = tb [ysp; dhat], y = ysp - Cd dhat; the shared matrices are formed once on the host (fp64).  One wave per
 * problem: primal-dual active-set iterations on the KKT system of the free inputs and the equalities (<= nu + nz
 * unknowns, Gaussian elimination with partial pivoting in LDS), fp64 throughout, KKT-certified. */
typedef struct nnmpc_ts nnmpc_ts;
/* Pr: nu x nu (symmetric positive definite), E: nz x nu, lb/ub: nu.  nu + nz <= 64 (the KKT system of a step sits on the 64 lanes of one wave), nz <= 16. */
int nnmpc_ts_create(nnmpc_ts** out, int32_t nu, int32_t nz, const double* Pr, const double* E,
                    const double* lb, const double* ub);
int nnmpc_ts_destroy(nnmpc_ts* h);
/* q: B x nu, e: B x nz  ->  us: B x nu, lam_eq: B x nz (multipliers of the equalities, may be NULL), active: B x nu
 * bytes (0 free / 1 at uub / 2 at ulb, may be NULL), status: B (NNMPC_ST_*) */
int nnmpc_ts_solve_batch(nnmpc_ts* h, int32_t B, const double* q, const double* e, double* us,
                         double* lam_eq, uint8_t* active, int32_t* status, int32_t ptr_kind);

/* ---- Device plumbing: HBM buffers, copies and synchronisation for host programs that bind only this library. */
int nnmpc_device_count(void);
int nnmpc_set_device(int32_t dev);
int nnmpc_device_synchronize(void);
int nnmpc_dev_mem_info(uint64_t* free_bytes, uint64_t* total_bytes);
int nnmpc_dev_malloc(void** out, uint64_t bytes);
int nnmpc_dev_free(void* p);
int nnmpc_dev_memset(void* p, int32_t value, uint64_t bytes);
int nnmpc_memcpy_h2d(void* dst, const void* src, uint64_t bytes);
int nnmpc_memcpy_d2h(void* dst, const void* src, uint64_t bytes);
int nnmpc_memcpy_d2d(void* dst, const void* src, uint64_t bytes);
int nnmpc_host_alloc_pinned(void** out, uint64_t bytes);
int nnmpc_host_free_pinned(void* p);

/* ---- One process per GPU, results gathered over xGMI (RCCL)  <-  the reference's parallel model: independent OS
 * processes / cluster jobs on contiguous slices of the scenario signal (OfflineSimulator._split_scenarios,
 * generate_data, lib/linearMPC.py:786-825) whose per-task files are concatenated afterwards (_post_process_data,
 * lib/controller_evaluation.py:273-295).  Here every rank solves its contiguous shard with no communication and the
 * first moves meet on one rank in ONE gather.  nnmpc_comm_unique_id is called by one rank, the 128 bytes reach the
 * others by any side channel (bench.py: a file under /tmp, all ranks are on one node), then every rank calls
 * nnmpc_comm_init with the HIP device it will use already current (nnmpc_set_device). */
typedef struct nnmpc_comm nnmpc_comm;
int nnmpc_comm_unique_id(void* id128);
int nnmpc_comm_init(nnmpc_comm** out, const void* id128, int32_t rank, int32_t world);
int nnmpc_comm_destroy(nnmpc_comm* c);
int nnmpc_comm_rank(nnmpc_comm* c);
int nnmpc_comm_world(nnmpc_comm* c);
/* send: rows[rank] x row_doubles (device) -> recv on root: sum(rows) x row_doubles (device), rank blocks in rank
 * order; rows: world entries (host), the same on every rank; ragged shards allowed.  Blocking. */
int nnmpc_comm_gather_rows(nnmpc_comm* c, const double* send, const int64_t* rows, int32_t row_doubles,
                           double* recv, int32_t root);
int nnmpc_comm_allreduce_max(nnmpc_comm* c, double* value);   /* host scalar in/out */
int nnmpc_comm_barrier(nnmpc_comm* c);                        /* device idle and every rank arrived */

#ifdef __cplusplus
}
#endif
#endif
