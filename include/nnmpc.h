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
  int32_t asm_max_rounds;    /* ... and its round budget; 0 = 200 (all-at-once exchanges settle in ~5 rounds; the
                                single-exchange fallback against cycling is finite but can take hundreds of
                                ~0.2 ms rounds on ill-conditioned Hessians with half of the bounds active: raise
                                it for such problems; what is left goes to the PDIP path) */
  int32_t asm_f32_rounds;    /* 0 = the rounds run in f32 until a problem's set settles, then in fp64 (only fp64
                                results are accepted); < 0 = fp64 from the first round */
  int32_t seg_max;           /* problems per segment (one lock-step pass; ~0.3 MB of workspace each at n = 4480);
                                0 = as many as a quarter of the free HBM holds (at most 2^20) */
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

/* Enables the shared-inverse active-set pass: Hinv = P^-1 (n x n, fp64), Kunc = -Hinv tq (n x n_aug).
 * All samples share P (reference lib/linearMPC.py:472), so on an active set A the equality-constrained
 * optimum is x = x_unc - Hinv[:,A] lam with lam = (Hinv_AA)^-1 (x_unc,A - b_A): a primal-dual active-set
 * iteration needs no n^3 factorisation.  The inverse is verified once on the device (|P Pinv - I|,
 * |P Kunc + tq|); those bounds certify each result's KKT residual and multiplier signs, results too
 * close to call are re-checked against P itself in fp64, and problems the pass cannot finish go
 * through the PDIP path. */
int nnmpc_qp_set_inverse(nnmpc_qp* h, const double* Hinv, const double* Kunc);

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
 * NULL = no clipping (the Keras layer). */
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

#ifdef __cplusplus
}
#endif
#endif
