/* =============================================================================
 * celerite2_amd.h -- C-ABI of libcelerite2_amd.so (MI355X / gfx950 HIP backend)
 *
 * Drop-in boundary for ONE hot path of exoplanet-dev/celerite2: the O(N)
 * semiseparable GP linear algebra (factor / solve_* / matmul_* /
 * general_matmul_* and the *_rev reverse-mode passes).  Each entry point cites
 * the reference interface it replaces (paths relative to the reference repo).
 *
 * Conventions (all entry points)
 *   - float64 only, C-contiguous row-major, exactly like the reference's
 *     `py::array_t<double, py::array::c_style>` arguments
 *     (python/celerite2/driver.cpp:13-21).
 *   - Caller allocates every output and workspace; the library keeps no state
 *     between calls and allocates no user-visible memory (driver.cpp:13-64).
 *     Internal scratch of a few entry points (time-parallel forms, many
 *     right-hand sides) is a stream-ordered temporary from a memory pool the
 *     LIBRARY owns (one per device, up to 1 GiB kept cached between calls); the
 *     device's default pool, which the process shares with everybody else, is
 *     never reconfigured.  The c2h_* host entry points keep one staging arena,
 *     pinned bounce buffer and stream PER CALLING THREAD (re-entrant).
 *   - Exact-pointer aliasing the reference allows is allowed here too:
 *     d == a and W == V for factor (forward.hpp:55-58), Z == Y for
 *     solve_* / matmul_* (numpy.py:95-108).
 *   - Return value: C2_OK (0) or a negative C2_ERR_* code.  A non positive
 *     definite matrix is NOT an error code: `flag[b]` receives the first row
 *     index n >= 1 with d[n] <= 0, or 0 on success (forward.hpp:128,134), and
 *     rows 0..n of d / 0..n-1 of W are already written, as in the reference.
 *
 * Two families
 *   c2_*   device entry points: every pointer is a DEVICE pointer, there is a
 *          leading batch dimension B (B independent series, contiguous
 *          batch-major: t (B,N) or shared (N,), c (B,J) or shared (J,),
 *          a (B,N), U (B,N,J), Y (B,N,nrhs), S (B,N,J,J), F (B,N,J,nrhs) ...),
 *          and the launch is asynchronous on `stream` (a hipStream_t passed as
 *          void*; NULL = the default stream).  `t_bs` / `c_bs` are the batch
 *          strides of t and c in elements (N / J, or 0 when shared by the batch).
 *   c2h_*  host entry points: every pointer is a HOST pointer, B == 1, the call
 *          stages through device memory, runs the same kernels and returns
 *          after the results are back on the host.  These are what the
 *          `driver` / `backprop` pybind11 modules bind.
 *
 * Width limit: 1 <= J <= C2_MAX_WIDTH = 128.  Up to C2_FAST_WIDTH = 32 (the
 * reference's CELERITE_MAX_WIDTH, c++/include/celerite2/terms.hpp:10-12: the
 * widths it instantiates at compile time) the tuned kernels run; 33 .. 128 --
 * the reference's Eigen::Dynamic path, python/celerite2/driver.hpp:98-99 --
 * run on the workgroup-per-series kernels of csrc/c2_wide.hip.  J > 128 returns
 * C2_ERR_UNSUPPORTED, and so do the 2-D and coefficient-level extensions
 * (c2_kron_*, c2_loglik_terms*) above 32.
 * ============================================================================= */
#ifndef CELERITE2_AMD_H_
#define CELERITE2_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2_OK 0
#define C2_ERR_INVALID (-1)     /* bad size / null pointer ("Invalid shape", driver.cpp:40-46) */
#define C2_ERR_UNSUPPORTED (-2) /* J > C2_MAX_WIDTH (or an extension entry point beyond C2_FAST_WIDTH) */
#define C2_ERR_HIP (-3)         /* HIP runtime error (no device, launch failure, OOM) */
#define C2_MAX_WIDTH 128 /* the reference's dynamic path takes any J (driver.hpp:98-99); here the J x J state must fit LDS */
#define C2_FAST_WIDTH 32 /* widths the tuned kernels cover; beyond: the workgroup-per-series kernels of csrc/c2_wide.hip.
                            The extensions without a reference counterpart (c2_kron_*, c2_loglik_terms*) stop here. */

typedef void *c2_stream_t; /* hipStream_t */

/* Library / device information. */
const char *c2_version(void);
int c2_device_count(void);         /* number of visible HIP devices, 0 if none */
const char *c2_last_error(void);   /* text of the last HIP error seen by this thread */

/* Dispatch options (celerite2_amd/csrc/c2_dispatch.hpp; no counterpart in the reference, whose code has one formulation
 * per op).  Which formulation an entry point runs -- row by row, parallel along time, which lane mapping -- follows from
 * the shape through one table of switches and measured thresholds.  The table is initialised from the environment ONCE,
 * when the library is loaded (variable names in INTEGRATION.md section 5); afterwards only c2_set_option changes it:
 * `name` is the option's name or its environment variable, `value` its new value as text, NULL / "" = back to the
 * default (switches: to the automatic choice).  Every alternative is parity-tested: options move speed, not results
 * (beyond rounding).  Not thread-safe against concurrent launches that depend on the option being changed.
 * c2_options_reload_env re-reads the environment (test harnesses that edit os.environ after loading the library). */
int c2_set_option(const char *name, const char *value);
int c2_get_option(const char *name, double *value, int *is_set);
int c2_option_count(void);
int c2_option_info(int index, const char **name, const char **env, double *default_value, int *is_switch,
                   const char **doc, const char **measured);
void c2_options_reload_env(void);

/* ---------------------------------------------------------------------------
 * DEVICE entry points (batched, asynchronous)
 * ------------------------------------------------------------------------- */

/* core::factor  -- c++/include/celerite2/forward.hpp:69-135 (with workspace S)
 * and interface.hpp:37-48 (S == NULL).  driver.factor (driver.cpp:13-64),
 * backprop.factor_fwd (backprop.cpp:12-67).
 * d (B,N), W (B,N,J), S (B,N,J,J) with S[n, i + J*j] = Sn(i,j), flag (B,) int32. */
int c2_factor(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
              const double *a, const double *U, const double *V, double *d, double *W, double *S /* nullable */,
              int32_t *flag, c2_stream_t stream);

/* core::solve_lower / solve_upper -- forward.hpp:156-170, 193-207 (F nullable:
 * interface.hpp:70-80, 102-112).  Z = L^-1 Y / L^-T Y, L = I + tril(U W^T).
 * driver.solve_lower/upper (driver.cpp:66-178).  F[n, j + J*k] = Fn(j,k). */
int c2_solve_lower(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                   int64_t c_bs, const double *U, const double *W, const double *Y, double *Z,
                   double *F /* nullable */, c2_stream_t stream);
int c2_solve_upper(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                   int64_t c_bs, const double *U, const double *W, const double *Y, double *Z,
                   double *F /* nullable */, c2_stream_t stream);

/* core::matmul_lower / matmul_upper -- forward.hpp:228-239, 260-271.
 * Z += tril(U V^T) Y / Z += triu(V U^T) Y: ACCUMULATES into the caller's Z
 * (driver.cpp:180-292).  The backprop *_fwd variants zero Z first
 * (backprop.cpp:505,511): pass zero_z != 0. */
int c2_matmul_lower(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                    int64_t c_bs, const double *U, const double *V, const double *Y, double *Z,
                    double *F /* nullable */, int zero_z, c2_stream_t stream);
int c2_matmul_upper(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                    int64_t c_bs, const double *U, const double *V, const double *Y, double *Z,
                    double *F /* nullable */, int zero_z, c2_stream_t stream);

/* core::general_matmul_lower / upper -- forward.hpp:285-332, 346-392
 * (driver.cpp:294-420, backprop.cpp:761-901).  t1 (B,N) / t2 (B,M) sorted;
 * U (B,N,J), V (B,M,J), Y (B,M,nrhs), Z (B,N,nrhs) accumulated,
 * F (B,M,J,nrhs) nullable, row-major F[m, j*nrhs + k]. */
int c2_general_matmul_lower(int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1,
                            int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c, int64_t c_bs,
                            const double *U, const double *V, const double *Y, double *Z, double *F /* nullable */,
                            int zero_z, c2_stream_t stream);
int c2_general_matmul_upper(int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1,
                            int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c, int64_t c_bs,
                            const double *U, const double *V, const double *Y, double *Z, double *F /* nullable */,
                            int zero_z, c2_stream_t stream);

/* core::factor_rev -- c++/include/celerite2/reverse.hpp:10-85
 * (backprop.factor_rev, backprop.cpp:68-150).  Outputs fully overwritten:
 * bt (B,N), bc (B,J), ba (B,N), bU (B,N,J), bV (B,N,J).  S must be the workspace of c2_factor for these d, W (as in the
 * reference); on small batches of series of 512 rows and more the states are replayed from d, W instead of read from S
 * (the reverse pass parallel along time), which is the same thing for a consistent S. */
int c2_factor_rev(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                  const double *a, const double *U, const double *V, const double *d, const double *W,
                  const double *S, const double *bd, const double *bW, double *bt, double *bc, double *ba,
                  double *bU, double *bV, c2_stream_t stream);

/* core::solve_lower_rev / solve_upper_rev / matmul_lower_rev / matmul_upper_rev
 * -- reverse.hpp:87-217 over internal::forward_rev / backward_rev
 * (internal.hpp:191-303); backprop.cpp:216-302, 368-454, 520-606, 672-758.
 * Outputs fully overwritten: bt (B,N), bc (B,J), bU, bW|bV (B,N,J), bY (B,N,nrhs). */
int c2_solve_lower_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                       int64_t c_bs, const double *U, const double *W, const double *Y, const double *Z,
                       const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bW, double *bY,
                       c2_stream_t stream);
int c2_solve_upper_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                       int64_t c_bs, const double *U, const double *W, const double *Y, const double *Z,
                       const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bW, double *bY,
                       c2_stream_t stream);
int c2_matmul_lower_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                        const double *c, int64_t c_bs, const double *U, const double *V, const double *Y,
                        const double *Z, const double *F, const double *bZ, double *bt, double *bc, double *bU,
                        double *bV, double *bY, c2_stream_t stream);
int c2_matmul_upper_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                        const double *c, int64_t c_bs, const double *U, const double *V, const double *Y,
                        const double *Z, const double *F, const double *bZ, double *bt, double *bc, double *bU,
                        double *bV, double *bY, c2_stream_t stream);

/* driver::get_celerite_matrices -- python/celerite2/driver.cpp:422-477.
 * Coefficients ar (B,Jr), ac/bc/dc (B,Jc) with batch stride coef_bs in
 * {0 = shared, 1 = per series}; x (B,N) / shared (N,) via x_bs; diag (B,N).
 * Outputs a (B,N), U (B,N,J), V (B,N,J), J = Jr + 2 Jc, complex terms at
 * interleaved columns (Jr+2j, Jr+2j+1).  No sortedness precondition, as in the
 * reference's elementwise recipe: phases |dc x| beyond 1.6e6 (raw Julian dates)
 * take the library's large-argument sincos -- whole terms when the two ENDS of
 * a series' grid say so, single rows of an unsorted grid otherwise (a second,
 * row-parallel kernel that returns at once in the common case). */
int c2_get_celerite_matrices(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                             const double *bc, const double *dc, int coef_batched, const double *x, int64_t x_bs,
                             const double *diag, double *a, double *U, double *V, c2_stream_t stream);

/* Term.get_value on two grids -- python/celerite2/terms.py:58-79 evaluated at
 * tau = t1[n] - t2[m]: K[b, n, m] = sum_r ar e^{-cr |tau|} + sum_k e^{-cc |tau|}
 * (ac cos(dc |tau|) + bc sin(dc |tau|)).  What the conditional distribution
 * forms before it calls the solves (core.py:46-54 KxsT, :142-148 the prior
 * covariance of the prediction grid).  Coefficients shared or per series
 * (coef_batched); t1 (B,N) / shared (N,) via t1_bs, t2 (B,M) / (M,) via t2_bs;
 * K (B,N,M). */
int c2_kernel_values(int64_t B, int64_t N, int64_t M, int64_t Jr, int64_t Jc, const double *ar, const double *cr,
                     const double *ac, const double *bc, const double *cc, const double *dc, int coef_batched,
                     const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs, double *K, c2_stream_t stream);

/* out[b, m] = sum_n Z[b, n, m]^2 / d[b, n]  (Z (B,N,M), d (B,N), out (B,M)): the quadratic form of the PREDICTIVE VARIANCE,
 * core.py:134-140 `kernel.get_value(0) - diagdot(KxsT, Kinv_KxsT)` (numpy.py:24-25), evaluated from the lower solve alone:
 * with K = L D L^T, diag(Kxs K^-1 Kxs^T)_m = sum_n (L^-1 KxsT)_nm^2 / d_n -- Z = c2_solve_lower(KxsT); the reference reaches
 * the same number through apply_inverse (both solves) and a second pass over the two N x M arrays. */
int c2_colsumsq_over_d(int64_t B, int64_t N, int64_t M, const double *Z, const double *d, double *out, c2_stream_t stream);

/* Fused log-likelihood -- the assembly the reference's callers perform around
 * factor + solve_lower (python/celerite2/numpy.py:66-87,104-109, core.py:407-428):
 *   ll[b] = -1/2 (sum log d + N log 2pi) - 1/2 sum z^2/d,  z = L^-1 y.
 * No d/W/z is materialised.  flag[b] != 0 -> ll[b] = -inf (numpy.py:78-82). */
int c2_loglik(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
              const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
              c2_stream_t stream);

/* Fused log-likelihood + reverse-mode gradient w.r.t. (t, c, a, U, V, y): the
 * chain factor_fwd -> solve_lower_fwd -> [seeds] -> solve_lower_rev -> factor_rev
 * an autodiff frontend runs (python/celerite2/pymc/ops.py:104-141,
 * pymc/distribution.py:123-128), with per-series outputs bt (B,N), bc (B,J),
 * ba (B,N), bU (B,N,J), bV (B,N,J), by (B,N).  `work` is caller-provided device
 * scratch of c2_loglik_grad_workspace_bytes(B,N,J) bytes (the query follows the
 * dispatch: checkpoints + W rows + (d,z) records for the row-by-row kernels, lane-major
 * records for chip-filling batches, d / W / z / state rows / chunk maps for small
 * batches of long series, which run parallel along time).
 * Agreement with the reference's operation order: the row-by-row forward passes repeat
 * it up to FMA contraction and reduction order (1e-13 on well-conditioned data); their
 * reverse sweeps recover the forward state S_n, F_n of reverse.hpp:52-84 / internal.hpp:225-245
 * by running forward.hpp:115-123 backward from a checkpoint at most 32 rows up wherever the
 * decays in between can be inverted (c_max * span <= 2: errors grow by at most e^4), and by
 * replaying the forward steps from a checkpoint where they cannot (gaps in time) -- decided
 * on the device, per wavefront; gradients within 1e-10 of the largest
 * entry of their array either way (1.5e-11 between the two forms on the bench's batch);
 * small batches of long series run PARALLEL ALONG TIME, verified
 * on the device -- what every chunk arrives at is compared with what its
 * neighbour was given, and the row-by-row kernels recompute the batch behind
 * that gate should they disagree beyond 2e-12 -- and are held to
 *   |x - x_ref| <= 1e-10 max|x_ref| + 4 max|x_ref - x_ext|   per gradient array,
 * x_ext = the reference recursion evaluated in extended precision: a float64
 * evaluation in ANY order (the reference's own included) moves by ~0.4 eps
 * kappa^2 of the largest entry, kappa = max a_n / d_n; chunked sums reorder the
 * additions, so entries far below the largest of their array are sums of large
 * terms (profiles/r03_timepar_verification.md; option timepar_cond_limit).
 * A series whose factorisation fails (flag[b] != 0; the reference raises,
 * driver.hpp:13-19) gets ll[b] = -inf and ALL SIX gradients filled with NaN
 * -- defined, never stale memory; the other series of the batch are unaffected. */
size_t c2_loglik_grad_workspace_bytes(int64_t B, int64_t N, int64_t J);
int c2_loglik_grad(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                   const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                   double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                   size_t work_bytes, c2_stream_t stream);

/* Conditioning of the factorisation, per series: kappa[b] = max_n a_n / d_n (+inf where the factorisation fails,
 * flag[b] = the failing row as in c2_factor).  d_n = a_n - U_n S_n U_n^T (forward.hpp:126-128) is a difference: a
 * float64 evaluation of the recursion IN ANY ORDER -- the reference's own included -- carries a rounding error of
 * ~0.4 eps kappa^2 relative to the largest entry of a gradient array (oracle vs its own extended-precision
 * evaluation: tests/test_oracle.py, tools/kappa_sweep.py).  north_star's 1e-10 agreement with the reference is
 * therefore attainable where kappa <~ 1e3 (0.4 * 2.2e-16 * kappa^2 <= 1e-10): a caller who needs to know whether a
 * result can be held to that tolerance asks here.  (bench.py reports the timed batch's largest kappa; the synthetic
 * series of SURVEY.md 8d at N = 4096, J = 8 sit at kappa ~ 290 median, 380 at most: a floor of 1.3e-11.)  Replaces nothing in the reference (which reports no conditioning);
 * costs one `factor` pass on library temporaries (slices of 4096 series). */
int c2_condition(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                 const double *a, const double *U, const double *V, double *kappa, int32_t *flag, c2_stream_t stream);

/* 2-D (multi-band) extension, rank-1 band covariance K = T (x) alpha alpha^T + diag over N epochs x M bands
 * (observations interleaved epoch-major: row n*M + m).  EXTENSION -- the reference has no 2-D code (no core2.hpp;
 * README.md:14-17 only cites the paper), so this entry point replaces nothing and its parity is pinned by the dense
 * Kronecker matrix and by the 1-D recursions on the interleaved series, not by the reference.
 * Inputs: the 1-D celerite matrices of the EPOCH grid built with zero white noise -- t (B,N)|(N,), c (B,J)|(J,),
 * a (B,N) = k(0), U, V (B,N,J) -- plus alpha (B,M)|(M,) (alpha_bs = M or 0), diag (B,N,M) and y (B,N,M).
 * method: C2_KRON_COLLAPSED (each epoch's M bands fold into one effective observation; needs diag > 0; the
 * recursion runs over N rows) or C2_KRON_INTERLEAVED (the 1-D recursions on the N*M series with U' = U (x) alpha).
 * flag[b]: first failing row in the method's own series (epoch index / interleaved row), -1 for a non-positive
 * band variance under the collapsed method -- or for alpha == 0 in every band (the collapse divides by A = sum alpha^2 /
 * D: an epoch whose bands carry no signal has no effective observation; the interleaved method has no such restriction).  Gradients: bt (B,N), bc (B,J), ba (B,N), bU, bV (B,N,J),
 * balpha (B,M) (per series, also when alpha is shared), bdiag (B,N,M), by (B,N,M). */
#define C2_KRON_COLLAPSED 0
#define C2_KRON_INTERLEAVED 1
size_t c2_kron_loglik_workspace_bytes(int64_t B, int64_t N, int64_t M, int64_t J, int method, int grad);
int c2_kron_loglik(int64_t B, int64_t N, int64_t M, int64_t J, const double *t, int64_t t_bs, const double *c,
                   int64_t c_bs, const double *a, const double *U, const double *V, const double *alpha,
                   int64_t alpha_bs, const double *diag, const double *y, double *ll, int32_t *flag, int method,
                   void *work, size_t work_bytes, c2_stream_t stream);
int c2_kron_loglik_grad(int64_t B, int64_t N, int64_t M, int64_t J, const double *t, int64_t t_bs, const double *c,
                        int64_t c_bs, const double *a, const double *U, const double *V, const double *alpha,
                        int64_t alpha_bs, const double *diag, const double *y, double *ll, double *bt, double *bc,
                        double *ba, double *bU, double *bV, double *balpha, double *bdiag, double *by, int32_t *flag,
                        int method, void *work, size_t work_bytes, c2_stream_t stream);

/* Log-likelihood (+ gradient) from the celerite COEFFICIENTS: the chain
 * get_celerite_matrices (driver.cpp:422-477, terms.py:117-177) -> factor -> solve_lower -> reductions and its
 * reverse, which the reference's jax / pymc frontends obtain by autodiff of their term code
 * (python/celerite2/jax/terms.py, pymc/terms.py), kept on the device.  Coefficients ar, cr (B,Jr), ac, bc, cc, dc
 * (B,Jc), all per series (coef_batched = 1) or all shared by the batch (0); x (B,N) / shared (N,) via x_bs;
 * diag, y (B,N).  J = Jr + 2 Jc <= C2_MAX_WIDTH.  Gradients (per series, also for shared coefficients):
 * bar, bcr (B,Jr); bac, bbc, bcc, bdc (B,Jc); bx, bdiag, by (B,N).  A failed series: ll = -inf, NaN gradients.
 * Widths J = 8, 4, 2 and chip-filling batches run kernels that form U_n, V_n inside the recursion (no matrices in memory);
 * everything else the composed chain on matrices kept in `work` (same results).  `work` is
 * sized for either. */
size_t c2_loglik_terms_workspace_bytes(int64_t B, int64_t N, int64_t Jr, int64_t Jc, int grad);
int c2_loglik_terms(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *cr, const double *ac,
                    const double *bc, const double *cc, const double *dc, int coef_batched, const double *x,
                    int64_t x_bs, const double *diag, const double *y, double *ll, int32_t *flag, void *work,
                    size_t work_bytes, c2_stream_t stream);
int c2_loglik_terms_grad(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *cr,
                         const double *ac, const double *bc, const double *cc, const double *dc, int coef_batched,
                         const double *x, int64_t x_bs, const double *diag, const double *y, double *ll, double *bar,
                         double *bcr, double *bac, double *bbc, double *bcc, double *bdc, double *bx, double *bdiag,
                         double *by, int32_t *flag, void *work, size_t work_bytes, c2_stream_t stream);

/* dot_tril -- python/celerite2/numpy.py:100-102: Z = Y * sqrt(d)[:,None];
 * Z += tril(U W^T) Z.  Y == Z allowed. */
int c2_dot_tril(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                int64_t c_bs, const double *U, const double *W, const double *d, const double *Y, double *Z,
                c2_stream_t stream);

/* ---------------------------------------------------------------------------
 * HOST entry points (B == 1, synchronous) -- what celerite2.driver /
 * celerite2.backprop bind.  Same argument meaning as the pybind11 functions of
 * python/celerite2/driver.cpp and backprop.cpp; shapes are passed explicitly.
 * c2h_factor returns C2_OK and stores the reference's flag in *flag.
 * ------------------------------------------------------------------------- */
int c2h_factor(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
               const double *V, double *d, double *W, double *S /* nullable */, int64_t *flag);
int c2h_solve_lower(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                    const double *W, const double *Y, double *Z, double *F /* nullable */);
int c2h_solve_upper(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                    const double *W, const double *Y, double *Z, double *F /* nullable */);
int c2h_matmul_lower(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                     const double *V, const double *Y, double *Z, double *F /* nullable */, int zero_z);
int c2h_matmul_upper(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                     const double *V, const double *Y, double *Z, double *F /* nullable */, int zero_z);
int c2h_general_matmul_lower(int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, const double *t2,
                             const double *c, const double *U, const double *V, const double *Y, double *Z,
                             double *F /* nullable */, int zero_z);
int c2h_general_matmul_upper(int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, const double *t2,
                             const double *c, const double *U, const double *V, const double *Y, double *Z,
                             double *F /* nullable */, int zero_z);
int c2h_factor_rev(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
                   const double *V, const double *d, const double *W, const double *S, const double *bd,
                   const double *bW, double *bt, double *bc, double *ba, double *bU, double *bV);
int c2h_solve_lower_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                        const double *W, const double *Y, const double *Z, const double *F, const double *bZ,
                        double *bt, double *bc, double *bU, double *bW, double *bY);
int c2h_solve_upper_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                        const double *W, const double *Y, const double *Z, const double *F, const double *bZ,
                        double *bt, double *bc, double *bU, double *bW, double *bY);
int c2h_matmul_lower_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                         const double *V, const double *Y, const double *Z, const double *F, const double *bZ,
                         double *bt, double *bc, double *bU, double *bV, double *bY);
int c2h_matmul_upper_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                         const double *V, const double *Y, const double *Z, const double *F, const double *bZ,
                         double *bt, double *bc, double *bU, double *bV, double *bY);
int c2h_get_celerite_matrices(int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                              const double *bc, const double *dc, const double *x, const double *diag, double *a,
                              double *U, double *V);

/* The c2h_* entry points stage their arguments through a per-thread device arena + pinned bounce buffer that is kept
 * between calls (memory beyond 256 MiB is given back when the call returns).  This releases everything the CALLING
 * thread holds; its next c2h_* call allocates again.  (The reference is stateless: nothing to replace.) */
void c2h_release_thread_cache(void);

#ifdef __cplusplus
}
#endif
#endif /* CELERITE2_AMD_H_ */
