/*
 * mxf_gp.h -- C ABI of libmxf_gp.so: the MI355X (gfx950) implementation of MXFusion's
 * Gaussian-process + SVI hot path.
 *
 * The reference (amzn/MXFusion v0.3.1) is pure Python on Apache MXNet; the arithmetic of this path
 * lives in MXNet operators called from four Python files.  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference root).  The reference-side
 * binding a maintainer would add (a ctypes stub inside the reference's kernel / module classes) is
 * shown in INTEGRATION.md.
 *
 * Conventions (all entry points)
 *   - extern "C", plain pointers and sizes; no C++ / torch types.
 *   - return int status: 0 ok; <0 bad argument / runtime error (text via mxf_last_error);
 *     LAPACK-style "first non-PD leading minor" is reported asynchronously through a device-side
 *     int* info argument (never a host sync inside the library, like MXNet's lazy error).
 *   - the CALLER owns every data buffer (device pointers); the library owns only the opaque handle
 *     and its scratch workspace.  Row-major, leading sample axis S.  Strides are in ELEMENTS;
 *     a sample stride of 0 broadcasts that operand over S (the reference physically broadcasts,
 *     components/variables/runtime_variable.py:82-118).
 *   - dtype: MXF_F32 or MXF_F64 for every array of the call.
 *   - every call takes the hipStream_t (as void*) it is enqueued on and is asynchronous w.r.t. the host.
 *   - one handle per (thread, device); calls on one handle are not re-entrant.
 */
#ifndef MXF_GP_H
#define MXF_GP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxf_ctx* mxf_handle;

enum { MXF_F32 = 0, MXF_F64 = 1 };

/* covariance-function kinds (components/distributions/gp/kernels/{rbf,matern,linear,static}.py) */
enum { MXF_K_RBF = 0, MXF_K_MATERN12 = 1, MXF_K_MATERN32 = 2, MXF_K_MATERN52 = 3,
       MXF_K_LINEAR = 4, MXF_K_BIAS = 5, MXF_K_WHITE = 6 };

/* how mxf_gram combines with the existing contents of K_out:
 * AddKernel / MultiplyKernel (kernels/add_kernel.py:44-68, multiply_kernel.py:44-67) */
enum { MXF_WRITE = 0, MXF_ACC_ADD = 1, MXF_ACC_MUL = 2 };

int mxf_version(void);
int mxf_create(int device, mxf_handle* out);
int mxf_destroy(mxf_handle h);
const char* mxf_last_error(mxf_handle h);
/* bytes of scratch currently held by the handle */
int64_t mxf_workspace_bytes(mxf_handle h);
/* Counts (re-)allocations of the handle's scratch.  The library owns its workspace and grows it on demand (hipFree + hipMalloc, never
 * inside a stream capture); a caller that captured launches into a hipGraph must re-capture when this number has changed, because the
 * captured kernels carry the old scratch addresses.  (The reference has no counterpart: MXNet owns its temporaries.) */
int64_t mxf_workspace_generation(mxf_handle h);
/* 1-norm condition number |Kuu + jitter I|_1 |(Kuu + jitter I)^-1|_1 of the last mxf_svgp_logpdf training call on this handle (both norms
 * are computed on the device next to the factorisation; this call copies them to the host, i.e. it synchronises).  The float32 streaming
 * form of the bound applies H0 = Kuu^-1 - Kuu^-1 Su Kuu^-1 explicitly, so its rounding error grows like cond * 2^-24: measured ELBO
 * agreement with float64 is 3e-6 at cond 1.4e3 and 2e-3 at 5e4 (tests/probes/f32_accuracy.py) -- above ~3e3 use float64.  0 if there was
 * no such call.  (No reference counterpart: svgp_regression.py:83-92 solves with the Cholesky factor, in whatever dtype the model has.) */
int mxf_svgp_last_cond(mxf_handle h, double* cond1_out);
/* The same quantity WITHOUT synchronising: the running maximum of the condition numbers that the training calls finished so far on this
 * handle have published (the last launch of every training call folds its cond_1 into one pinned, device-visible host word); reset != 0
 * clears it after the read.  A caller polls it every step for free and lags by at most the calls still in flight -- the float32 guard of
 * mxfusion_amd's SVGP module (automatic switch of the streaming stage to float64 above ~3e3) is built on it.  No reference counterpart. */
int mxf_svgp_cond_nowait(mxf_handle h, double* cond1_max_out, int reset);
/* Float32 streaming form of the NEXT mxf_svgp_logpdf / _sampled training calls on this handle, and the slot (0 <= cond_slot < 64) they
 * publish their condition number into.  MXF_SVGP_EXPLICIT (default): T = H0 Kuf with the explicit inverse H0 = Kuu^-1 - Kuu^-1 Su Kuu^-1
 * (two full-width split GEMMs; error ~ cond 2^-24).  MXF_SVGP_WHITENED: the factorised form the reference evaluates
 * (svgp_regression.py:83-92: trsm with the Cholesky factor) on the split GEMMs -- V = L^-1 Kuf (triangular product, written directly as
 * f16 planes), Phi = V V^T, T = L^-T (I - A_s A_s^T) V, U = (L^-1 mu)^T V -- three full-width products; every operand is bounded by
 * |L^-1| ~ sqrt(cond) and |v_n|^2 <= k_nn, so the float32 error grows like sqrt(cond) 2^-24 (ELBO 1e-7 at cond_2 1e5, 1e-6 at 4e6).
 * Applies to float32 training calls (want_grad) on shapes mxf_svgp_whitened_ok accepts; such a call on another shape fails with -3.
 * float64 calls ignore the form.  No reference counterpart (the reference has one dtype and one form).                                   */
enum { MXF_SVGP_EXPLICIT = 0, MXF_SVGP_WHITENED = 1 };
int mxf_svgp_configure(mxf_handle h, int form, int cond_slot);
/* 1 if the whitened float32 form covers this call shape (M % 128 == 0, S B % 256 == 0, Q <= 16, P <= 8, sampled X or S == 1), else 0 */
int mxf_svgp_whitened_ok(int dtype, int S, int64_t B, int64_t M, int Q, int P, int64_t strideS_X);
/* One slot's condition words without synchronising: the LAST value a finished call published into it and the running maximum (either
 * pointer may be NULL); reset != 0 clears both after the read.  A module instance that owns a slot sees its own Kuu only, whatever other
 * SVGP modules run on the same handle (mxf_svgp_cond_nowait = the maximum over all slots).                                               */
int mxf_svgp_cond_slot(mxf_handle h, int slot, double* last_out, double* max_out, int reset);
/* In-step durations of the bulk kernels of the LAST mxf_svgp_logpdf training call on this handle, measured with HIP events on the stream
 * each kernel runs on (enable with mxf_svgp_timing(h, 1); off by default).  mxf_svgp_timing_read synchronises the device and fills
 * ms_out[8] (-1 = not part of that call): [0] first Gram-planes pass (explicit form: Kuf planes; whitened: Kfu planes), [1] Psi2 / Phi
 * product, [2] second planes pass (explicit: Kfu planes + U; whitened: transposition of V + U), [3] T product, [4] fused reverse pass,
 * [5] the float64 core chain (Kuu ... H0 planes), [6] V = L^-1 Kuf (whitened), [7] the whole call on the caller's stream.
 * bench.py divides algorithmic bytes / flops by these (roofline_planes, step_breakdown_ms).  No reference counterpart.                  */
int mxf_svgp_timing(mxf_handle h, int enable);
int mxf_svgp_timing_read(mxf_handle h, double* ms_out);

/* out[0] = sum_i g[i] if the n values agree to 1e-6 relative, NaN otherwise.  The fused composites return the gradients of
 * gscale * sum_s logL[s] with ONE weight: the reverse-mode bridge uses this to scale them by the upstream gradient of mean_S(logL)
 * (factor_graph.py:233) and to poison the result -- on the device, no host sync -- if a caller weights the samples unequally.            */
int mxf_uniform_sum(mxf_handle h, int dtype, int64_t n, const void* g, void* out, void* stream);

/* MXNet SGD as driven by gluon.Trainer.step with optimizer='sgd' (the `optimizer` argument of batch_loop.py:29-44 is handed to the Trainer
 * by name): g = rescale_grad * grad + wd * w;  mom == NULL: w -= lr g;  else mom = momentum * mom - lr * g, w += mom.                     */
int mxf_sgd_step(mxf_handle h, int dtype, int64_t n, void* w, const void* g, void* mom, double lr, double momentum, double wd,
                 double rescale_grad, void* stream);

/* The other update rules gluon.Trainer is driven with by name through the same seam (batch_loop.py:46-49, minibatch_loop.py:71-74): MXNet 1.x
 * 'rmsprop' (non-centred), 'adagrad', 'adadelta', 'nag' on a flat buffer -- s1 / s2 are the optimiser's state buffers (n elements each,
 * zero-initialised by the caller; s2 only for adadelta), p1 = gamma1 / rho / momentum.  Rules in csrc/elementwise.hip (API knowledge of
 * MXNet; the reference holds no vector for them).                                                                                      */
#define MXF_OPT_RMSPROP 1
#define MXF_OPT_ADAGRAD 2
#define MXF_OPT_ADADELTA 3
#define MXF_OPT_NAG 4
int mxf_opt_step(mxf_handle h, int kind, int dtype, int64_t n, void* w, const void* g, void* s1, void* s2, double lr, double p1, double epsilon,
                 double wd, double rescale_grad, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU gradient exchange (RCCL over xGMI), SURVEY.md section 8(b)/(e).  The reference has no multi-device path (one MXNet context
 * per Inference object); north_star shards the Monte-Carlo samples of StochasticVariationalInference.compute
 * (inference/variational.py:15-26) over the GPUs -- one process and one handle per GPU -- and sums the flat gradient ONCE per step
 * before the optimiser update (what Trainer.step of inference/batch_loop.py:46-60 would do with a kvstore).  librccl is opened on first
 * use; single-GPU callers never load it.
 *   mxf_comm_unique_id : rank 0 creates the 128-byte rendezvous id; the caller distributes it to the other ranks by its own means
 *                        (MPI, a file, the launcher's environment).
 *   mxf_comm_init      : collective over all ranks; binds a communicator of `nranks` ranks to this handle's device.
 *   mxf_allreduce_sum  : in-place sum over ranks of `count` elements (MXF_F32 / MXF_F64) at the device pointer `buf`, ordered on `stream`.
 *   mxf_bcast          : in-place broadcast from `root` (initial parameters; the minibatch permutation of config 4).
 *   mxf_comm_destroy   : releases the communicator (mxf_destroy does it too).
 * Returns 0, or < 0 with mxf_last_error (-6: librccl not loadable, -7: an RCCL call failed). */
#define MXF_COMM_ID_BYTES 128
int mxf_comm_unique_id(mxf_handle h, void* id_out /* MXF_COMM_ID_BYTES, host */);
int mxf_comm_init(mxf_handle h, int nranks, int rank, const void* id /* MXF_COMM_ID_BYTES, host */);
int mxf_allreduce_sum(mxf_handle h, int dtype, void* buf, int64_t count, void* stream);
int mxf_bcast(mxf_handle h, int dtype, void* buf, int64_t count, int root, void* stream);
int mxf_comm_destroy(mxf_handle h);

/* ---------------------------------------------------------------------------------------------
 * Gram build.  Replaces Kernel.K -> _compute_K (kernels/kernel.py:96-123), i.e.
 * StationaryKernel._compute_R2 (kernels/stationary.py:74-107) + RBF._compute_K (rbf.py:71-72) /
 * Matern{12,32,52}._compute_K (matern.py:84-88,116-120,148-151) / Linear (linear.py:59-89) /
 * Bias, White (static.py:56-74,125-150), fused into ONE pass that writes K once.
 *   X  : (S|1, N,  Q)   X2 : (S|1, N2, Q) or NULL (=> X2 = X, square Gram)
 *   lengthscale : (S|1, Q) if ard else (S|1, 1); for MXF_K_LINEAR it is `variances`; unused for BIAS/WHITE
 *   variance    : (S|1, 1)                       (unused for MXF_K_LINEAR)
 *   diag_add    : optional (S|1, 1) device scalar added to the diagonal (square Gram only):
 *                 the "+ eye*noise_var" of gp_regression.py:55-57;  jitter: host scalar, same place
 *                 (gp_regression.py:58-60, svgp_regression.py:70-72)
 *   K_out : (S, N, N2) with row stride ldk.
 * r^2 is evaluated as sum_q ((x_q - z_q)/l_q)^2 (difference form; the reference's expansion form
 * agrees to rounding in float64 and is less accurate in float32).                                  */
int mxf_gram(mxf_handle h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q,
             const void* X, int64_t strideS_X, const void* X2, int64_t strideS_X2,
             const void* lengthscale, int ard, int64_t strideS_ls,
             const void* variance, int64_t strideS_var,
             const void* diag_add, int64_t strideS_diag, double jitter, int mode,
             void* K_out, int64_t ldk, int64_t strideS_K, void* stream);

/* K = k1(X, X2) + k2(X, X2) (op = MXF_ACC_ADD) or k1 * k2 (MXF_ACC_MUL) for TWO stationary kernels on the same inputs in ONE pass and ONE
 * write: AddKernel / MultiplyKernel._compute_K (kernels/add_kernel.py:44-68, multiply_kernel.py:44-67), which in the reference materialise
 * every sub-kernel's Gram and combine them (three N x N2 passes for two kernels).  Both covariances are formed from the same coordinate
 * differences (difference first, each kernel's length-scales after).  Arguments as mxf_gram, one (kind, lengthscale, ard, variance) set per
 * kernel; diag_add / jitter on the diagonal of a square Gram; Q <= 16.  Its reverse mode is mxf_gram_bwd per sub-kernel (ADD: with dK;
 * MUL: with dK * the other kernel's Gram).                                                                                              */
int mxf_gram2(mxf_handle h, int kind1, int kind2, int op, int dtype, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t strideS_X,
              const void* X2, int64_t strideS_X2, const void* lengthscale1, int ard1, int64_t strideS_ls1, const void* variance1,
              int64_t strideS_var1, const void* lengthscale2, int ard2, int64_t strideS_ls2, const void* variance2, int64_t strideS_var2,
              const void* diag_add, int64_t strideS_diag, double jitter, void* K_out, int64_t ldk, int64_t strideS_K, void* stream);

/* Reverse mode of mxf_gram for the stationary kinds (what MXNet autograd does through
 * stationary.py:92-106 + rbf.py:71-72 / matern.py): given dK (S,N,N2) accumulates
 *   dX (S,N,Q), dX2 (S,N2,Q) [NULL when X2==NULL: both roles flow into dX], dls (S, Q|1), dvar (S,1).
 * Outputs are ACCUMULATED INTO (caller zeroes them); any output pointer may be NULL.               */
int mxf_gram_bwd(mxf_handle h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q,
                 const void* X, int64_t strideS_X, const void* X2, int64_t strideS_X2,
                 const void* lengthscale, int ard, int64_t strideS_ls,
                 const void* variance, int64_t strideS_var,
                 const void* dK, int64_t lddk, int64_t strideS_dK,
                 void* dX, void* dX2, void* dls, void* dvar, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense building blocks (MXNet linalg.* call sites).  All batched over S with strides.  */

/* C = alpha*op(A)*op(B) + beta*C  -- linalg.gemm2 / linalg.syrk (svgp_regression.py:76,82,89,90 ...) */
int mxf_gemm(mxf_handle h, int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
             double alpha, const void* A, int64_t lda, int64_t strideA,
             const void* B, int64_t ldb, int64_t strideB,
             double beta, void* C, int64_t ldc, int64_t strideC, int batch, void* stream);

/* The same product, C = alpha A B^T + beta C for k-contiguous float32 operands A (M x K), B (N x K), on the bf16 matrix pipe:
 * every f32 operand is split exactly into three bf16 terms and the six leading bf16 products accumulate in f32 (f32-equivalent
 * accuracy, 6/16 of the f32-MFMA cost; gemm_split.hip).  Replaces linalg.gemm2(A, B, False, True) / linalg.syrk call sites whose
 * operands are float32 (svgp_regression.py:88-107).  lower_only: only blocks / entries on or below the diagonal are written.     */
int mxf_gemm_f32x3(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda, const void* B, int64_t ldb,
                   double beta, void* C, int64_t ldc, int lower_only, void* stream);

/* The same product from TWO scaled f16 terms per operand and THREE matrix-pipe products (hi hi' + hi lo' + lo hi'; each operand is
 * scaled by the power of two that puts its largest magnitude at [2^13, 2^14), so the low term keeps its 11 bits over 2^18 of dynamic
 * range): the f32 MFMA's product accuracy at 3/16 of its cost.  This is the form the SVGP training step uses for Psi2 = Kuf Kuf^T and
 * T = H0 Kuf (MXF_SPLIT_MODE=bf16x3 selects the three-term form there).  Normwise f32 accuracy.                                      */
int mxf_gemm_f16x2(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda, const void* B, int64_t ldb,
                   double beta, void* C, int64_t ldc, int lower_only, void* stream);
/* Its two halves: mxf_f16x2_split writes the two planes of an (R x K) operand (2 * mxf_f32x3_plane_elems(R, K) 16-bit elements) and
 * the 32-bit word holding the bit pattern of max |X| (the operand's scale); mxf_gemm_f16x2_planes multiplies two split operands.
 * mxf_gemm_f16x2_planes only: lower_only = 2 writes the FULL product with C in 16-column blocks -- element (m, n) at
 * ((n / 16) * M + m) * 16 + n % 16, ldc ignored, N % 16 == 0, beta == 0 -- the layout the SVGP training step keeps T = H0 Kuf in.         */
int mxf_f16x2_split(mxf_handle h, int64_t R, int64_t K, const void* X, int64_t ld, void* planes, void* maxword, void* stream);
int mxf_gemm_f16x2_planes(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* A_maxword,
                          const void* B_planes, const void* B_maxword, double beta, void* C, int64_t ldc, int lower_only, void* stream);
/* The same product with the second operand stored the OTHER way round (r06): C (M x N) = alpha * A (M x K) * Bt (K x N), Bt_planes =
 * mxf_f16x2_split of the (K x N) matrix Bt itself (rows = contraction index).  The SVGP training step uses it for T = H0 Kuf on the SAME
 * planes of Kuf that Psi2 = Kuf Kuf^T reads (svgp_regression.py:85-90 needs Kuf in both roles; until r05 it was written twice).
 * M % 256 == 0, N % 256 == 0, K % 16 == 0, K >= 48.  blocked != 0: C in 16-column blocks (layout of lower_only = 2 above), else row-major
 * with ldc == N.  w (K floats) and U (N floats), both or neither: the same launch forms U[n] = sum_k w[k] Bt[k][n] (K <= 2048).          */
int mxf_gemm_f16x2_planes_kmajor(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* A_maxword,
                                 const void* Bt_planes, const void* Bt_maxword, void* C, int blocked, const void* w, void* U, void* stream);
/* Chained split products (the whitened SVGP tier, svgp_regression.py:83-92 in factorised float32 form):
 * mxf_gemm_f16x2_planes_out writes alpha * A B^T DIRECTLY as the two f16 planes (hi + lo, UNSCALED: the caller picks alpha so that the
 * largest magnitude sits near 2^13..2^14; as an operand of the next product its maxword is a word holding 8192.0f = scale 1) of the (M x N) operand whose contraction
 * index is its column -- element (m, n) at ((n / 16) * M + m) * 16 + n % 16 -- ready to be the operand of the next product; M % 128 == 0,
 * N % 256 == 0.  a_lower != 0: A is lower triangular (A[m][k] = 0 for k > m), the k loop of a row tile stops at its last row.
 * Ct_planes != NULL: the same launch ALSO writes the planes of the transposed (N x M) operand -- element (n, m) at
 * ((m / 16) * N + n) * 16 + m % 16 -- and, with a (M floats) and U (N floats) given, U[n] = sum_m a[m] (hi + lo)(m, n) in the planes' units
 * (what the whitened tier needs of V = L^-1 Kuf: V for Phi = V V^T, V^T for T = Hh V, a^T V for the mean term -- one pass).
 * mxf_f16x2_planes_transpose turns the planes of an (R x K) operand into those of its transpose (K x R) (R, K multiples of 64; 4 bytes
 * read + 4 written per element, HBM bound); U != NULL: the same pass forms U[k] = scale[0] * sum_r a[r] x(r, k) (a: R floats, scale: one
 * float, both on the device).                                                                                                            */
int mxf_gemm_f16x2_planes_out(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* A_maxword,
                              const void* B_planes, const void* B_maxword, void* C_planes, void* Ct_planes, const void* a, void* U,
                              int a_lower, void* stream);
int mxf_f16x2_planes_transpose(mxf_handle h, int64_t R, int64_t K, const void* planes_in, void* planes_out, const void* a, const void* scale,
                               void* U, void* stream);

/* The two halves of mxf_gemm_f32x3 for callers that reuse split operands (the SVGP step splits Kuf once for two products):
 * mxf_f32x3_split writes the three bf16 planes of an (R x K) float32 matrix (k16-blocked, see gemm_split.hip) into `planes`
 * (3 * mxf_f32x3_plane_elems(R, K) 16-bit elements); mxf_gemm_f32x3_planes multiplies two split operands.                         */
int64_t mxf_f32x3_plane_elems(int64_t R, int64_t K);
int mxf_f32x3_split(mxf_handle h, int64_t R, int64_t K, const void* X, int64_t ld, void* planes, void* stream);
int mxf_gemm_f32x3_planes(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* B_planes,
                          double beta, void* C, int64_t ldc, int lower_only, void* stream);

/* in-place lower Cholesky, strictly-upper part zeroed -- linalg.potrf (gp_regression.py:61,
 * svgp_regression.py:83-84).  info: device int[S], 0 or (1-based) index of the first bad pivot.   */
int mxf_potrf(mxf_handle h, int dtype, int S, int64_t n, void* A, int64_t lda, int64_t strideS_A,
              int* info, void* stream);

/* B <- op(L)^-1 B, L lower (n x n), B (n x nrhs) -- linalg.trsm(L,B,transpose)
 * (gp_regression.py:66,172; svgp_regression.py:85-87,153-156,164)                                   */
int mxf_trsm(mxf_handle h, int dtype, int transpose, int S, int64_t n, int64_t nrhs,
             const void* L, int64_t ldl, int64_t strideS_L, void* B, int64_t ldb, int64_t strideS_B,
             void* stream);

/* Linv <- L^-1 (lower) ; used for K^-1 = L^-T L^-1 in the closed-form gradients                    */
int mxf_trtri(mxf_handle h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t strideS_L,
              void* Linv, int64_t ldi, int64_t strideS_I, void* stream);

/* make_diagonal custom operator (mxfusion/util/customop.py:22-81): out (batch, n, n) = diag-embed of a (batch, n); mxf_diag_of is its
 * reverse mode, out (batch, n) = diagonal of g (batch, n, n) (customop.py:49-61).  Used for S = W W^T + make_diagonal(s)
 * (svgp_regression.py:76, :145).                                                                                                    */
int mxf_make_diagonal(mxf_handle h, int dtype, int64_t batch, int64_t n, const void* a, void* out, void* stream);
int mxf_diag_of(mxf_handle h, int dtype, int64_t batch, int64_t n, const void* g, void* out, void* stream);

/* out[s] = sum_i log|L_ii| -- linalg.sumlogdiag(abs(L)) (gp_regression.py:67)                       */
int mxf_sumlogdiag(mxf_handle h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t strideS_L,
                   void* out, void* stream);

/* out[s][n] = sum_m A[s][m][n]*B[s][m][n] -- F.sum(A*B, axis=-2) of the predictive variances
 * (gp_regression.py:181, svgp_regression.py:166-169, sparsegp_regression.py:153-155)                */
int mxf_coldot(mxf_handle h, int dtype, int S, int64_t M, int64_t N, const void* A, int64_t lda, int64_t strideS_A,
               const void* B, int64_t ldb, int64_t strideS_B, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Prediction composites (for a binder with no module code of its own; the Python mirror composes the same kernels).           */

/* Kernel.Kdiag: out (S, N) = diagonal of K(X[s], X[s]).  Stationary kinds, Bias, White: the variance (stationary.py:123-124,
 * static.py:76-86,152-162); Linear: sum_q variances_q x_q^2 with `lengthscale` carrying the variances (linear.py:91-104).           */
int mxf_kdiag(mxf_handle h, int kind, int dtype, int S, int64_t N, int Q, const void* X, int64_t strideS_X,
              const void* lengthscale, int ard, int64_t strideS_ls, const void* variance, int64_t strideS_var,
              void* out, void* stream);

/* GPRegressionMeanVariancePrediction.compute (gp_regression.py:146-196) for ONE posterior (L (N,N) lower, LinvY (N,P) as the inference
 * stored them, :203-234) and S samples of the test inputs X_test (S, Nt, Q): mean_out (S, Nt, P); var_out (S, Nt) -- the reference
 * broadcasts it over P -- or, full_cov, (S, Nt, Nt).  noise_free = 0 adds noise_var (1,).  Stationary kinds.                          */
int mxf_gp_predict(mxf_handle h, int kind, int dtype, int S, int64_t N, int64_t Nt, int Q, int P,
                   const void* X_cond, const void* X_test, const void* lengthscale, int ard, const void* variance,
                   const void* L, int64_t ldl, const void* LinvY, const void* noise_var, int noise_free, int full_cov,
                   void* mean_out, void* var_out, void* stream);

/* SVGPRegressionMeanVariancePrediction.compute (svgp_regression.py:121-189) from the variational parameters themselves: Z (M,Q),
 * qU_mean (M,P), qU_cov_W (M,M), qU_cov_diag (M,) (constrained values), S samples of X_test (S, Nt, Q).  Outputs as mxf_gp_predict;
 * info (2 ints, may be null): potrf status of Kuu + jitter I and of S = W W^T + diag.                                                */
int mxf_svgp_predict(mxf_handle h, int kind, int dtype, int S, int64_t M, int64_t Nt, int Q, int P,
                     const void* Z, const void* X_test, const void* lengthscale, int ard, const void* variance,
                     const void* qU_mean, const void* qU_cov_W, const void* qU_cov_diag, const void* noise_var,
                     double jitter, int noise_free, int full_cov, void* mean_out, void* var_out, void* info, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise / reduction pieces of the MC-ELBO loop.                                             */

/* y = log(1+exp(x)) and its reverse mode -- PositiveTransformation (var_trans.py:63-91)             */
int mxf_softplus_fwd(mxf_handle h, int dtype, int64_t n, const void* x, void* y, void* stream);
int mxf_softplus_bwd(mxf_handle h, int dtype, int64_t n, const void* x, const void* dy, void* dx_acc, void* stream);

/* x[s,i] = mean[i] + eps[s,i]*sqrt(var[i]) -- Normal.draw_samples_impl (normal.py:72-92); eps is the
 * caller-injected noise buffer (random_gen.py:26-28 seam).  n = elements per sample.               */
int mxf_normal_reparam(mxf_handle h, int dtype, int S, int64_t n, const void* mean, const void* var,
                       const void* eps, void* x, void* stream);

/* out += scale * sum_{s,i} logN(x[s,i] | mean[i], var[i])  (normal.py:52-70 then
 * factor_graph.py:223: sum(mean_S(.)) => scale = +-log_pdf_scaling/S), with wave-shuffle reductions;
 * optional reverse mode accumulated into dx (S,n), dmean (n), dvar (n) (all scaled by `scale`).
 * mean/var may be single-element broadcasts (n_mean, n_var in {1, n}).                            */
int mxf_normal_logpdf(mxf_handle h, int dtype, int S, int64_t n, const void* x,
                      const void* mean, int64_t n_mean, const void* var, int64_t n_var, double scale,
                      void* out_acc, void* dx_acc, void* dmean_acc, void* dvar_acc, void* stream);

/* reverse mode of mxf_normal_reparam: dmean += sum_s dx ; dvar += sum_s dx*eps/(2 sqrt(var))        */
int mxf_normal_reparam_bwd(mxf_handle h, int dtype, int S, int64_t n, const void* var, const void* eps,
                           const void* dx, void* dmean_acc, void* dvar_acc, void* stream);

/* MXNet Adam as driven by gluon.Trainer.step (batch_loop.py:46-60, minibatch_loop.py:71-91):
 * g*=rescale; m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)      */
int mxf_adam_step(mxf_handle h, int dtype, int64_t n, void* w, const void* g, void* m, void* v,
                  double lr, double beta1, double beta2, double epsilon, double rescale_grad, int t,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused composites: one call = one reference `compute` body, value AND reverse mode.               */

/* GPRegressionLogPdf.compute (modules/gp_modules/gp_regression.py:42-76), stationary kernel.
 *   X (S|1,N,Q)  Y (S|1,N,P) [already minus mean]  noise_var (S|1,1)
 * outputs: logL (S);  L (S,N,N) and LinvY (S,N,P) (the posterior side effect of :72-75);
 * if want_grad: dX (S,N,Q), dY (S,N,P), dnoise (S), dls (S,Q|1), dvar (S) = d logL[s] / d(.) , WRITTEN.
 * info: device int[S].                                                                             */
int mxf_gp_logpdf(mxf_handle h, int kind, int dtype, int S, int64_t N, int Q, int P,
                  const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y,
                  const void* noise_var, int64_t strideS_noise,
                  const void* lengthscale, int ard, int64_t strideS_ls,
                  const void* variance, int64_t strideS_var, double jitter,
                  void* logL, void* L, void* LinvY, int* info, int want_grad,
                  void* dX, void* dY, void* dnoise, void* dls, void* dvar, void* stream);

/* SVGPRegressionLogPdf.compute (modules/gp_modules/svgp_regression.py:43-109), homoscedastic noise,
 * stationary kernel, streaming sufficient-statistics form: Kuf is never the operand of a
 * triangular solve; all (M x M) factorisation work is done once (not S times) in float64.
 *   X (S|1,B,Q)  Y (S|1,B,P) [minus mean]  Z (M,Q)  noise_var (1)  qU_mean (M,P)  qU_cov_W (M,M)
 *   qU_cov_diag (M) [positive]  lengthscale (Q|1)  variance (1); scaling = log_pdf_scaling (:108)
 * outputs: logL (S) per-sample bound.  If want_grad: gradients of  sum_s gw[s]*logL[s]  (gw: host
 * weights folded as a single scalar `gscale` applied to every sample, i.e. the mean_S of
 * factor_graph.py:233 => gscale = 1/S): dX (S,B,Q) dY (S,B,P) dZ (M,Q) dnoise (1) dmu (M,P) dW (M,M)
 * dSdiag (M) dls (Q|1) dvar (1), all WRITTEN.                                                      */
int mxf_svgp_logpdf(mxf_handle h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                    const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y,
                    const void* Z, const void* noise_var, const void* qU_mean, const void* qU_cov_W,
                    const void* qU_cov_diag, const void* lengthscale, int ard, const void* variance,
                    double jitter, double scaling, double gscale,
                    void* logL, int* info, int want_grad,
                    void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW, void* dSdiag,
                    void* dls, void* dvar, void* stream);

/* The same bound with SAMPLED parameters: every operand of svgp_regression.py:43-109 may carry the sample axis (the reference broadcasts
 * all of them to S, components/variables/runtime_variable.py:96-118, exercised by its tests with sampled variables): sample s uses slice
 * s of every operand with a non-zero sample stride (in elements; 0 = shared), incl. its own Kuu / q(u) core.
 *   X (S|1,B,Q)  Y (S|1,B,P)  Z (S|1,M,Q)  noise_var (S|1,1)  qU_mean (S|1,M,P)  qU_cov_W (S|1,M,M)  qU_cov_diag (S|1,M)
 *   lengthscale (S|1,Q|1)  variance (S|1,1)
 * outputs: logL (S), info (S ints).  If want_grad: dX (S,B,Q) dY (S,B,P) dZ (S,M,Q) dnoise (S) dmu (S,M,P) dW (S,M,M) dSdiag (S,M)
 * dls (S,Q|1) dvar (S), all WRITTEN: slice s = gradient of gscale * logL[s] w.r.t. the operands sample s used (a shared operand's
 * gradient is the sum of its slices).                                                                                            */
int mxf_svgp_logpdf_sampled(mxf_handle h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                            const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y, const void* Z, int64_t strideS_Z,
                            const void* noise_var, int64_t strideS_noise, const void* qU_mean, int64_t strideS_mu,
                            const void* qU_cov_W, int64_t strideS_W, const void* qU_cov_diag, int64_t strideS_sd,
                            const void* lengthscale, int ard, int64_t strideS_ls, const void* variance, int64_t strideS_var,
                            double jitter, double scaling, double gscale, void* logL, int* info, int want_grad,
                            void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW, void* dSdiag, void* dls, void* dvar,
                            void* stream);

/* The same bound with heteroscedastic and/or per-output noise (svgp_regression.py:61-67: noise_var of shape (N, D'), D' in {1, D};
 * testing/modules/svgpregression_test.py:142-167).  noise_var is (noise_rows, noise_cols) with noise_rows in {1, B} and noise_cols in {1, P},
 * shared by all samples; dnoise has the same shape.  Per-row noise (B, 1) with one output column in float32 (want_grad, Q <= 8, B % 16 == 0,
 * M % 16 == 0, M >= 128, explicit form) STREAMS (r04): with nmin = min noise and r_n = nmin / noise_n the Gram planes are written as
 * Kuf diag(sqrt r) and diag(r) Kfu, the homoscedastic split products and fused reverse pass run with noise := nmin, and one extra pass
 * over the Kfu planes and T forms sum_n e_n^2 / noise_n and the (B, 1) noise gradient (8.4 vs 26.2 ms at B = 65 536, M = 1 024, S = 8).
 * Every other shape: the generic (materialised dKuf) path, same results.                                                             */
int mxf_svgp_logpdf_het(mxf_handle h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                        const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y,
                        const void* Z, const void* noise_var, int64_t noise_rows, int noise_cols,
                        const void* qU_mean, const void* qU_cov_W, const void* qU_cov_diag, const void* lengthscale, int ard,
                        const void* variance, double jitter, double scaling, double gscale,
                        void* logL, int* info, int want_grad,
                        void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW, void* dSdiag,
                        void* dls, void* dvar, void* stream);

/* The same bound from MATERIALISED Gram matrices, ONE sample per call: for kernels whose K is composed on the host
 * (AddKernel / MultiplyKernel, add_kernel.py:44-68, multiply_kernel.py:44-67 -- the deep-GP configuration's Matern52+RBF).
 *   Kuu (M,M) WITHOUT jitter (added here, svgp_regression.py:70-72), Kuf (M,B), Kdiag (B), Y (S,B,P) [minus mean]: S samples of the
 *   outputs over the SAME inputs (a hidden layer of a deep GP); logL (S), dY (S,B,P); the other gradients are summed over the samples;
 *   noise_var (noise_rows, noise_cols) as above.
 * if want_grad: gscale * d logL / d(.) WRITTEN into dKuu (M,M) dKuf (M,B) dKdiag (B) dY dnoise dmu dW dSdiag; the caller chains
 * dKuu / dKuf / dKdiag into the kernels' own reverse mode (mxf_gram_bwd).                                                        */
int mxf_svgp_logpdf_mat(mxf_handle h, int dtype, int S, int64_t B, int64_t M, int P, const void* Kuu, const void* Kuf, const void* Kdiag,
                        const void* Y, int64_t strideS_Y, const void* noise_var, int64_t noise_rows, int noise_cols, const void* qU_mean,
                        const void* qU_cov_W, const void* qU_cov_diag, double jitter, double scaling, double gscale, void* logL,
                        int* info, int want_grad, void* dKuu, void* dKuf, void* dKdiag, void* dY, void* dnoise, void* dmu,
                        void* dW, void* dSdiag, void* stream);

/* SparseGPRegressionLogPdf.compute (modules/gp_modules/sparsegp_regression.py:42-108), Titsias bound, ONE sample
 * (callers loop over samples), stationary kernel, sufficient-statistics form with C = Kuu + Psi2/noise.
 *   X (B,Q)  Y (B,P) [minus mean]  Z (M,Q)  noise_var (1)  lengthscale (Q|1)  variance (1)
 * outputs: logL (1); the posterior side products of :99-106: wv (M,P), L (M,M), LA (M,M) (any may be NULL);
 * if want_grad: gscale * d logL / d(.) WRITTEN into dX (B,Q) dY (B,P) dZ (M,Q) dnoise (1) dls (Q|1) dvar (1).       */
int mxf_sgp_logpdf(mxf_handle h, int kind, int dtype, int64_t B, int64_t M, int Q, int P, const void* X, const void* Y,
                   const void* Z, const void* noise_var, const void* lengthscale, int ard, const void* variance,
                   double jitter, double gscale, void* logL, void* wv, void* L, void* LA, int* info, int want_grad,
                   void* dX, void* dY, void* dZ, void* dnoise, void* dls, void* dvar, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MXF_GP_H */
