// Internal (non-exported) typed launchers shared between translation units of libmxf_gp.so.
#pragma once
#include "common.h"

void mxf_comm_release(mxf_ctx* h);      // comm.hip: destroys the handle's RCCL communicator, if any

// C = alpha op(A) op(B) + beta C; lower_only: skip blocks / entries strictly above the diagonal (syrk-style update)
int mxf_gemm_internal(mxf_ctx* h, int dtype, int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                      const void* A, int64_t lda, int64_t sA, const void* B, int64_t ldb, int64_t sB, double beta,
                      void* C, int64_t ldc, int64_t sC, int batch, int lower_only, hipStream_t st, int reserve_cus = 0, int k_from_m = 0);

// zero_upper = false: leave the strict upper triangle outside the 64 x 64 diagonal blocks as it was (callers that only read the lower part)
// zero_info = false: the caller has zeroed `info` already (the SVGP composite clears its status words in one launch off the critical path)
int mxf_potrf_internal(mxf_ctx* h, int dtype, int S, int64_t n, void* A, int64_t lda, int64_t sA, int* info, hipStream_t st, bool zero_upper = true,
                       bool zero_info = true, void* Linv_eager = nullptr, int64_t ldie = 0, bool* eager_done = nullptr,
                       void* Kacc = nullptr, int64_t ldk = 0, double kcoef = 0.0, bool* kacc_done = nullptr);
// (Kacc, with Linv_eager: kcoef L^-T L^-1 (lower triangle) is accumulated into this ZEROED n x n buffer row block by row block of L^-1, next to
//  the factorisation as well -- L^-T L^-1 = sum over row blocks b of Linv[b, :]^T Linv[b, :]; *kacc_done says whether it was)
// (Linv_eager: float64, S = 1, large n: L^-1 is formed into this (n x n, leading dimension ldie) buffer NEXT TO the factorisation, row block by row
//  block on a third stream; *eager_done says whether it was -- if not, the caller runs mxf_trtri_internal as before)
// rhs_lower: B is block-lower-triangular (trtri); only columns < (k+1)*64 of block row k are touched
int mxf_trsm_internal(mxf_ctx* h, int dtype, int transpose, int S, int64_t n, int64_t nrhs, const void* L, int64_t ldl,
                      int64_t sL, void* B, int64_t ldb, int64_t sB, int rhs_lower, hipStream_t st);
int mxf_trtri_internal(mxf_ctx* h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t sL, void* Linv, int64_t ldi,
                       int64_t sI, hipStream_t st);
int mxf_sumlogdiag_internal(mxf_ctx* h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t sL, void* out, hipStream_t st);

int mxf_gram_bwd_internal(mxf_ctx* h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t sX,
                          const void* X2, int64_t sX2, const void* ls, int ard, int64_t sls, const void* var, int64_t svar,
                          const void* dK, int64_t lddk, int64_t sdK, void* dX, void* dX2, void* dls, void* dvar, hipStream_t st, int dk_symmetric = 0);
// (dk_symmetric, square case only: the caller vouches that dK is symmetric -- the row-side sums are skipped and the column side counts twice)

// true if mxf_svgp_bwd_fused_internal takes the matrix-pipe pass for these arguments; that pass reads T in 16-column blocks
// (element (m, n) at ((n / 16) * M + m) * 16 + n % 16: mxf_gemm_split_internal's c_blocked output) when called with t_blocked = 1
bool mxf_svgp_bwd_is_mfma(int kind, int dtype, int64_t SB, int64_t B, int Q, int P, const void* Text);
bool mxf_svgp_bwd_reads_blocked(int kind, int dtype, int64_t SB, int64_t B, int Q, int P, const void* Text);
// SVGP-fused reverse pass over Text = [H0; w^T] Kuf_all (never materialises dKuf); see gram_bwd.hip
int mxf_svgp_bwd_fused_internal(mxf_ctx* h, int kind, int dtype, int64_t M, int64_t SB, int64_t B, int Q, int P, const void* Z,
                                const void* Xall, const void* ls, int ard, const void* var, const void* Text, const void* Y,
                                int64_t sY, const void* w, const void* noise, double a1, void* dZ, void* dXall, void* dls,
                                void* dvar, void* dY, int dY_shared, void* R, double* scal, hipStream_t st, int t_blocked = 0,
                                const unsigned* h0max = nullptr, const unsigned* tmax = nullptr);
// (h0max: bit pattern of max |H0| when T = H0 Kuf came from the f16x2 split GEMM -- the operand bound that lets the matrix-pipe pass
//  accumulate its RBF weights as hi + lo f16; nullptr: float32 accumulation.  tmax: bit pattern of max |T| if the GEMM reported it
//  (word != 0): the tight bound)

// r05: the SVGP reverse pass FUSED into the epilogue of the T product (gemm_split.hip wide_body<..., FUSE>): the 256 x 256 accumulator tile IS T --
// the pass's weights W = -(c1 variance) (T + w e) k, its row sums [B | S] = W [X | 1], R and its column sums [D | C] = W^T [Z | 1] are formed
// from registers, T is never written (8.6 GB less written and 8.6 GB less read per 32-sample step, one bulk kernel less).  RBF, Q <= 8, one
// output column, M % 256 == 0, B % 256 == 0.  Filled by mxf_svgp_bwd_fuse_prepare (gram_bwd.hip), consumed by mxf_gemm_split_internal.
struct mxf_fuse_args {
    const float* Zs; const float* Zn;    // scaled, centred inducing inputs (8 per row) and their squared norms
    const float* Xs; const float* Xn;    // the same for the data columns
    const float* U; const float* Y; const float* w; const float* ls; const float* var; const float* noise;
    float* dX; double* zacc; double* dls3; double* scal;
    const unsigned* h0max; const unsigned* mx;      // bit patterns: max |A operand| of the product, {max |w|, max |y - U|}
    int64_t B, sY;
    int Q, ard;
    double a1;
};
int mxf_svgp_bwd_fuse_ok(int kind, int dtype, int64_t M, int64_t SB, int64_t B, int Q, int P);
int mxf_svgp_bwd_fuse_prepare(mxf_ctx* h, int64_t M, int64_t SB, int64_t B, int Q, const float* Z, const float* X, const float* ls, int ard,
                              const float* var, const float* U, const float* Y, int64_t sY, const float* w, const float* noise, double a1, float* dX,
                              float* dY, int dY_shared, double* scal, const unsigned* h0max, mxf_fuse_args* out, hipStream_t st);
int mxf_svgp_bwd_fuse_finish(mxf_ctx* h, int64_t M, int Q, int ard, const float* ls, const float* var, const mxf_fuse_args* fz, float* dZ, float* dls,
                             float* dvar, float* R, hipStream_t st);

// f32-accurate GEMM on the bf16 matrix pipe (three-term bf16 splitting, gemm_split.hip)
size_t mxf_split_plane_elems(int64_t R, int64_t K);    // elements (bf16) of ONE plane of an (R x K) operand
// operand formats of the split GEMM (gemm_split.hip): three bf16 terms / two scaled f16 terms
#define MXF_SPLIT_BF16X3 0
#define MXF_SPLIT_F16X2 1
int mxf_maxabs_internal(mxf_ctx* h, int64_t R, int64_t K, const float* x, int64_t ld, unsigned* out, hipStream_t st, bool zero = true);   // out[0] = bit pattern of max |x| (zero = false: the caller cleared the word)
int mxf_split_planes_internal(mxf_ctx* h, int64_t R, int64_t K, const float* X, int64_t ld, unsigned short* planes, hipStream_t st,
                              int mode = MXF_SPLIT_BF16X3, const unsigned* maxbits = nullptr);
int mxf_gemm_split_internal(mxf_ctx* h, int64_t M, int64_t N, int64_t K, double alpha, const unsigned short* A, int64_t pA,
                            const unsigned short* B, int64_t pB, double beta, float* C, int64_t ldc, int lower_only, hipStream_t st,
                            int reserve_cus = 0, int mode = MXF_SPLIT_BF16X3, const float* ad0 = nullptr, int pow0 = 0,
                            const unsigned* maxbits = nullptr, const unsigned* maxbits2 = nullptr, int c_blocked = 0,
                            unsigned* maxout = nullptr, unsigned short* Cplanes = nullptr, int64_t pC = 0, int a_lower = 0,
                            unsigned short* Ct = nullptr, int64_t pCt = 0, const float* avec = nullptr, float* Upart = nullptr,
                            const mxf_fuse_args* fuse = nullptr);
// (Ct: the planes output ALSO in the transposed orientation, ((m / 16) * N + n) * 16 + m % 16, plane stride pCt; avec (M floats) / Upart
//  ((M / 128) x N floats): per 128-row band the sums of avec[m] * (hi + lo)(m, n), in the planes' units -- deterministic, summed by the caller)
// (Cplanes != nullptr: the product is written as two f16 planes (hi + lo of alpha * A B^T, plane stride pC) in the layout of an (M x K' = N)
//  operand, ((n / 16) * M + m) * 16 + n % 16 -- C / ldc are ignored; a_lower: A is lower triangular, the k loop of a row tile stops at its last row)
// (maxout: the wide kernel's plain products raise this word (atomicMax) to the bit pattern of max |C|; every other path leaves it untouched)
// gemm_bt.hip (r06): the same product with the second operand stored K-MAJOR -- C (M x N) = alpha * ad0[0] / scale(maxbits) * A (M x K) Bt (K x N),
// Bt = the planes of the (btR >= K rows, k' = N) operand, element (k, n) at ((n / 16) * btR + k) * 16 + n % 16: the T product of the SVGP step
// reads the SAME Kuf planes as Psi2.  w / U / wscratch (2 K halves + one word): optional row U[n] = uscale * ad0[0] * sum_k w[k] Bt[k][n].
bool mxf_gemm_bt_ok(int64_t M, int64_t N, int64_t K);
int mxf_gemm_bt_internal(mxf_ctx* h, int64_t M, int64_t N, int64_t K, double alpha, const unsigned short* A, int64_t pA, const unsigned short* Bt,
                         int64_t pB, int64_t btR, float* C, int64_t ldc, int c_blocked, hipStream_t st, int reserve_cus = 0, const float* ad0 = nullptr,
                         const unsigned* maxbits = nullptr, unsigned* maxout = nullptr, const float* w = nullptr, float* U = nullptr,
                         double uscale = 1.0, void* wscratch = nullptr, const unsigned* maxbits2 = nullptr);
size_t mxf_gram_planes_scratch_bytes(int64_t R, int64_t Kn, int Q);
int mxf_gram_planes_internal(mxf_ctx* h, int kind, int64_t R, int64_t Kn, int Q, const float* Xmin, const float* Xmaj, const float* ls,
                             int ard, const float* var, unsigned short* planes, int64_t pstride, float* scratch, hipStream_t st,
                             int mode = 0 /* MXF_SPLIT_BF16X3 */, const float* wk = nullptr, int Pw = 0, float* U = nullptr, int64_t ldU = 0,
                             const float* majs = nullptr, const float* mins = nullptr, int64_t period = 1);
// (majs / mins: optional per-row weights (period entries, index taken modulo period) on the major / minor index -- f16x2 lean kernel only)
// (wk (Kn x Pw), U (Pw x ldU): optional fused product U[p][r] = sum_k wk[k][p] cov(xmin[r], xmaj[k]); see gram_planes_kernel)

// whiten.hip: two f16 planes of an (R x K) operand -> the planes of its transpose (K x R); U != nullptr: U[k] = scale[0] * sc2 * sum_r a[r] x(r, k)
int mxf_upart_reduce_internal(mxf_ctx* h, int64_t N, int nparts, const float* Upart, const float* scale, float sc2, float* U, hipStream_t st);
int mxf_planes_transpose_internal(mxf_ctx* h, int64_t R, int64_t K, const unsigned short* in, int64_t pin, unsigned short* out, int64_t pout,
                                  const float* a, const float* scale, float sc2, float* U, hipStream_t st);
int mxf_tril_copy_internal(mxf_ctx* h, int64_t n, const double* src, double* dst, hipStream_t st);      // dst = tril(src), float64
