// Internal (non-exported) typed launchers shared between translation units of libmxf_gp.so.
#pragma once
#include "common.h"

// C = alpha op(A) op(B) + beta C; lower_only: skip blocks / entries strictly above the diagonal (syrk-style update)
int mxf_gemm_internal(mxf_ctx* h, int dtype, int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                      const void* A, int64_t lda, int64_t sA, const void* B, int64_t ldb, int64_t sB, double beta,
                      void* C, int64_t ldc, int64_t sC, int batch, int lower_only, hipStream_t st);
