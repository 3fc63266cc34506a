#pragma once
#include <hip/hip_runtime.h>
enum { UENC_PROF_GEMM_NT = 0, UENC_PROF_GEMM_TN = 1, UENC_PROF_WATTN_FWD = 2, UENC_PROF_WATTN_BWD = 3, UENC_PROF_GEMM_NT256 = 4, UENC_PROF_GEMM_NT128 = 5 };
bool uenc_prof_on();
void uenc_prof_begin(int kind, double flops, hipStream_t stream, double bytes = 0.0);
void uenc_prof_end(hipStream_t stream);
