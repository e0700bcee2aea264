// dW (M, N) = dY^T X over many rows with a tiny output (attention projections of a d_model = 32 encoder layer): see skinny_tn.hip
#pragma once
#include "common.hpp"

// applies: 16 <= M <= 192, 16 <= N <= 64 (multiples of 16), R >= 8192, 16-byte aligned dense-enough rows (IMMTSF_SKINNY_TN=0 disables)
bool skinny_tn_ok(int M, int N, int R, int ldy, int ldx, const void* dY, const void* X, const void* dW);
size_t skinny_tn_scratch_floats(int M, int N, int R);
// dW (M x N, row pitch N) and db (M, may be null) are OVERWRITTEN; scratch: skinny_tn_scratch_floats floats
int launch_skinny_tn(const float* dY, int ldy, int M, const float* X, int ldx, int N, int R, float* dW, float* db, float* scratch,
                     hipStream_t s);
