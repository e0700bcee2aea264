// TTF_RecAvg recency-weighted average over packed ragged note rows.
#pragma once
#include "common.hpp"

// Vp packed [R,d]; Eraw out [B*T,d]; denom out [B*T] (un-clamped sum of weights)
int launch_recavg_fwd(int B, int T, int d, int N, const int* offsets, const int* rowmap, const float* tau_pad,
                      const float* t_hat, const float* log_sigma, const float* Vp, float* Eraw, float* denom,
                      hipStream_t s);
// dEraw [B*T,d] -> dVp [R,d], dls_part [B] (per-window d/d log_sigma)
int launch_recavg_bwd(int B, int T, int d, const int* offsets, const int* rowmap, const float* tau_pad, const float* t_hat,
                      const float* log_sigma, const float* Vp, const float* Eraw, const float* denom, const float* dEraw,
                      float* dVp, float* dls_part, hipStream_t s, int precision = 0);     // precision 1: the contractions on MFMA tiles (T <= 32)
