// Fused feed-forward (d_model = 32, ReLU) of the tPatchGNN encoder layer at many rows: see ffn32.hip
#pragma once
#include "common.hpp"

// the fused path applies (bf16 precision, d_model 32, ReLU, >= 2048 rows, F a multiple of 256; IMMTSF_FFN32=0 disables)
bool ffn32_ok(int R, int D, int F, int act, int prec);
// bytes of `saved` (forward -> backward: weight images + mask words) and of the backward's `scratch` (gradient slabs)
size_t ffn32_saved_bytes(int R, int F);
size_t ffn32_scratch_bytes(int R, int F);
// ff (R, 32) = Dropout(relu(x1 W1^T + b1)) W2^T + b2
int ffn32_forward(int R, int F, const DropCfg& dd, uint64_t site, const float* x1, const float* w1, const float* b1, const float* w2,
                  const float* b2, void* saved, float* ff, hipStream_t s);
// d1 (R, 32) += d ff / d x1 ;  gw1 (F, 32), gb1 (F), gw2 (32, F) written
int ffn32_backward(int R, int F, const DropCfg& dd, const float* x1, const float* b1, const float* dff, const void* saved, void* scratch,
                   float* d1, float* gw1, float* gb1, float* gw2, hipStream_t s);
