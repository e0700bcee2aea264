// MFMA GEMM used by every projection of the fusion path (forward NT, data-gradient NN, weight-gradient TN).
#pragma once
#include "common.hpp"

#define IMMTSF_GEMM_MAX_PROBLEMS 4

struct GemmProblem {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // length N, may be null
    float* bias_grad;    // TN + ones_col: length M, receives sum_k opA(m,k)
    const void* Bh;      // bf16 copy of B (same element offsets): given by the caller, else looked up in the twin registry
    const void* Ah;      // bf16 copy of A (same element offsets, same lda), written by A's producer; may be null
    void* Ch;            // optional bf16 copy of the result (row pitch GemmArgs::ldch); C may then be null
    float* Cpre;         // optional: the value BEFORE the activation (after bias / row flag / add_vec), same pitch as C --
                         // what a GELU backward needs (gemm.hip only)
};

// C[m,n] = epilogue( alpha * sum_k opA(m,k) * opB(n,k) )
//   opA(m,k) = TA ? A[k*lda + m] : A[m*lda + k]        (TA: reduction index is the slow one)
//   opB(n,k) = TB ? B[k*ldb + n] : B[n*ldb + k]
// epilogue(v) = act( rowflag(v + bias[n]) + add_vec[n] ) (+ C_old if accumulate)
struct GemmArgs {
    GemmProblem p[IMMTSF_GEMM_MAX_PROBLEMS];
    int nprob;
    int M, N, K;
    int lda, ldb, ldc;
    int ldch;              // row pitch of the bf16 result copies p[i].Ch (0: same as ldc)
    const int* dyn;        // device int overriding M (dyn_which==0) or K (dyn_which==1); may be null
    int dyn_which;
    const int* a_rowmap;   // !TA: source row of A for logical row m.  TA: source row for reduction index k
    const int* b_rowmap;   // TB only: source row of B for reduction index k
    float alpha;
    int accumulate;
    const unsigned char* row_flag;   // optional: rows with row_flag[m / row_flag_div] == 0 are zeroed
    int row_flag_div;
    const int* row_flag32;           // optional int32 array with the same zero / non-zero pattern as row_flag (e.g. the per-window note
                                     // counts behind M_txt): lets the many-rows kernel (gemm3.hip) fetch the flags by LDS-DMA
    const float* add_vec;  // optional, length N
    int act;               // 0 none, 1 relu, 2 gelu(erf)
    int vecA, vecB, vecC;  // host-verified: 16-byte aligned base and ld % 4 == 0
    // batched form (dense attention): grid.z = nprob * nbatch; batch index bi -> (bo, bin) = (bi / batch_inner,
    // bi % batch_inner); each operand pointer advances by bo*s?_o + bin*s?_i elements.  nbatch <= 1: no batching.
    int nbatch, batch_inner;
    long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    // TN only: when p[i].bias_grad != null the workgroups of the first tile column also reduce the A chunks they stage
    // over the reduction index (= column sums of dY, the bias gradient, exact fp32) into bias_grad[m].
    int ones_col;
    // optional relu-backward mask: results whose relu_ref[m*ld_ref + n] <= 0 are zeroed (relu_ref = forward output)
    const float* relu_ref;
    int ld_ref;
    int ref_kind;          // 0: relu_ref is a ReLU output (mask where <= 0); 2: relu_ref is a GELU PRE-activation z, results are
                           // multiplied by gelu'(z) (gemm.hip only)
    // the caller guarantees C (and bias_grad) are already zero: split-K skips its own zero-fill
    int c_prezeroed;
    // optional split-K workspace (TN weight gradients with a long reduction): immtsf_gemm3_tn_ws_bytes(M, N, K) bytes, private to
    // this launch until it has completed; without it such products run on gemm2's tiles
    void* ws;
    size_t ws_bytes;
    int no_split;          // never split K (tiny products whose zero-fill + atomics cost more than the serial K loop)
    // optional dropout on the result (after the activation): element (m, n) uses Philox index m * N + n of site epi_site
    DropCfg epi_drop;
    uint64_t epi_site;
    int xcd_remap;         // set by the launcher
    int xcd_gm;            // gemm2: > 0 = the 8 XCDs own an xcd_gm x (8 / xcd_gm) arrangement of equal rectangles of the tile grid
    int dbg;               // ablation flags (tools/gemm_bench.py), 0 in production
    // gemm2: the tile-index arithmetic of the kernel's first instructions, precomputed by the launcher (g2_fast != 0): every
    // workgroup used to open with ~8 run-time integer divisions (tile grid, XCD partition, K split), ~0.5 us before its first load.
    // n / d for n, d < 65536 as __umulhi(n, ceil(2^32 / d)); magic 0 = divisor 1
    int g2_fast;
    int g2_tiles_n;
    unsigned g2_tn_magic;  // lin / tiles_n
    int g2_rows_x, g2_cols_x, g2_xc_shift;
    unsigned g2_cx_magic;  // idx / cols_x
    int g2_per;            // K steps per split (0: K is a device value, divide in the kernel)
    // gemm2 only: ONE problem repeated zbatch times along grid.z (0: off) -- batch z reads A + z zsA, B + z zsB (bf16 elements), writes
    // C / Ch / Cpre + z zsC and takes its device-side M or K from dyn[z dyn_stride].  atomic_c: every batch ADDS its product (and bias
    // gradient) into the same zeroed C by fp32 atomics (zsC = 0: a weight gradient summed over the batches)
    int zbatch, dyn_stride, atomic_c;
    long zsA, zsB, zsC;
};

enum { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };

// bf16 twins (immtsf_bf16_twin_register): a registered fp32 range [base, base+count) has a bf16 copy with the same
// element offsets, kept current by its owner (the fused Adam kernel writes it).  Returns the twin of `p` or null.
const void* immtsf_twin_lookup(const float* p, size_t min_elems);

// ---- gemm2.hip: the bf16-in-memory path.  Both operands are bf16 in HBM and go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4) through a multi-stage ring; same GemmArgs, reading p[i].Ah / p[i].Bh and writing p[i].C
// (fp32, may be null) and/or p[i].Ch (bf16, may be null).  Returns IMMTSF_EUNSUPPORTED for argument combinations it
// does not implement (batched form, TN with a row map, unaligned operands): the caller then uses immtsf_launch_gemm.
bool immtsf_gemm2_supported(int layout, const GemmArgs& g);
void immtsf_gemm_note_grid(long threads);      // timing tap: threads of the launch just made
int immtsf_launch_gemm2(int layout, GemmArgs& g, hipStream_t stream);
// several TN products of different shapes in one launch (64 x 64 K-group tiles); IMMTSF_EUNSUPPORTED: launch them one by one
int immtsf_launch_gemm2_group_tn(GemmArgs* list, int n, hipStream_t stream);
bool immtsf_gemm_group_enabled();      // IMMTSF_GEMM_GROUP=1 (off by default: measured slower inside the cfg2 step)

// ---- gemm3.hip: the persistent many-rows kernel (M >> 256, K > 64, plain epilogue: alpha, bias, row flags, add_vec).
// IMMTSF_EUNSUPPORTED for anything else: the caller falls back to gemm2.
int immtsf_launch_gemm3(int layout, const void* A, int lda, const void* B, int ldb, float* C, int ldc, void* Ch, int ldch,
                        const float* bias, const float* add_vec, const int* row_flag, int row_flag_div, int M, int N, int K,
                        float alpha, int act, const int* dyn_rows, hipStream_t stream);
// TN with a long reduction (K >= 8192) and few tiles: split-K over the persistent workgroups + one reduce launch.
size_t immtsf_gemm3_tn_ws_bytes(int M, int N, int K);     // 0: this product would not take the split path
int immtsf_launch_gemm3_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, void* Ch, int ldch, float* bias_grad, int M, int N,
                           int K, float alpha, int accumulate, const int* dynk, void* ws, size_t ws_bytes, hipStream_t stream);

// precision: 0 = exact fp32 (v_mfma_f32_16x16x4_f32), 1 = bf16 operands / fp32 accumulate (v_mfma_f32_16x16x32_bf16)
int immtsf_launch_gemm(int layout, int precision, GemmArgs& g, hipStream_t stream);
// the weight-gradient products a block collected (immtsf_launch_gemm's arguments): one grouped launch when they qualify, else one by one
int immtsf_launch_gemm_tn_list(int precision, GemmArgs* list, int n, hipStream_t stream);

// linear_small.hip: backward of a small linear layer (N <= 32 outputs, K <= 64 inputs, M <= 4096 rows) as ONE launch; dW / db are
// ADDED to by atomics (the buffers read zero), dx is overwritten
bool linear_small_ok(int M, int N, int K);
int launch_linear_small_bwd(const float* x, const float* W, const float* dy, int M, int N, int K, float* dx, const float* relu_x, float* dW,
                            float* db, hipStream_t s);
