// Feed-forward half of tPatchGNN's transformer encoder layer at many rows, without the (rows x 2048) intermediate
// (reference models/tPatchGNN.py:118-121: nn.TransformerEncoderLayer(d_model = hid_dim = 32, dim_feedforward = 2048, ReLU)).
//
//   h = Dropout(relu(x W1^T + b1)) ;  ff = h W2^T + b2            x: (R, 32)  W1: (F, 32)  W2: (32, F)
//
// As GEMMs this is 2 forward + 4 backward launches around a (R x F) fp32 tensor that is written once and read four times:
// at 4096 windows (R = 65 536) that is 512 MB per pass, 3.7 ms per step (profiles/r03_w4096_kernel_stats.csv, the
// gemm_kernel rows), for 51 GFLOP.  d_model = 32 is exactly one K-step of v_mfma_f32_16x16x32_bf16, so h never has to exist:
//
//   * rows kernel (forward, and the data gradient d1 += (dff W2 . mask) W1 with the same code): a wave owns RT 16-row tiles
//     and walks the F / 32 chunks of the hidden dimension.  Per chunk and row tile: 2 MFMAs give h^T (lane = row, 4 consecutive
//     hidden units per register quad) -- exactly the B-operand layout of the second product once the 32 hidden units of the
//     chunk are taken in the order (4q..4q+3, 16+4q..16+4q+3) per lane group q, the order the weight image of the second
//     product is stored in -- bias / ReLU / Philox dropout on the accumulators, 2 MFMAs into the 16 x 32 output tile.
//     The combined mask (kept AND pre-activation > 0) of every accumulator register is the v_cmp result itself: 8 64-bit
//     words per (chunk, row tile), stored (R F / 8 bytes) and applied in the backward as the SGPR operand of v_cndmask.
//   * weight-gradient kernel: grid (F / 256, row blocks); wave w owns 2 chunks, keeps their W1 / W2 fragments and the
//     10 gradient tiles (gW2 2x2, gW1 2x2, gb1 2) per chunk in registers and walks the row block.  h and dh are recomputed
//     in the other orientation (operands swapped: lane = hidden unit, 4 consecutive rows per register quad) which is the
//     B-operand layout of the products over rows; x^T / dff^T come from one bf16 image of the row sub-block in LDS through
//     ds_read_b64_tr_b16 (the same image gives the plain fragments by ds_read_b128).  Partial tiles go to a slab
//     [row block][chunk][tile] in MFMA register order (1 KB coalesced stores), summed by the reduce kernel.
//   * the four weight images (fragment order, bf16) are rebuilt by one small launch per forward.
//
// Dropout uses the same Philox stream and element index (row * F + unit) as the GEMM epilogue it replaces.
#include "ffn32.hpp"

namespace {

constexpr int FD = 32;            // d_model
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float sel_mask(float x, uint64_t m) {       // lane's bit of m ? x : 0
    float r;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}
template <int L> __device__ __forceinline__ void put_lane(uint32_t& v, uint32_t sval) {       // v[lane L] = sval (wave-uniform)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sval), "n"(L));
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
    r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
    r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
    return r;
}
__device__ __forceinline__ bf16x8 cvt8(const float4& a, const float4& b) {
    bf16x8 r;
    r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
    r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
    return r;
}

// ---- weight images: img[which][chunk c][tile t][lane l][slot s], bf16, (q = l / 16)
//   0  W1f   W1[c*32 + t*16 + l%16][8q + s]                         (hidden unit x model dim, natural k order)
//   1  W2t   W2[8q + s][c*32 + t*16 + l%16]                         (the same of W2^T)
//   2  W2f   W2[t*16 + l%16][c*32 + 16*(s/4) + 4q + s%4]            (model dim x hidden unit, accumulator k order)
//   3  W1p   W1[c*32 + 16*(s/4) + 4q + s%4][t*16 + l%16]            (the same of W1^T)
__global__ __launch_bounds__(256) void ffn32_images_kernel(const float* __restrict__ W1, const float* __restrict__ W2, int F,
                                                           bf16_t* __restrict__ img) {
    const int per = F * FD, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 4 * per) return;
    const int which = idx / per, rem = idx - which * per;
    const int c = rem >> 10, t = (rem >> 9) & 1, l = (rem >> 3) & 63, s = rem & 7, q = l >> 4, r = l & 15;
    const int nat_f = c * 32 + t * 16 + r, nat_k = 8 * q + s;
    const int prm_f = c * 32 + 16 * (s >> 2) + 4 * q + (s & 3), prm_m = t * 16 + r;
    float v;
    switch (which) {
        case 0: v = W1[(size_t)nat_f * FD + nat_k]; break;
        case 1: v = W2[(size_t)nat_k * F + nat_f]; break;
        case 2: v = W2[(size_t)prm_m * F + prm_f]; break;
        default: v = W1[(size_t)prm_f * FD + prm_m]; break;
    }
    img[idx] = (bf16_t)v;
}

struct RowsArgs {
    const float* x;           // (R, 32): x1 (forward) / dff (backward)
    const float* b1;          // forward: (F)
    const float* b2;          // forward: (32)
    float* out;               // (R, 32): ff (forward, written) / d1 (backward, added to)
    const bf16_t* imgA;       // first product's weight fragments: W1f (forward) / W2t (backward)
    const bf16_t* imgB;       // second product's: W2f (forward) / W1p (backward)
    uint64_t* mask;           // [chunk][row tile][8] words: bit l of word t*4 + e <=> register e of tile t of lane l is live
    int R, F, nrt;            // nrt = ceil(R / 16)
    DropCfg drop;
    uint64_t site;
    float scale;              // applied to the second product (1 / keep)
};

// grid ceil(nrt / (4 RT)), 256 threads; no LDS, no barrier.
// SPLIT (few rows: R <= 4096, RT = 1): grid nrt, 512 threads -- the eight waves of a workgroup share ONE row tile and split the
// hidden dimension (F / 256 chunks each instead of F / 32: the wave's walk over the chunks is the kernel's whole duration), their
// partial output tiles meet in LDS.  At 1024 rows (64 windows) the feed-forward then is 2 + 3 short launches on 64 CUs instead of
// 3 + 4 GEMM launches of up to 512 workgroups.
template <int RT, bool BWD, bool SPLIT = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void ffn32_rows_kernel(const RowsArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const int rt0 = __builtin_amdgcn_readfirstlane(SPLIT ? (int)blockIdx.x : (int)(blockIdx.x * 4 + wave) * RT);
    if (!SPLIT && rt0 >= a.nrt) return;
    const int F = a.F, NC = F >> 5;
    const int c0 = SPLIT ? __builtin_amdgcn_readfirstlane(wave * (NC / NW)) : 0, c1 = SPLIT ? c0 + (NC / NW) : NC;
    bf16x8 xb[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = (rt0 + rt) * 16 + fr;
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
        if (row < a.R) {
            const float4* p = reinterpret_cast<const float4*>(a.x + (size_t)row * FD + fq * 8);
            u = p[0]; v = p[1];
        }
        xb[rt] = cvt8(u, v);
    }
    f32x4 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* fa = reinterpret_cast<const bf16x8*>(a.imgA) + lane;
    const bf16x8* fb = reinterpret_cast<const bf16x8*>(a.imgB) + lane;
    const bool drop = !BWD && a.drop.p > 0.f;
    // the next chunk's weight fragments, bias and (backward) mask words are requested before the current chunk is computed:
    // one chunk is ~4 RT MFMAs, far shorter than a trip to L2
    bf16x8 n1_0 = fa[(c0 * 2) * 64], n1_1 = fa[(c0 * 2 + 1) * 64], n2_0 = fb[(c0 * 2) * 64], n2_1 = fb[(c0 * 2 + 1) * 64];
    float4 nb0 = make_float4(0.f, 0.f, 0.f, 0.f), nb1 = nb0;
    if (!BWD) {
        nb0 = *reinterpret_cast<const float4*>(a.b1 + c0 * 32 + fq * 4);
        nb1 = *reinterpret_cast<const float4*>(a.b1 + c0 * 32 + 16 + fq * 4);
    }
    uint64_t nm[BWD ? RT : 1][8];
    if (BWD) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const uint64_t* rec = a.mask + ((size_t)c0 * a.nrt + min(rt0 + rt, a.nrt - 1)) * 8;
#pragma unroll
            for (int w = 0; w < 8; ++w) nm[rt][w] = rec[w];
        }
    }
    for (int c = c0; c < c1; ++c) {
        const bf16x8 a1_0 = n1_0, a1_1 = n1_1, a2_0 = n2_0, a2_1 = n2_1;
        const float4 bb0 = nb0, bb1 = nb1;
        uint64_t cm[BWD ? RT : 1][8];
        if (BWD) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int w = 0; w < 8; ++w) cm[rt][w] = nm[rt][w];
        }
        {
            const int cn = c + 1 < c1 ? c + 1 : c;
            n1_0 = fa[(cn * 2 + 0) * 64]; n1_1 = fa[(cn * 2 + 1) * 64];
            n2_0 = fb[(cn * 2 + 0) * 64]; n2_1 = fb[(cn * 2 + 1) * 64];
            if (!BWD) {
                nb0 = *reinterpret_cast<const float4*>(a.b1 + cn * 32 + fq * 4);
                nb1 = *reinterpret_cast<const float4*>(a.b1 + cn * 32 + 16 + fq * 4);
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const uint64_t* rec = a.mask + ((size_t)cn * a.nrt + min(rt0 + rt, a.nrt - 1)) * 8;
#pragma unroll
                    for (int w = 0; w < 8; ++w) nm[rt][w] = rec[w];
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (rt > 0 && rt0 + rt >= a.nrt) break;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 h0 = mfma(a1_0, xb[rt], z), h1 = mfma(a1_1, xb[rt], z);
            uint64_t* rec = a.mask + ((size_t)c * a.nrt + (rt0 + rt)) * 8;
            if (!BWD) {
                h0[0] += bb0.x; h0[1] += bb0.y; h0[2] += bb0.z; h0[3] += bb0.w;
                h1[0] += bb1.x; h1[1] += bb1.y; h1[2] += bb1.z; h1[3] += bb1.w;
                float k0[4] = {1.f, 1.f, 1.f, 1.f}, k1[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) {
                    const uint64_t idx4 = (uint64_t)((rt0 + rt) * 16 + fr) * (uint64_t)F + (uint64_t)(c * 32 + fq * 4);
                    dropout_scale4(a.drop, a.site, idx4, k0);
                    dropout_scale4(a.drop, a.site, idx4 + 16, k1);
                }
                uint32_t mv = 0;
                uint64_t m0[4], m1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m0[e] = __builtin_amdgcn_ballot_w64(h0[e] > 0.f && k0[e] != 0.f);
                    m1[e] = __builtin_amdgcn_ballot_w64(h1[e] > 0.f && k1[e] != 0.f);
                    h0[e] = sel_mask(h0[e], m0[e]);
                    h1[e] = sel_mask(h1[e], m1[e]);
                }
                put_lane<0>(mv, (uint32_t)m0[0]);  put_lane<1>(mv, (uint32_t)(m0[0] >> 32));
                put_lane<2>(mv, (uint32_t)m0[1]);  put_lane<3>(mv, (uint32_t)(m0[1] >> 32));
                put_lane<4>(mv, (uint32_t)m0[2]);  put_lane<5>(mv, (uint32_t)(m0[2] >> 32));
                put_lane<6>(mv, (uint32_t)m0[3]);  put_lane<7>(mv, (uint32_t)(m0[3] >> 32));
                put_lane<8>(mv, (uint32_t)m1[0]);  put_lane<9>(mv, (uint32_t)(m1[0] >> 32));
                put_lane<10>(mv, (uint32_t)m1[1]); put_lane<11>(mv, (uint32_t)(m1[1] >> 32));
                put_lane<12>(mv, (uint32_t)m1[2]); put_lane<13>(mv, (uint32_t)(m1[2] >> 32));
                put_lane<14>(mv, (uint32_t)m1[3]); put_lane<15>(mv, (uint32_t)(m1[3] >> 32));
                if (lane < 16) reinterpret_cast<uint32_t*>(rec)[lane] = mv;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h0[e] = sel_mask(h0[e], cm[rt][e]);
                    h1[e] = sel_mask(h1[e], cm[rt][4 + e]);
                }
            }
            const bf16x8 hb = pack8(h0, h1);
            acc[rt][0] = mfma(a2_0, hb, acc[rt][0]);
            acc[rt][1] = mfma(a2_1, hb, acc[rt][1]);
        }
    }
    if (SPLIT) {          // the eight waves' partial tiles of the one row tile: summed by wave 0
        __shared__ __attribute__((aligned(16))) float red[NW][64][8];
        *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc[0][0];
        *reinterpret_cast<f32x4*>(&red[wave][lane][4]) = acc[0][1];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            acc[0][0] += *reinterpret_cast<const f32x4*>(&red[w][lane][0]);
            acc[0][1] += *reinterpret_cast<const f32x4*>(&red[w][lane][4]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = (rt0 + rt) * 16 + fr;
        if (row >= a.R) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float4* dst = reinterpret_cast<float4*>(a.out + (size_t)row * FD + nt * 16 + fq * 4);
            float4 o = make_float4(acc[rt][nt][0] * a.scale, acc[rt][nt][1] * a.scale, acc[rt][nt][2] * a.scale, acc[rt][nt][3] * a.scale);
            if (BWD) {
                const float4 old = *dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            } else {
                const float4 b = *reinterpret_cast<const float4*>(a.b2 + nt * 16 + fq * 4);
                o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
            }
            *dst = o;
        }
    }
}

struct WgArgs {
    const float* x;           // (R, 32) x1
    const float* dff;         // (R, 32)
    const float* b1;          // (F)
    const bf16_t* W1f;
    const bf16_t* W2t;
    const uint16_t* mask16;   // the rows kernel's mask words as 16-bit pieces
    float* slab;              // [row block][chunk][10][64 lanes][4]
    int R, F, nrt, RB;        // RB: rows per row block (multiple of 128)
};
constexpr int WG_SB = 128;    // rows per LDS sub-block

// byte offset of 16-byte chunk q of row r in a (rows x 32 bf16) image: 64-byte rows, chunk slots xor-ed with (r >> 2) & 3
__device__ __forceinline__ unsigned img_off(int r, int q) { return (unsigned)(r * 64 + ((q ^ ((r >> 2) & 3)) << 4)); }

// grid (F / 256, row blocks), 256 threads: wave w owns chunks blockIdx.x * 8 + 2 w, + 1
__global__ __launch_bounds__(256, 2) void ffn32_wgrad_kernel(const WgArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char sx[WG_SB * 64], sd[WG_SB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int F = a.F, NC = F >> 5;
    const int c0 = blockIdx.x * 8 + wave * 2;
    const int row_lo = blockIdx.y * a.RB, row_hi = min(a.R, row_lo + a.RB);
    // this wave's weight fragments and bias
    bf16x8 w1[2][2], w2[2][2];
    float bf[2][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            w1[ch][ft] = reinterpret_cast<const bf16x8*>(a.W1f)[((c0 + ch) * 2 + ft) * 64 + lane];
            w2[ch][ft] = reinterpret_cast<const bf16x8*>(a.W2t)[((c0 + ch) * 2 + ft) * 64 + lane];
            bf[ch][ft] = a.b1[(c0 + ch) * 32 + ft * 16 + fr];
        }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    f32x4 g[2][10];           // per chunk: gW2 [nt*2 + ft], gW1 4 + [dt*2 + ft], gb1 8 + [ft]
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int t = 0; t < 10; ++t) g[ch][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: thread t loads row t >> 1, 16 floats (half t & 1) of x and of dff
    const int srow = tid >> 1, shalf = tid & 1;
    float4 px[4], pd[4];
#define FFN32_FETCH(base_)                                                                                   \
    {                                                                                                        \
        const bool ok_ = (base_) + srow < row_hi;                                                            \
        const int row_ = ok_ ? (base_) + srow : row_lo;                                                      \
        const float4* p_ = reinterpret_cast<const float4*>(a.x + (size_t)row_ * FD + shalf * 16);            \
        const float4* d_ = reinterpret_cast<const float4*>(a.dff + (size_t)row_ * FD + shalf * 16);          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
            px[i] = p_[i]; pd[i] = d_[i];                                                                    \
            if (!ok_) { px[i] = make_float4(0.f, 0.f, 0.f, 0.f); pd[i] = make_float4(0.f, 0.f, 0.f, 0.f); }  \
        }                                                                                                    \
    }
    FFN32_FETCH(row_lo)
    // this lane's 16-bit mask piece of (chunk c0 + ch, row tile rt, ft): mbase[(ch * nrt + rt) * 32 + ft * 16]
    const uint16_t* mbase = a.mask16 + (size_t)c0 * a.nrt * 32 + (fr & 3) * 4 + (fr >> 2);
    unsigned nmk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = i >> 2, ft = (i >> 1) & 1, t = i & 1;
        nmk[i] = mbase[((size_t)ch * a.nrt + min((row_lo >> 4) + t, a.nrt - 1)) * 32 + ft * 16];
    }
    const int tq = fr >> 2, tp = fr & 3;          // transposed reads: this lane addresses k-line (row) 4 fq + tq, columns 4 tp .. + 3
    for (int base = row_lo; base < row_hi; base += WG_SB) {
        if (base != row_lo) __syncthreads();
        *reinterpret_cast<bf16x8*>(sx + img_off(srow, shalf * 2)) = cvt8(px[0], px[1]);
        *reinterpret_cast<bf16x8*>(sx + img_off(srow, shalf * 2 + 1)) = cvt8(px[2], px[3]);
        *reinterpret_cast<bf16x8*>(sd + img_off(srow, shalf * 2)) = cvt8(pd[0], pd[1]);
        *reinterpret_cast<bf16x8*>(sd + img_off(srow, shalf * 2 + 1)) = cvt8(pd[2], pd[3]);
        __syncthreads();
        if (base + WG_SB < row_hi) FFN32_FETCH(base + WG_SB)
        const int nkb = min(4, (row_hi - base + 31) >> 5);
        for (int kb = 0; kb < nkb; ++kb) {
            const int r0 = kb * 32;
            bf16x8 xb[2], db[2], xT[2], dT[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xb[t] = *reinterpret_cast<const bf16x8*>(sx + img_off(r0 + 16 * t + fr, fq));
                db[t] = *reinterpret_cast<const bf16x8*>(sd + img_off(r0 + 16 * t + fr, fq));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {           // A operand [i = column 16 t + fr][k slots: rows 4 fq .. + 3 of both row tiles]
                const int rl = r0 + 4 * fq + tq, rh = rl + 16;
                const unsigned ol = img_off(rl, t * 2 + (tp >> 1)) + (tp & 1) * 8, oh = img_off(rh, t * 2 + (tp >> 1)) + (tp & 1) * 8;
                const s16x4 xl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sx + ol));
                const s16x4 xh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sx + oh));
                const s16x4 dl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sd + ol));
                const s16x4 dh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sd + oh));
                xT[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xl, xh, 0, 1, 2, 3, 4, 5, 6, 7));
                dT[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(dl, dh, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            // this block's mask pieces (requested one block ahead): [ch][ft][t]
            unsigned mk[2][2][2];
#pragma unroll
            for (int i = 0; i < 8; ++i) mk[i >> 2][(i >> 1) & 1][i & 1] = nmk[i];
            {
                const int rtile = (base + r0 + 32) >> 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ch = i >> 2, ft = (i >> 1) & 1, t = i & 1;
                    const int rtc = min(rtile + t, a.nrt - 1);
                    nmk[i] = mbase[((size_t)ch * a.nrt + rtc) * 32 + ft * 16];
                }
            }
            const int rtile = (base + r0) >> 4;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                bf16x8 hB[2], dB[2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    f32x4 h[2], d[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        h[t] = mfma(xb[t], w1[ch][ft], z);            // [i = row 4 fq + e of tile t][j = hidden unit fr of tile ft]
                        d[t] = mfma(db[t], w2[ch][ft], z);
                        // mask piece of (chunk, row tile, ft, e' = fr & 3, q' = fr >> 2): bits = the 16 rows of the tile
                        const unsigned m = (rtile + t < a.nrt ? mk[ch][ft][t] : 0u) >> (4 * fq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool live = (m >> e) & 1u;
                            h[t][e] = live ? h[t][e] + bf[ch][ft] : 0.f;
                            d[t][e] = live ? d[t][e] : 0.f;
                        }
                    }
                    hB[ft] = pack8(h[0], h[1]);
                    dB[ft] = pack8(d[0], d[1]);
                }
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        g[ch][t * 2 + ft] = mfma(dT[t], hB[ft], g[ch][t * 2 + ft]);          // gW2[n][f] += dff[r][n] h[r][f]
                        g[ch][4 + t * 2 + ft] = mfma(xT[t], dB[ft], g[ch][4 + t * 2 + ft]);  // gW1[f][d] += dh[r][f] x[r][d]
                    }
                    g[ch][8 + ft] = mfma(ones, dB[ft], g[ch][8 + ft]);                       // gb1[f] += dh[r][f]
                }
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float* dst = a.slab + ((((size_t)blockIdx.y * NC + (c0 + ch)) * 10 + t) * 64 + lane) * 4;
            *reinterpret_cast<f32x4*>(dst) = g[ch][t];
        }
}

// one thread per (chunk, tile, lane) register quad; sums the row blocks and scatters to gW2 (32, F), gW1 (F, 32), gb1 (F)
__global__ __launch_bounds__(256) void ffn32_wgrad_reduce_kernel(const float* __restrict__ slab, int F, int nrb, float scale,
                                                                 float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2) {
    const int NC = F >> 5, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NC * 640) return;
    const int c = idx / 640, rem = idx - c * 640, t = rem >> 6, l = rem & 63, r = l & 15, q = l >> 4;
    const size_t stride = (size_t)NC * 640 * 4;
    const float* p = slab + (size_t)idx * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int b = 0; b < nrb; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float o[4] = {s.x * scale, s.y * scale, s.z * scale, s.w * scale};
    const int f = c * 32 + (t & 1) * 16 + r;
    if (t < 4) {
        const int n = (t >> 1) * 16 + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) gw2[(size_t)(n + e) * F + f] = o[e];
    } else if (t < 8) {
        const int d = ((t - 4) >> 1) * 16 + 4 * q;
        *reinterpret_cast<float4*>(gw1 + (size_t)f * FD + d) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (q == 0) {
        gb1[f] = o[0];
    }
}

inline int rows_block(int R) {      // rows per row block of the weight-gradient kernel: ~64 blocks (x F / 256 workgroups), whole sub-blocks
    int rb = (R + 63) / 64;
    rb = (rb + WG_SB - 1) / WG_SB * WG_SB;
    return rb < WG_SB ? WG_SB : rb;
}

}  // namespace

static int ffn32_split_max() {        // row tiles up to which the rows kernels split the hidden dimension (IMMTSF_FFN32_SPLIT_MAX: A/B runs)
    constexpr int v = 256;
    return v;
}
bool ffn32_ok(int R, int D, int F, int act, int prec) {
    constexpr bool on = true;
    return on && D == FD && act == 1 && prec == 1 && R >= 256 && F >= 256 && (F % 256) == 0;
}
size_t ffn32_saved_bytes(int R, int F) { return (size_t)4 * F * FD * 2 + (size_t)(F / 32) * ((R + 15) / 16) * 64; }
size_t ffn32_scratch_bytes(int R, int F) {
    const int rb = rows_block(R);
    return (size_t)((R + rb - 1) / rb) * (F / 32) * 640 * 4 * sizeof(float);
}

int ffn32_forward(int R, int F, const DropCfg& dd, uint64_t site, const float* x1, const float* w1, const float* b1, const float* w2,
                  const float* b2, void* saved, float* ff, hipStream_t s) {
    bf16_t* img = static_cast<bf16_t*>(saved);
    hipLaunchKernelGGL(ffn32_images_kernel, dim3((4 * F * FD + 255) / 256), dim3(256), 0, s, w1, w2, F, img);
    IMMTSF_LAUNCH_CHECK();
    RowsArgs a;
    a.x = x1; a.b1 = b1; a.b2 = b2; a.out = ff;
    a.imgA = img; a.imgB = img + (size_t)2 * F * FD;
    a.mask = reinterpret_cast<uint64_t*>(img + (size_t)4 * F * FD);
    a.R = R; a.F = F; a.nrt = (R + 15) / 16;
    a.drop = dd; a.site = site; a.scale = dd.p > 0.f ? dd.inv_keep : 1.f;
    const int rt = a.nrt >= 16384 ? 4 : a.nrt >= 4096 ? 2 : 1;       // >= 2 waves per SIMD first (the Philox draws are VALU work)
    const int grid = (a.nrt + 4 * rt - 1) / (4 * rt);
    // few rows: split the hidden dimension -- over EIGHT waves in the forward, whose chunk is ~450 instructions of Philox, ballots and
    // mask packing issued back to back by a lone wave per SIMD (17.8 us for 1024 rows on four waves); the backward stays on four (an
    // eight-wave backward produced wrong data gradients: DESIGN 6)
    constexpr bool fwd8 = true;
    if (a.nrt <= ffn32_split_max() && fwd8 && ((a.F >> 5) % 8) == 0)
        hipLaunchKernelGGL((ffn32_rows_kernel<1, false, true, 8>), dim3(a.nrt), dim3(512), 0, s, a);
    else if (a.nrt <= ffn32_split_max()) hipLaunchKernelGGL((ffn32_rows_kernel<1, false, true>), dim3(a.nrt), dim3(256), 0, s, a);
    else if (rt == 4) hipLaunchKernelGGL((ffn32_rows_kernel<4, false>), dim3(grid), dim3(256), 0, s, a);
    else if (rt == 2) hipLaunchKernelGGL((ffn32_rows_kernel<2, false>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ffn32_rows_kernel<1, false>), dim3(grid), dim3(256), 0, s, a);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int ffn32_backward(int R, int F, const DropCfg& dd, const float* x1, const float* b1, const float* dff, const void* saved, void* scratch,
                   float* d1, float* gw1, float* gb1, float* gw2, hipStream_t s) {
    const bf16_t* img = static_cast<const bf16_t*>(saved);
    const float scale = dd.p > 0.f ? dd.inv_keep : 1.f;
    const int nrt = (R + 15) / 16;
    {   // d1 += ((dff W2) . mask / keep) W1
        RowsArgs a;
        a.x = dff; a.b1 = nullptr; a.b2 = nullptr; a.out = d1;
        a.imgA = img + (size_t)1 * F * FD; a.imgB = img + (size_t)3 * F * FD;
        a.mask = reinterpret_cast<uint64_t*>(const_cast<bf16_t*>(img) + (size_t)4 * F * FD);
        a.R = R; a.F = F; a.nrt = nrt;
        a.drop = dd; a.site = 0; a.scale = scale;
        const int rt = nrt >= 4096 ? 2 : 1;          // (the mask words of RT row tiles x 2 chunks live in SGPRs: RT <= 2)
        const int grid = (nrt + 4 * rt - 1) / (4 * rt);
        if (nrt <= ffn32_split_max()) hipLaunchKernelGGL((ffn32_rows_kernel<1, true, true>), dim3(nrt), dim3(256), 0, s, a);
        else if (rt == 2) hipLaunchKernelGGL((ffn32_rows_kernel<2, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ffn32_rows_kernel<1, true>), dim3(grid), dim3(256), 0, s, a);
        IMMTSF_LAUNCH_CHECK();
    }
    WgArgs w;
    w.x = x1; w.dff = dff; w.b1 = b1;
    w.W1f = img; w.W2t = img + (size_t)1 * F * FD;
    w.mask16 = reinterpret_cast<const uint16_t*>(img + (size_t)4 * F * FD);
    w.slab = static_cast<float*>(scratch);
    w.R = R; w.F = F; w.nrt = nrt; w.RB = rows_block(R);
    const int nrb = (R + w.RB - 1) / w.RB;
    hipLaunchKernelGGL(ffn32_wgrad_kernel, dim3(F / 256, nrb), dim3(256), 0, s, w);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ffn32_wgrad_reduce_kernel, dim3(((F / 32) * 640 + 255) / 256), dim3(256), 0, s, w.slab, F, nrb, scale, gw1, gb1, gw2);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
