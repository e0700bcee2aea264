// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the IMM-TSF fusion hot path.
// Wavefront = 64 lanes everywhere; nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IMMTSF_WAVE 64

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- status codes of the C ABI (see include/immtsf.h) ------------------------------------------
#define IMMTSF_OK 0
#define IMMTSF_EINVAL (-1)
#define IMMTSF_EWORKSPACE (-2)
#define IMMTSF_EUNSUPPORTED (-3)

#define IMMTSF_LAUNCH_CHECK()                           \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- wave / block reductions ----------------------------------------------------------------------
// Cross-lane steps without the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipeline round trip per step and
// an address register); inside a row of 16 lanes the DPP modifiers of the VALU add / max do the exchange for free, and
// gfx950's v_permlane16_swap / v_permlane32_swap exchange rows (swap(v, v) returns (v of the even rows | v of the odd rows)
// replicated, so their sum / max is the value combined with lane ^ 16, resp. lane ^ 32).
typedef unsigned immtsf_u2 __attribute__((ext_vector_type(2)));
#define IMMTSF_DPP(v, ctrl) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a row, in every lane of the row
    v += IMMTSF_DPP(v, 0xB1);      // quad_perm [1,0,3,2]
    v += IMMTSF_DPP(v, 0x4E);      // quad_perm [2,3,0,1]
    v += IMMTSF_DPP(v, 0x141);     // row_half_mirror
    v += IMMTSF_DPP(v, 0x140);     // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, IMMTSF_DPP(v, 0xB1));
    v = fmaxf(v, IMMTSF_DPP(v, 0x4E));
    v = fmaxf(v, IMMTSF_DPP(v, 0x141));
    v = fmaxf(v, IMMTSF_DPP(v, 0x140));
    return v;
}
__device__ __forceinline__ float xor16_sum(float v) {      // v + v[lane ^ 16]
    const immtsf_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {      // v + v[lane ^ 32]
    const immtsf_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const immtsf_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const immtsf_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum over the lanes with the same lane % CT (CT a power of two <= 64): the row steps as rotations (a rotation orbit of a row is
// the coset)
__device__ __forceinline__ float coset_sum(float v, int CT) {
    if (CT <= 32) v = xor32_sum(v);
    if (CT <= 16) v = xor16_sum(v);
    if (CT <= 8) v += IMMTSF_DPP(v, 0x128);     // row_ror:8
    if (CT <= 4) v += IMMTSF_DPP(v, 0x124);     // row_ror:4
    if (CT <= 2) v += IMMTSF_DPP(v, 0x122);     // row_ror:2
    if (CT <= 1) v += IMMTSF_DPP(v, 0x121);     // row_ror:1
    return v;
}
// sum over the aligned group of CT consecutive lanes a lane belongs to (CT a power of two <= 64), in every lane of the group
__device__ __forceinline__ float group_sum(float v, int CT) {
    if (CT >= 2) v += IMMTSF_DPP(v, 0xB1);      // quad_perm [1,0,3,2]
    if (CT >= 4) v += IMMTSF_DPP(v, 0x4E);      // quad_perm [2,3,0,1]
    if (CT >= 8) v += IMMTSF_DPP(v, 0x141);     // row_half_mirror
    if (CT >= 16) v += IMMTSF_DPP(v, 0x140);    // row_mirror
    if (CT >= 32) v = xor16_sum(v);
    if (CT >= 64) v = xor32_sum(v);
    return v;
}
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, IMMTSF_DPP(v, 0xB1));
    v = fmaxf(v, IMMTSF_DPP(v, 0x4E));
    return fmaxf(v, IMMTSF_DPP(v, 0x141));
}
// the value of lane l (wave-uniform l) in every lane: v_readlane instead of ds_bpermute
__device__ __forceinline__ float lane_bcast(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` is >= 16 floats of LDS.
// Every thread gets the result.  Contains two barriers.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

// ---- Philox4x32-10 counter RNG for in-kernel dropout ------------------------------------------------
// One call yields 4 x 32 random bits for (seed, subsequence = dropout site, counter = element/4).
// The same (seed, site, element) always gives the same bit, so backward recomputes masks instead of
// storing them, and tests can export them (immtsf_dropout_mask).
struct Philox4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t site, uint64_t ctr) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)site, c3 = (uint32_t)(site >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 product each (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the integer multiplies are
        // quarter rate, and they are what dropout inside the many-element kernels costs (ffn32 forward: 77 of them per 32 x 16 chunk)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

// keep-probability test for element `idx` of dropout site `site`: returns 1.f/(1-p) or 0.f.
// p == 0 never calls the generator.  The uniform is the top 24 bits so that the comparison is exact
// in float on both host and device.
__host__ __device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t site, uint64_t idx, float p,
                                                        float inv_keep) {
    if (p <= 0.f) return 1.f;
    const Philox4 r = philox4x32_10(seed, site, idx >> 2);
    const uint32_t lane = (uint32_t)(idx & 3);
    const uint32_t bits = lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w;
    const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.f;
}

struct DropCfg {
    uint64_t seed;
    float p;          // drop probability (0 => identity)
    float inv_keep;   // 1/(1-p)
    // optional DEVICE counter added to `seed` at run time: lets a captured hipGraph draw fresh masks on every
    // replay (the host-side seed is baked into the graph's kernel arguments, the counter is not)
    const uint64_t* seed_dev;
};

__device__ __forceinline__ float dropout_scale(const DropCfg& d, uint64_t site, uint64_t idx) {
    if (d.p <= 0.f) return 1.f;
    const uint64_t seed = d.seed + (d.seed_dev ? *d.seed_dev : 0ull);
    return dropout_scale(seed, site, idx, d.p, d.inv_keep);
}

// four consecutive elements idx4 .. idx4+3 (idx4 % 4 == 0) share ONE Philox call: same bits as four dropout_scale calls
__device__ __forceinline__ void dropout_scale4(const DropCfg& d, uint64_t site, uint64_t idx4, float (&s)[4]) {
    if (d.p <= 0.f) { s[0] = s[1] = s[2] = s[3] = 1.f; return; }
    const uint64_t seed = d.seed + (d.seed_dev ? *d.seed_dev : 0ull);
    const Philox4 r = philox4x32_10(seed, site, idx4 >> 2);
    const uint32_t bits[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = (float)(bits[e] >> 8) * (1.0f / 16777216.0f) >= d.p ? d.inv_keep : 0.f;
}

// dropout sites (Philox subsequence ids) -- one per dropout call of the reference modules
enum : uint64_t {
    SITE_T2V_ATTN = 1,   // fusions/TTF_T2V_XAttn.py:79-84  (attention-weight dropout inside MHA)
    SITE_T2V_OUT = 2,    // fusions/TTF_T2V_XAttn.py:179
    SITE_REC_OUT = 3,    // fusions/TTF_RecAvg.py:106
    SITE_XADD_ATTN = 4,  // fusions/MMF_XAttn_Add.py:42-47
    SITE_XADD_OUT = 5,   // fusions/MMF_XAttn_Add.py:95
    SITE_GR_OUT = 6,     // fusions/MMF_GR_Add.py:51
    SITE_LAYER_BASE = 16 // layers/* dropout sites: SITE_LAYER_BASE + caller-chosen id
};
