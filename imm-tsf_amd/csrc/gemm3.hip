// Persistent "ping-pong" bf16 GEMM for the many-rows regime of the fusion path (forward / data-gradient projections at
// >= 512 windows per GPU: M = 16 k ... 260 k rows, N = 768 ... 1536, K = 768 ... 4096; reference call sites
// fusions/TTF_T2V_XAttn.py:70-84 input_proj / KV_proj / MHA in-proj, fusions/MMF_XAttn_Add.py:36-47).
//
// Why a third GEMM: gemm2.hip's one-barrier-per-K-step ring runs these shapes at 4.5-9 % of the bf16 MFMA peak
// (profiles/r02_*): per 256-row tile the ring prologue and the fp32 epilogue through LDS (~24 us per tile at K = 768) are as
// long as the K loop itself and overlap with nothing, because a 128 KB ring leaves one workgroup per CU.  Here
//   * ONE persistent 512-thread workgroup per CU walks its tiles; the LDS-DMA stream simply continues into the next tile's
//     first K-tiles, so there is no ring prologue after the first tile;
//   * the 8 waves are two groups of 4 (waves 0-3 own the top half of the tile's rows, waves 4-7 the bottom half); wave w and
//     wave w+4 share a SIMD.  The groups run the same program one s_barrier apart: while one group issues its 16 MFMAs of a
//     phase (one 64 x 32 quadrant of the wave's 128 x 64 tile over the 64-deep K-tile), the other reads its next fragments
//     from LDS and issues its share of the LDS-DMA loads -- the matrix pipe of every SIMD always has one wave in its
//     MFMA section (MI355X_MICROARCH "two waves per SIMD": split roles by wave >= 4);
//   * LDS-DMA (buffer_load_dwordx4 ... lds) half-tiles (16 KB: 128 rows x 64 k) are issued one per phase, 1-1.25 K-tiles
//     ahead of their first read, and retired by counted s_waitcnt vmcnt(N) twice per K-tile; raw s_barrier only;
//   * the epilogue never touches LDS or a barrier: the product is formed transposed (acc = B-fragment x A-fragment), so a lane
//     holds 4 consecutive columns of one row and stores 16 bytes (fp32) / 8 bytes (bf16) straight from its accumulators; the
//     stores of quadrant q of tile i are issued in phase q of tile i+1's FIRST K-tile, in the wave's LDS-read section,
//     just before that phase's MFMAs overwrite the quadrant (first MFMA of a tile takes C = 0);
//   * bias / add_vec / row flags of a tile come through LDS by LDS-DMA as well ("epilogue vectors"), so no ordinary
//     global load ever makes hipcc drain the LDS-DMA queue with a vmcnt(0).
// Buffer descriptors do all the edge handling: rows past M and k-lines past K are out of range of the descriptor and
// read as zeros; chunks past the K tail / column edge get an out-of-range offset; stores past M are dropped.
//
// LDS images are gemm2.hip's (same conflict-free permutations, verified there):
//   R image (rows x 64 k, k contiguous): chunk (row, c) at position row*8 + (c ^ ((row>>1)&7)); fragment = ds_read_b128.
//   T image (64 k-lines x 64-column blocks): chunk (k, n8) at k*8 + (n8 ^ sw(k)); fragment = 2 x ds_read_b64_tr_b16.
#include "gemm.hpp"
#include <string.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }


// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}
// ds_read_b64_tr_b16 as inline asm: through the builtin hipcc cannot tell the read from the LDS-DMA writes in flight and puts
// an s_waitcnt vmcnt(0) in front of every group of transposed reads (the whole load pipeline drained four times per K-tile).
// The data is waited for by the read section's own s_waitcnt lgkmcnt(0) + sched_barrier ahead of the MFMAs that use it.
// ds_read_b128 the same way: hipcc orders an LDS-DMA behind every pending ds_read it knows of (s_waitcnt lgkmcnt before the
// buffer_load ... lds), which serialises a read section into "reads, their latency, then the DMA issue"; as asm the reads go
// first and the DMA issue overlaps their latency.  (The DMA targets of a phase never overlap what the phase reads.)
template <int OFF> __device__ __forceinline__ bf16x8 lds_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset immediate");
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ s16x4 lds_tr16(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset immediate");
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

constexpr unsigned G3_OOB = 0xFFFFFF00u;        // a voffset no descriptor below reaches (num_records are clamped to it)

struct G3Args {
    const void* A;            // bf16
    const void* B;            // bf16
    float* C;                 // fp32 result or null
    void* Ch;                 // bf16 result or null
    const float* bias;        // [N] or null
    const float* add_vec;     // [N] or null
    const int* row_flag;      // int32 per row group: rows with row_flag[m / row_flag_div] == 0 are zeroed (before add_vec), or null
    const int* dyn;           // device row count overriding M, or null
    const int* dynk;          // split mode: device reduction length overriding K, or null
    float* slab;              // split mode: [items][BM x 256] fp32 partial tiles (item = split * tiles + tile: neighbouring items --
                              // an XCD's share of a round -- are the tiles of ONE K-range and find each other's A / B panels in L2)
    float* slab_b;            // split mode, bias gradient requested: [items][BM] partial column sums of A (tiles of column 0 only)
    int splits;               // split mode: K-ranges per tile
    unsigned s_magic;         // id / tiles as __umulhi(id, magic) (split mode); 0: one tile
    int tiles_total;          // split mode: tiles (items = tiles_total * splits)
    int row_flag_div;
    unsigned rf_magic;        // floor(2^32 / row_flag_div) (0: div == 1): m / div = __umulhi(m, magic) (+1 after one check)
    int M, N, K;
    int lda, ldb, ldc, ldch;
    float alpha;
    int tiles_n;
    unsigned tn_magic;        // id / tiles_n as __umulhi(id, magic); 0: tiles_n == 1
};

struct G3Page { unsigned int w[512]; };
constexpr G3Page g3_make_page() {
    G3Page p{};
    for (int i = 256; i < 512; ++i) p.w[i] = 1u;
    return p;
}
__device__ __attribute__((aligned(16))) const G3Page g3_zero_page = g3_make_page();     // 1 KiB of zeros, 1 KiB of int32 ones

// where the LDS-DMA loads of one K-tile come from (all wave-uniform)
struct G3Cur {
    int id, kt;               // work item (tile, or tile * splits + split; >= number of items: past the end), K-tile inside it
    int kt0;                  // first K-tile of the item's reduction range
    int row0, col0;
    __amdgpu_buffer_rsrc_t ra, rb;
    int krem0;                // K range of the tile's reduction (0 past the end: every chunk invalid)
    int ncolA, ncolB;         // T images: valid columns from the tile's first column on
};

// EPI: 0 = x = alpha * acc only (no epilogue vectors are loaded or read); 1 = bias / add_vec / row flags through LDS
// VAR: 0 the kernel; 2 / 3 / 4 = timing experiments of tools/gemm3_bench.py probe (no LDS fragment reads / no LDS-DMA / neither:
//      they compute garbage)
// SPLIT (TN weight gradients: few tiles, long reduction): a work item is (tile, K-range); its fp32 partial tile goes to
//      slab[item] and -- for the tiles of column 0, first wave column -- the column sums of A (the bias gradient, one more MFMA
//      against a fragment of ones) to slab_b[item]; gemm3_reduce_kernel sums the ranges.  OUT = 1, EPI = 0.
// BN: tile width, 256 or 192 (768 = 4 x 192: 512 tiles = exactly two rounds at 32 768 rows where 256-wide tiles make 1.5)
template <bool TA, bool TB, int BM, int OUT, int EPI, int VAR = 0, bool SPLIT = false, int BN = 256>
__global__ __launch_bounds__(512) void gemm3_kernel(const G3Args g) {
    constexpr int GH = BM / 2;                  // rows of a wave group
    constexpr int TMW = GH / 16;                // 16-row MFMA tiles per wave (8 or 4)
    constexpr int HM = TMW / 2;                 // ... per phase (row half mh of the wave's tile)
    constexpr int GQ = GH / 2;                  // rows of one group in one row half (64 or 32)
    constexpr int LPA = GH * 8 / 512;           // LDS-DMA loads per thread per A half-tile (2 or 1)
    constexpr int TNW = BN / 64;                // 16-column MFMA tiles per wave (4 or 3): wave wc owns columns wc * BN/4 .. of the tile
    constexpr int WCOLS = BN / 4;
    constexpr int LPB = BN * 8 / 512;           // LDS-DMA loads per thread for the B region of a stage (4 or 3)
    constexpr int A_HALF = GH * 128, B_REG = BN * 128;
    constexpr int STAGE = 2 * A_HALF + B_REG;
    constexpr int EVB = 8 * 1024;               // epilogue vectors of one tile: 1 KiB per wave
    constexpr int NKIND = (OUT & 1) + ((OUT >> 1) & 1);
    constexpr int NS = HM * TNW * NKIND + (SPLIT ? HM : 0);    // stores per phase of a storing K-tile
    constexpr int XB = LPA + LPB;               // loads a thread issues in phase B (A row half 0 + the B region of K-tile t+2)
    constexpr int XA = LPA;                     // ... in phase A (A row half 1 of K-tile t+1)
    constexpr int EV1 = EPI ? 1 : 0;
    constexpr int ST_AUX = 0;        // default cache policy: nt result stores measured 1.1-1.5x slower, sc1 (write-through) 0-10 % slower (r03)
    static_assert(BM == 256 || BM == 128, "tile heights");
    static_assert(BN == 256 || BN == 192, "tile widths");
    static_assert(!SPLIT || (OUT == 1 && EPI == 0 && TA && TB), "split mode: fp32 partial tiles of a TN product");
    auto cap63 = [](int v) constexpr { return v > 63 ? 63 : v; };      // a smaller count only waits for more

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 2 * EVB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    int M = g.M;
    if (g.dyn) M = *g.dyn;
    int K = g.K;
    if (SPLIT && g.dynk) K = *g.dynk;
    const int N = g.N;
    const int S = SPLIT ? g.splits : 1;
    int nk = (K + 63) >> 6;                     // K-tiles per work item
    if (SPLIT) {
        nk = (nk + S - 1) / S;
        nk = nk < 2 ? 2 : nk;                   // (the loop needs two K-tiles per item; ranges past K read zeros)
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = g.tiles_n;
    const int nt = tiles_m * tiles_n * S;       // work items
    const int G = gridDim.x;                    // a multiple of 8
    const int first_id = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);    // round r: tile r*G + first_id, neighbours share an XCD
    if (first_id >= nt) return;

    // ---------------------------------------------------------------- LDS map of one stage (64-deep K-tile)
    //   [ Am0 | Am1 | B ]   Am<mh>: row half mh of BOTH wave groups (group 0's GQ rows, then group 1's), what phase mh reads;
    //                       B: the tile's BN columns (wave wc owns columns wc * BN/4 .. + BN/4 - 1), all read in phase A
    // ---------------------------------------------------------------- per-thread constants of the LDS-DMA loads
    // position p = tid (+ 512 for a thread's second load) of a half-tile image
    const int lda2 = g.lda * 2, ldb2 = g.ldb * 2;
    int voA, kcA, c8A = 0, voB, kcB, c8B = 0;
    int stepA;                                  // voffset of a thread's second A load relative to its first
    {
        const int row = tid >> 3, slot = tid & 7;
        const int cR = slot ^ ((row >> 1) & 7);
        const int k = tid >> 3, n8 = slot ^ ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
        if (!TA) {
            // image row r of Am<mh> is tile row (r / GQ) * GH + mh * GQ + r % GQ.  BM = 256: a thread's two loads are rows
            // r and r + 64 = the same row of the two groups; BM = 128: one load, r = tid >> 3 covers both groups
            const int trow = LPA == 2 ? row : (row / GQ) * GH + (row % GQ);
            voA = trow * lda2 + cR * 16;
            kcA = cR * 8;
            stepA = GH * lda2;
        } else {
            // image column c of Am<mh>: same mapping on columns; a 64-column block is one group's (BM = 256) or both groups' 32
            const int tcol = LPA == 2 ? n8 * 8 : ((n8 * 8) / GQ) * GH + ((n8 * 8) % GQ);
            voA = k * lda2 + tcol * 2;
            kcA = k;
            c8A = tcol;
            stepA = GH * 2;
        }
        if (!TB) { voB = row * ldb2 + cR * 16; kcB = cR * 8; }
        else     { voB = k * ldb2 + n8 * 16;   kcB = k; c8B = n8 * 8; }
    }
    const int stepB = TB ? 128 : 64 * ldb2;                     // a thread's second B load: 64 rows / the next 64-column block
    const int halfA = TA ? GQ * 2 : GQ * lda2;                  // row half 1 of A: GQ rows (columns) further inside each group
    const int kstepA = TA ? 64 * lda2 : 128, kstepB = TB ? 64 * ldb2 : 128;      // bytes per K-tile

    // ---------------------------------------------------------------- fragment read offsets (inside a half-tile)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // R image: byte offset of (row fr, k-chunk kk*4 + fq) of a 16-row tile; tiles are 2048 bytes apart
    // T image: lane (fq, q = fr>>2, p = fr&3) reads k-lines kk*32 + fq*8 + q (and + 4), columns 16*tl + 4p .. 4p+3: chunk
    //          n8 = 2*(tl&3) + (p>>1) of the k-line's block (tl>>2), stored at slot n8 ^ sw
    int ofA[4], ofB[4];       // R: [kk]; T: A by tile ii of the phase (this wave's group folded in), B by tile j of the wave
    {
        const int q = fr >> 2, p = fr & 3, sw = ((q >> 1) << 1) | ((fq & 1) << 2);
        const int tbase = ((fq * 8 + q) << 7) + (p & 1) * 8;
        if (!TA) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) ofA[kk] = (grp * GQ + fr) * 128 + (((kk * 4 + fq) ^ (fr >> 1)) << 4);
            ofA[2] = ofA[3] = 0;
        } else {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {          // tile grp*HM + ii of the half-tile
                const int tl = grp * HM + (ii < HM ? ii : 0);
                ofA[ii] = (tl >> 2) * 8192 + tbase + (((2 * (tl & 3) + (p >> 1)) ^ sw) << 4);
            }
        }
        if (!TB) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) ofB[kk] = (wc * WCOLS + fr) * 128 + (((kk * 4 + fq) ^ (fr >> 1)) << 4);
            ofB[2] = ofB[3] = 0;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {             // tile wc * TNW + j of the region: 64-column block tl >> 2, position tl & 3
                const int tl = wc * TNW + (j < TNW ? j : 0);
                ofB[j] = (tl >> 2) * 8192 + tbase + (((2 * (tl & 3) + (p >> 1)) ^ sw) << 4);
            }
        }
    }
    // A fragment: 16-row tile II of this wave's rows in the Am half-tile at LDS byte offset `base`, k-half KK
    auto fragA = [&](unsigned base, auto ii_c, auto kk_c) __attribute__((always_inline)) -> bf16x8 {
        constexpr int ii = decltype(ii_c)::value, kk = decltype(kk_c)::value;
        if constexpr (VAR == 2 || VAR == 4) {
            bf16x8 d;
            asm volatile("" : "=v"(d));
            return d;
        } else if constexpr (!TA) {
            return lds_b128<ii * 2048>(lds0 + base + (unsigned)ofA[kk]);
        } else {
            const unsigned a0 = lds0 + base + (unsigned)ofA[ii];
            const s16x4 lo = lds_tr16<kk * 4096>(a0);
            const s16x4 hi = lds_tr16<kk * 4096 + 512>(a0);
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto fragB = [&](unsigned base, auto jj_c, auto kk_c) __attribute__((always_inline)) -> bf16x8 {     // tile j of this wave's columns
        constexpr int jj = decltype(jj_c)::value, kk = decltype(kk_c)::value;
        if constexpr (VAR == 2 || VAR == 4) {
            bf16x8 d;
            asm volatile("" : "=v"(d));
            return d;
        } else if constexpr (!TB) {
            return lds_b128<jj * 2048>(lds0 + base + (unsigned)ofB[kk]);
        } else {
            const unsigned a0 = lds0 + base + (unsigned)ofB[jj];
            const s16x4 lo = lds_tr16<kk * 4096>(a0);
            const s16x4 hi = lds_tr16<kk * 4096 + 512>(a0);
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };

    // ---------------------------------------------------------------- load cursors
    // bytes of `rows` rows of pitch `pitch2` bytes (minus `sub`), clamped to the largest num_records used here -- on the scalar unit
    auto nrec_of = [&](int rows, int pitch2, int sub) __attribute__((always_inline)) -> int {
        const unsigned r = rows > 0 ? (unsigned)rows : 0u;
        const unsigned lo = r * (unsigned)pitch2, hi = __umulhi(r, (unsigned)pitch2);
        unsigned v = hi ? G3_OOB : (lo < G3_OOB ? lo : G3_OOB);
        v -= (unsigned)sub;       // sub < v by construction (a tile's first column lies inside the first k-line); NOT a saturating
                                  // subtract: that one only exists on the VALU and drags the descriptor into VGPRs (waterfall loops)
        return (int)v;
    };
    auto set_tile = [&](G3Cur& c, int id) __attribute__((always_inline)) {
        c.id = id;
        c.kt = 0;
        const bool ok = id < nt;
        const int idc = ok ? id : 0;
        const int split = SPLIT ? (g.s_magic ? (int)__umulhi((unsigned)idc, g.s_magic) : idc) : 0;      // (magic 0: one tile, id / 1)
        const int tile = SPLIT ? idc - split * g.tiles_total : idc;
        const int tm = g.tn_magic ? (int)__umulhi((unsigned)tile, g.tn_magic) : tile;
        const int tn = tile - tm * tiles_n;
        c.row0 = tm * BM;
        c.col0 = tn * BN;
        c.kt0 = split * nk;
        const int kend = (c.kt0 + nk) * 64;
        c.krem0 = ok ? (kend < K ? kend : K) : 0;
        const char* pa = reinterpret_cast<const char*>(g.A);
        const char* pb = reinterpret_cast<const char*>(g.B);
        if (!TA) c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(pa + (size_t)(unsigned)c.row0 * (unsigned)lda2), (short)0, nrec_of(M - c.row0, lda2, 0), 0x00020000);
        else     c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(pa + (size_t)c.row0 * 2), (short)0, nrec_of(K, lda2, c.row0 * 2), 0x00020000);
        if (!TB) c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)(pb + (size_t)(unsigned)c.col0 * (unsigned)ldb2), (short)0, nrec_of(N - c.col0, ldb2, 0), 0x00020000);
        else     c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)(pb + (size_t)c.col0 * 2), (short)0, nrec_of(K, ldb2, c.col0 * 2), 0x00020000);
        c.ncolA = M - c.row0;
        c.ncolB = N - c.col0;
    };
    auto advance = [&](G3Cur& c) __attribute__((always_inline)) {
        c.kt += 1;
        if (c.kt == nk) set_tile(c, c.id + G);
    };
    // cursor c's K-tile into LDS at `dst` (wave-uniform): A row half mh (a half-tile) / the whole B region
    auto issueA = [&](const G3Cur& c, int mh, unsigned char* dst) __attribute__((always_inline)) {
        if (VAR == 3 || VAR == 4) return;
        const int krem = c.krem0 - (c.kt0 + c.kt) * 64;
        const unsigned uadd = (unsigned)((c.kt0 + c.kt) * kstepA + mh * halfA);
#pragma unroll
        for (int i = 0; i < LPA; ++i) {
            bool ok = kcA < krem;
            if (TA) ok = ok & (mh * GQ + i * GH + c8A + 8 <= c.ncolA);
            const unsigned v = ok ? (unsigned)voA + (unsigned)(i * stepA) + uadd : G3_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_vp)(dst + i * 8192 + wave * 1024), 16, (int)v, 0, 0, 0);
        }
    };
    auto issueB = [&](const G3Cur& c, unsigned char* dst) __attribute__((always_inline)) {
        if (VAR == 3 || VAR == 4) return;
        const int krem = c.krem0 - (c.kt0 + c.kt) * 64;
        const unsigned uadd = (unsigned)((c.kt0 + c.kt) * kstepB);
#pragma unroll
        for (int i = 0; i < LPB; ++i) {
            bool ok = kcB < krem;
            if (TB) ok = ok & (i * 64 + c8B + 8 <= c.ncolB);
            const unsigned v = ok ? (unsigned)voB + (unsigned)(i * stepB) + uadd : G3_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rb, (lds_vp)(dst + i * 8192 + wave * 1024), 16, (int)v, 0, 0, 0);
        }
    };
    // epilogue vectors of the tile at (row0, col0): ONE 16-byte-per-lane LDS-DMA load per wave (every wave's vmcnt sees the
    // same count), source picked per wave without a branch:
    //   wave 0  bias[col0 .. col0+255]      -> ev + 0       (absent: zeros)
    //   wave 1  add_vec[col0 .. col0+255]   -> ev + 1024    (absent: zeros)
    //   waves 2-5  lane l: the 16 bytes at &row_flag[(row0 + (w-2)*64 + l) / div] -> ev + 2048 + (w-2)*1024 + l*16: the flag of
    //           tile row r is the int at ev + 2048 + r*16 (absent: ones)
    //   waves 6, 7  zeros (dummy)
    auto issueEV = [&](int row0, int col0, unsigned char* ev) __attribute__((always_inline)) {
        const char* zp = reinterpret_cast<const char*>(g3_zero_page.w);      // 1 KiB of zeros, then 1 KiB of int32 ones
        const float* vec = wave == 0 ? g.bias : g.add_vec;
        const bool isvec = wave < 2 && vec != nullptr;
        const bool isflag = wave >= 2 && wave < 6;
        const bool realflag = isflag && g.row_flag != nullptr;
        const char* base = isvec ? reinterpret_cast<const char*>(vec + col0)
                         : realflag ? reinterpret_cast<const char*>(g.row_flag)
                         : isflag ? zp + 1024 : zp;
        const int nrec = isvec ? (N - col0) * 4 : realflag ? (int)(((unsigned)(M - 1) / (unsigned)g.row_flag_div + 1) * 4) : 1024;
        const unsigned row = (unsigned)(row0 + (wave - 2) * 64 + lane);
        unsigned q = g.rf_magic ? __umulhi(row, g.rf_magic) : row;
        q += (row - q * (unsigned)g.row_flag_div >= (unsigned)g.row_flag_div) ? 1u : 0u;
        const int vo = realflag ? (int)(q * 4) : lane * 16;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, nrec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vp)(ev + wave * 1024), 16, vo, 0, 0, 0);
    };

    // ---------------------------------------------------------------- accumulators and the epilogue
    f32x4 acc[TMW][TNW];
    // split mode stores the item's tile at rows item*BM .. of the slab (pitch 256 floats, all 256 columns, alpha applied later)
    const __amdgpu_buffer_rsrc_t rC = SPLIT ? __builtin_amdgcn_make_buffer_rsrc((void*)g.slab, (short)0, nrec_of(nt * BM, BN * 4, 0), 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)g.C, (short)0, nrec_of(M, g.ldc * 4, 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)g.Ch, (short)0, nrec_of(M, g.ldch * 2, 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rSB = __builtin_amdgcn_make_buffer_rsrc((void*)g.slab_b, (short)0, SPLIT && g.slab_b ? nt * BM * 4 : 0, 0x00020000);
    const unsigned ldc4 = SPLIT ? (unsigned)(BN * 4) : (unsigned)g.ldc * 4u, ldh2 = (unsigned)g.ldch * 2u;
    const int Nst = SPLIT ? BN : N;            // columns that exist in the store target
    const float alpha = SPLIT ? 1.f : g.alpha;
    f32x4 accb[SPLIT ? TMW : 1];                // split mode: column sums of A for this wave's rows (tiles of column 0, wave column 0)
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    // store row half mh of this wave's accumulators (HM x 4 MFMA tiles) of the tile at (row0, col0); ev = that tile's epilogue
    // vectors in LDS.  x = rowflag(alpha * acc + bias) + add_vec.  Rows past M are out of range of the descriptors (dropped);
    // columns past N get an out-of-range offset.
    auto store_half = [&](auto mh_c, int row0, int col0, const unsigned char* ev, bool with_b) __attribute__((always_inline)) {
        constexpr int mh = decltype(mh_c)::value;
        // an opaque zero: keeps the offsets below from being hoisted out of the tile loop into two dozen long-lived VGPRs
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        const int colw = wc * WCOLS + fq * 4 + oz;                  // column inside the tile of this lane's 4 values, j = 0
        const int rloc = grp * GH + mh * GQ + fr;                   // row inside the tile, ii = 0
        const unsigned vrow = (unsigned)(row0 + rloc);
        const unsigned voC = vrow * ldc4 + (unsigned)(col0 + colw) * 4u, voH = vrow * ldh2 + (unsigned)(col0 + colw) * 2u;
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int ct = colw + j * 16;
            const bool cok = col0 + ct < Nst;                        // N % 4 == 0: a lane's 4 columns are in or out together
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, a4 = {0.f, 0.f, 0.f, 0.f};
            if (EPI) {
                b4 = *reinterpret_cast<const f32x4*>(ev + ct * 4);
                a4 = *reinterpret_cast<const f32x4*>(ev + 1024 + ct * 4);
            }
#pragma unroll
            for (int ii = 0; ii < HM; ++ii) {
                f32x4 v = acc[mh * HM + ii][j];
                if (EPI) {
                    const float live = *reinterpret_cast<const int*>(ev + 2048 + (rloc + ii * 16) * 16) != 0 ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (alpha * v[e] + b4[e]) * live + a4[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = alpha * v[e];
                }
                if (OUT & 1) {
                    const unsigned vo = cok ? voC + (unsigned)(ii * 16) * ldc4 + (unsigned)(j * 64) : G3_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rC, (int)vo, 0, ST_AUX);
                }
                if (OUT & 2) {
                    const bf16x4 h = __builtin_convertvector(v, bf16x4);
                    const unsigned vo = cok ? voH + (unsigned)(ii * 16) * ldh2 + (unsigned)(j * 32) : G3_OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, h), rH, (int)vo, 0, ST_AUX);
                }
            }
        }
        if (SPLIT) {        // the column sums (every register of a lane holds its row's sum); every wave issues the stores, only the
                            // waves that formed the sums with an in-range offset (each wave's vmcnt must see the same count)
#pragma unroll
            for (int ii = 0; ii < HM; ++ii) {
                const unsigned vo = with_b ? (vrow + (unsigned)(ii * 16)) * 4u : G3_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, accb[SPLIT ? mh * HM + ii : 0][0]), rSB, (int)vo, 0, 0);
            }
        }
    };
    // a tile's accumulators restart from zero: plain v_mov in the LDS-read section (while the SIMD's other wave has the matrix
    // pipe), so every MFMA of the kernel accumulates in place -- an MFMA with C = 0 is a second definition of the accumulator
    // and made hipcc spill all 128 of them around the first K-tile of every tile
    auto zero_half = [&](auto mh_c) __attribute__((always_inline)) {
        constexpr int mh = decltype(mh_c)::value;
#pragma unroll
        for (int ii = 0; ii < HM; ++ii)
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                float z0, z1, z2, z3;       // (opaque to the optimiser, which would otherwise fold the zero back into the MFMA's C)
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                acc[mh * HM + ii][j] = f32x4{z0, z1, z2, z3};
            }
        if (SPLIT) {
#pragma unroll
            for (int ii = 0; ii < HM; ++ii) {
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                accb[SPLIT ? mh * HM + ii : 0] = f32x4{z0, z1, z2, z3};
            }
        }
    };

    // ---------------------------------------------------------------- prologue
    G3Cur c1, c2;             // c1: the K-tile after the one being computed; c2: the one after that
    set_tile(c1, first_id);
    // K-tile 0 of the first tile completely; A row half 0 and the B region of K-tile 1
    issueA(c1, 0, smem);
    issueA(c1, 1, smem + A_HALF);
    issueB(c1, smem + 2 * A_HALF);
    advance(c1);
    issueA(c1, 0, smem + STAGE);
    issueB(c1, smem + STAGE + 2 * A_HALF);
    c2 = c1;
    advance(c2);
    wait_vm<(VAR == 3 || VAR == 4) ? 0 : XB>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();         // the second group runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    // ---------------------------------------------------------------- one K-tile = 2 phases (row halves of the wave's tile)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    bf16x8 af[2][HM], bf[2][TNW];        // A fragments of the current row half, B fragments of the wave's column tiles (kept for phase B)
    int buf = 0, evsel = 0;
    int prow0 = 0, pcol0 = 0;            // the tile whose accumulators are complete (being stored)
    bool do_b = false, pdo_b = false;      // split mode: this wave forms the bias-gradient sums of the current / the finished item
    auto mfma_half = [&](auto mh_c) __attribute__((always_inline)) {
        constexpr int mh = decltype(mh_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < HM; ++ii)
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    f32x4& a = acc[mh * HM + ii][j];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[kk][j], af[kk][ii], a, 0, 0, 0);
                }
        if (SPLIT) {
            if (do_b) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int ii = 0; ii < HM; ++ii) {
                        f32x4& a = accb[SPLIT ? mh * HM + ii : 0];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[kk][ii], a, 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto close_read = [&]() __attribute__((always_inline)) {          // end of a read section: own LDS reads retired, then the barrier
        wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto close_mfma = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---------------------------------------------------------------- the K-tile loop, flattened over this workgroup's tiles
    // ONE loop body (hipcc's register allocation of several specialised copies of it spilled the accumulators); what differs
    // between K-tiles is decided by wave-uniform flags:
    //   first : first K-tile of an output tile -- loads the tile's epilogue vectors, zeroes each row half before its MFMAs
    //   store : first && a previous tile exists -- that tile's row half mh is stored in phase mh, before the zeroing
    //   pstore: the previous K-tile was a `store` one (its stores sit between the loads the counted waits skip)
    // (the launcher guarantees nk >= 2, so a `store` K-tile is never followed by another `first` one)
    //
    // Loads in flight.  Phase A of K-tile t issues A row half 1 of K-tile t+1 (XA loads); phase B issues A row half 0 and both B
    // halves of K-tile t+2 (XB loads) into the stage being computed: phase A was the last reader of those three regions and every
    // wave has retired its phase-A reads (lgkmcnt(0) before the barrier that closes a read section).  Every load gets one full
    // K-tile of flight: the wait of phase B retires what phase B of the previous K-tile issued (next read: phase A of t+1),
    // the wait of phase A retires what phase A of the previous K-tile issued (next read: phase B of t).  Each wave waits for its
    // own loads before the barrier that closes its read section; the reads come one read section later.
    int id = first_id, kt = 0, row0 = 0, col0 = 0;
    bool have_prev = false;
    for (;;) {
        const bool first = kt == 0;
        const bool store = first && have_prev;
        const bool pstore = kt == 1 && have_prev;
        if (first) {
            if (SPLIT) {        // store coordinates = the item's slab rows; the tile's column decides who sums A's columns
                const int tile = g.s_magic ? id - (int)__umulhi((unsigned)id, g.s_magic) * g.tiles_total : 0;
                const int tm = g.tn_magic ? (int)__umulhi((unsigned)tile, g.tn_magic) : tile;
                row0 = id * BM;
                col0 = 0;
                do_b = g.slab_b != nullptr && tile - tm * tiles_n == 0 && wc == 0;
            } else {
                const int tm = g.tn_magic ? (int)__umulhi((unsigned)id, g.tn_magic) : id;
                row0 = tm * BM;
                col0 = (id - tm * tiles_n) * BN;
            }
        }
        unsigned char* st = smem + buf * STAGE;
        unsigned char* nx = smem + (buf ^ 1) * STAGE;
        const unsigned sA0 = (unsigned)(buf * STAGE), sA1 = sA0 + A_HALF;
        const unsigned sB = sA0 + 2 * A_HALF;
        const unsigned char* evp = smem + 2 * STAGE + (evsel ^ 1) * EVB;        // vectors of the tile being stored
        // ---- phase A: row half 0 x all 4 column tiles
        static_for<2>([&](auto kk) {
            static_for<TNW>([&](auto jj) { bf[kk][jj] = fragB(sB, jj, kk); });
            static_for<HM>([&](auto ii) { af[kk][ii] = fragA(sA0, ii, kk); });
        });
        issueA(c1, 1, nx + A_HALF);
        if (EPI && first) issueEV(row0, col0, smem + 2 * STAGE + evsel * EVB);
        if (VAR != 3 && VAR != 4) {
            // outstanding and younger than phase A's loads of the previous K-tile: [stores of its phase A] phase B's XB [stores of its
            // phase B] this phase's XA [+ the epilogue vectors]
            if (first) wait_vm<cap63(XB + XA + EV1)>();
            else if (pstore) wait_vm<cap63(XB + XA + 2 * NS)>();
            else wait_vm<XB + XA>();
        }
        if (store) store_half(I0{}, prow0, pcol0, evp, pdo_b);
        if (first) zero_half(I0{});
        close_read();
        mfma_half(I0{});
        close_mfma();
        // ---- phase B: row half 1
        static_for<2>([&](auto kk) { static_for<HM>([&](auto ii) { af[kk][ii] = fragA(sA1, ii, kk); }); });
        issueA(c2, 0, st);
        issueB(c2, st + 2 * A_HALF);
        if (VAR != 3 && VAR != 4) {
            // younger than phase B's loads of the previous K-tile: [stores of the previous K-tile's phase B] phase A's XA [+ vectors]
            // [stores of phase A] this phase's XB
            if (store) wait_vm<cap63(XA + EV1 + NS + XB)>();
            else if (first) wait_vm<cap63(XA + EV1 + XB)>();
            else if (pstore) wait_vm<cap63(NS + XA + XB)>();
            else wait_vm<XA + XB>();
        }
        if (store) store_half(I1{}, prow0, pcol0, evp, pdo_b);
        if (first) zero_half(I1{});
        close_read();
        mfma_half(I1{});
        close_mfma();
        c1 = c2;
        advance(c2);
        buf ^= 1;
        kt += 1;
        if (kt == nk) {
            kt = 0;
            have_prev = true;
            prow0 = row0;
            pcol0 = col0;
            pdo_b = do_b;
            evsel ^= 1;
            id += G;
            if (id >= nt) break;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();         // pairs with the second group's last barrier
    __builtin_amdgcn_sched_barrier(0);
    {   // the last tile's accumulators
        const unsigned char* evp = smem + 2 * STAGE + (evsel ^ 1) * EVB;
        wait_vm<0>();           // (its epilogue vectors landed long ago; the dummy loads past the end still target the ring)
        store_half(I0{}, prow0, pcol0, evp, pdo_b);
        store_half(I1{}, prow0, pcol0, evp, pdo_b);
    }
}


// C[m, n] = alpha * sum_s slab[s * tiles + tile][m % BM][n % 256] (+ C[m, n]); bias_grad[m] = alpha * sum_s slab_b[s * tiles + tile(m, 0)][m % BM]
__global__ __launch_bounds__(256) void gemm3_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_b, float* __restrict__ C,
                                                           int ldc, bf16_t* __restrict__ Ch, int ldch, float* __restrict__ bias_grad, int M, int N,
                                                           int BM, int tiles_n, int S, float alpha, int accumulate) {
    const int n4 = N >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)M * n4;
    if (idx < total) {
        const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
        const int tm = m / BM, r = m - tm * BM, tn = n >> 8, c = n & 255;
        const size_t tiles = (size_t)((M + BM - 1) / BM) * tiles_n;
        const float* p = slab + ((size_t)(tm * tiles_n + tn) * BM + r) * 256 + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)s * tiles * BM * 256);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float4 o = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
        if (C) {
            float4* dst = reinterpret_cast<float4*>(C + (size_t)m * ldc + n);
            if (accumulate) {
                const float4 old = *dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *dst = o;
        }
        if (Ch) {
            bf16x4 h;
            h[0] = (bf16_t)o.x; h[1] = (bf16_t)o.y; h[2] = (bf16_t)o.z; h[3] = (bf16_t)o.w;
            *reinterpret_cast<bf16x4*>(Ch + (size_t)m * ldch + n) = h;
        }
    } else if (bias_grad && idx - total < M) {
        const int m = (int)(idx - total), tm = m / BM, r = m - tm * BM;
        const size_t tiles = (size_t)((M + BM - 1) / BM) * tiles_n;
        const float* p = slab_b + (size_t)(tm * tiles_n) * BM + r;
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += p[(size_t)s * tiles * BM];
        bias_grad[m] = alpha * acc + (accumulate ? bias_grad[m] : 0.f);
    }
}

constexpr int G3_BN = 256;                   // tile width of the split-K path and the default
int g3_bm = 0, g3_grid = 0, g3_var = 0, g3_bn = 0;      // tool overrides (immtsf_debug_gemm3_config)

template <bool TA, bool TB, int BM, int EPI, int BN>
int launch3_epi(const G3Args& g, int grid, hipStream_t stream) {
    const int out = (g.C ? 1 : 0) | (g.Ch ? 2 : 0);
    switch (out) {
        case 1: hipLaunchKernelGGL((gemm3_kernel<TA, TB, BM, 1, EPI, 0, false, BN>), dim3(grid), dim3(512), 0, stream, g); break;
        case 2:
            if constexpr (BM == 256 && EPI == 0 && !TA && !TB && BN == 256) {     // timing experiments (tools/gemm3_bench.py probe); 2-4 compute garbage
                if (g3_var == 2) { hipLaunchKernelGGL((gemm3_kernel<TA, TB, BM, 2, EPI, 2>), dim3(grid), dim3(512), 0, stream, g); break; }
                if (g3_var == 3) { hipLaunchKernelGGL((gemm3_kernel<TA, TB, BM, 2, EPI, 3>), dim3(grid), dim3(512), 0, stream, g); break; }
                if (g3_var == 4) { hipLaunchKernelGGL((gemm3_kernel<TA, TB, BM, 2, EPI, 4>), dim3(grid), dim3(512), 0, stream, g); break; }
            }
            hipLaunchKernelGGL((gemm3_kernel<TA, TB, BM, 2, EPI, 0, false, BN>), dim3(grid), dim3(512), 0, stream, g);
            break;
        case 3: hipLaunchKernelGGL((gemm3_kernel<TA, TB, BM, 3, EPI, 0, false, BN>), dim3(grid), dim3(512), 0, stream, g); break;
        default: return IMMTSF_EINVAL;
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
template <bool TA, bool TB, int BM, int BN = 256>
int launch3_out(const G3Args& g, int grid, hipStream_t stream) {
    if (g.bias || g.add_vec || g.row_flag) return launch3_epi<TA, TB, BM, 1, BN>(g, grid, stream);
    return launch3_epi<TA, TB, BM, 0, BN>(g, grid, stream);
}

inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int immtsf_debug_gemm3_config(int bm, int grid) {
    g3_bm = bm & 0xfff;          // bits 12..15: tile width override (1: 256, 2: 192); bits 16..: experimental kernel variant (tools only)
    g3_bn = (bm >> 12) & 0xf;
    g3_var = bm >> 16;
    g3_grid = grid;
    return 0;
}

// layout: GEMM_NT / GEMM_NN / GEMM_TN.  Returns IMMTSF_EUNSUPPORTED for what this kernel does not do (the caller falls back
// to gemm2): K <= 64, unaligned operands, leading dimensions that are not whole 16-byte chunks, results beyond 4 GiB.
int immtsf_launch_gemm3(int layout, const void* A, int lda, const void* B, int ldb, float* C, int ldc, void* Ch, int ldch,
                        const float* bias, const float* add_vec, const int* row_flag, int row_flag_div, int M, int N, int K,
                        float alpha, int act, const int* dyn_rows, hipStream_t stream) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 64 || act != 0) return IMMTSF_EUNSUPPORTED;
    if (!A || !B || (!C && !Ch) || !al16p(A) || !al16p(B) || (lda % 8) || (ldb % 8) || (N % 8)) return IMMTSF_EUNSUPPORTED;
    if (layout == GEMM_NT && (K % 8)) return IMMTSF_EUNSUPPORTED;
    if (layout == GEMM_NN && (K % 8)) return IMMTSF_EUNSUPPORTED;
    if (layout == GEMM_TN && (M % 8)) return IMMTSF_EUNSUPPORTED;
    if (C && (!al16p(C) || (ldc % 4) || (long)M * ldc * 4 > (long)G3_OOB)) return IMMTSF_EUNSUPPORTED;
    if (Ch && ((reinterpret_cast<uintptr_t>(Ch) & 7) || (ldch % 4) || (long)M * ldch * 2 > (long)G3_OOB)) return IMMTSF_EUNSUPPORTED;
    if (bias && !al16p(bias)) return IMMTSF_EUNSUPPORTED;
    if (add_vec && !al16p(add_vec)) return IMMTSF_EUNSUPPORTED;
    if (row_flag && row_flag_div <= 0) return IMMTSF_EINVAL;
    if (dyn_rows && layout == GEMM_TN) return IMMTSF_EUNSUPPORTED;
    // operand extents must fit a 32-bit buffer offset
    const long ea = layout == GEMM_TN ? (long)K * lda * 2 : (long)M * lda * 2;
    const long eb = layout == GEMM_NT ? (long)N * ldb * 2 : (long)K * ldb * 2;
    if (ea > (long)G3_OOB || eb > (long)G3_OOB) return IMMTSF_EUNSUPPORTED;
    G3Args g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.Ch = Ch;
    g.bias = bias; g.add_vec = add_vec; g.row_flag = row_flag; g.dyn = dyn_rows;
    g.row_flag_div = row_flag_div > 0 ? row_flag_div : 1;
    g.rf_magic = g.row_flag_div > 1 ? (unsigned)(0x100000000ull / (unsigned)g.row_flag_div) : 0u;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldch = ldch;
    g.alpha = alpha;
    // tile shape by the time model rounds(tiles / 256 workgroups) x tile area (192-wide tiles cost ~6 % more per flop: fewer
    // MFMAs per byte staged); 128-row tiles only when the 256-row rounds are poorly filled.  192-wide tiles exist for NT / NN.
    int bm = g3_bm, bn = g3_bn == 1 ? 256 : g3_bn == 2 ? 192 : 0;
    {
        double best = 0;
        int bbm = 256, bbn = 256;
        for (int cand = 0; cand < 4; ++cand) {
            const int cbm = cand & 1 ? 128 : 256, cbn = cand & 2 ? 192 : 256;
            if ((bm && cbm != bm) || (bn && cbn != bn)) continue;
            if (cbn == 192 && (layout == GEMM_TN || cbm != 256)) continue;
            const long tiles = (long)cdiv(M, cbm) * cdiv(N, cbn);
            const double cost = (double)cdiv((int)((tiles + 255) / 256 * 256), 256) * cbm * cbn * (cbm == 128 ? 1.18 : 1.0) * (cbn == 192 ? 1.06 : 1.0);
            if (best == 0 || cost < best) { best = cost; bbm = cbm; bbn = cbn; }
        }
        if (best == 0) return IMMTSF_EINVAL;
        bm = bbm; bn = bbn;
    }
    g.tiles_n = cdiv(N, bn);
    g.tn_magic = g.tiles_n <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)g.tiles_n - 1) / (unsigned)g.tiles_n);
    const long nt = (long)cdiv(M, bm) * g.tiles_n;
    if (nt * 8 >= 0x7fffffff) return IMMTSF_EUNSUPPORTED;
    int grid = g3_grid > 0 ? g3_grid : 256;
    if (!dyn_rows && nt < grid) grid = (int)((nt + 7) / 8 * 8);
    grid = (grid + 7) / 8 * 8;
    immtsf_gemm_note_grid((long)grid * 512);
#define G3_LAUNCH(TA_, TB_) (bn == 192 ? launch3_out<TA_, TB_, 256, 192>(g, grid, stream) \
                            : bm == 256 ? launch3_out<TA_, TB_, 256>(g, grid, stream) : launch3_out<TA_, TB_, 128>(g, grid, stream))
    switch (layout) {
        case GEMM_NT: return G3_LAUNCH(false, false);
        case GEMM_NN: return G3_LAUNCH(false, true);
        default: return bm == 256 ? launch3_out<true, true, 256>(g, grid, stream) : launch3_out<true, true, 128>(g, grid, stream);
    }
#undef G3_LAUNCH
}

// debug / test / tool entry (declared in include/immtsf.h)
extern "C" int immtsf_gemm3_bf16(int32_t layout, const void* A, int32_t lda, const void* B, int32_t ldb, float* C, int32_t ldc,
                                 void* Ch, int32_t ldch, const float* bias, const float* add_vec, const int32_t* row_flag,
                                 int32_t row_flag_div, int32_t M, int32_t N, int32_t K, float alpha, int32_t act,
                                 const int32_t* dyn_rows, void* stream) {
    return immtsf_launch_gemm3(layout, A, lda, B, ldb, C, ldc, Ch, ldch, bias, add_vec, row_flag, row_flag_div, M, N, K, alpha, act,
                               dyn_rows, static_cast<hipStream_t>(stream));
}

// ---- TN with a long reduction and few tiles (weight gradients at >= 256 windows per GPU): split-K over the persistent
// workgroups, fp32 partial tiles in `ws`, one reduce launch.  C = alpha * A(K,M)^T B(K,N) (+ C); bias_grad (M) = alpha * column
// sums of A.  dynk: optional device int32 overriding K (the ragged note count).
static void g3_tn_plan(int M, int N, int K, int* bm, int* splits, size_t* bytes) {
    *bm = 256;
    const long tiles = (long)cdiv(M, 256) * cdiv(N, G3_BN);
    const int nk = cdiv(K, 64);
    int S = tiles >= 256 ? 1 : (int)(256 / tiles);
    while (S > 1 && nk / S < 6) --S;         // at least ~6 K-tiles per item: the partial tile's store is worth ~1
    *splits = S;
    *bytes = (size_t)tiles * S * 256 * (256 + 1) * sizeof(float);
}
size_t immtsf_gemm3_tn_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 8192 || (M % 8) || (N % 8)) return 0;
    int bm, S;
    size_t bytes;
    g3_tn_plan(M, N, K, &bm, &S, &bytes);
    return S >= 2 ? bytes : 0;
}
int immtsf_launch_gemm3_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, void* Ch, int ldch, float* bias_grad, int M, int N,
                           int K, float alpha, int accumulate, const int* dynk, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (!A || !B || (!C && !Ch) || !ws || M <= 0 || N <= 0 || K < 8192 || (accumulate && !C)) return IMMTSF_EUNSUPPORTED;
    if (!al16p(A) || !al16p(B) || !al16p(ws) || (lda % 8) || (ldb % 8) || (M % 8) || (N % 8)) return IMMTSF_EUNSUPPORTED;
    if (C && (!al16p(C) || (ldc % 4))) return IMMTSF_EUNSUPPORTED;
    if (Ch && ((reinterpret_cast<uintptr_t>(Ch) & 7) || (ldch % 4))) return IMMTSF_EUNSUPPORTED;
    if ((long)K * lda * 2 > (long)G3_OOB || (long)K * ldb * 2 > (long)G3_OOB) return IMMTSF_EUNSUPPORTED;
    int bm, S;
    size_t bytes;
    g3_tn_plan(M, N, K, &bm, &S, &bytes);
    if (S < 2 || bytes > ws_bytes || bytes > (size_t)G3_OOB) return IMMTSF_EUNSUPPORTED;
    G3Args g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb;
    g.alpha = 1.f;
    g.row_flag_div = 1;
    g.tiles_n = cdiv(N, G3_BN);
    g.tn_magic = g.tiles_n <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)g.tiles_n - 1) / (unsigned)g.tiles_n);
    g.splits = S;
    g.tiles_total = cdiv(M, 256) * g.tiles_n;
    // (one tile: 2^32 / 1 does not fit the magic -- 0 there, the kernel takes id / 1 = id)
    g.s_magic = g.tiles_total <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)g.tiles_total - 1) / (unsigned)g.tiles_total);
    g.dynk = dynk;
    const long items = (long)cdiv(M, 256) * g.tiles_n * S;
    g.slab = static_cast<float*>(ws);
    g.slab_b = bias_grad ? g.slab + (size_t)items * 256 * 256 : nullptr;
    int grid = g3_grid > 0 ? g3_grid : 256;
    if (items < grid) grid = (int)((items + 7) / 8 * 8);
    immtsf_gemm_note_grid((long)grid * 512);
    hipLaunchKernelGGL((gemm3_kernel<true, true, 256, 1, 0, 0, true>), dim3(grid), dim3(512), 0, stream, g);
    IMMTSF_LAUNCH_CHECK();
    const long threads = (long)M * (N / 4) + (bias_grad ? M : 0);
    hipLaunchKernelGGL(gemm3_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, g.slab, g.slab_b, C, ldc,
                       reinterpret_cast<bf16_t*>(Ch), ldch, bias_grad, M, N, 256, g.tiles_n, S, alpha, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// debug / test / tool entry (declared in include/immtsf.h): TN with split-K; ws from immtsf_gemm3_tn_workspace_bytes
extern "C" size_t immtsf_gemm3_tn_workspace_bytes(int32_t M, int32_t N, int32_t K) { return immtsf_gemm3_tn_ws_bytes(M, N, K); }
extern "C" int immtsf_gemm3_tn_bf16(const void* A, int32_t lda, const void* B, int32_t ldb, float* C, int32_t ldc, float* bias_grad, int32_t M,
                                    int32_t N, int32_t K, float alpha, int32_t accumulate, const int32_t* dyn_k, void* ws, size_t ws_bytes,
                                    void* stream) {
    return immtsf_launch_gemm3_tn(A, lda, B, ldb, C, ldc, nullptr, 0, bias_grad, M, N, K, alpha, accumulate, dyn_k, ws, ws_bytes,
                                  static_cast<hipStream_t>(stream));
}
