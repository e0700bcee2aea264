// Device-side batch builder: the dataset's windows live in HBM as window-major CSR arrays (the
// "resident store"); a batch is a list of window ids.  These kernels emit exactly what the reference's collate
// functions return -- zero-padded history / prediction tensors with normalised times
// (lib/parse_datasets.py:252-295), tPatchGNN's per-(patch, variable) compacted patches (:298-366 with
// lib/utils.py:359-413) and the multimodal part, tau + zero-padded note embeddings (:764-824) -- plus the packed
// ragged note index (lengths / offsets / row map into the resident embedding matrix) that the fusion kernels use,
// so the padded embeddings and the |V|-sum mask re-derivation can be skipped.  Pure gathers: HBM-bound, bit-exact.
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

// the reference normalises with (tp - 0.0) / scale, scale = time_max + (time_max == 0) * 1e-8 (lib/utils.py:335-347)
__device__ __forceinline__ float norm_tp(float t, float scale) { return (t - 0.0f) / scale; }

// grid (B, chunks): history rows are the prefix [0, hist_len) of a window (times ascending), prediction rows the rest
__global__ __launch_bounds__(256) void collate_series_kernel(immtsf_store s, const int32_t* __restrict__ wid, int Lmax, int Lpmax,
                                                              float scale, float* __restrict__ otp, float* __restrict__ odat,
                                                              float* __restrict__ omsk, float* __restrict__ ptp,
                                                              float* __restrict__ pdat, float* __restrict__ pmsk) {
    const int b = blockIdx.x, w = wid[b], C = s.C;
    const long r0 = s.row_off[w];
    const int len = (int)(s.row_off[w + 1] - r0), hl = s.hist_len[w], pl = len - hl;
    const int stride = gridDim.y * 256, t0 = blockIdx.y * 256 + threadIdx.x;
    if (otp) {
        for (int i = t0; i < Lmax; i += stride) otp[(size_t)b * Lmax + i] = i < hl ? norm_tp(s.tt[r0 + i], scale) : 0.f;
        for (int i = t0; i < Lmax * C; i += stride) {
            const int l = i / C;
            const bool live = l < hl;
            odat[(size_t)b * Lmax * C + i] = live ? s.vals[(r0 + l) * C + (i - l * C)] : 0.f;
            omsk[(size_t)b * Lmax * C + i] = live ? s.mask[(r0 + l) * C + (i - l * C)] : 0.f;
        }
    }
    for (int i = t0; i < Lpmax; i += stride) ptp[(size_t)b * Lpmax + i] = i < pl ? norm_tp(s.tt[r0 + hl + i], scale) : 0.f;
    for (int i = t0; i < Lpmax * C; i += stride) {
        const int l = i / C;
        const bool live = l < pl;
        pdat[(size_t)b * Lpmax * C + i] = live ? s.vals[(r0 + hl + l) * C + (i - l * C)] : 0.f;
        pmsk[(size_t)b * Lpmax * C + i] = live ? s.mask[(r0 + hl + l) * C + (i - l * C)] : 0.f;
    }
}

// grid (B, npatch), one wave per variable (waves loop over C): ballot-compaction of the observed rows of variable d
// whose time lies in patch i's range, in time order, into slots 0..count-1 of out[b, i, :, d]; remaining slots zero
__global__ __launch_bounds__(256) void collate_patches_kernel(immtsf_store s, const int32_t* __restrict__ wid, int npatch,
                                                               float patch_size, float patch_stride, float history, int Lp,
                                                               float scale, float* __restrict__ otp, float* __restrict__ odat,
                                                               float* __restrict__ omsk) {
    const int b = blockIdx.x, i = blockIdx.y, w = wid[b], C = s.C;
    const long r0 = s.row_off[w];
    const int hl = s.hist_len[w];
    const float st = (float)i * patch_stride, ed = (i == npatch - 1) ? history : st + patch_size;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t base = ((size_t)b * npatch + i) * Lp * C;
    for (int d = wave; d < C; d += nw) {
        int count = 0;
        for (int j0 = 0; j0 < hl; j0 += 64) {
            const int j = j0 + lane;
            float t = 0.f, m = 0.f;
            bool hit = false;
            if (j < hl) {
                t = s.tt[r0 + j];
                m = s.mask[(r0 + j) * C + d];
                hit = (t >= st) && (t < ed) && (m != 0.f);
            }
            const unsigned long long bal = __ballot(hit);
            if (hit) {
                const int slot = count + __popcll(bal & ((1ull << lane) - 1ull));
                const size_t o = base + (size_t)slot * C + d;
                otp[o] = norm_tp(t, scale);
                odat[o] = s.vals[(r0 + j) * C + d];
                omsk[o] = m;
            }
            count += __popcll(bal);
        }
        for (int l = count + lane; l < Lp; l += 64) {
            const size_t o = base + (size_t)l * C + d;
            otp[o] = 0.f;
            odat[o] = 0.f;
            omsk[o] = 0.f;
        }
    }
}

// one block: lengths[b] = notes of window b, offsets = exclusive scan (B+1 entries)
__global__ __launch_bounds__(256) void note_index_kernel(immtsf_store s, const int32_t* __restrict__ wid, int B,
                                                          int32_t* __restrict__ lengths, int32_t* __restrict__ offsets) {
    __shared__ int part[256];
    const int tid = threadIdx.x, per = (B + 255) / 256, b0 = tid * per, b1 = min(B, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) {
        const int w = wid[b];
        const int n = (int)(s.note_off[w + 1] - s.note_off[w]);
        lengths[b] = n;
        sum += n;
    }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
        offsets[B] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int b = b0; b < b1; ++b) {
        offsets[b] = run;
        run += lengths[b];
    }
}

// grid (B, chunks): tau / padded embeddings / packed row map
__global__ __launch_bounds__(256) void collate_notes_kernel(immtsf_store s, const int32_t* __restrict__ wid, int Nmax,
                                                             const int32_t* __restrict__ offsets, float* __restrict__ tau,
                                                             float* __restrict__ notes, int32_t* __restrict__ rowmap) {
    const int b = blockIdx.x, w = wid[b], d_m = s.d_m;
    const long n0 = s.note_off[w];
    const int n = (int)(s.note_off[w + 1] - n0);
    const int stride = gridDim.y * 256, t0 = blockIdx.y * 256 + threadIdx.x;
    for (int i = t0; i < Nmax; i += stride) {
        if (tau) tau[(size_t)b * Nmax + i] = i < n ? s.note_tau[n0 + i] : 0.f;
        if (rowmap && i < n) rowmap[offsets[b] + i] = (int32_t)s.note_src[n0 + i];
    }
    if (notes) {
        const int q = d_m / 4;
        if ((d_m & 3) == 0 && ((reinterpret_cast<uintptr_t>(s.emb) | reinterpret_cast<uintptr_t>(notes)) & 15) == 0) {
            for (long i = t0; i < (long)Nmax * q; i += stride) {
                const int r = (int)(i / q), c = (int)(i - (long)r * q);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < n) v = reinterpret_cast<const float4*>(s.emb + (size_t)s.note_src[n0 + r] * d_m)[c];
                reinterpret_cast<float4*>(notes + ((size_t)b * Nmax + r) * d_m)[c] = v;
            }
        } else {
            for (long i = t0; i < (long)Nmax * d_m; i += stride) {
                const int r = (int)(i / d_m), c = (int)(i - (long)r * d_m);
                notes[((size_t)b * Nmax + r) * d_m + c] = r < n ? s.emb[(size_t)s.note_src[n0 + r] * d_m + c] : 0.f;
            }
        }
    }
}

bool store_ok(const immtsf_store* s) {
    return s && s->tt && s->vals && s->mask && s->row_off && s->hist_len && s->C > 0;
}

}  // namespace

extern "C" {

int immtsf_collate_series(const immtsf_store* s, const int32_t* window_ids, int32_t B, int32_t Lmax, int32_t Lpmax,
                          float time_max, float* obs_tp, float* obs_data, float* obs_mask, float* pred_tp,
                          float* pred_data, float* pred_mask, immtsf_stream_t stream) {
    if (!store_ok(s) || !window_ids || B < 0 || Lmax < 0 || Lpmax < 0 || !pred_tp || !pred_data || !pred_mask) return IMMTSF_EINVAL;
    if ((obs_tp != nullptr) != (obs_data != nullptr) || (obs_tp != nullptr) != (obs_mask != nullptr)) return IMMTSF_EINVAL;
    if (B == 0) return IMMTSF_OK;
    float scale = time_max - 0.0f;
    scale = scale + (scale == 0.f ? 1.f : 0.f) * 1e-8f;
    const int work = (Lmax > Lpmax ? Lmax : Lpmax) * s->C;
    hipLaunchKernelGGL(collate_series_kernel, dim3(B, work > 0 ? cdiv(work, 256) : 1), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *s, window_ids, Lmax, Lpmax, scale, obs_tp, obs_data, obs_mask, pred_tp,
                       pred_data, pred_mask);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_collate_patches(const immtsf_store* s, const int32_t* window_ids, int32_t B, int32_t npatch, float patch_size,
                           float patch_stride, float history, int32_t Lp, float time_max, float* obs_tp, float* obs_data,
                           float* obs_mask, immtsf_stream_t stream) {
    if (!store_ok(s) || !window_ids || B < 0 || npatch <= 0 || Lp < 0 || !obs_tp || !obs_data || !obs_mask) return IMMTSF_EINVAL;
    if (B == 0 || Lp == 0) return IMMTSF_OK;
    float scale = time_max - 0.0f;
    scale = scale + (scale == 0.f ? 1.f : 0.f) * 1e-8f;
    hipLaunchKernelGGL(collate_patches_kernel, dim3(B, npatch), dim3(256), 0, static_cast<hipStream_t>(stream), *s, window_ids,
                       npatch, patch_size, patch_stride, history, Lp, scale, obs_tp, obs_data, obs_mask);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_collate_notes(const immtsf_store* s, const int32_t* window_ids, int32_t B, int32_t Nmax, float* tau, float* notes,
                         int32_t* lengths, int32_t* offsets, int32_t* rowmap, immtsf_stream_t stream) {
    if (!s || !s->note_off || !s->note_tau || !s->note_src || !window_ids || B < 0 || Nmax < 0 || !lengths || !offsets)
        return IMMTSF_EINVAL;
    if (notes && (!s->emb || s->d_m <= 0)) return IMMTSF_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(note_index_kernel, dim3(1), dim3(256), 0, st, *s, window_ids, B, lengths, offsets);
    IMMTSF_LAUNCH_CHECK();
    if (B == 0 || Nmax == 0) return IMMTSF_OK;
    const long work = notes ? (long)Nmax * (s->d_m / 4 > 0 ? s->d_m / 4 : s->d_m) : Nmax;
    int gy = (int)((work + 255) / 256);
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(collate_notes_kernel, dim3(B, gy < 1 ? 1 : gy), dim3(256), 0, st, *s, window_ids, Nmax, offsets, tau, notes,
                       rowmap);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // extern "C"
