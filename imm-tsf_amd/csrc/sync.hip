// Device-side flags between two streams of one captured step (immtsf.train.FlagStep).  A hipGraph branch that has to WAIT for
// another branch -- a barrier packet on a not-yet-signalled event -- resumes 110 - 175 us late on this runtime (ROCm 7.2, measured in
// the cfg2 step: profiles/r03_step_kernel_sequence.txt), while a dependency that is already satisfied when the waiter gets there
// costs nothing.  So inside the step the two branches carry NO graph edges between the fork at its start and the join at its end;
// where one needs the other's result, the producer's stream runs flag_set behind its last kernel (kernel-end release makes
// the results visible device-wide) and the consumer's stream runs flag_wait, a one-lane spin on the flag, in front of its first.
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

// optional trace of the flag kernels (immtsf_flag_trace / immtsf_flag_trace_read; tools/flag_timeline.py): who waited for whom, and
// how long, inside a replayed step -- on the 100 MHz wall clock, without a profiler serialising the two branches
constexpr int FT_N = 1024;
__device__ int ft_on;
__device__ unsigned int ft_count;
__device__ long long ft_ring[FT_N][3];       // flag address | kind (0 set, 1 wait entered, 2 wait left, 3 clear) | wall clock
__device__ __forceinline__ void ft_note(const void* flag, int kind) {
    if (!ft_on) return;
    const unsigned int i = atomicAdd(&ft_count, 1u);
    if (i < FT_N) { ft_ring[i][0] = (long long)(uintptr_t)flag; ft_ring[i][1] = kind; ft_ring[i][2] = wall_clock64(); }
}

__global__ void flag_set_kernel(int* flag) {
    ft_note(flag, 0);
    __threadfence_system();
    __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// spins until *flag != 0; gives up after `ticks` of the 100 MHz wall clock (a scheduling accident must not hang the GPU) and
// reports it in *err
__global__ void flag_wait_kernel(int* flag, int* err, long long ticks) {
    const long long t0 = wall_clock64();
    ft_note(flag, 1);
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > ticks) {
            atomicExch(err, 1);
            break;
        }
    }
    ft_note(flag, 2);
    __threadfence_system();
}
// counting form, for a consumer OUTSIDE the captured step (the communication stream of the data-parallel step, which is enqueued
// eagerly beside the replaying graph): the producer inside the graph adds 1 per replay, the consumer of replay k waits for >= k.
// Nothing is ever cleared, so a consumer that runs late can never miss (or double-count) a hand-over.
__global__ void flag_bump_kernel(int* flag) {
    ft_note(flag, 0);
    __threadfence_system();
    __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_ge_kernel(int* flag, int target, int* err, long long ticks) {
    const long long t0 = wall_clock64();
    ft_note(flag, 1);
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > ticks) {
            atomicExch(err, 1);
            break;
        }
    }
    ft_note(flag, 2);
    __threadfence_system();
}
// flag_wait_ge on up to four flags at once (a collective over several buckets that complete together), then -- slot != null -- this rank's
// guard word into `slot` (the step's last collective sums it over the ranks): ONE launch on the communication stream's exposed tail
struct FlagSet { int* f[4]; int n; };
__global__ void flag_wait_ge_multi_kernel(FlagSet fs, int target, int* err, long long ticks, void* slot, int is_bf16) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < fs.n; ++i) {
        ft_note(fs.f[i], 1);
        while (__hip_atomic_load(fs.f[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > ticks) {
                atomicExch(err, 1);
                break;
            }
        }
        ft_note(fs.f[i], 2);
    }
    if (slot) {
        const float v = (__hip_atomic_load(err, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1.f : 0.f;
        if (is_bf16) *static_cast<__bf16*>(slot) = (__bf16)v;
        else *static_cast<float*>(slot) = v;
    }
    __threadfence_system();
}
__global__ void flags_clear_kernel(int* flags, int n, int* set_flag = nullptr) {
    if (threadIdx.x == 0) ft_note(flags, 3);
    if ((int)threadIdx.x < n) flags[threadIdx.x] = 0;
    if (set_flag && threadIdx.x == 0) *set_flag = 1;
}

}  // namespace

extern "C" {

int immtsf_flag_set(int32_t* flag, immtsf_stream_t stream) {
    if (!flag) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), flag);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_flag_wait(int32_t* flag, int32_t* err, int32_t timeout_ms, immtsf_stream_t stream) {
    if (!flag || !err || timeout_ms <= 0) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), flag, err, (long long)timeout_ms * 100000ll);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_flag_bump(int32_t* flag, immtsf_stream_t stream) {
    if (!flag) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flag_bump_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), flag);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_flag_wait_ge(int32_t* flag, int32_t target, int32_t* err, int32_t timeout_ms, immtsf_stream_t stream) {
    if (!flag || !err || timeout_ms <= 0) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flag_wait_ge_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), flag, target, err,
                       (long long)timeout_ms * 100000ll);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_flag_wait_ge_multi(int32_t n, int32_t* const* flags, int32_t target, int32_t* err, int32_t timeout_ms, void* guard_slot,
                              int32_t is_bf16, immtsf_stream_t stream) {
    if (n < 1 || n > 4 || !flags || !err || timeout_ms <= 0) return IMMTSF_EINVAL;
    FlagSet fs;
    fs.n = n;
    for (int i = 0; i < 4; ++i) {
        fs.f[i] = i < n ? flags[i] : nullptr;
        if (i < n && !flags[i]) return IMMTSF_EINVAL;
    }
    hipLaunchKernelGGL(flag_wait_ge_multi_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), fs, target, err,
                       (long long)timeout_ms * 100000ll, guard_slot, is_bf16 ? 1 : 0);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_flags_clear(int32_t* flags, int32_t n, immtsf_stream_t stream) {
    if (!flags || n <= 0 || n > 64) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flags_clear_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), flags, n, (int*)nullptr);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_flags_clear_set(int32_t* flags, int32_t n, int32_t* set_flag, immtsf_stream_t stream) {
    if (!flags || n <= 0 || n > 64 || !set_flag) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flags_clear_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), flags, n, set_flag);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

/* trace of the flag kernels: enable != 0 empties the ring and switches recording on, 0 switches it off (synchronises the device) */
int immtsf_flag_trace(int32_t enable) {
    const int on = enable ? 1 : 0;
    const unsigned int zero = 0u;
    if (hipDeviceSynchronize() != hipSuccess) return IMMTSF_EINVAL;
    if (hipMemcpyToSymbol(HIP_SYMBOL(ft_count), &zero, sizeof(zero)) != hipSuccess) return IMMTSF_EINVAL;
    if (hipMemcpyToSymbol(HIP_SYMBOL(ft_on), &on, sizeof(on)) != hipSuccess) return IMMTSF_EINVAL;
    return IMMTSF_OK;
}
/* the recorded entries, three int64 each (flag address, kind: 0 set / 1 wait entered / 2 wait left / 3 clear, 100 MHz wall clock), at
 * most max_entries (<= 256) of them; returns the count, < 0 on error (synchronises the device) */
int immtsf_flag_trace_read(int64_t* out, int32_t max_entries) {
    if (!out || max_entries <= 0) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(ft_count), sizeof(n)) != hipSuccess) return -1;
    if (n > (unsigned)FT_N) n = FT_N;
    if (n > (unsigned)max_entries) n = (unsigned)max_entries;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(ft_ring), sizeof(long long) * 3 * n) != hipSuccess) return -1;
    return (int)n;
}

}  // extern "C"
