// Device-side flags between two streams of one captured step (immtsf.train.FlagStep).  A hipGraph branch that has to WAIT for
// another branch -- a barrier packet on a not-yet-signalled event -- resumes 110 - 175 us late on this runtime (ROCm 7.2, measured in
// the cfg2 step: profiles/r03_step_kernel_sequence.txt), while a dependency that is already satisfied when the waiter gets there
// costs nothing.  So inside the step the two branches carry NO graph edges between the fork at its start and the join at its end;
// where one needs the other's result, the producer's stream runs flag_set behind its last kernel (kernel-end release makes
// the results visible device-wide) and the consumer's stream runs flag_wait, a one-lane spin on the flag, in front of its first.
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

__global__ void flag_set_kernel(int* flag) {
    __threadfence_system();
    __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// spins until *flag != 0; gives up after `ticks` of the 100 MHz wall clock (a scheduling accident must not hang the GPU) and
// reports it in *err
__global__ void flag_wait_kernel(int* flag, int* err, long long ticks) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > ticks) {
            atomicExch(err, 1);
            break;
        }
    }
    __threadfence_system();
}
__global__ void flags_clear_kernel(int* flags, int n) {
    if ((int)threadIdx.x < n) flags[threadIdx.x] = 0;
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
int immtsf_flags_clear(int32_t* flags, int32_t n, immtsf_stream_t stream) {
    if (!flags || n <= 0 || n > 64) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(flags_clear_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), flags, n);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // extern "C"
