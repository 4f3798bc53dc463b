// staged_pull.hip — the upload of ONE pageable image as a kernel that pulls the staged bytes over the link.
//
// jpegenc_encoder_encode reads the caller's slice where it lies (encoder.rs:440-454); across a PCIe link the pixels go through the
// handle's page-locked staging buffer first (host_frame.cpp, upload_in_stripes: the caller's pageable memory never reaches the runtime).
// With DMA commands that costs one command per run of staged chunks - ~10 us of engine latency each, and whatever the LAST command
// covers still has to cross the link after the last byte was copied (half the frame when the copy is the slower side).  Here the
// transfer is one kernel, launched before the first byte is copied: its workgroups follow the "chunks ready so far" word the copier
// threads advance in page-locked memory and copy each chunk - slice g of it per workgroup - as soon as it is there.  A kernel reads
// page-locked host memory at the link's rate from 16 workgroups on (csrc/tools/host_read_rates.hip: 57-61 GB/s); nothing is enqueued per
// chunk, and what remains after the last chunk lands is one chunk's transfer (csrc/tools/pull_probe.hip: a cold 4K frame through two
// copier threads 719 -> 576 us, 1080p 231 -> 205).
//
// The kernel waits for the HOST, so it carries a deadline: a workgroup that sees no progress for kStagedPullTimeoutTicks of the 100 MHz
// wall clock sets *timed_out (page-locked memory) and leaves; the host looks at the word once the frame's stream has been waited for and
// fails the call.  The host publishes every chunk before upload_in_stripes returns, whatever happens in between.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "host_common.h"

namespace jpegenc {

typedef uint32_t pull_u32x4 __attribute__((ext_vector_type(4)));

// ready: epoch << 32 | chunks staged so far, contiguous from chunk 0 (a word of another epoch counts as 0).
// One launch moves the bytes [from, to) of the image (the whole image, or one stripe of a frame that is coded stripe by stripe).
__global__ void __launch_bounds__(kStagedPullThreads) k_pull_staged(const uint8_t *h, uint8_t *d, uint32_t chunk, const uint64_t *ready, uint32_t epoch,
                                                                    uint32_t *timed_out, size_t from, size_t to) {
    __shared__ uint32_t s_have;
    const uint32_t slice = chunk / kStagedPullGroups;                              // (chunk: a multiple of kStagedPullGroups * 64)
    const uint32_t first = (uint32_t)(from / chunk), last = (uint32_t)((to - 1u) / chunk);
    uint32_t have = 0;
    for (uint32_t k = first; k <= last; k++) {
        if (k >= have) {                                                           // (uniform over the workgroup)
            if (threadIdx.x == 0) {
                const uint64_t t0 = wall_clock64();
                uint32_t r;
                for (;;) {
                    const uint64_t w = __hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    r = (uint32_t)(w >> 32) == epoch ? (uint32_t)w : 0u;
                    if (r > k) break;
                    if (wall_clock64() - t0 > kStagedPullTimeoutTicks) {
                        __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        r = 0xFFFFFFFFu;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(64);
                }
                s_have = r;
            }
            __syncthreads();
            have = s_have;
            __syncthreads();
            if (have == 0xFFFFFFFFu) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                         // the chunk's bytes are read after the word that announces them
        }
        const size_t at = (size_t)k * chunk + (size_t)blockIdx.x * slice;
        const size_t lo = at > from ? at : from, hi = at + slice < to ? at + slice : to;
        if (lo >= hi) continue;
        // 16-byte units on the image's own grid (both buffers are aligned to it); the units a range begins or ends inside go byte by byte
        for (size_t i = (lo & ~(size_t)15u) + (size_t)threadIdx.x * 16u; i < hi; i += (size_t)kStagedPullThreads * 16u * 4u) {
            pull_u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const size_t a = i + (size_t)j * kStagedPullThreads * 16u;
                if (a >= lo && a + 16u <= hi) v[j] = __builtin_nontemporal_load((const pull_u32x4 *)(h + a));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const size_t a = i + (size_t)j * kStagedPullThreads * 16u;
                if (a >= lo && a + 16u <= hi) *(pull_u32x4 *)(d + a) = v[j];
                else if (a < hi && a + 16u > lo) for (size_t b = a > lo ? a : lo; b < a + 16u && b < hi; b++) d[b] = h[b];
            }
        }
    }
}

hipError_t launch_staged_pull(const uint8_t *h_staged, uint8_t *d_pixels, size_t from, size_t to, uint32_t chunk, const uint64_t *h_ready,
                              uint32_t epoch, uint32_t *h_timed_out, hipStream_t stream) {
    if (to <= from) return hipSuccess;
    if (!chunk || chunk % (kStagedPullGroups * 64u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pull_staged, dim3(kStagedPullGroups), dim3(kStagedPullThreads), 0, stream, h_staged, d_pixels, chunk, h_ready, epoch, h_timed_out, from, to);
    return hipGetLastError();
}

}  // namespace jpegenc
