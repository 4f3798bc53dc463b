// copy_rates.hip — what HBM rate can a read-N / write-N kernel reach on this GPU?  (calibration only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PER>
__global__ void __launch_bounds__(256) k_copy(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n) {
    // each thread moves PER 16-byte chunks, block-contiguous
    size_t base = ((size_t)blockIdx.x * PER) * 256 + threadIdx.x;
    u32x4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        size_t idx = base + (size_t)i * 256;
        if (idx < n) v[i] = (MODE & 1) ? __builtin_nontemporal_load(&in[idx]) : in[idx];
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        size_t idx = base + (size_t)i * 256;
        if (idx < n) { if (MODE & 2) __builtin_nontemporal_store(v[i], &out[idx]); else out[idx] = v[i]; }
    }
}
template <int PER>
__global__ void __launch_bounds__(256) k_read(const u32x4 *__restrict__ in, uint32_t *sink, size_t n) {
    size_t base = ((size_t)blockIdx.x * PER) * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PER; i++) { size_t idx = base + (size_t)i * 256; if (idx < n) acc ^= in[idx]; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int PER>
__global__ void __launch_bounds__(256) k_write(u32x4 *__restrict__ out, size_t n) {
    size_t base = ((size_t)blockIdx.x * PER) * 256 + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
#pragma unroll
    for (int i = 0; i < PER; i++) { size_t idx = base + (size_t)i * 256; if (idx < n) __builtin_nontemporal_store(v, &out[idx]); }
}
template <class F>
static float best_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 8; r++) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t bytes = (size_t)3 << 29;   // 1.5 GiB each way
    const size_t n = bytes / 16;
    u32x4 *src, *dst; uint32_t *sink;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 2, bytes));
#define RUN(NAME, PER, BYTES_MOVED, ...) { unsigned blocks = (unsigned)((n + 256ull * PER - 1) / (256ull * PER)); \
        float ms = best_ms([&] { __VA_ARGS__; }); printf("%-34s %8.3f ms  %.2f TB/s\n", NAME, ms, (BYTES_MOVED) / (ms * 1e-3) / 1e12); }
    RUN("copy plain x1", 1, 2.0 * bytes, k_copy<0, 1><<<blocks, 256>>>(src, dst, n))
    RUN("copy plain x4", 4, 2.0 * bytes, k_copy<0, 4><<<blocks, 256>>>(src, dst, n))
    RUN("copy plain x8", 8, 2.0 * bytes, k_copy<0, 8><<<blocks, 256>>>(src, dst, n))
    RUN("copy nt-store x4", 4, 2.0 * bytes, k_copy<2, 4><<<blocks, 256>>>(src, dst, n))
    RUN("copy nt-load+store x4", 4, 2.0 * bytes, k_copy<3, 4><<<blocks, 256>>>(src, dst, n))
    RUN("copy nt-load+store x8", 8, 2.0 * bytes, k_copy<3, 8><<<blocks, 256>>>(src, dst, n))
    RUN("read only x4", 4, 1.0 * bytes, k_read<4><<<blocks, 256>>>(src, sink, n))
    RUN("read only x8", 8, 1.0 * bytes, k_read<8><<<blocks, 256>>>(src, sink, n))
    RUN("write only nt x4", 4, 1.0 * bytes, k_write<4><<<blocks, 256>>>(dst, n))
    RUN("hipMemcpyDtoD", 1, 2.0 * bytes, (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0))
    return 0;
}
