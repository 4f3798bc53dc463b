// host_read_rates.hip — what a KERNEL gets when it reads pinned host memory itself (calibration only): how long W workgroups
// take to pull N bytes across the host link with coalesced 16-byte loads, each byte once, against the same from device memory
// and against an empty launch.  The question behind it: a small frame's pixels are read from the handle's pinned buffer by the
// pixels -> bits kernel (no DMA node in the launch sequence) - is a frame better read once, up front, in wide requests?
//   hipcc -O2 --offload-arch=gfx950 -o host_read_rates host_read_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__global__ void __launch_bounds__(256) k_read(const uint4 *src, size_t chunks, uint32_t *sink, int per_thread) {
    // every workgroup takes a contiguous share; a thread has per_thread loads in flight before it looks at any
    const size_t share = (chunks + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * share, hi = min(chunks, lo + share);
    uint32_t acc = 0;
    for (size_t base = lo + threadIdx.x; base < hi; base += (size_t)256 * per_thread) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t i = base + (size_t)k * 256;
            v[k] = k < per_thread && i < hi ? src[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// the frame's copy to device memory by the kernel itself (what a staging prologue would do)
__global__ void __launch_bounds__(256) k_copy(const uint4 *src, uint4 *dst, size_t chunks, int per_thread) {
    const size_t share = (chunks + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * share, hi = min(chunks, lo + share);
    for (size_t base = lo + threadIdx.x; base < hi; base += (size_t)256 * per_thread) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t i = base + (size_t)k * 256;
            v[k] = k < per_thread && i < hi ? src[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t i = base + (size_t)k * 256;
            if (k < per_thread && i < hi) dst[i] = v[k];
        }
    }
}
__global__ void k_empty(uint32_t *sink) { if (threadIdx.x == 9999) sink[0] = 1; }

template <class F>
static double median_us(F launch, hipStream_t st, int reps = 41) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int r = 0; r < reps + 5; r++) {
        hipEventRecord(a, st);
        launch();
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (r >= 5) t.push_back(ms * 1000.f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const size_t cap = (size_t)8 << 20;
    uint8_t *host = nullptr, *dev = nullptr, *dst = nullptr;
    uint32_t *sink = nullptr;
    hipHostMalloc((void **)&host, cap, hipHostMallocDefault);
    hipMalloc((void **)&dev, cap); hipMalloc((void **)&dst, cap); hipMalloc((void **)&sink, 64);
    for (size_t i = 0; i < cap; i++) host[i] = (uint8_t)(i * 7);
    hipMemcpy(dev, host, cap, hipMemcpyHostToDevice);
    const double empty = median_us([&] { hipLaunchKernelGGL(k_empty, dim3(4), dim3(256), 0, st, sink); }, st);
    printf("empty launch between two events: %.1f us\n", empty);
    const size_t sizes[] = {(size_t)196608, (size_t)921600, (size_t)2764800, (size_t)6220800};
    for (size_t n : sizes) {
        const size_t chunks = n / 16;
        const double dma = median_us([&] { hipMemcpyAsync(dst, host, n, hipMemcpyHostToDevice, st); }, st);
        printf("%8zu bytes: hipMemcpyAsync H2D %.1f us (%.1f GB/s)\n", n, dma, n / dma / 1e3);
        for (int wgs : {4, 16, 64, 256})
            for (int per : {2, 8}) {
                const double h = median_us([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, st, (const uint4 *)host, chunks, sink, per); }, st);
                const double d = median_us([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, st, (const uint4 *)dev, chunks, sink, per); }, st);
                const double c = median_us([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, st, (const uint4 *)host, (uint4 *)dst, chunks, per); }, st);
                printf("   %3d workgroups, %d loads in flight per thread: host read %.1f us (%.1f GB/s beyond the empty launch), device read %.1f us, host -> device copy %.1f us\n",
                       wgs, per, h, n / std::max(h - empty, 0.1) / 1e3, d, c);
            }
    }
    return 0;
}
