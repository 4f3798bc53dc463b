// pull_probe.hip — calibration tool (not part of the library): a pageable frame -> device memory through a page-locked staging buffer,
// (a) as the library does it in round 6: copier threads fill 512 KB chunks, DMA commands over runs of chunks that double, or
// (b) with ONE kernel that PULLS the staged bytes over the link: launched before the first byte is copied, its workgroups follow a
//     "chunks ready so far" word in page-locked memory that the copier threads advance, and copy each chunk's slice as soon as it is there -
//     no DMA command, nothing to enqueue per chunk, and what is left after the last chunk lands is ONE chunk's transfer.
// Timed: from the first copied byte until a consumer kernel behind the upload (a checksum of the device copy) has finished; the checksum
// is compared with the host's.  Sources: the same buffer every call (warm in the last-level cache) and a ring of buffers larger than it (cold).
//   hipcc -O3 --offload-arch=gfx950 pull_probe.hip -o pull_probe -lpthread
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kPullGroups = 32, kPullThreads = 256;

// word: epoch << 32 | chunks ready (contiguous from 0).  Every workgroup copies slice g of every chunk.
__global__ void __launch_bounds__(kPullThreads) k_pull(const uint8_t *h, uint8_t *d, size_t bytes, uint32_t chunk, uint32_t nchunks,
                                                       const uint64_t *word, uint32_t epoch, uint32_t *err, uint64_t timeout_ticks) {
    __shared__ uint32_t s_have;
    const uint32_t slice = chunk / kPullGroups;                                    // (chunk is a multiple of kPullGroups * 16)
    uint32_t have = 0;
    for (uint32_t k = 0; k < nchunks; k++) {
        if (k >= have) {
            if (threadIdx.x == 0) {
                const uint64_t t0 = wall_clock64();
                uint32_t r = 0;
                for (;;) {
                    const uint64_t w = __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    r = (uint32_t)(w >> 32) == epoch ? (uint32_t)w : 0u;
                    if (r > k) break;
                    if (wall_clock64() - t0 > timeout_ticks) { atomicOr(err, 1u); r = 0xFFFFFFFFu; break; }
                    __builtin_amdgcn_s_sleep(64);
                }
                s_have = r;
            }
            __syncthreads();
            have = s_have;
            __syncthreads();
            if (have == 0xFFFFFFFFu) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        const size_t at = (size_t)k * chunk + (size_t)blockIdx.x * slice;
        const size_t end = std::min(bytes, at + slice);
        for (size_t i = at + (size_t)threadIdx.x * 16u; i < end; i += (size_t)kPullThreads * 16u * 4u) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const size_t a = i + (size_t)j * kPullThreads * 16u; if (a + 16u <= end) v[j] = __builtin_nontemporal_load((const u32x4 *)(h + a)); }
#pragma unroll
            for (int j = 0; j < 4; j++) { const size_t a = i + (size_t)j * kPullThreads * 16u; if (a + 16u <= end) *(u32x4 *)(d + a) = v[j]; }
        }
    }
}

__global__ void __launch_bounds__(256) k_sum(const uint32_t *d, size_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += d[i];
    for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

static void stream_copy(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), e = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c); _mm256_stream_si256((__m256i *)(dst + i + 96), e);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}

struct Pool {                                   // persistent helpers woken through a condition variable, like the library's
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv, cv_done;
    std::function<void()> task; uint64_t gen = 0; int pending = 0; bool stop = false;
    explicit Pool(int n) {
        for (int i = 0; i < n; i++) th.emplace_back([this] {
            uint64_t seen = 0;
            for (;;) {
                std::function<void()> t;
                { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; t = task; }
                t();
                { std::lock_guard<std::mutex> l(mu); if (--pending == 0) cv_done.notify_all(); }
            }
        });
    }
    void run(std::function<void()> t) { { std::lock_guard<std::mutex> l(mu); task = std::move(t); pending = (int)th.size(); gen++; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> l(mu); cv_done.wait(l, [&] { return pending == 0; }); }
    ~Pool() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
};

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 4;
    const size_t chunk = (size_t)512 << 10;
    const size_t sizes[] = {2764800, 6220800, 24883200, 132710400};
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint8_t *hst, *d; uint64_t *word; uint32_t *err; unsigned long long *dsum, *hsum;
    CHECK(hipHostMalloc((void **)&hst, sizes[3], hipHostMallocDefault)); CHECK(hipMalloc((void **)&d, sizes[3]));
    CHECK(hipHostMalloc((void **)&word, 64, hipHostMallocDefault)); CHECK(hipHostMalloc((void **)&hsum, 64, hipHostMallocDefault));
    CHECK(hipMalloc((void **)&err, 4)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMalloc((void **)&dsum, 8));
    *word = 0;
    Pool pool(threads - 1);
    const size_t ring_bytes = (size_t)1200 << 20;                                  // the cold sources: a ring larger than any last-level cache
    uint8_t *ring = (uint8_t *)malloc(ring_bytes);
    for (size_t i = 0; i < ring_bytes; i += 4) *(uint32_t *)(ring + i) = (uint32_t)(i * 2654435761u >> 7);
    uint32_t epoch = 0;
    printf("staged upload of one pageable frame, %d copier threads (this one included), 512 KB chunks; microseconds until a kernel behind the upload is done (median of 15)\n", threads);
    for (size_t bytes : sizes) {
        const uint32_t nchunks = (uint32_t)((bytes + chunk - 1) / chunk);
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[nchunks]);
        for (int cold = 0; cold < 2; cold++) {
            for (int mode = 0; mode < 2; mode++) {
                std::vector<double> us; bool ok = true; size_t ring_at = 0;
                for (int it = 0; it < 19; it++) {
                    const uint8_t *src = ring;
                    if (cold) { if (ring_at + bytes > ring_bytes) ring_at = 0; src = ring + ring_at; ring_at += (bytes + 4095) & ~(size_t)4095; }
                    unsigned long long want = 0;
                    if (it == 18) for (size_t i = 0; i + 4 <= bytes; i += 4) want += *(const uint32_t *)(src + i);      // (checked on the last, untimed, pass)
                    for (uint32_t k = 0; k < nchunks; k++) done[k].store(0);
                    std::atomic<uint32_t> next(0);
                    CHECK(hipMemsetAsync(dsum, 0, 8, st)); CHECK(hipStreamSynchronize(st));
                    epoch++;
                    const auto t0 = std::chrono::steady_clock::now();
                    auto publish = [&] {                                               // advance the "ready so far" word past every finished chunk
                        std::atomic<uint64_t> *w = reinterpret_cast<std::atomic<uint64_t> *>(word);
                        for (;;) {
                            uint64_t cur = w->load();
                            const uint32_t r = (uint32_t)(cur >> 32) == epoch ? (uint32_t)cur : 0u;
                            uint32_t n = r;
                            while (n < nchunks && done[n].load()) n++;
                            if (n == r) return;
                            if (w->compare_exchange_strong(cur, ((uint64_t)epoch << 32) | n)) continue;
                        }
                    };
                    auto copy_one = [&](uint32_t k) {
                        const size_t at = (size_t)k * chunk, n = std::min(chunk, bytes - at);
                        stream_copy(hst + at, src + at, n);
                        done[k].store(1);                                               // (sequentially consistent: the scan below must not pass it)
                        if (mode == 1) publish();
                    };
                    auto copier = [&] { for (;;) { const uint32_t k = next.fetch_add(1); if (k >= nchunks) break; copy_one(k); } };
                    if (threads > 1) pool.run(copier);                                  // (the helpers take ~20 us to wake: first)
                    if (mode == 1) hipLaunchKernelGGL(k_pull, dim3(kPullGroups), dim3(kPullThreads), 0, st, hst, d, bytes, (uint32_t)chunk, nchunks, word, epoch, err, (uint64_t)200000000);
                    if (mode == 0) {
                        uint32_t unit_begin = 0, unit_len = 1, ready = 0;
                        while (unit_begin < nchunks) {
                            const uint32_t unit_end = std::min(nchunks, unit_begin + unit_len);
                            while (ready < unit_end && done[ready].load(std::memory_order_acquire)) ready++;
                            if (ready >= unit_end) {
                                const size_t at = (size_t)unit_begin * chunk, end = std::min(bytes, (size_t)unit_end * chunk);
                                CHECK(hipMemcpyAsync(d + at, hst + at, end - at, hipMemcpyHostToDevice, st));
                                unit_begin = unit_end; unit_len = std::min(64u, unit_len * 2u);
                                continue;
                            }
                            const uint32_t k = next.fetch_add(1);
                            if (k < nchunks) copy_one(k); else _mm_pause();
                        }
                    } else {
                        copier();
                    }
                    if (threads > 1) pool.wait();
                    if (mode == 1) publish();
                    hipLaunchKernelGGL(k_sum, dim3(512), dim3(256), 0, st, (const uint32_t *)d, bytes / 4, dsum);
                    CHECK(hipMemcpyAsync(hsum, dsum, 8, hipMemcpyDeviceToHost, st));
                    CHECK(hipStreamSynchronize(st));
                    const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (it >= 3 && it < 18) us.push_back(t);
                    if (it == 18 && *hsum != want) ok = false;
                }
                uint32_t e = 0; CHECK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
                std::sort(us.begin(), us.end());
                printf("  %10zu bytes  %-4s source  %-34s  %7.0f us  (min %5.0f; the link alone at 57 GB/s: %5.0f)  %s%s\n", bytes, cold ? "cold" : "warm",
                       mode ? "pull kernel behind a ready counter" : "DMA commands over doubling runs", us[us.size() / 2], us[0], bytes / 57e3, ok ? "checksum ok" : "CHECKSUM DIFFERS", e ? "  TIMED OUT" : "");
            }
        }
    }
    return 0;
}
