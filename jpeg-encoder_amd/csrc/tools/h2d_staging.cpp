// h2d_staging.cpp — what the batch workers' upload pattern delivers over the host link (calibration only): T threads, each
// with its own stream and pinned buffer, loop { memcpy(pageable frame -> pinned) in C chunks, hipMemcpyAsync of each chunk,
// stream sync }.  Prints aggregate GB/s for frame sizes 6.2 MB (1080p) and 24.9 MB (4K).
//   hipcc -O2 -o h2d_staging h2d_staging.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static double run(size_t frame, int threads, int chunks, bool stage, double seconds) {
    std::atomic<long> frames(0);
    std::atomic<bool> stop(false);
    std::vector<std::thread> pool;
    std::vector<char *> src((size_t)threads);
    for (int t = 0; t < threads; t++) { src[t] = (char *)malloc(frame); memset(src[t], t + 1, frame); }
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            char *pinned, *dev;
            hipHostMalloc((void **)&pinned, frame, hipHostMallocDefault);
            hipMalloc((void **)&dev, frame);
            if (!stage) memcpy(pinned, src[t], frame);
            while (!stop.load()) {
                for (int c = 0; c < chunks; c++) {
                    const size_t lo = frame * c / chunks, hi = frame * (c + 1) / chunks;
                    if (stage) memcpy(pinned + lo, src[t] + lo, hi - lo);
                    hipMemcpyAsync(dev + lo, pinned + lo, hi - lo, hipMemcpyHostToDevice, st);
                }
                hipStreamSynchronize(st);
                frames.fetch_add(1);
            }
            hipFree(dev); hipHostFree(pinned); hipStreamDestroy(st);
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    const long f0 = frames.load();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const long f1 = frames.load();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    for (auto &th : pool) th.join();
    for (auto p : src) free(p);
    return (double)(f1 - f0) * (double)frame / dt / 1e9;
}

int main() {
    const size_t sizes[2] = {1920u * 1080u * 3u, 3840u * 2160u * 3u};
    for (size_t frame : sizes)
        for (int stage = 1; stage >= 0; stage--)
            for (int threads : {4, 8, 16, 24})
                for (int chunks : {1, 4, 8}) {
                    if (!stage && chunks != 1) continue;
                    printf("frame %5.1f MB  %s  threads %2d  chunks %d : %6.1f GB/s\n", frame / 1e6, stage ? "memcpy+H2D" : "H2D only  ", threads, chunks,
                           run(frame, threads, chunks, stage != 0, 1.0));
                    fflush(stdout);
                }
    return 0;
}
