// host_register_rates.cpp — three ways of getting a PAGEABLE frame to the GPU from T host threads (calibration for the batch
// workers' upload): (a) the staging copy into the worker's page-locked buffer + async DMA (what the workers do),
// (b) hipHostRegister of the caller's frame in place + async DMA + hipHostUnregister (no host copy: the DMA engine reads the
// caller's pages), (c) a plain hipMemcpyAsync from pageable memory (the runtime stages it itself).  Prints GB/s, frames/s and the
// CPUs the process kept busy (cgroup cpu.stat) - on a host whose eight ranks share two sockets' DRAM and, in a container, a CPU
// quota, the CPU time and the DRAM moves per frame byte are what is scarce, not the link.
//   hipcc -O2 -o host_register_rates host_register_rates.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static long long cpu_usage_usec() {
    FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r");
    if (!f) return -1;
    char key[64]; long long v = -1, out = -1;
    while (fscanf(f, "%63s %lld", key, &v) == 2) if (!strcmp(key, "usage_usec")) out = v;
    fclose(f);
    return out;
}

static void run(size_t frame, int threads, int mode, int frames_per_thread, double seconds) {
    std::atomic<long> frames(0);
    std::atomic<bool> stop(false);
    std::vector<std::thread> pool;
    // every thread cycles through its own set of distinct pageable frames (page-aligned, as a frame allocator would hand them out)
    std::vector<std::vector<char *>> src((size_t)threads);
    for (int t = 0; t < threads; t++)
        for (int k = 0; k < frames_per_thread; k++) {
            char *p = nullptr;
            if (posix_memalign((void **)&p, 4096, frame)) exit(1);
            memset(p, t + k + 1, frame);
            src[t].push_back(p);
        }
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            char *pinned = nullptr, *dev;
            if (mode == 0) hipHostMalloc((void **)&pinned, frame, hipHostMallocDefault);
            hipMalloc((void **)&dev, frame);
            for (long k = 0; !stop.load(); k++) {
                char *f = src[t][(size_t)(k % frames_per_thread)];
                if (mode == 0) { memcpy(pinned, f, frame); hipMemcpyAsync(dev, pinned, frame, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
                else if (mode == 1) {
                    if (hipHostRegister(f, frame, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); break; }
                    hipMemcpyAsync(dev, f, frame, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
                    hipHostUnregister(f);
                } else { hipMemcpyAsync(dev, f, frame, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
                frames.fetch_add(1);
            }
            hipFree(dev); if (pinned) hipHostFree(pinned); hipStreamDestroy(st);
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    const long f0 = frames.load();
    const long long c0 = cpu_usage_usec();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const long f1 = frames.load();
    const long long c1 = cpu_usage_usec();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    for (auto &th : pool) th.join();
    for (auto &v : src) for (auto p : v) free(p);
    const char *names[3] = {"staging copy + DMA     ", "register in place + DMA", "pageable hipMemcpyAsync"};
    printf("frame %5.1f MB  %s  threads %2d : %6.1f GB/s  %8.0f frames/s  %5.1f CPUs busy  (%.0f us of CPU per frame)\n", frame / 1e6, names[mode], threads,
           (double)(f1 - f0) * (double)frame / dt / 1e9, (f1 - f0) / dt, c0 >= 0 ? (c1 - c0) / 1e6 / dt : -1.0, c0 >= 0 && f1 > f0 ? (double)(c1 - c0) / (double)(f1 - f0) : -1.0);
    fflush(stdout);
}

int main() {
    const size_t sizes[2] = {1920u * 1080u * 3u, 3840u * 2160u * 3u};
    for (size_t frame : sizes)
        for (int mode = 0; mode < 3; mode++)
            for (int threads : {1, 2, 4, 8, 14})
                run(frame, threads, mode, 8, 1.0);
    return 0;
}
