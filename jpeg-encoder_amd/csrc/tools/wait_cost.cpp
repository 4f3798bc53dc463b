// wait_cost.cpp — what does a host thread COST while it waits for the GPU (calibration only)?  Four threads each keep a stream busy
// with 6.2 MB page-locked uploads (a C3 frame) and wait for every one of them in one of several ways; per way: uploads/s, and the
// CPU time the waiting threads burnt (CLOCK_THREAD_CPUTIME_ID) as CPUs busy, split user / system.
//   hipStreamSynchronize | hipEventSynchronize on a default event | ... on a hipEventBlockingSync event | hipEventQuery + nanosleep
//   (timer slack 1 us) | hipLaunchHostFunc -> condition variable
// Round 5 assumed the blocking-sync event sleeps; profiles/r06_rank_cpu_budget.txt shows every batch worker at 1.00 CPUs of USER time
// inside libhsa-runtime64 while it waits.  Environment knobs worth a run each: HSA_ENABLE_MWAITX=0/1, HSA_ENABLE_INTERRUPT=0/1.
//   hipcc -O2 -o wait_cost wait_cost.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/resource.h>
#include <time.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double thread_cpu_s(int which) {         // 0 = user + system of the calling thread
    rusage ru;
    getrusage(RUSAGE_THREAD, &ru);
    const double u = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, s = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
    return which == 1 ? u : which == 2 ? s : u + s;
}

struct Waiter { std::mutex m; std::condition_variable cv; int done = 0; };
static void host_fn(void *p) { Waiter *w = (Waiter *)p; { std::lock_guard<std::mutex> l(w->m); w->done++; } w->cv.notify_one(); }

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 4, per_thread = argc > 2 ? atoi(argv[2]) : 400;
    const size_t bytes = 6220800;
    const char *names[] = {"hipStreamSynchronize", "hipEventSynchronize (default event)", "hipEventSynchronize (hipEventBlockingSync)",
                           "hipEventQuery + nanosleep 20 us (timer slack 1 us)", "hipLaunchHostFunc -> condition variable",
                           "hipEventQuery + nanosleep 50 us (timer slack 1 us)"};
    printf("%d threads x %d uploads of %zu page-locked bytes each, one stream per thread\n", threads, per_thread, bytes);
    for (int way = 0; way < 6; way++) {
        std::vector<std::thread> pool;
        std::vector<double> user((size_t)threads), sys((size_t)threads);
        std::atomic<int> ready(0);
        std::atomic<bool> go(false);
        const double t_all = now_s();
        double t0 = 0;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                (void)hipSetDevice(0);
                hipStream_t st; hipEvent_t ev;
                (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming | (way == 2 ? hipEventBlockingSync : 0u));
                char *h, *d;
                (void)hipHostMalloc((void **)&h, bytes, hipHostMallocDefault); memset(h, t, bytes);
                (void)hipMalloc((void **)&d, bytes);
                if (way == 3 || way == 5) prctl(PR_SET_TIMERSLACK, 1000UL);
                Waiter w;
                (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st);
                ready++;
                while (!go.load()) std::this_thread::yield();
                const double u0 = thread_cpu_s(1), s0 = thread_cpu_s(2);
                for (int i = 0; i < per_thread; i++) {
                    (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
                    switch (way) {
                    case 0: (void)hipStreamSynchronize(st); break;
                    case 1: case 2: (void)hipEventRecord(ev, st); (void)hipEventSynchronize(ev); break;
                    case 3: case 5: {
                        (void)hipEventRecord(ev, st);
                        const timespec ts = {0, way == 3 ? 20000 : 50000};
                        while (hipEventQuery(ev) == hipErrorNotReady) nanosleep(&ts, nullptr);
                        break;
                    }
                    case 4: {
                        (void)hipLaunchHostFunc(st, host_fn, &w);
                        std::unique_lock<std::mutex> l(w.m);
                        w.cv.wait(l, [&] { return w.done > i; });
                        break;
                    }
                    }
                }
                user[(size_t)t] = thread_cpu_s(1) - u0; sys[(size_t)t] = thread_cpu_s(2) - s0;
                (void)hipStreamSynchronize(st);
                (void)hipFree(d); (void)hipHostFree(h); (void)hipEventDestroy(ev); (void)hipStreamDestroy(st);
            });
        while (ready.load() < threads) std::this_thread::yield();
        rusage r0; getrusage(RUSAGE_SELF, &r0);
        t0 = now_s();
        go.store(true);
        for (auto &th : pool) th.join();
        const double wall = now_s() - t0;
        rusage r1; getrusage(RUSAGE_SELF, &r1);
        double u = 0, s = 0;
        for (int t = 0; t < threads; t++) { u += user[(size_t)t]; s += sys[(size_t)t]; }
        const double pu = (r1.ru_utime.tv_sec - r0.ru_utime.tv_sec) + (r1.ru_utime.tv_usec - r0.ru_utime.tv_usec) * 1e-6;
        const double ps = (r1.ru_stime.tv_sec - r0.ru_stime.tv_sec) + (r1.ru_stime.tv_usec - r0.ru_stime.tv_usec) * 1e-6;
        printf("%-52s %7.0f uploads/s  %5.1f GB/s   waiting threads: %.2f CPUs user + %.2f system;  whole process (runtime threads included): %.2f user + %.2f system\n",
               names[way], threads * per_thread / wall, threads * per_thread * (double)bytes / wall / 1e9, u / wall, s / wall, pu / wall, ps / wall);
        (void)t_all;
    }
    return 0;
}
