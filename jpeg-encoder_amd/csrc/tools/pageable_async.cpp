// pageable_async.cpp — does hipMemcpyAsync on PAGEABLE host memory return before the copy is done (calibration only)?
// Prints, per size and direction: microseconds until the call returns, until the stream is idle, and whether an upload and a
// download issued back to back from one thread on two streams overlap.
//   hipcc -O2 -o pageable_async pageable_async.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double us(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); }
int main() {
    hipStream_t s1, s2;
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const size_t cap = (size_t)32 << 20;
    char *h1 = (char *)malloc(cap), *h2 = (char *)malloc(cap), *d1, *d2, *p1;
    memset(h1, 1, cap); memset(h2, 2, cap);
    (void)hipMalloc((void **)&d1, cap); (void)hipMalloc((void **)&d2, cap); (void)hipHostMalloc((void **)&p1, cap, hipHostMallocDefault);
    for (size_t n : {(size_t)1 << 20, (size_t)10800000, (size_t)14400000, (size_t)24883200}) {
        for (int rep = 0; rep < 3; rep++) {
            auto t = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s1);
            const double r1 = us(t); (void)hipStreamSynchronize(s1); const double c1 = us(t);
            t = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2);
            const double r2 = us(t); (void)hipStreamSynchronize(s2); const double c2 = us(t);
            t = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s1);
            (void)hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2);
            const double rb = us(t); (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); const double cb = us(t);
            t = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(d1, p1, n, hipMemcpyHostToDevice, s1);
            (void)hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2);
            const double rp = us(t); (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); const double cp = us(t);
            if (rep == 2)
                printf("%9zu B: H2D pageable returns %.0f us, done %.0f | D2H pageable returns %.0f, done %.0f | both (2 streams) return %.0f, done %.0f | pinned H2D + pageable D2H return %.0f, done %.0f\n",
                       n, r1, c1, r2, c2, rb, cb, rp, cp);
        }
    }
    return 0;
}
