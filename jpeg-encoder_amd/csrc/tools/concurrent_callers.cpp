// concurrent_callers.cpp — T host threads, one Encoder each, every thread encoding the same image one call at a time
// (jpegenc_encoder_encode_to_buffer from pageable host pixels, or jpegenc_encoder_encode_device from pixels in HBM): frames/s of all
// threads and the latency a caller sees.  The library is opened at run time (JPEGENC_LIB, default the in-tree build) so that
// the diagnostic build's switches can be compared: JPEGENC_NO_FINISH=1 = the launched k_push / k_stuff sequence instead of the
// kernel that finishes the scan itself (whose workgroups wait for their predecessors on their CU slots: a latency path).
//   hipcc -O2 -o concurrent_callers concurrent_callers.cpp -ldl -lpthread && ./concurrent_callers [device]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

typedef struct jpegenc_encoder jpegenc_encoder;
typedef int (*write_fn)(void *, const uint8_t *, size_t);
static int drop(void *user, const uint8_t *, size_t n) { *(size_t *)user += n; return 0; }

int main(int argc, char **argv) {
    const bool device = argc > 1 && !strcmp(argv[1], "device");
    const char *path = getenv("JPEGENC_LIB");
    void *lib = dlopen(path ? path : "jpeg-encoder_amd/libjpegenc_mi355x.so", RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    auto enc_new = (jpegenc_encoder * (*)(int)) dlsym(lib, "jpegenc_encoder_new");
    auto enc_free = (void (*)(jpegenc_encoder *))dlsym(lib, "jpegenc_encoder_free");
    auto set_sampling = (int (*)(jpegenc_encoder *, int))dlsym(lib, "jpegenc_encoder_set_sampling_factor");
    auto to_buffer = (int (*)(jpegenc_encoder *, const uint8_t *, size_t, int, int, int, uint8_t *, size_t, size_t *))dlsym(lib, "jpegenc_encoder_encode_to_buffer");
    auto enc_device = (int (*)(jpegenc_encoder *, const void *, int, int, int, write_fn, void *))dlsym(lib, "jpegenc_encoder_encode_device");
    if (!enc_new || !to_buffer || !enc_device) { printf("symbols missing\n"); return 1; }
    const int sizes[3][2] = {{256, 256}, {1920, 1080}, {3840, 2160}};
    for (auto &wh : sizes) {
        const int w = wh[0], h = wh[1];
        const size_t bytes = (size_t)w * h * 3;
        std::vector<uint8_t> px(bytes);
        uint32_t s = 1;
        for (size_t i = 0; i < bytes; i++) {                          // the reference's gradient with a little noise: entropy-codes like a photograph
            const size_t p = i / 3, x = p % (size_t)w, y = p / (size_t)w, c = i % 3;
            s = s * 1664525u + 1013904223u;
            const int base = c == 0 ? (int)std::min<size_t>(x, 255) : c == 1 ? (int)((2 * y) & 255) : (int)(((std::min<size_t>(x, 255) + 2 * y) / 2) & 255);
            px[i] = (uint8_t)std::min(255, std::max(0, base + (int)(s >> 28) - 8));
        }
        void *d_px = nullptr;
        if (device) { if (hipMalloc(&d_px, bytes) != hipSuccess || hipMemcpy(d_px, px.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 2; }
        for (int T : {1, 2, 4, 8, 16}) {
            std::vector<jpegenc_encoder *> encs((size_t)T);
            std::vector<std::vector<double>> lat((size_t)T);
            std::vector<std::vector<uint8_t>> outs((size_t)T, std::vector<uint8_t>(bytes + 65536));
            for (auto &e : encs) { e = enc_new(85); set_sampling(e, 0x22); }
            std::atomic<int> ready(0);
            std::atomic<bool> go(false), stop(false);
            std::vector<std::thread> pool;
            for (int t = 0; t < T; t++)
                pool.emplace_back([&, t] {
                    size_t n = 0;
                    auto call = [&]() { return device ? enc_device(encs[t], d_px, w, h, 1, drop, &n) : to_buffer(encs[t], px.data(), bytes, w, h, 1, outs[t].data(), outs[t].size(), &n); };
                    for (int k = 0; k < 5; k++) if (call()) { printf("encode failed\n"); exit(3); }
                    ready.fetch_add(1);
                    while (!go.load()) std::this_thread::yield();
                    while (!stop.load()) {
                        const auto t0 = std::chrono::steady_clock::now();
                        call();
                        lat[t].push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                    }
                });
            while (ready.load() < T) std::this_thread::yield();
            go.store(true);
            const auto t0 = std::chrono::steady_clock::now();
            std::this_thread::sleep_for(std::chrono::seconds(1));
            stop.store(true);
            for (auto &th : pool) th.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<double> all;
            for (auto &l : lat) all.insert(all.end(), l.begin(), l.end());
            std::sort(all.begin(), all.end());
            printf("{\"image\": \"%dx%d\", \"input\": \"%s\", \"threads\": %d, \"frames_per_s\": %.1f, \"median_us\": %.1f, \"p95_us\": %.1f, \"self_finishing_kernel\": %s}\n",
                   w, h, device ? "device-resident" : "host", T, all.size() / dt, all[all.size() / 2] * 1e6, all[(size_t)(all.size() * 0.95)] * 1e6,
                   getenv("JPEGENC_NO_FINISH") ? "false" : "true");
            fflush(stdout);
            for (auto e : encs) enc_free(e);
        }
        if (d_px) (void)hipFree(d_px);
    }
    return 0;
}
