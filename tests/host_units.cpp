// Host-side units of the library that need no GPU (tests/test_host_units.py compiles this with hipcc for the host only and runs it):
// DeviceCtx::StripeTuner (how many stripes a large frame between page-locked buffers takes) and WorkerThreads (the threads a handle
// keeps for its batch calls).  Exit status 0 = every check held; the first failing check is printed.
#include <atomic>
#include <cstdio>
#include <set>

#include "host_internal.h"

#define CHECK(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

using jpegenc::DeviceCtx;
using jpegenc::WorkerThreads;

static int stripe_tuner() {
    DeviceCtx::StripeTuner t;
    // the first six calls of a geometry try 4, 2 and 1 stripes twice each
    int tried[3] = {0, 0, 0};
    auto cost = [](int stripes, float base4, float base2, float base1) { return stripes == 4 ? base4 : stripes == 2 ? base2 : base1; };
    for (int i = 0; i < 6; i++) {
        const int s = t.choose(77);
        CHECK(s == 4 || s == 2 || s == 1);
        tried[s == 4 ? 0 : s == 2 ? 1 : 2]++;
        t.record(i < 3 ? 5000.f : cost(s, 900.f, 760.f, 640.f));      // (first samples: warm-up costs, replaced by the second)
    }
    CHECK(tried[0] == 2 && tried[1] == 2 && tried[2] == 2);
    // then the fastest (one piece here), with another option every 32nd call
    int others = 0;
    for (int i = 6; i < 200; i++) {
        const int s = t.choose(77);
        if (s != 1) { others++; CHECK(t.calls % 32u == 1u); }              // (choose() has counted the call already)
        t.record(cost(s, 900.f, 760.f, 640.f));
    }
    CHECK(others >= 4 && others <= 7);
    // the frames change (larger files: stripes pay now): the re-trials move the choice
    int fours = 0;
    for (int i = 0; i < 400; i++) {
        const int s = t.choose(77);
        t.record(cost(s, 450.f, 520.f, 580.f));
        if (i >= 300 && s == 4) fours++;
    }
    CHECK(fours >= 95);
    // another geometry starts over
    CHECK(t.choose(78) == 4 && t.calls == 1u);
    return 0;
}

static int worker_threads() {
    WorkerThreads pool;
    std::atomic<int> sum(0);
    std::set<std::thread::id> ids;
    std::mutex mu;
    for (int round = 0; round < 200; round++) {
        const int n = 1 + round % 9;                                           // widths 1 .. 9, growing and shrinking
        std::atomic<int> seen_mask(0);
        pool.run(n, [&](int w) {
            seen_mask.fetch_or(1 << w);
            sum.fetch_add(w + 1);
            std::lock_guard<std::mutex> lock(mu);
            ids.insert(std::this_thread::get_id());
        });
        CHECK(seen_mask.load() == (1 << n) - 1);                               // every index exactly once ... (sum below)
    }
    int want = 0;
    for (int round = 0; round < 200; round++) { const int n = 1 + round % 9; want += n * (n + 1) / 2; }
    CHECK(sum.load() == want);
    CHECK(ids.size() == 9);                                                    // the caller's thread + eight that persisted
    pool.stop();
    pool.run(3, [&](int w) { sum.fetch_add(100 * (w + 1)); });                  // usable again after a stop
    CHECK(sum.load() == want + 600);
    pool.run(0, [&](int) { sum.fetch_add(1); });
    CHECK(sum.load() == want + 600);
    return 0;
}

int main() {
    if (stripe_tuner()) return 1;
    if (worker_threads()) return 1;
    printf("host units ok\n");
    return 0;
}
