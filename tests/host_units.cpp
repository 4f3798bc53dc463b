// Host-side units of the library that need no GPU (tests/test_host_units.py compiles this with hipcc for the host only and runs it):
// DeviceCtx::StripeTuner (how many stripes a large frame between page-locked buffers takes) and WorkerThreads (the threads a handle
// keeps for its batch calls).  Exit status 0 = every check held; the first failing check is printed.
#include <atomic>
#include <cstdio>
#include <set>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include "host_internal.h"
#include "host_register_ahead.h"

#define CHECK(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

using jpegenc::DeviceCtx;
using jpegenc::WorkerThreads;

static int stripe_tuner() {
    DeviceCtx::StripeTuner t;
    // the first nine calls of a geometry try 4, 2 and 1 stripes three times each
    int tried[3] = {0, 0, 0};
    auto cost = [](int stripes, float base4, float base2, float base1) { return stripes == 4 ? base4 : stripes == 2 ? base2 : base1; };
    for (int i = 0; i < 9; i++) {
        const int s = t.choose(77);
        CHECK(s == 4 || s == 2 || s == 1);
        tried[s == 4 ? 0 : s == 2 ? 1 : 2]++;
        // (first samples: warm-up costs, replaced by the second; the second of one piece is a hiccup that its third corrects)
        t.record(i < 3 ? 5000.f : i == 5 ? 2000.f : cost(s, 900.f, 760.f, 640.f));
    }
    CHECK(tried[0] == 3 && tried[1] == 3 && tried[2] == 3);
    // then the fastest (one piece here), with another option every 32nd call
    int others = 0;
    for (int i = 9; i < 200; i++) {
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

// ThreadBinding (bind_thread_near_device): a persistent worker follows its handle's (device, switch) - on, off, another device -
// with a fake "sysfs": device 0 sits on the first half of the CPUs this process may use, device 1 on the second half, device 2 on
// CPUs the process does not have.
static int thread_binding() {
    cpu_set_t start;
    CHECK(sched_getaffinity(0, sizeof start, &start) == 0);
    std::vector<int> cpus;
    for (int i = 0; i < CPU_SETSIZE; i++) if (CPU_ISSET(i, &start)) cpus.push_back(i);
    if (cpus.size() < 2) { printf("thread binding: one CPU only, skipped\n"); return 0; }
    std::string lists[2];
    for (size_t i = 0; i < cpus.size(); i++) {
        std::string &l = lists[i < cpus.size() / 2 ? 0 : 1];
        if (!l.empty()) l += ",";
        l += std::to_string(cpus[i]);
    }
    auto lookup = [&](int device, cpu_set_t *out) {
        if (device == 2) { CPU_ZERO(out); CPU_SET(CPU_SETSIZE - 1, out); return true; }      // a node whose CPUs are closed to us
        if (device < 0 || device > 2) return false;
        return jpegenc::parse_cpulist(lists[device].c_str(), out);
    };
    auto mask = [] { cpu_set_t m; CPU_ZERO(&m); (void)sched_getaffinity(0, sizeof m, &m); return m; };
    auto same = [](const cpu_set_t &a, const cpu_set_t &b) { return CPU_EQUAL(&a, &b) != 0; };
    cpu_set_t node[2];
    CHECK(jpegenc::parse_cpulist(lists[0].c_str(), &node[0]) && jpegenc::parse_cpulist(lists[1].c_str(), &node[1]));
    cpu_set_t ranges;
    CHECK(jpegenc::parse_cpulist("0-2,5,7-8", &ranges) && CPU_COUNT(&ranges) == 6 && CPU_ISSET(5, &ranges) && !CPU_ISSET(6, &ranges));
    jpegenc::ThreadBinding b;
    b.apply(0, false, lookup);                                   // off and never bound: nothing happens
    CHECK(!b.bound && same(mask(), start));
    b.apply(0, true, lookup);
    CHECK(b.bound && b.device == 0 && same(mask(), node[0]));
    b.apply(0, true, lookup);                                    // the same again: still there
    CHECK(b.bound && same(mask(), node[0]));
    b.apply(0, false, lookup);                                   // the switch goes off: back to where the thread started
    CHECK(!b.bound && same(mask(), start));
    b.apply(1, true, lookup);
    CHECK(b.bound && b.device == 1 && same(mask(), node[1]));
    b.apply(0, true, lookup);                                    // the handle now serves another device: straight over, not an empty intersection
    CHECK(b.bound && b.device == 0 && same(mask(), node[0]));
    b.apply(2, true, lookup);                                    // a node with none of our CPUs: unbound, not left on the old node
    CHECK(!b.bound && same(mask(), start));
    b.apply(1, true, lookup);
    b.apply(7, true, lookup);                                    // a device without a known node: unbound as well
    CHECK(!b.bound && same(mask(), start));
    b.apply(1, true, lookup);
    b.restore();
    CHECK(same(mask(), start));
    return 0;
}

// StripeTuner: an option whose trial calls were never recorded has no cost - it is tried again, it does not win with 0
static int stripe_tuner_untimed_option() {
    DeviceCtx::StripeTuner t;
    for (int i = 0; i < 9; i++) {
        const int s = t.choose(5);
        if (s != 2) t.record(s == 4 ? 700.f : 650.f);               // the two-stripe calls "failed": nothing recorded
    }
    CHECK(t.choose(5) == 2);                                     // still owed a measurement
    t.record(900.f);
    for (int i = 0; i < 20; i++) {
        const int s = t.choose(5);
        if (t.calls % 32u != 1u) CHECK(s == 1);                  // the measured best, not the option that cost "0"
        t.record(s == 4 ? 700.f : s == 2 ? 900.f : 650.f);
    }
    return 0;
}

// BackgroundPool: tasks run behind the submitter's back, wait(group) returns only when every task of THAT group has run - tasks that
// submit more tasks of their group included (the pieces of a large scan, BatchRun::assemble_frames) -, groups are independent, a task may
// wait for another group's tasks (the small-frame staging: the task that drives a round waits for its copiers), the threads persist.
static int background_pool() {
    BackgroundPool pool;
    pool.ensure_threads(4);
    std::atomic<int> done[BackgroundPool::kGroups];
    for (auto &d : done) d.store(0);
    std::set<std::thread::id> ids;
    std::mutex mu;
    for (int round = 0; round < 100; round++) {
        const int g = round % 3;
        for (int i = 0; i < 5; i++)
            pool.submit(g, [&, g] {
                { std::lock_guard<std::mutex> lock(mu); ids.insert(std::this_thread::get_id()); }
                for (int piece = 0; piece < 3; piece++) pool.submit(g, [&, g] { done[g].fetch_add(1); });   // nested: counted before this task ends
                done[g].fetch_add(1);
            });
        if (round % 3 == 2) {
            for (int k = 0; k < 3; k++) pool.wait(k);
            const int rounds_of_group = (round + 1) / 3;
            for (int k = 0; k < 3; k++) CHECK(done[k].load() == rounds_of_group * 5 * 4);
        }
    }
    pool.wait(0);
    CHECK(done[0].load() == 34 * 20);
    CHECK(ids.size() <= 4 && ids.count(std::this_thread::get_id()) == 0);      // the pool's own threads, never the submitter's
    // a task of group 1 waits for tasks of group 0 it submitted itself
    std::atomic<int> inner(0), outer(0);
    pool.submit(1, [&] {
        for (int i = 0; i < 6; i++) pool.submit(0, [&] { inner.fetch_add(1); });
        pool.wait(0);
        outer.store(inner.load());
    });
    pool.wait(1);
    CHECK(outer.load() == 6);
    pool.wait(3);                                                              // a group nothing was ever submitted to
    pool.stop();
    pool.ensure_threads(2);                                                    // usable again after a stop
    pool.submit(2, [&] { inner.fetch_add(10); });
    pool.wait(2);
    CHECK(inner.load() == 16);
    return 0;
}

// jpegenc_encoder_set_batch_workers: one rule sizes every pool of a handle (pool_threads, host_internal.h)
static int thread_budget() {
    using jpegenc::pool_threads;
    // the caller's number wins over everything but the work there is
    CHECK(pool_threads(3, 16, 2, 2, 100) == 3);
    CHECK(pool_threads(1, 16, 2, 2, 100) == 1);                       // the reference's single thread
    CHECK(pool_threads(64, 4, 2, 2, 5) == 5);
    CHECK(pool_threads(2, 8, 1, 1, 0) == 1);                          // (never 0 threads)
    // automatic: the pool's own cap, the CPUs the process may use less the reserve, the floor, the items
    const int cpus = jpegenc::usable_cpus();
    CHECK(cpus >= 1);
    const int want = std::max(2, std::min(4, cpus - 1));
    CHECK(pool_threads(0, 4, 1, 2, 1000) == want);
    CHECK(pool_threads(0, 4, 1, 2, 1) == 1);
    CHECK(pool_threads(0, 16, 2, 2, 1000) == std::max(2, std::min(16, cpus - 2)));
    CHECK(jpegenc::batch_pool_size(0, jpegenc::kDeviceEntropyWorkers, 1000, 1) == want);
    // a thread pinned to one CPU does not shrink the process's pools (the wider of the leader's and the caller's masks counts)
    cpu_set_t before, one;
    CHECK(sched_getaffinity(0, sizeof before, &before) == 0);
    if (CPU_COUNT(&before) >= 2) {
        std::atomic<int> seen(0);
        std::thread t([&] {
            CPU_ZERO(&one);
            for (int i = 0; i < CPU_SETSIZE; i++) if (CPU_ISSET(i, &before)) { CPU_SET(i, &one); break; }
            (void)sched_setaffinity(0, sizeof one, &one);
            seen.store(jpegenc::usable_cpus_now());
        });
        t.join();
        CHECK(seen.load() == jpegenc::usable_cpus_now());
    }
    return 0;
}

// the process-wide registry of register-ahead registrations: interval look-up (the registering itself needs a GPU)
static int register_ahead_registry() {
    jpegenc::RegisterAheadRegistry r;
    CHECK(!r.overlaps(0x1000, 0x2000));
    { std::lock_guard<std::mutex> l(r.mu); r.owned[0x4000] = 0x8000; r.owned[0x10000] = 0x11000; }
    CHECK(!r.overlaps(0x1000, 0x4000));                                // ends where a range begins
    CHECK(r.overlaps(0x1000, 0x4001));
    CHECK(r.overlaps(0x5000, 0x6000));                                 // inside
    CHECK(r.overlaps(0x7fff, 0x9000));                                 // its last byte
    CHECK(!r.overlaps(0x8000, 0x10000));                               // the gap between two ranges
    CHECK(r.overlaps(0x8000, 0x10001));
    CHECK(r.overlaps(0x0, 0x20000));                                   // covers both
    CHECK(!r.overlaps(0x11000, 0x12000));
    return 0;
}

// advance_staged_word: four copiers finish 4 096 chunks in whatever order the scheduler gives them; an observer - the pull kernel's
// view - only ever sees the word grow, every chunk below it finished, and at the end all of them; a word left by another epoch is 0.
static int staged_word() {
    for (int round = 0; round < 20; round++) {
        const uint32_t nchunks = 4096, epoch = 7u + (uint32_t)round;
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[nchunks]);
        for (uint32_t k = 0; k < nchunks; k++) done[k].store(0);
        std::atomic<uint64_t> word(((uint64_t)(epoch - 1u) << 32) | 4096u);         // (what the upload before left behind)
        std::atomic<uint32_t> next(0);
        std::atomic<bool> stop(false), bad(false);
        std::thread observer([&] {
            uint32_t last = 0;
            while (!stop.load()) {
                const uint64_t w = word.load();
                const uint32_t r = (uint32_t)(w >> 32) == epoch ? (uint32_t)w : 0u;
                if (r < last || r > nchunks) bad.store(true);
                for (uint32_t k = last; k < r; k++) if (!done[k].load()) bad.store(true);
                last = r;
            }
        });
        std::vector<std::thread> copiers;
        for (int t = 0; t < 4; t++) copiers.emplace_back([&, t] {
            for (;;) {
                const uint32_t k = next.fetch_add(1);
                if (k >= nchunks) break;
                if (((k * 2654435761u) >> 28) == (uint32_t)t) std::this_thread::yield();      // (neighbours finish out of order)
                done[k].store(1);
                jpegenc::advance_staged_word(&word, epoch, done.get(), nchunks);
            }
        });
        for (auto &c : copiers) c.join();
        jpegenc::advance_staged_word(&word, epoch, done.get(), nchunks);
        stop.store(true);
        observer.join();
        CHECK(!bad.load());
        CHECK(word.load() == (((uint64_t)epoch << 32) | nchunks));
    }
    return 0;
}

int main() {
    if (staged_word()) return 1;
    if (thread_budget()) return 1;
    if (register_ahead_registry()) return 1;
    if (background_pool()) return 1;
    if (thread_binding()) return 1;
    if (stripe_tuner_untimed_option()) return 1;
    if (stripe_tuner()) return 1;
    if (worker_threads()) return 1;
    printf("host units ok\n");
    return 0;
}
