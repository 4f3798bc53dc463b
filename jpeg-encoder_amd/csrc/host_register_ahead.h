// host_register_ahead.h — JPEGENC_UPLOAD_REGISTER_AHEAD (jpegenc_encoder_set_batch_upload): the pageable frames of a host-fed batch are
// page-locked a few ahead of the workers by ONE thread of the handle (hipHostRegister of whole pages, in frame order), uploaded by the
// workers where they lie, and released behind the workers by a second thread - never more than one thread inside hipHostRegister and
// one inside hipHostUnregister, never a pageable hipMemcpyAsync from a worker.  The reference reads the caller's slice in place
// (encoder.rs:440-454); this is its closest equivalent across a PCIe link.
//
// Lifetime of a registration (round 6; until then a range was released once the frame AFTER it was done, which is only right for
// distinct frames in ascending order): every registration counts the frames that upload from it - its own frame and every later
// frame whose first or last page, or whole extent, lies inside it (frames that share a page; a batch that repeats a pointer:
// [A, A, A, ...], [A, B, A]) - and is released when that count is zero.  Look-up, counting and the decision to release happen under
// one mutex, so a frame can only ever be pointed at a registration that stays until the frame is done; a registration that is being
// released (`dying`) or a layout the pieces cannot describe sends the frame through the staged path instead.
#pragma once
#include <map>

#include "host_internal.h"

namespace jpegenc {

// Every registration any RegisterAhead of this process holds (several handles - the per-device children of a multi-device batch, an
// application's own threads - may run batches over neighbouring or identical memory at the same time): a frame that touches ANOTHER
// batch's registration is staged - that registration goes when its own batch says so, not when this frame is done.
// Look-up and hipHostRegister happen under ONE lock: two batches that found the same pages free at the same time both "succeeded" in
// registering them, the second release then handed the runtime a pointer it no longer knew - and the runtime ABORTS on that ("Memobj map
// does not have ptr", rocclr device.cpp:359; tests/test_gpu_batch_multi.py::test_register_ahead_from_two_handles_over_the_same_memory).
// A range stays in the map until its hipHostUnregister has returned.
struct RegisterAheadRegistry {
    std::mutex mu;
    std::map<uintptr_t, uintptr_t> owned;          // start -> end (being released ones included)
    static RegisterAheadRegistry &get() { static RegisterAheadRegistry r; return r; }
    bool overlaps_locked(uintptr_t a, uintptr_t b) const {
        auto it = owned.upper_bound(a);
        if (it != owned.begin()) { auto before = std::prev(it); if (before->second > a) return true; }
        return it != owned.end() && it->first < b;
    }
    bool overlaps(uintptr_t a, uintptr_t b) { std::lock_guard<std::mutex> l(mu); return overlaps_locked(a, b); }
    // 0 = registered (and recorded), 1 = another batch's registration in the way, 2 = the runtime declined
    int try_register(uintptr_t a, uintptr_t b) {
        std::lock_guard<std::mutex> l(mu);
        if (overlaps_locked(a, b)) return 1;
        if (hipHostRegister((void *)a, b - a, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 2; }
        owned[a] = b;
        return 0;
    }
    void release(uintptr_t a) {                     // (the unregistering itself outside the lock: it can take milliseconds)
        if (hipHostUnregister((void *)a) != hipSuccess) (void)hipGetLastError();
        std::lock_guard<std::mutex> l(mu);
        owned.erase(a);
    }
};

struct RegisterAhead {
    const uint8_t *const *frames;
    const size_t bytes;
    const int n, depth, device;
    std::atomic<int> &taken;                       // frames handed to workers so far (the batch's own counter)
    // per frame: 0 = not looked at yet, 1 = page-locked here (pieces[] says where), 2 = to be staged (the caller's in part, shares pages
    // in a way the pieces cannot describe, or not lockable), 3 = page-locked by the caller as a whole: uploaded in place
    std::unique_ptr<std::atomic<int>[]> state;
    struct Pieces { size_t n[3]; int range[3]; };  // ascending addresses: bytes of the frame inside registration range[k] (-1: none)
    std::vector<Pieces> pieces;
    std::atomic<bool> finished{false}, reg_done{false};
    std::atomic<uint64_t> registered_bytes{0}, register_ns{0}, unregister_ns{0};
    bool gave_up = false;                          // (registrar thread only; read by the caller after finish())

    RegisterAhead(const uint8_t *const *f, size_t b, int count, int ahead, int dev, std::atomic<int> &next)
        : frames(f), bytes(b), n(count), depth(ahead), device(dev), taken(next), state(new std::atomic<int>[(size_t)count]),
          pieces((size_t)count, Pieces{{0, 0, 0}, {-1, -1, -1}}), ranges_((size_t)count, Range{0, 0, 0, false}) {
        for (int i = 0; i < count; i++) state[i].store(0);
        // Two threads: locking is cheap (1 000 1080p frames on huge pages: 5 ms in all), RELEASING is what costs (110 ms for the same
        // frames); with both halves on one thread the workers wait for frames that are not locked yet because the thread is busy
        // unlocking (6 700 against 7 500 frames/s staged; two threads: 8 100).  JPEGENC_REGISTER_AHEAD_THREADS=1 (diagnostic build):
        // both halves on one thread, profiles/r05_upload_modes.txt.
        static const bool one = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_REGISTER_AHEAD_THREADS"); return v && atoi(v) == 1; }();
        two_threads_ = !one;
        registrar_ = std::thread([this] { register_loop(); });
        if (two_threads_) releaser_ = std::thread([this] { release_loop(); });
    }
    ~RegisterAhead() { finish(); }

    void finish() {                                 // every worker is done (or has given up): release what is still locked
        if (!registrar_.joinable()) return;
        finished.store(true);
        { std::lock_guard<std::mutex> lock(mu_); }
        cv_.notify_all();
        registrar_.join();
        if (releaser_.joinable()) releaser_.join();
    }
    void wait_ready(int i) {                        // worker: frame i has been looked at
        if (state[i].load(std::memory_order_acquire)) return;
        std::unique_lock<std::mutex> lock(mu_);
        cv_.notify_all();                           // (the registrar may be waiting for `taken` to move)
        cv_.wait(lock, [&] { return state[i].load(std::memory_order_acquire) != 0 || finished.load(); });
    }
    void frame_done(int i) {                        // worker: frame i's upload has been waited for
        {
            std::lock_guard<std::mutex> lock(mu_);
            drop_users(pieces[(size_t)i]);
        }
        cv_.notify_all();
    }

  private:
    struct Range { uintptr_t a, b; int users; bool dying; };   // a registration made for frame [index]: pages [a, b), frames still to upload from it
    std::vector<Range> ranges_;
    std::map<uintptr_t, int> live_;                // start address -> index into ranges_, the dying ones included (guarded by mu_)
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread registrar_, releaser_;
    bool two_threads_ = false;

    static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void drop_users(const Pieces &pc) {             // (mu_ held)
        for (int k = 0; k < 3; k++)
            if (pc.range[k] >= 0 && pc.n[k]) ranges_[(size_t)pc.range[k]].users--;
    }

    // Release every registration no frame uploads from any more (all of them once `everything` is set: the batch is over).
    bool release_pass(bool everything) {
        std::vector<int> victims;
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (const auto &kv : live_) {
                Range &r = ranges_[(size_t)kv.second];
                if (!r.dying && (everything || r.users <= 0)) { r.dying = true; victims.push_back(kv.second); }
            }
        }
        for (int v : victims) {
            const uint64_t t0 = now_ns();
            RegisterAheadRegistry::get().release(ranges_[(size_t)v].a);
            unregister_ns += now_ns() - t0;
        }
        if (!victims.empty()) {
            std::lock_guard<std::mutex> lock(mu_);
            for (int v : victims) { live_.erase(ranges_[(size_t)v].a); ranges_[(size_t)v] = Range{0, 0, 0, false}; }
        }
        return !victims.empty();
    }
    void release_loop() {
        if (hipSetDevice(device) != hipSuccess) (void)hipGetLastError();
        for (;;) {
            const bool over = finished.load() && reg_done.load();
            const bool progressed = release_pass(over);
            if (over) break;
            if (!progressed) {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait_for(lock, std::chrono::microseconds(200));
            }
        }
    }

    // Frame `i`: where its bytes lie relative to the registrations that are live, what is left to register, and the users it adds.
    // Returns the state (1 / 2 / 3) - with mu_ NOT held; takes it for the look-up and the bookkeeping.
    int place_frame(int i, uintptr_t page) {
        const uint8_t *p = frames[i];
        if (!p || !bytes) return 2;
        const uintptr_t first = (uintptr_t)p, end = first + bytes;
        const uintptr_t a = first & ~(page - 1), b = (end + page - 1) & ~(page - 1);
        Pieces pc = {{0, 0, 0}, {-1, -1, -1}};
        uintptr_t ua = a, ub = b;                                   // what no live registration covers
        bool describable = true;
        {
            std::lock_guard<std::mutex> lock(mu_);
            auto it = live_.upper_bound(a);                         // ranges are disjoint: the one that may contain `a` starts at or before it
            if (it != live_.begin()) --it;
            for (; it != live_.end() && it->first < b && describable; ++it) {
                const Range &r = ranges_[(size_t)it->second];
                if (r.b <= a) continue;
                if (r.dying) { describable = false; break; }
                if (r.a <= a && r.b >= b) { pc.n[0] = bytes; pc.range[0] = it->second; ua = ub = b; }                 // the whole frame
                else if (r.a <= a) { pc.n[0] = (size_t)(std::min(r.b, end) - first); pc.range[0] = it->second; ua = r.b; }   // its first page(s)
                else if (r.b >= b && pc.range[2] < 0) { pc.n[2] = (size_t)(end - std::max(r.a, first)); pc.range[2] = it->second; ub = r.a; }   // its last page(s)
                else describable = false;                           // a registration in the middle of the frame
            }
            if (describable) {
                if (pc.n[0] + pc.n[2] > bytes) describable = false;
                else {
                    pc.n[1] = bytes - pc.n[0] - pc.n[2];
                    for (int k : {0, 2}) if (pc.range[k] >= 0 && pc.n[k]) ranges_[(size_t)pc.range[k]].users++;
                }
            }
        }
        if (!describable) return 2;
        const bool shares = (pc.range[0] >= 0 && pc.n[0]) || (pc.range[2] >= 0 && pc.n[2]);
        int st = 2;
        if (ub > ua && RegisterAheadRegistry::get().overlaps(ua, ub)) {
            st = 2;                                                 // another batch's registration (this batch's own are not in [ua, ub))
        } else if (ub > ua) {
            // page-locked by someone else - the caller - in whole or in part: left as it is (looked at outside this batch's own registrations)
            const bool pinned_head = is_pinned_host((const uint8_t *)std::max(ua, first)), pinned_tail = is_pinned_host((const uint8_t *)std::min(ub, end) - 1);
            if (pinned_head || pinned_tail) {
                // (as a whole = inside ONE registration that is not this batch's: a frame that merely starts and ends in page-locked
                //  memory - another handle's register-ahead next door, two registrations of the caller's - is staged)
                // (... and not ANOTHER batch's either: a registration that is visible is in the registry - both happen under its lock)
                st = pinned_head && pinned_tail && !shares && is_pinned_host_range(p, bytes) && !RegisterAheadRegistry::get().overlaps(a, b) ? 3 : 2;
            } else {
                const uint64_t t0 = now_ns();
                if (RegisterAheadRegistry::get().try_register(ua, ub) == 0) {
                    registered_bytes += ub - ua; register_ns += now_ns() - t0;
                    std::lock_guard<std::mutex> lock(mu_);
                    ranges_[(size_t)i] = Range{ua, ub, pc.n[1] ? 1 : 0, false};
                    live_[ua] = i;
                    pc.range[1] = i;
                    st = 1;
                } else {
                    (void)hipGetLastError();                        // (someone else's registration in the way, a limit: the worker stages this frame)
                }
            }
        } else {
            st = 1;                                                 // wholly inside pages earlier frames brought along
        }
        if (st == 1) {
            pieces[(size_t)i] = pc;
        } else if (shares) {                                        // not uploaded from the registrations after all
            std::lock_guard<std::mutex> lock(mu_);
            drop_users(pc);
        }
        return st;
    }

    void register_loop() {
        if (hipSetDevice(device) != hipSuccess) (void)hipGetLastError();
        const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
        int reg = 0;
        for (;;) {
            bool progressed = false;
            const bool fin = finished.load();
            if (!two_threads_) {
                progressed = release_pass(fin);
                if (fin) break;
            } else if (fin || reg >= n) {
                break;                              // (the releaser sees to the rest)
            }
            if (reg < n && reg < taken.load() + depth) {
                // How fast pages can be locked depends on what backs them: frames on transparent huge pages lock at > 1 TB/s (5 ms
                // for 6.2 GB), frames on 4 KB pages at 9-13 GB/s - a quarter of what the link moves (profiles/r05_upload_modes.txt).
                // The thread times itself: once at least three frames and 16 MB are on record at under 30 GB/s, it stops locking and
                // the rest of the batch is staged by the workers as in the default mode.
                if (!gave_up && reg >= 3 && registered_bytes.load() >= ((uint64_t)16 << 20) &&
                    (double)registered_bytes.load() / (double)(register_ns.load() ? register_ns.load() : 1) < 30.0) gave_up = true;
                const int st = gave_up ? 2 : place_frame(reg, page);
                state[reg].store(st, std::memory_order_release);
                reg++; progressed = true;
                { std::lock_guard<std::mutex> lock(mu_); }
                cv_.notify_all();
            }
            if (!progressed) {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait_for(lock, std::chrono::microseconds(200));
            }
        }
        reg_done.store(true);
        { std::lock_guard<std::mutex> lock(mu_); }
        cv_.notify_all();
    }
};

}  // namespace jpegenc
