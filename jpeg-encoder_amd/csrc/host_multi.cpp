// host_multi.cpp — one process, several GPUs: frame-wise shards (frame k -> GPU k mod N, no collective) with a
// child encoder per device, NUMA placement of the host threads that feed a GPU, and page-locked host memory for frames.
#include "host_internal.h"

namespace jpegenc {

// Host threads that feed a GPU should run on the NUMA node its PCIe root complex hangs off (pinned staging memory is
// then first touched there and the uploads do not cross the socket interconnect) - what matters once eight ranks, or one
// process driving eight GPUs, share a two-socket host.  Best effort: any failure leaves the thread where
// it was.  The node's CPU list is read from sysfs once per device.  OPT-IN (JPEGENC_NUMA_BIND=1): on the one host it
// could be measured on (2 x EPYC 9575F, one GPU) binding the 16 workers of a batch to the GPU's node LOST throughput
// (1000 1080p frames: 4 800 vs 6 200 frames/s; the caller's pageable frames live wherever its own thread put them), and
// an eight-GPU node was not available to show the opposite.
static bool device_cpus(int device, cpu_set_t *out) {
    static std::mutex mu;
    static cpu_set_t sets[64];
    static int state[64];              // 0 = unknown, 1 = known, -1 = none
    if (device < 0 || device >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (state[device] == 0) {
        state[device] = -1;
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) == hipSuccess) {
            for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
            char path[160];
            snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
            int node = -1;
            if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
            if (node >= 0) {
                snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
                char list[4096] = {0};
                size_t len = 0;
                if (FILE *f = fopen(path, "r")) { len = fread(list, 1, sizeof list - 1, f); fclose(f); }
                cpu_set_t want;
                CPU_ZERO(&want);
                if (len) (void)parse_cpulist(list, &want);                  // "0-31,128-159"
                if (CPU_COUNT(&want) > 0) { sets[device] = want; state[device] = 1; }
            }
        }
    }
    if (state[device] != 1) return false;
    *out = sets[device];
    return true;
}

void bind_thread_near_device(int device, bool on) {
    thread_local ThreadBinding binding;                                     // (persistent workers: remembered from batch to batch)
    binding.apply(device, on, device_cpus);
}


}  // namespace jpegenc

extern "C" {

// ---- multi-GPU batches (frame k -> GPU k mod N, no collective) ---------------------------------
// Page-locked host memory for frames (and outputs): what the batch entry points upload without a staging copy.
int jpegenc_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null result pointer");
    *out = nullptr;
    if (bytes == 0) return JPEGENC_OK;
    JPEGENC_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return JPEGENC_OK;
}
int jpegenc_host_free(void *p) {
    if (!p) return JPEGENC_OK;
    JPEGENC_HIP(hipHostFree(p));
    return JPEGENC_OK;
}
int jpegenc_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "nothing to register");
    JPEGENC_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return JPEGENC_OK;
}
int jpegenc_host_unregister(void *p) {
    if (!p) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    JPEGENC_HIP(hipHostUnregister(p));
    return JPEGENC_OK;
}

int jpegenc_shard_frames(int num_frames, int num_shards, int shard, int *indices, int capacity) {
    if (num_frames < 0 || num_shards < 1 || shard < 0 || shard >= num_shards)
        return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad shard arguments");
    int n = 0;
    for (int k = shard; k < num_frames; k += num_shards, n++)
        if (indices && n < capacity) indices[n] = k;
    return n;
}

}  // extern "C"

namespace jpegenc {

static int encode_batch_multi(jpegenc_encoder *e, const int *devices, int num_devices, const uint8_t *const *frames, size_t frame_len,
                              int num_frames, int width, int height, int color_type, jpegenc_write_fn sink, void *const *users) {
    if (!devices || num_devices < 1 || num_devices > 64) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad device list");
    if (num_frames < 0 || (num_frames && (!frames || !users)) || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    if (num_frames == 0) return JPEGENC_OK;
    int rc = validate_image(frame_len, width, height, color_type);         // before any device work
    if (rc) return rc;
    for (int d = 0; d < num_devices; d++) {
        rc = ensure_device_ready(devices[d]);
        if (rc) return rc;
    }
    for (int i = 0; i < num_frames; i++)
        if (!frames[i]) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame");
    if ((int)e->shards.size() > num_devices) e->shards.resize((size_t)num_devices);
    while ((int)e->shards.size() < num_devices) e->shards.emplace_back(nullptr);
    const unsigned hw = (unsigned)usable_cpus();
    int per_shard = (int)(hw / (unsigned)num_devices);
    if (per_shard < 4) per_shard = 4;
    if (per_shard > 16) per_shard = 16;
    for (int d = 0; d < num_devices; d++) {
        auto &child = e->shards[(size_t)d];
        if (!child || child->device != devices[d]) {       // its buffers live on the device it was made for
            child.reset(new (std::nothrow) jpegenc_encoder());
            if (!child) return fail(JPEGENC_ERR_HIP, "out of memory");
            child->device = devices[d];
        }
        // everything the parent was told about how batches run applies to each device's share of the batch
        child->cfg = e->cfg;
        child->max_batch_workers = per_shard;
        if (child->batch_workers != e->batch_workers) { child->batch_workers = e->batch_workers; child->release_idle_threads(); }
        child->numa_bind = e->numa_bind;
        child->batch_upload = e->batch_upload;
        if (child->reg_cache.budget != e->reg_cache.budget) {
            if (child->reg_cache.held > e->reg_cache.budget) child->reg_cache.clear();
            child->reg_cache.budget = e->reg_cache.budget;
        }
    }
    std::vector<int> status((size_t)num_devices, JPEGENC_OK);
    std::vector<std::string> messages((size_t)num_devices);
    auto shard_body = [&](int d) {
        // (this driving thread only - it ends with the call; the child's persistent workers place themselves at the top of every
        //  batch body, host_batch.cpp, and un-place themselves when the switch is off or the child is re-made for another device)
        bind_thread_near_device(devices[d], e->numa_bind);
        const int n = jpegenc_shard_frames(num_frames, num_devices, d, nullptr, 0);
        if (n <= 0) { status[(size_t)d] = n < 0 ? -n : JPEGENC_OK; return; }
        std::vector<int> idx((size_t)n);
        (void)jpegenc_shard_frames(num_frames, num_devices, d, idx.data(), n);
        std::vector<const uint8_t *> sub_frames((size_t)n);
        std::vector<void *> sub_users((size_t)n);
        for (int i = 0; i < n; i++) { sub_frames[(size_t)i] = frames[idx[(size_t)i]]; sub_users[(size_t)i] = users[idx[(size_t)i]]; }
        const int r = jpegenc_encoder_encode_batch(e->shards[(size_t)d].get(), sub_frames.data(), frame_len, n, width, height, color_type,
                                                   sink, sub_users.data());
        status[(size_t)d] = r;
        if (r) messages[(size_t)d] = jpegenc_last_error();
    };
    std::vector<std::thread> pool;
    for (int d = 0; d < num_devices; d++) pool.emplace_back(shard_body, d);   // (shard threads only drive; the caller's affinity is left alone)
    for (auto &th : pool) th.join();
    for (int d = 0; d < num_devices; d++)
        if (status[(size_t)d] != JPEGENC_OK) { set_last_error("device " + std::to_string(devices[d]) + ": " + messages[(size_t)d]); return status[(size_t)d]; }
    return JPEGENC_OK;
}

}  // namespace jpegenc

extern "C" {

int jpegenc_encoder_encode_batch_multi(jpegenc_encoder *e, const int *devices, int num_devices, const uint8_t *const *frames,
                                       size_t frame_len, int num_frames, int width, int height, int color_type,
                                       jpegenc_write_fn sink, void *const *users) {
    REQUIRE(e);
    return encode_batch_multi(e, devices, num_devices, frames, frame_len, num_frames, width, height, color_type, sink, users);
}

int jpegenc_encoder_batch_shard_info(jpegenc_encoder *e, int shard, int *device, int *batch_workers, int *upload_mode,
                                     size_t *register_cache_bytes, int *pool_workers) {
    if (!e) return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "null encoder");
    const int n = (int)e->shards.size();
    const jpegenc_encoder *c = shard >= 0 && shard < n ? e->shards[(size_t)shard].get() : nullptr;
    if (device) *device = c ? c->device : -1;
    if (batch_workers) *batch_workers = c ? c->batch_workers : 0;
    if (upload_mode) *upload_mode = c ? c->batch_upload : 0;
    if (register_cache_bytes) *register_cache_bytes = c ? c->reg_cache.budget : 0;
    if (pool_workers) *pool_workers = c ? (int)c->workers.size() : 0;
    return n;
}

int jpegenc_encoder_encode_batch_multi_to_buffers(jpegenc_encoder *e, const int *devices, int num_devices, const uint8_t *const *frames,
                                                  size_t frame_len, int num_frames, int width, int height, int color_type,
                                                  uint8_t *const *outs, const size_t *capacities, size_t *lengths) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!outs || !capacities || !lengths))) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    std::vector<BufferSink> sinks((size_t)num_frames);
    std::vector<void *> users((size_t)num_frames);
    for (int i = 0; i < num_frames; i++) {
        sinks[(size_t)i] = BufferSink{outs[i], outs[i] ? capacities[i] : 0, 0};
        users[(size_t)i] = &sinks[(size_t)i];
    }
    int rc = encode_batch_multi(e, devices, num_devices, frames, frame_len, num_frames, width, height, color_type, buffer_sink, users.data());
    bool fits = true;
    for (int i = 0; i < num_frames; i++) {
        lengths[i] = sinks[(size_t)i].len;
        if (sinks[(size_t)i].len > sinks[(size_t)i].cap) fits = false;
    }
    if (rc) return rc;
    return fits ? JPEGENC_OK : fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "at least one output buffer is too small");
}

}  // extern "C"

