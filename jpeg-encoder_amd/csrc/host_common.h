// host_common.h — internals shared by the C-ABI translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/jpegenc_mi355x.h"
#include "device_params.h"
#include "entropy_params.h"

namespace jpegenc {

void set_last_error(const std::string &msg);
int fail(int status, const std::string &msg);
int hip_fail(hipError_t e, const char *what);

#define JPEGENC_HIP(call)                                        \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return jpegenc::hip_fail(e__, #call); \
    } while (0)

// init_components (encoder.rs:569-631) for a JpegColorType.
int components_for(int jpeg_color_type, int hs, int vs, jpegenc_layout *L);
int jpeg_color_type_of(int color_type);

// Fill the kernel parameter block for interleaved pixel input.
int build_block_params(BlockKernelParams *p, const jpegenc_layout &L, int width, int height,
                       int color_type, const jpegenc_qtable tables[2], int order);
// Same for planar, already-converted input (user ImageBuffer path).
int build_block_params_planes(BlockKernelParams *p, const jpegenc_layout &L, int width, int height,
                              const jpegenc_qtable tables[2], int order);

// block_kernels.hip
hipError_t launch_blocks_generic(const BlockKernelParams &p, int num_frames, int variant, hipStream_t stream);
hipError_t launch_histogram(const HistKernelParams &p, hipStream_t stream);
hipError_t launch_hist_finish(const HistFinishParams &p, hipStream_t stream, int frames = 1);
// fast_kernels.hip: returns false when the configuration has no specialised kernel
bool launch_blocks_fast(const BlockKernelParams &p, int num_frames, int variant, hipStream_t stream,
                        hipError_t *err);
// fast_kernels_planes.hip: a described planar source in ONE launch (sampling factors 1 and 2); false = take the per-plane launches
bool launch_blocks_planes_once(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled, int num_frames, int variant,
                               hipStream_t stream, hipError_t *err);
// (a batch of described surfaces: p.pixels = device table [frame][8] of plane addresses and pitches, p.pixel_frame_stride = kPlaneTableStrideHost)
constexpr uint64_t kPlaneTableStrideHost = ~0ull;
bool fused_planes_supported(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled);
hipError_t launch_group_planes(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled, const EntropyParams *d_params,
                               int frames, int variant, hipStream_t st);
// fast_kernels.hip: a device-resident planar source, one launch per component plane (jpegenc_encoder_encode_planes_device)
hipError_t launch_blocks_planes(const BlockKernelParams &base, const jpegenc_plane planes[4], bool planes_subsampled, int variant,
                                hipStream_t stream);

// entropy_kernels.hip
// stored: what the parameter blocks at d_params currently hold (nullptr: always store) - a caller that codes the same
// scan again and again (an Encoder fed frames of one geometry) skips the store launch
hipError_t store_entropy_params(const EntropyParams *jobs, int njobs, EntropyParams *d_params, int frames, hipStream_t stream, std::string *stored);
// fused_kernels.hip: the Encoder's interleaved baseline scan coded straight from the pixels (no coefficients in HBM)
struct FusedSource {
    const BlockKernelParams *blocks; int variant; const jpegenc_plane *planes; bool planes_subsampled;   // planes: a described planar source (else null)
    uint32_t *chain = nullptr, *finish_abort = nullptr;      // both set: the kernel may finish the scan itself (finish_run.hip.h) where the scan qualifies
    uint32_t *finish_done = nullptr;                         // (the scan goes to pinned host memory) the host word the kernel sets when all of it is there
    uint32_t *stripe_ends = nullptr;                         // (the frame is coded in several launches: blocks->group_base / group_count / stripe_index) where each launch's bytes end
};
bool fused_supported(const BlockKernelParams &b);      // the layout has a fused kernel (interleaved order, 3 to 6 blocks per MCU, sampling factors 1 and 2)
bool fused_enabled();                                    // the Encoder uses it (default; JPEGENC_FUSED=0 keeps block kernel + k_block_code)
uint32_t fused_run_blocks(const BlockKernelParams &b);
uint32_t fused_runs(const BlockKernelParams &b);
uint32_t fused_slot_words(const BlockKernelParams &b, uint32_t slot_words_64);   // slot of a run, given the slot of 64 blocks
hipError_t launch_fused_code(const BlockKernelParams &b, const EntropyParams *d_params, int restart_interval, int frames, int variant,
                             hipStream_t st);
// fused != nullptr (one scan): the first kernel of the sequence reads pixels instead of coefficients
hipError_t launch_entropy_scans(const EntropyParams *jobs, int njobs, EntropyParams *d_params, int frames, hipStream_t stream,
                                std::string *stored = nullptr, const FusedSource *fused = nullptr, int group_stride = 0);

hipError_t launch_batch_gather(const BatchGatherArgs &a, const void *d_src, const uint32_t *d_len, uint64_t *d_pos, void *d_dst,
                               hipStream_t st, uint32_t *h_len = nullptr);
hipError_t launch_gather_scans(const GatherArgs &a, const void *d_src, const uint32_t *d_len, void *d_dst, hipStream_t stream);

// staged_pull.hip: the upload of one pageable image as a kernel that follows the copier threads through the page-locked staging buffer
constexpr uint32_t kStagedPullGroups = 32u, kStagedPullThreads = 256u;
constexpr uint64_t kStagedPullTimeoutTicks = 200000000ull;      // 2 s of the 100 MHz wall clock without a new chunk: the kernel gives up
hipError_t launch_staged_pull(const uint8_t *h_staged, uint8_t *d_pixels, size_t from, size_t to, uint32_t chunk, const uint64_t *h_ready,
                              uint32_t epoch, uint32_t *h_timed_out, hipStream_t stream);      // bytes [from, to) of the image

// capi_entropy.hip
int scan_device(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L,
                const jpegenc_scan &sc, const jpegenc_huffman_spec (*tables)[2], const void *d_lut, void *d_out,
                size_t out_frame_stride, uint32_t *d_out_lengths, void *d_ws, size_t ws_bytes, hipStream_t st,
                std::string *stored_params = nullptr, const FusedSource *fused = nullptr, bool lut_per_frame = false);
// only fills and (unless `stored_params` says they are there) stores the scan's parameter blocks: what a replayed launch
// sequence needs done outside of it
int scan_store_params(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L,
                      const jpegenc_scan &sc, const void *d_lut, void *d_out, size_t out_frame_stride, uint32_t *d_out_lengths,
                      void *d_ws, size_t ws_bytes, hipStream_t st, std::string *stored_params, const FusedSource *fused = nullptr);
struct ScanJob {                 // one scan of scan_device_multi
    jpegenc_scan sc;
    void *d_out; size_t out_frame_stride; uint32_t *d_out_lengths;
    void *d_ws; size_t ws_bytes;
};
int scan_device_multi(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L, const ScanJob *jobs,
                      int njobs, const void *d_lut, hipStream_t st, bool lut_per_frame = false);
int upload_huffman_luts(const jpegenc_huffman_spec (*tables)[2], void *d_lut, hipStream_t st);
// one table set per frame of a batch in one launch (h_specs page-locked: frames x huffman_lut_batch_spec_bytes())
size_t huffman_lut_batch_spec_bytes();
void fill_huffman_lut_spec(void *h_specs, int frame, const jpegenc_huffman_spec (*tables)[2]);
int upload_huffman_luts_batch(const void *h_specs, void *d_specs, void *d_luts, int frames, hipStream_t st);
size_t scan_workspace_size(const jpegenc_layout &L, const jpegenc_scan &sc, int frames);
size_t scan_max_bytes(const jpegenc_layout &L, const jpegenc_scan &sc);

int ensure_device_ready(int device);
// Dense content: blocks that code to more than this many bits on average outgrow their strips in most workgroups of the pixels -> bits
// kernel and the block kernel + k_block_code pair is faster (profiles/r04_fused_quality_matrix.txt; DeviceCtx::dense_last_time,
// jpegenc_pixels_scan_dense).
constexpr uint64_t kDenseBitsPerBlock = 390;
bool is_pinned_host_range(const void *p, size_t bytes);          // inside ONE page-locked registration
bool pinned_registration_of(const void *p, uintptr_t *start, size_t *size);
bool is_pinned_host(const void *p);      // page-locked host memory (hipHostMalloc / hipHostRegister / jpegenc_host_*): DMA reads it in place

}  // namespace jpegenc
