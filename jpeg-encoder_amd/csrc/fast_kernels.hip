// fast_kernels.hip — specialised block-encode kernels (placeholder until the tuned path lands).
#include "host_common.h"
namespace jpegenc {
bool launch_blocks_fast(const BlockKernelParams &, int, int, hipStream_t, hipError_t *) { return false; }
}  // namespace jpegenc
