// occupancy_probe.hip — how many 384-thread workgroups does a CU of this GPU keep resident for a given dynamic LDS size and
// VGPR allocation?  (calibration for the pixels -> bits kernel's LDS budget: fused_kernel_impl.hip.h)
// Every workgroup spins for a fixed time; with W workgroups per CU launched, the kernel takes ceil(W / resident) spins.
//   hipcc --offload-arch=gfx950 -O2 -o occupancy_probe occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VGPRS>
__global__ void __launch_bounds__(384) k_spin(uint32_t *out, uint32_t ticks) {
    extern __shared__ uint32_t lds[];
    if (VGPRS >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    else if (VGPRS >= 88) asm volatile("v_mov_b32 v87, 0" ::: "v87");
    else if (VGPRS >= 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    else if (VGPRS >= 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    lds[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) % 384] == 0xFFFFFFFFu) out[0] = 1;
}

template <int VGPRS>
static int sweep(uint32_t *d_out, int cus, const std::vector<size_t> &sizes) {
    CHECK(hipFuncSetAttribute((const void *)k_spin<VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t ticks = 20000;                 // 200 us of the 100 MHz clock
    const int per_cu = 6;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t lds : sizes) {
        hipLaunchKernelGGL(k_spin<VGPRS>, dim3(cus * per_cu), dim3(384), lds, 0, d_out, 100u);     // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<VGPRS>, dim3(cus * per_cu), dim3(384), lds, 0, d_out, ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double rounds = ms / 0.2;
        printf("vgprs %3d  lds %6zu B (%5.2f granules of 1280, %6.2f of 512): %.3f ms = %.2f spins -> ~%.1f workgroups (%.0f waves) resident per CU\n", VGPRS, lds,
               lds / 1280.0, lds / 512.0, ms, rounds, per_cu / rounds, 6.0 * per_cu / rounds);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, %zu B of LDS per workgroup at most\n", prop.name, cus, (size_t)prop.sharedMemPerBlock);
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, 64));
    std::vector<size_t> fine;
    for (size_t b = 51200; b <= 56320; b += 256) fine.push_back(b);
    if (sweep<32>(d_out, cus, {16384, 32768, 40960, 49152, 53248, 53760, 54080, 54272, 54613, 55040, 56320, 65536, 79872, 81920})) return 1;
    if (sweep<32>(d_out, cus, fine)) return 1;
    if (sweep<64>(d_out, cus, {32768, 40960, 53760})) return 1;
    if (sweep<80>(d_out, cus, {28416, 32768, 40960, 53760})) return 1;
    if (sweep<88>(d_out, cus, {28416, 32768, 40960, 49152, 53760})) return 1;
    if (sweep<96>(d_out, cus, {16384, 28416, 32768, 36864, 40960, 49152, 53760, 54080})) return 1;
    return 0;
}
