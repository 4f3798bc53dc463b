// occupancy_probe2.hip — resident workgroups per CU as a function of workgroup size (waves) and VGPR allocation, small LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VGPRS>
__global__ void __launch_bounds__(1024) k_spin(uint32_t *out, uint32_t ticks) {
    extern __shared__ uint32_t lds[];
    if (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VGPRS == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VGPRS == 88) asm volatile("v_mov_b32 v87, 0" ::: "v87");
    if (VGPRS == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VGPRS == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (VGPRS == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (VGPRS == 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    lds[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) % blockDim.x] == 0xFFFFFFFFu) out[0] = 1;
}

template <int VGPRS>
static int sweep(uint32_t *d_out, int cus) {
    const uint32_t ticks = 5000;                  // 50 us
    const int per_cu = 48;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves : {2, 3, 4, 5, 6, 8, 10, 12}) {
        hipLaunchKernelGGL(k_spin<VGPRS>, dim3(cus * per_cu), dim3(waves * 64), 4096, 0, d_out, 100u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<VGPRS>, dim3(cus * per_cu), dim3(waves * 64), 4096, 0, d_out, ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double r = per_cu / (ms / 0.05);
        printf("vgprs %3d  %2d waves per workgroup: %.3f ms -> %.2f workgroups = %.1f waves resident per CU (%.2f per SIMD)\n", VGPRS, waves, ms, r, r * waves, r * waves / 4);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, 64));
    if (sweep<32>(d_out, cus) || sweep<56>(d_out, cus) || sweep<64>(d_out, cus) || sweep<72>(d_out, cus) || sweep<80>(d_out, cus) || sweep<88>(d_out, cus) || sweep<96>(d_out, cus) || sweep<104>(d_out, cus) || sweep<128>(d_out, cus)) return 1;
    return 0;
}
