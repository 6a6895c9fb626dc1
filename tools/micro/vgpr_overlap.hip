// dev probe: do two waves that share a SIMD keep disjoint registers when the kernel descriptor says 248 VGPRs (accum_offset 248)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(unsigned* bad, int rounds) {
    extern __shared__ unsigned char smem[];
    unsigned nbad = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned tag = (blockIdx.x * 64u + threadIdx.x) * 65536u + (unsigned)r * 2048u + 1u;
        unsigned res;
        __asm__ volatile(
        "v_mov_b32 v232, %1\n"
        "v_mov_b32 v233, %1\n"
        "v_mov_b32 v234, %1\n"
        "v_mov_b32 v235, %1\n"
        "v_mov_b32 v236, %1\n"
        "v_mov_b32 v237, %1\n"
        "v_mov_b32 v238, %1\n"
        "v_mov_b32 v239, %1\n"
        "v_mov_b32 v240, %1\n"
        "v_mov_b32 v241, %1\n"
        "v_mov_b32 v242, %1\n"
        "v_mov_b32 v243, %1\n"
        "v_mov_b32 v244, %1\n"
        "v_mov_b32 v245, %1\n"
        "v_mov_b32 v246, %1\n"
        "v_mov_b32 v247, %1\n"
        "s_mov_b32 s40, 3000\n"
        "1: s_sleep 10\n s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
        "v_xor_b32 %0, v232, %1\n"
        "v_xor_b32 v233, v233, %1\n v_or_b32 %0, %0, v233\n"
        "v_xor_b32 v234, v234, %1\n v_or_b32 %0, %0, v234\n"
        "v_xor_b32 v235, v235, %1\n v_or_b32 %0, %0, v235\n"
        "v_xor_b32 v236, v236, %1\n v_or_b32 %0, %0, v236\n"
        "v_xor_b32 v237, v237, %1\n v_or_b32 %0, %0, v237\n"
        "v_xor_b32 v238, v238, %1\n v_or_b32 %0, %0, v238\n"
        "v_xor_b32 v239, v239, %1\n v_or_b32 %0, %0, v239\n"
        "v_xor_b32 v240, v240, %1\n v_or_b32 %0, %0, v240\n"
        "v_xor_b32 v241, v241, %1\n v_or_b32 %0, %0, v241\n"
        "v_xor_b32 v242, v242, %1\n v_or_b32 %0, %0, v242\n"
        "v_xor_b32 v243, v243, %1\n v_or_b32 %0, %0, v243\n"
        "v_xor_b32 v244, v244, %1\n v_or_b32 %0, %0, v244\n"
        "v_xor_b32 v245, v245, %1\n v_or_b32 %0, %0, v245\n"
        "v_xor_b32 v246, v246, %1\n v_or_b32 %0, %0, v246\n"
        "v_xor_b32 v247, v247, %1\n v_or_b32 %0, %0, v247\n"
        : "=&v"(res) : "v"(tag) : "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "s40", "scc");
        nbad += res != 0;
    }
    if (smem[threadIdx.x] == 77 && rounds < 0) nbad++;
    if (nbad) atomicAdd(bad, nbad);
}
int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 3;
    const size_t lds = (size_t)160 * 1024 / per_cu - 512;
    unsigned* bad;
    (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nblk = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, probe, 64, lds);
    hipLaunchKernelGGL(probe, dim3(256 * per_cu), dim3(64), lds, 0, bad, 20);
    hipError_t e = hipDeviceSynchronize();
    unsigned h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("per_cu %d (occupancy query %d), v232..v247 pinned: %s, corrupted lanes-rounds %u\n", per_cu, nblk, hipGetErrorString(e), h);
    return 0;
}
