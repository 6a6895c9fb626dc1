// dev probe: is the data of an in-flight vector-memory store safe from a following DPP (or plain) overwrite of its register,
// when two waves share a SIMD?  Each lane stores NR registers (tag ^ register number) and overwrites them at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(unsigned* buf, unsigned* bad, int rounds) {
    extern __shared__ unsigned char smem[];
    unsigned* mine = buf + ((size_t)blockIdx.x * 16) * 64 + threadIdx.x;   // [reg][lane]
    unsigned nbad = 0;
    const unsigned zero = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned tag = (blockIdx.x * 64u + threadIdx.x) * 4096u + (unsigned)r * 16u + 1u;
        if (MODE == 0) __asm__ volatile(
        "v_xor_b32 v200, 200, %[tag]\n"
        "v_xor_b32 v201, 201, %[tag]\n"
        "v_xor_b32 v202, 202, %[tag]\n"
        "v_xor_b32 v203, 203, %[tag]\n"
        "v_xor_b32 v204, 204, %[tag]\n"
        "v_xor_b32 v205, 205, %[tag]\n"
        "v_xor_b32 v206, 206, %[tag]\n"
        "v_xor_b32 v207, 207, %[tag]\n"
        "v_xor_b32 v208, 208, %[tag]\n"
        "v_xor_b32 v209, 209, %[tag]\n"
        "v_xor_b32 v210, 210, %[tag]\n"
        "v_xor_b32 v211, 211, %[tag]\n"
        "v_xor_b32 v212, 212, %[tag]\n"
        "v_xor_b32 v213, 213, %[tag]\n"
        "v_xor_b32 v214, 214, %[tag]\n"
        "v_xor_b32 v215, 215, %[tag]\n"
        "global_store_dword %[addr], v200, off offset:0\n"
        "global_store_dword %[addr], v201, off offset:256\n"
        "global_store_dword %[addr], v202, off offset:512\n"
        "global_store_dword %[addr], v203, off offset:768\n"
        "global_store_dword %[addr], v204, off offset:1024\n"
        "global_store_dword %[addr], v205, off offset:1280\n"
        "global_store_dword %[addr], v206, off offset:1536\n"
        "global_store_dword %[addr], v207, off offset:1792\n"
        "global_store_dword %[addr], v208, off offset:2048\n"
        "global_store_dword %[addr], v209, off offset:2304\n"
        "global_store_dword %[addr], v210, off offset:2560\n"
        "global_store_dword %[addr], v211, off offset:2816\n"
        "global_store_dword %[addr], v212, off offset:3072\n"
        "global_store_dword %[addr], v213, off offset:3328\n"
        "global_store_dword %[addr], v214, off offset:3584\n"
        "global_store_dword %[addr], v215, off offset:3840\n"
        "v_mov_b32_dpp v200, v200 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v201, v201 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v202, v202 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v203, v203 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v204, v204 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v205, v205 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v206, v206 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v207, v207 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v208, v208 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v209, v209 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v210, v210 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v211, v211 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v212, v212 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v213, v213 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v214, v214 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v215, v215 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_waitcnt vmcnt(0)\n"
        :: [tag] "v"(tag), [addr] "v"(mine), [zero] "v"(zero) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "memory");
        else if (MODE == 1) __asm__ volatile(
        "v_xor_b32 v200, 200, %[tag]\n"
        "v_xor_b32 v201, 201, %[tag]\n"
        "v_xor_b32 v202, 202, %[tag]\n"
        "v_xor_b32 v203, 203, %[tag]\n"
        "v_xor_b32 v204, 204, %[tag]\n"
        "v_xor_b32 v205, 205, %[tag]\n"
        "v_xor_b32 v206, 206, %[tag]\n"
        "v_xor_b32 v207, 207, %[tag]\n"
        "v_xor_b32 v208, 208, %[tag]\n"
        "v_xor_b32 v209, 209, %[tag]\n"
        "v_xor_b32 v210, 210, %[tag]\n"
        "v_xor_b32 v211, 211, %[tag]\n"
        "v_xor_b32 v212, 212, %[tag]\n"
        "v_xor_b32 v213, 213, %[tag]\n"
        "v_xor_b32 v214, 214, %[tag]\n"
        "v_xor_b32 v215, 215, %[tag]\n"
        "global_store_dword %[addr], v200, off offset:0\n"
        "global_store_dword %[addr], v201, off offset:256\n"
        "global_store_dword %[addr], v202, off offset:512\n"
        "global_store_dword %[addr], v203, off offset:768\n"
        "global_store_dword %[addr], v204, off offset:1024\n"
        "global_store_dword %[addr], v205, off offset:1280\n"
        "global_store_dword %[addr], v206, off offset:1536\n"
        "global_store_dword %[addr], v207, off offset:1792\n"
        "global_store_dword %[addr], v208, off offset:2048\n"
        "global_store_dword %[addr], v209, off offset:2304\n"
        "global_store_dword %[addr], v210, off offset:2560\n"
        "global_store_dword %[addr], v211, off offset:2816\n"
        "global_store_dword %[addr], v212, off offset:3072\n"
        "global_store_dword %[addr], v213, off offset:3328\n"
        "global_store_dword %[addr], v214, off offset:3584\n"
        "global_store_dword %[addr], v215, off offset:3840\n"
        "v_mov_b32_dpp v200, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v201, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v202, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v203, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v204, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v205, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v206, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v207, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v208, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v209, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v210, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v211, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v212, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v213, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v214, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_mov_b32_dpp v215, %[zero] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_waitcnt vmcnt(0)\n"
        :: [tag] "v"(tag), [addr] "v"(mine), [zero] "v"(zero) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "memory");
        else __asm__ volatile(
        "v_xor_b32 v200, 200, %[tag]\n"
        "v_xor_b32 v201, 201, %[tag]\n"
        "v_xor_b32 v202, 202, %[tag]\n"
        "v_xor_b32 v203, 203, %[tag]\n"
        "v_xor_b32 v204, 204, %[tag]\n"
        "v_xor_b32 v205, 205, %[tag]\n"
        "v_xor_b32 v206, 206, %[tag]\n"
        "v_xor_b32 v207, 207, %[tag]\n"
        "v_xor_b32 v208, 208, %[tag]\n"
        "v_xor_b32 v209, 209, %[tag]\n"
        "v_xor_b32 v210, 210, %[tag]\n"
        "v_xor_b32 v211, 211, %[tag]\n"
        "v_xor_b32 v212, 212, %[tag]\n"
        "v_xor_b32 v213, 213, %[tag]\n"
        "v_xor_b32 v214, 214, %[tag]\n"
        "v_xor_b32 v215, 215, %[tag]\n"
        "global_store_dword %[addr], v200, off offset:0\n"
        "global_store_dword %[addr], v201, off offset:256\n"
        "global_store_dword %[addr], v202, off offset:512\n"
        "global_store_dword %[addr], v203, off offset:768\n"
        "global_store_dword %[addr], v204, off offset:1024\n"
        "global_store_dword %[addr], v205, off offset:1280\n"
        "global_store_dword %[addr], v206, off offset:1536\n"
        "global_store_dword %[addr], v207, off offset:1792\n"
        "global_store_dword %[addr], v208, off offset:2048\n"
        "global_store_dword %[addr], v209, off offset:2304\n"
        "global_store_dword %[addr], v210, off offset:2560\n"
        "global_store_dword %[addr], v211, off offset:2816\n"
        "global_store_dword %[addr], v212, off offset:3072\n"
        "global_store_dword %[addr], v213, off offset:3328\n"
        "global_store_dword %[addr], v214, off offset:3584\n"
        "global_store_dword %[addr], v215, off offset:3840\n"
        "v_mov_b32 v200, 0\n"
        "v_mov_b32 v201, 0\n"
        "v_mov_b32 v202, 0\n"
        "v_mov_b32 v203, 0\n"
        "v_mov_b32 v204, 0\n"
        "v_mov_b32 v205, 0\n"
        "v_mov_b32 v206, 0\n"
        "v_mov_b32 v207, 0\n"
        "v_mov_b32 v208, 0\n"
        "v_mov_b32 v209, 0\n"
        "v_mov_b32 v210, 0\n"
        "v_mov_b32 v211, 0\n"
        "v_mov_b32 v212, 0\n"
        "v_mov_b32 v213, 0\n"
        "v_mov_b32 v214, 0\n"
        "v_mov_b32 v215, 0\n"
        "s_waitcnt vmcnt(0)\n"
        :: [tag] "v"(tag), [addr] "v"(mine), [zero] "v"(zero) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "memory");
        for (int k = 0; k < 16; k++) nbad += __builtin_nontemporal_load(mine + k * 64) != ((200u + k) ^ tag);
    }
    if (smem[threadIdx.x] == 77 && rounds < 0) nbad++;
    if (nbad) atomicAdd(bad, nbad);
}
int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 8;
    const size_t lds = (size_t)160 * 1024 / per_cu - 512;
    const int nblk = 256 * per_cu;
    unsigned *bad, *buf;
    (void)hipMalloc(&bad, 12); (void)hipMemset(bad, 0, 12);
    (void)hipMalloc(&buf, (size_t)nblk * 16 * 64 * 4);
    (void)hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe<0>, dim3(nblk), dim3(64), lds, 0, buf, bad, 2000);
    hipLaunchKernelGGL(probe<1>, dim3(nblk), dim3(64), lds, 0, buf, bad + 1, 2000);
    hipLaunchKernelGGL(probe<2>, dim3(nblk), dim3(64), lds, 0, buf, bad + 2, 2000);
    hipError_t e = hipDeviceSynchronize();
    unsigned h[3] = {0, 0, 0}; (void)hipMemcpy(h, bad, 12, hipMemcpyDeviceToHost);
    printf("per_cu %d: %s; stored words that differ after overwrite by: in-place DPP %u, DPP from another register %u, plain v_mov %u\n", per_cu, hipGetErrorString(e), h[0], h[1], h[2]);
    return 0;
}
