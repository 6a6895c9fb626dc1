// dev probe: do two waves that share a SIMD get disjoint scratch when the per-lane private segment is large?
// Each wave fills a private array (forced into scratch by a run-time index) with a wave-unique pattern, waits, verifies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef WORDS
#define WORDS 1280  // 5 KB per lane
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(unsigned* bad, int rounds, int spin, const int* perm) {
    extern __shared__ unsigned char smem[];
    volatile unsigned buf[WORDS];
    unsigned nbad = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned tag = (blockIdx.x * 64u + threadIdx.x) * 65536u + (unsigned)r * 2048u;
        for (int i = 0; i < WORDS; i++) buf[perm[i]] = tag + (unsigned)i;
        for (int s = 0; s < spin; s++) __builtin_amdgcn_s_sleep(10);
        for (int i = 0; i < WORDS; i++) nbad += buf[perm[i]] != tag + (unsigned)i;
    }
    if (smem[threadIdx.x] == 77 && rounds < 0) nbad++;
    if (nbad) atomicAdd(bad, nbad);
}
int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 3;
    const size_t lds = (size_t)160 * 1024 / per_cu - 512;
    unsigned* bad; int* perm;
    hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    int hp[WORDS]; for (int i = 0; i < WORDS; i++) hp[i] = (i * 7) % WORDS;
    hipMalloc(&perm, sizeof(hp)); hipMemcpy(perm, hp, sizeof(hp), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nblk = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, probe, 64, lds);
    hipLaunchKernelGGL(probe, dim3(256 * per_cu), dim3(64), lds, 0, bad, 20, 200, perm);
    hipError_t e = hipDeviceSynchronize();
    unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("per_cu %d (occupancy query %d), scratch %d B/lane: %s, mismatching words %u\n", per_cu, nblk, WORDS * 4, hipGetErrorString(e), h);
    return 0;
}
