// What a CU's memory pipeline gives the reach-set walks (pz_tv.h): waves that, batch after batch, issue U "terms" of three consecutive
// 512-byte rows (64 lanes x 8 B) at wave-uniform pseudo-random positions of a private table, wait for all of them, and then spend
// `work` dependent multiply-adds per term -- the walk's access pattern without its arithmetic.  One block per CU, W waves per block,
// table of R rows per block (R x 512 B).  Prints cycles per term and wave, and the rows/cycle a CU sustains.
//   row_gather <W> <U> <R> <work> [batches] [active lanes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U>
__global__ __launch_bounds__(512) void gather(const double* __restrict__ table, size_t rows_per_block, int batches, int work, double* out, long long* cycles, int act) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // act < 64: lanes >= act read lane act - 1's element (a group of fewer than 64 time steps: no byte of the row beyond 8 * act is touched)
    const double* base = table + (size_t)blockIdx.x * rows_per_block * 64 + min(lane, act - 1);
    const unsigned R = (unsigned)rows_per_block - 3;
    unsigned state = __builtin_amdgcn_readfirstlane(1234567u + 7919u * blockIdx.x + 104729u * wid);
    double acc = 0.0;
    const long long t0 = clock64();
    for (int b = 0; b < batches; b++) {
        double x[U][3];
#pragma unroll
        for (int u = 0; u < U; u++) {
            state = state * 1664525u + 1013904223u;
            const unsigned row = (unsigned)(((unsigned long long)state * R) >> 32);   // wave-uniform
            const double* p = base + (size_t)row * 64;
#pragma unroll
            for (int e = 0; e < 3; e++) x[u][e] = p[e * 64];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double v = x[u][0] + x[u][1] * x[u][2];
            for (int k = 0; k < work; k++) v = v * 1.0000001 + 0.5;   // dependent chain: `work` x 2 instructions
            acc += v;
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 8 + wid] = t1 - t0;
    if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 4, U = argc > 2 ? atoi(argv[2]) : 16;
    const size_t R = argc > 3 ? (size_t)atoll(argv[3]) : 20000;
    const int work = argc > 4 ? atoi(argv[4]) : 0, batches = argc > 5 ? atoi(argv[5]) : 400, act = argc > 6 ? atoi(argv[6]) : 64;
    const int blocks = 256;
    double *d, *out; long long* cyc;
    if (hipMalloc(&d, blocks * R * 512) != hipSuccess || hipMalloc(&out, 8) != hipSuccess || hipMalloc(&cyc, blocks * 8 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, blocks * R * 512);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (U == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(64 * W), 0, 0, d, R, batches, work, out, cyc, act);
        else if (U == 8) hipLaunchKernelGGL(gather<8>, dim3(blocks), dim3(64 * W), 0, 0, d, R, batches, work, out, cyc, act);
        else if (U == 16) hipLaunchKernelGGL(gather<16>, dim3(blocks), dim3(64 * W), 0, 0, d, R, batches, work, out, cyc, act);
        else hipLaunchKernelGGL(gather<32>, dim3(blocks), dim3(64 * W), 0, 0, d, R, batches, work, out, cyc, act);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long h[blocks * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int b = 0; b < blocks; b++) for (int w = 0; w < W; w++) sum += (double)h[b * 8 + w];
    const double per_wave = sum / (blocks * W), terms = (double)batches * U;
    printf("W=%d U=%2d R=%6zu (%.1f MB/block) work=%3d lanes=%2d: %.3f ms, %.0f cycles/term/wave, %.4f rows/cycle/CU, %.2f TB/s\n", W, U, R, R * 512 / 1048576.0, work, act, ms, per_wave / terms,
           3.0 * terms * W / per_wave, 3.0 * terms * W * blocks * 512 / ms / 1e9);
    return 0;
}
