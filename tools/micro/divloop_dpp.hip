// Probe for the round-2 "two waves per SIMD" failure of the per-step reach-set kernel, narrowed in round 3 to the reduce / emit loop of
// pz_wave.h running under DIVERGENT loop control (its bound reached it as a per-lane value read through a by-reference argument).
// This kernel rebuilds that loop shape in isolation: a non-inlined function receives its loop bound by reference (a flat load from the
// caller's stack frame: a per-lane value to the compiler), walks chunks of 64 sorted keys held in LDS, and sums every run of equal keys
// into its head lane by shifting the terms down the wave with `v_mov_b32_dpp wave_shl:1` under `__ballot` control -- exactly
// sort_reduce_emit's inner loop.  Every block does the same work; results are compared with a reference computed without DPP and
// without the divergent bound.  Launched with many one-wave blocks per CU so that waves share SIMDs.
//   divloop_dpp [blocks_per_cu] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define NOINL __attribute__((noinline))
__device__ inline double dpp_shl1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
struct Args { int N; int pad[7]; const uint64_t* keys; const double* vals; };

template <bool DIVERGENT>
__device__ NOINL double reduce_runs(const Args& a_, const uint64_t* lkeys, int lane) {
    int N;
    if (DIVERGENT) N = a_.N;                                           // per-lane value (flat load through the reference)
    else N = __builtin_amdgcn_readfirstlane(a_.N);                     // scalar
    double total = 0.0;
    for (int base = 0; base < N; base += 64) {
        const int p = base + lane;
        bool head = false;
        uint64_t key = 0;
        double acc[3] = {0.0, 0.0, 0.0};
        if (p < N) {
            key = lkeys[p];
            head = (p == 0) || lkeys[p - 1] != key;
            for (int e = 0; e < 3; e++) acc[e] = a_.vals[3 * p + e];
        }
        double sh[3] = {acc[0], acc[1], acc[2]};
        for (int j = 1;; j++) {
            const int q = p + j;
            const bool more = head && q < N && lkeys[q] == key;
            if (__ballot(more) == 0ull) break;
            for (int e = 0; e < 3; e++) sh[e] = dpp_shl1(sh[e]);
            if (more) {
                double c[3] = {sh[0], sh[1], sh[2]};
                if (lane + j >= 64) for (int e = 0; e < 3; e++) c[e] = a_.vals[3 * q + e];
                for (int e = 0; e < 3; e++) acc[e] += c[e];
            }
        }
        if (head) total += acc[0] * 1.0 + acc[1] * 3.0 + acc[2] * 7.0 + (double)(key & 1023);
    }
    return total;
}

template <bool DIVERGENT>
__global__ __launch_bounds__(64) void probe(const uint64_t* keys, const double* vals, int N, int reps, double* out) {
    extern __shared__ uint64_t lkeys[];
    const int lane = threadIdx.x;
    Args a;
    a.N = N; a.keys = keys; a.vals = vals;
    double s = 0.0;
    for (int r = 0; r < reps; r++) {
        for (int p = lane; p < N; p += 64) lkeys[p] = keys[p];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        s += reduce_runs<DIVERGENT>(a, lkeys, lane);
        __builtin_amdgcn_wave_barrier();
    }
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 200;
    const int N = 1500, blocks = 256 * per_cu;
    uint64_t* hk = new uint64_t[N]; double* hv = new double[3 * N];
    uint64_t k = 5; unsigned st = 12345;
    for (int p = 0; p < N; p++) { st = st * 1664525u + 1013904223u; if ((st >> 28) < 11) k += 1 + (st >> 20 & 7); hk[p] = k; for (int e = 0; e < 3; e++) { st = st * 1664525u + 1013904223u; hv[3 * p + e] = (double)(st >> 8) / 16777216.0 - 0.5; } }
    uint64_t* dk; double *dv, *dout;
    hipMalloc(&dk, N * 8); hipMalloc(&dv, 3 * N * 8); hipMalloc(&dout, (size_t)blocks * 64 * 8);
    hipMemcpy(dk, hk, N * 8, hipMemcpyHostToDevice); hipMemcpy(dv, hv, 3 * N * 8, hipMemcpyHostToDevice);
    double* h = new double[(size_t)blocks * 64];
    double ref[64];
    for (int pass = 0; pass < 2; pass++) {
        for (int rep = 0; rep < 4; rep++) {
            hipMemset(dout, 0, (size_t)blocks * 64 * 8);
            if (pass == 0) hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(64), N * 8, 0, dk, dv, N, reps, dout);
            else hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(64), N * 8, 0, dk, dv, N, reps, dout);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(h, dout, (size_t)blocks * 64 * 8, hipMemcpyDeviceToHost);
            if (pass == 0 && rep == 0) for (int l = 0; l < 64; l++) ref[l] = h[l];
            long bad = 0;
            for (int b = 0; b < blocks; b++) for (int l = 0; l < 64; l++) bad += h[(size_t)b * 64 + l] != ref[l];
            printf("%s loop bound, %d one-wave blocks per CU, launch %d: %ld of %d lane results differ from the reference\n", pass ? "per-lane (divergent)" : "scalar", per_cu, rep, bad, blocks * 64);
        }
    }
    return 0;
}
