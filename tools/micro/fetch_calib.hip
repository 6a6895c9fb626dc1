// Calibration of rocprofv3's FETCH_SIZE for THIS repository's access pattern (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  The fused evaluation reads its plane table with 8-byte loads, 64 consecutive
// doubles (512 B) per wave-instruction, consecutive instructions of a wave `stride` bytes apart (planes[b][c][p][q], q fastest).
//   read8   that pattern: every double of a 2 GiB buffer (>> the 256 MiB Infinity Cache) read exactly once
//   read16  the guide's reference pattern: 16 B per lane, fully streaming
// Run each under `rocprofv3 --pmc FETCH_SIZE` (tools/gpu_fetch_calib.sh) and divide the known bytes by FETCH_SIZE * 1024.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void read8(const double* __restrict__ p, size_t rows, size_t planes, double* out) {
    // block = 64-row tile x 4 waves; wave w reads planes w, w + 4, ... of its tile's rows (what collision_block does)
    const size_t q = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    double acc = 0.0;
    if (q < rows)
        for (size_t pl = w; pl < planes; pl += 4) acc += p[pl * rows + q];
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(256) void read16(const double2* __restrict__ p, size_t n2, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 8;
    const size_t rows = 35000ull * 128, planes = 60;   // configs[2]: Q = 35 000 rows x 128 problems, 60 doubles read per row
    const size_t n = rows * planes;                    // 268.8 M doubles = 2.15 GB
    double *d, *out;
    if (hipMalloc(&d, n * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, n * 8);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (which == 8) hipLaunchKernelGGL(read8, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, 0, d, rows, planes, out);
        else hipLaunchKernelGGL(read16, dim3(256 * 16), dim3(256), 0, 0, (const double2*)d, n / 2, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("read%d: %.0f bytes in %.3f ms = %.2f TB/s\n", which, (double)n * 8, ms, n * 8 / ms / 1e9);
    }
    return 0;
}
