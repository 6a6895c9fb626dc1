// Micro-benchmark: HBM write bandwidth of the half-space table's store pattern (armour_p1_planes_kernel) against a linear fill.
//   pattern 0: block x of problem y writes 180 segments of `rows` x 8 B, segment s at offset (s * Q + x * rows) doubles (the table's layout)
//   pattern 1: the same bytes, every block one contiguous range
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/write_pattern.hip -o tools/micro/write_pattern ; run: ./write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void fill(double* out, int Q, int pattern, int rows_per_block) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* base = out + (size_t)blockIdx.y * 180 * Q;
    for (int s = wv; s < 180; s += 4) {
        for (int r = lane; r < rows_per_block; r += 64) {
            const size_t q = (size_t)blockIdx.x * rows_per_block + r;
            if (q >= (size_t)Q) continue;
            const size_t off = pattern == 0 ? (size_t)s * Q + q : ((size_t)blockIdx.x * 180 + s) * rows_per_block + r;
            base[off] = (double)(s + r);
        }
    }
}
int main() {
    const int B = 128, Q = 7 * 100 * 20;
    double* d;
    hipMalloc(&d, (size_t)B * 180 * Q * 8 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rows : {64, 256, 1024})
        for (int pattern : {0, 1}) {
            dim3 grid((Q + rows - 1) / rows, B);
            for (int it = 0; it < 3; it++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(fill, grid, dim3(256), 0, 0, d, Q, pattern, rows);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("rows/block %4d pattern %d: %.3f ms, %.2f TB/s\n", rows, pattern, ms, (double)B * 180 * Q * 8 / ms * 1e-9);
        }
    return 0;
}
