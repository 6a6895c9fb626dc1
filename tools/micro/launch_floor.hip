// Development micro-benchmark: back-to-back launch period of an (almost) empty kernel as a function of grid size, block
// size, kernarg bytes and dynamic LDS.  hipcc --offload-arch=gfx950 -O2 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

struct Big { double a[48]; };  // 384 B of kernarg, about what the P2 kernel takes

__global__ void k_small(double* out) { if (out && threadIdx.x == 999999) out[0] = 1.0; }
__global__ void k_big(Big b, double* out) { if (out && threadIdx.x == 999999) out[0] = b.a[5]; }
__global__ void k_touch(Big b, const double* in, double* out) {
    extern __shared__ double sm[];
    // every block reads 8 B from a fresh address and writes one value: the minimum a real kernel does
    const double v = in[blockIdx.x * 32];
    if (threadIdx.x == 0) out[blockIdx.x] = v + b.a[3];
}

int main() {
    double *d_in, *d_out;
    hipMalloc(&d_in, 1 << 22); hipMalloc(&d_out, 1 << 22);
    hipMemset(d_in, 0, 1 << 22);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Big b{};
    const int grids[] = {1, 308, 1024};
    const int blocks[] = {256};
    for (int mode = 0; mode < 4; mode++)
        for (int blk : blocks)
            for (int g : grids) {
                const int reps = 2000;
                auto go = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k_small, dim3(g), dim3(blk), 0, s, d_out);
                    else if (mode == 1) hipLaunchKernelGGL(k_big, dim3(g), dim3(blk), 0, s, b, d_out);
                    else if (mode == 2) hipLaunchKernelGGL(k_touch, dim3(g), dim3(blk), 0, s, b, d_in, d_out);
                    else hipLaunchKernelGGL(k_touch, dim3(g), dim3(blk), 40000, s, b, d_in, d_out);
                };
                for (int i = 0; i < 200; i++) go();
                hipStreamSynchronize(s);
                hipEventRecord(e0, s);
                const auto h0 = std::chrono::steady_clock::now();
                for (int i = 0; i < reps; i++) go();
                const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / reps;
                hipEventRecord(e1, s);
                hipStreamSynchronize(s);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // the same launches as one graph
                hipGraph_t graph; hipGraphExec_t exec;
                hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
                for (int i = 0; i < 500; i++) go();
                hipStreamEndCapture(s, &graph);
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                hipGraphLaunch(exec, s); hipStreamSynchronize(s);
                hipEventRecord(e0, s);
                for (int i = 0; i < 4; i++) hipGraphLaunch(exec, s);
                hipEventRecord(e1, s);
                hipStreamSynchronize(s);
                float gms; hipEventElapsedTime(&gms, e0, e1);
                hipGraphExecDestroy(exec); hipGraphDestroy(graph);
                printf("mode %d (%s) block %3d grid %4d: %.2f us/launch (host enqueue %.2f us), in a graph %.2f us/node\n", mode, mode == 0 ? "8 B kernarg" : mode == 1 ? "392 B kernarg" : mode == 2 ? "392 B + load/store" : "392 B + load/store + 40 KB LDS", blk, g, ms * 1e3 / reps, host_us, gms * 1e3 / 2000);
            }
    return 0;
}
