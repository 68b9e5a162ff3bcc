// Microbenchmark: dependent-issue latency and throughput of v_fma_f64 / v_add_u32 on gfx950.
// hipcc --offload-arch=gfx950 -O3 fma_latency.hip -o fma_latency && ./fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void fma_chain(double* out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
static void run(int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fma_chain<ILP><<<blocks, 64>>>(d, 100, 0.999, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_chain<ILP><<<blocks, 64>>>(d, iters, 0.999, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = (double)iters * 16 * ILP;  // per wave
    const double cyc = ms * 1e-3 * 2.4e9;            // at the 2.4 GHz peak clock
    printf("ILP %d  waves/SIMD %d : %.2f cycles per wave-instruction per wave, %.2f cycles per instruction per SIMD, %.1f TFLOP/s\n", ILP,
           waves_per_simd, cyc / n_inst, cyc / (n_inst * waves_per_simd),
           2.0 * 64 * n_inst * blocks / (ms * 1e-3) * 1e-12);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<1>(w);
        run<2>(w);
        run<4>(w);
    }
    return 0;
}
