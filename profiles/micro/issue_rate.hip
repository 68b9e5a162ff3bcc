// Microbenchmark: what ONE wavefront alone on its SIMD pays per FP64 instruction (gfx950), by s_memtime inside the kernel.
// A latency-bound kernel (one ODE row per lane, fewer wavefronts than SIMDs) lives in exactly this regime.
// hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, int KIND>
__global__ void chain(double* out, long long* cyc, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-9 + i + 1.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) x[i] = __builtin_fma(x[i], a, b);
                if (KIND == 1) x[i] = x[i] * a;
                if (KIND == 2) x[i] = x[i] + b;
                if (KIND == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
                if (KIND == 4) x[i] = __builtin_amdgcn_rsq(x[i]);
                if (KIND == 5) x[i] = __builtin_amdgcn_ldexp(x[i], (r & 1) ? 1 : -1);
                if (KIND == 6) x[i] = x[i] > 1.5 ? a : x[i] + b;  // cmp + cndmask pair + add
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP, int KIND>
static void run(const char* name, int lanes) {
    double* d;
    long long* c;
    hipMalloc(&d, sizeof(double) * 64);
    hipMalloc(&c, sizeof(long long));
    const int iters = 2000;
    chain<ILP, KIND><<<1, lanes>>>(d, c, 10, 0.999, 1e-3);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    chain<ILP, KIND><<<1, lanes>>>(d, c, iters, 0.999, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long cy;
    hipMemcpy(&cy, c, sizeof cy, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * ILP;
    printf("%-8s ILP %d lanes %2d: %.2f counter ticks per instruction; kernel %.3f ms -> %.2f ns per instruction, counter %.1f MHz\n", name, ILP,
           lanes, cy / n, ms, ms * 1e6 / n, cy / (ms * 1e3));
    hipFree(d);
    hipFree(c);
}

int main() {
    run<1, 0>("fma", 64);
    run<2, 0>("fma", 64);
    run<4, 0>("fma", 64);
    run<8, 0>("fma", 64);
    run<1, 0>("fma", 1);
    run<4, 0>("fma", 1);
    run<4, 0>("fma", 16);
    run<1, 1>("mul", 64);
    run<4, 1>("mul", 64);
    run<1, 2>("add", 64);
    run<4, 2>("add", 64);
    run<1, 3>("rcp", 64);
    run<4, 3>("rcp", 64);
    run<1, 4>("rsq", 64);
    run<4, 4>("rsq", 64);
    run<1, 5>("ldexp", 64);
    run<4, 5>("ldexp", 64);
    run<1, 6>("cmpsel", 64);
    run<4, 6>("cmpsel", 64);
    return 0;
}
