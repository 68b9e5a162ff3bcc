// Microbenchmark: what a wave64 `ds_add_f64` costs the CU's LDS pipe as a function of how its lanes' addresses collide -- the
// row-per-lane flux kernels (vag_grid_rows.h, vag_fit_rows.h) add every interpolated value to a per-wavefront sum in LDS this way.
// Chip full (3 workgroups of 256 lanes per CU); cycles per instruction and CU = elapsed * clock / (instructions per CU).
//   hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic && ./lds_atomic
// PATTERN: 0 every lane its own double (consecutive)      1 all 64 lanes one address      2 two addresses (lane & 1)
//          3 four addresses (lane & 3)   4 sixteen addresses (lane & 15)   5 sixteen consecutive doubles, four lanes each (lane >> 2)
//          6 every lane its own double, stride 2 (16-byte apart)           7 ds_write_b64 of the same values (no read-modify-write)
//          8 lanes in pairs on one address, pairs consecutive (lane >> 1)  9 ds_add_f32 every lane its own float
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PATTERN>
__global__ void __launch_bounds__(256) kern(double* out, int iters) {
    __shared__ double lds[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 1024; i += 256) lds[i] = 0;
    __syncthreads();
    double* base = lds + wave * 1024;
    int idx = PATTERN == 0 ? lane : PATTERN == 1 ? 0 : PATTERN == 2 ? (lane & 1) : PATTERN == 3 ? (lane & 3) : PATTERN == 4 ? (lane & 15)
              : PATTERN == 5 ? (lane >> 2) : PATTERN == 6 ? 2 * lane : PATTERN == 7 ? lane : PATTERN == 8 ? (lane >> 1) : lane;
    const unsigned addr = (unsigned)(size_t)(base + idx) & 0xffff;
    const double v = 1.0 + lane;
    const float vf = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (PATTERN == 7)
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(0) : "memory");
            else if constexpr (PATTERN == 9)
                asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(vf) : "memory");
            else
                asm volatile("ds_add_f64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (iters < 0) out[threadIdx.x] = lds[threadIdx.x];
}

template <int PATTERN>
static void run(const char* what, int cus, double mhz) {
    double* out;
    hipMalloc(&out, 8 * 256);
    const int iters = 2000, blocks = cus * 3;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<PATTERN>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<PATTERN>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 3.0 * 4 * 16.0 * iters;  // workgroups * wavefronts * unroll * iterations
    printf("%-62s %7.1f LDS cycles per wave64 instruction and CU (%.3f ms)\n", what, ms * 1e-3 * mhz * 1e6 / instr_per_cu, ms);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
    run<0>("ds_add_f64, 64 consecutive doubles", p.multiProcessorCount, mhz);
    run<6>("ds_add_f64, 64 doubles 16 B apart", p.multiProcessorCount, mhz);
    run<8>("ds_add_f64, 32 consecutive doubles, two lanes each", p.multiProcessorCount, mhz);
    run<5>("ds_add_f64, 16 consecutive doubles, four lanes each", p.multiProcessorCount, mhz);
    run<4>("ds_add_f64, 16 doubles, lane & 15", p.multiProcessorCount, mhz);
    run<3>("ds_add_f64, 4 doubles, lane & 3", p.multiProcessorCount, mhz);
    run<2>("ds_add_f64, 2 doubles, lane & 1", p.multiProcessorCount, mhz);
    run<1>("ds_add_f64, one double for all 64 lanes", p.multiProcessorCount, mhz);
    run<7>("ds_write_b64, 64 consecutive doubles", p.multiProcessorCount, mhz);
    run<9>("ds_add_f32, 64 consecutive floats", p.multiProcessorCount, mhz);
    return 0;
}
