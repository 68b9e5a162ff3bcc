// Microbenchmark: THROUGHPUT cost of the VALU / LDS instruction kinds the flux kernels are made of, with the chip full (4 wavefronts
// per SIMD, every CU busy): SIMD cycles per wave64 instruction = elapsed * clock * SIMDs / instructions.  The roofline of a VALU-bound
// kernel is sum over its instruction mix of these costs, not 4 cycles per instruction.
// hipcc --offload-arch=gfx950 -O3 valu_throughput.hip -o valu_throughput && ./valu_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) kern(double* out, int iters, double a, double b, int ia) {
    // eight independent register sets, so that nothing waits on a dependency
    double x0 = threadIdx.x * 1e-9 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    __shared__ double lds[2048];
    lds[threadIdx.x] = x0;
    __syncthreads();
    unsigned la = (threadIdx.x * 16) & 8191;
    const unsigned la8 = threadIdx.x * 8, la_dup = (threadIdx.x >> 3) * 8;  // contiguous words; eight lanes per word
    for (int it = 0; it < iters; ++it) {
#define OP8(INS) \
    { REP8(asm volatile(INS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), \
                      "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a), "v"(b), "v"(ia), "v"(la) : "vcc", "memory");) }
        if constexpr (KIND == 0) OP8("v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17")
        if constexpr (KIND == 1) OP8("v_add_f64 %0, %0, %17\n v_add_f64 %1, %1, %17\n v_add_f64 %2, %2, %17\n v_add_f64 %3, %3, %17\n v_add_f64 %4, %4, %17\n v_add_f64 %5, %5, %17\n v_add_f64 %6, %6, %17\n v_add_f64 %7, %7, %17")
        if constexpr (KIND == 2) OP8("v_mul_f64 %0, %0, %16\n v_mul_f64 %1, %1, %16\n v_mul_f64 %2, %2, %16\n v_mul_f64 %3, %3, %16\n v_mul_f64 %4, %4, %16\n v_mul_f64 %5, %5, %16\n v_mul_f64 %6, %6, %16\n v_mul_f64 %7, %7, %16")
        if constexpr (KIND == 3) OP8("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7")
        if constexpr (KIND == 4) OP8("v_ldexp_f64 %0, %0, %18\n v_ldexp_f64 %1, %1, %18\n v_ldexp_f64 %2, %2, %18\n v_ldexp_f64 %3, %3, %18\n v_ldexp_f64 %4, %4, %18\n v_ldexp_f64 %5, %5, %18\n v_ldexp_f64 %6, %6, %18\n v_ldexp_f64 %7, %7, %18")
        if constexpr (KIND == 5) OP8("v_cvt_i32_f64 %8, %0\n v_cvt_i32_f64 %9, %1\n v_cvt_i32_f64 %10, %2\n v_cvt_i32_f64 %11, %3\n v_cvt_i32_f64 %12, %4\n v_cvt_i32_f64 %13, %5\n v_cvt_i32_f64 %14, %6\n v_cvt_i32_f64 %15, %7")
        if constexpr (KIND == 6) OP8("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n v_cvt_f64_i32 %4, %12\n v_cvt_f64_i32 %5, %13\n v_cvt_f64_i32 %6, %14\n v_cvt_f64_i32 %7, %15")
        if constexpr (KIND == 7) OP8("v_cmp_gt_f64 vcc, %0, %16\n v_cmp_gt_f64 vcc, %1, %16\n v_cmp_gt_f64 vcc, %2, %16\n v_cmp_gt_f64 vcc, %3, %16\n v_cmp_gt_f64 vcc, %4, %16\n v_cmp_gt_f64 vcc, %5, %16\n v_cmp_gt_f64 vcc, %6, %16\n v_cmp_gt_f64 vcc, %7, %16")
        if constexpr (KIND == 8) OP8("v_cndmask_b32 %8, %8, %18, vcc\n v_cndmask_b32 %9, %9, %18, vcc\n v_cndmask_b32 %10, %10, %18, vcc\n v_cndmask_b32 %11, %11, %18, vcc\n v_cndmask_b32 %12, %12, %18, vcc\n v_cndmask_b32 %13, %13, %18, vcc\n v_cndmask_b32 %14, %14, %18, vcc\n v_cndmask_b32 %15, %15, %18, vcc")
        if constexpr (KIND == 9) OP8("v_max_f64 %0, %0, %16\n v_max_f64 %1, %1, %16\n v_max_f64 %2, %2, %16\n v_max_f64 %3, %3, %16\n v_max_f64 %4, %4, %16\n v_max_f64 %5, %5, %16\n v_max_f64 %6, %6, %16\n v_max_f64 %7, %7, %16")
        if constexpr (KIND == 10) OP8("v_add_u32 %8, %8, %18\n v_add_u32 %9, %9, %18\n v_add_u32 %10, %10, %18\n v_add_u32 %11, %11, %18\n v_add_u32 %12, %12, %18\n v_add_u32 %13, %13, %18\n v_add_u32 %14, %14, %18\n v_add_u32 %15, %15, %18")
        if constexpr (KIND == 11) OP8("v_lshl_add_u32 %8, %8, 3, %18\n v_lshl_add_u32 %9, %9, 3, %18\n v_lshl_add_u32 %10, %10, 3, %18\n v_lshl_add_u32 %11, %11, 3, %18\n v_lshl_add_u32 %12, %12, 3, %18\n v_lshl_add_u32 %13, %13, 3, %18\n v_lshl_add_u32 %14, %14, 3, %18\n v_lshl_add_u32 %15, %15, 3, %18")
        if constexpr (KIND == 12) OP8("v_mad_u32_u24 %8, %8, %18, %18\n v_mad_u32_u24 %9, %9, %18, %18\n v_mad_u32_u24 %10, %10, %18, %18\n v_mad_u32_u24 %11, %11, %18, %18\n v_mad_u32_u24 %12, %12, %18, %18\n v_mad_u32_u24 %13, %13, %18, %18\n v_mad_u32_u24 %14, %14, %18, %18\n v_mad_u32_u24 %15, %15, %18, %18")
        if constexpr (KIND == 13) OP8("v_mov_b64 %0, %16\n v_mov_b64 %1, %16\n v_mov_b64 %2, %16\n v_mov_b64 %3, %16\n v_mov_b64 %4, %16\n v_mov_b64 %5, %16\n v_mov_b64 %6, %16\n v_mov_b64 %7, %16")
        if constexpr (KIND == 14) OP8("v_cmp_class_f64 vcc, %0, %18\n v_cmp_class_f64 vcc, %1, %18\n v_cmp_class_f64 vcc, %2, %18\n v_cmp_class_f64 vcc, %3, %18\n v_cmp_class_f64 vcc, %4, %18\n v_cmp_class_f64 vcc, %5, %18\n v_cmp_class_f64 vcc, %6, %18\n v_cmp_class_f64 vcc, %7, %18")
        if constexpr (KIND == 15) OP8("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7")
        if constexpr (KIND == 16) OP8("v_fma_f32 %8, %8, %18, %18\n v_fma_f32 %9, %9, %18, %18\n v_fma_f32 %10, %10, %18, %18\n v_fma_f32 %11, %11, %18, %18\n v_fma_f32 %12, %12, %18, %18\n v_fma_f32 %13, %13, %18, %18\n v_fma_f32 %14, %14, %18, %18\n v_fma_f32 %15, %15, %18, %18")
        if constexpr (KIND == 17) OP8("v_min_u32 %8, %8, %18\n v_min_u32 %9, %9, %18\n v_min_u32 %10, %10, %18\n v_min_u32 %11, %11, %18\n v_min_u32 %12, %12, %18\n v_min_u32 %13, %13, %18\n v_min_u32 %14, %14, %18\n v_min_u32 %15, %15, %18")
        if constexpr (KIND == 18) OP8("v_fma_f64 %0, |%0|, %16, %17\n v_fma_f64 %1, |%1|, %16, %17\n v_fma_f64 %2, -%2, %16, %17\n v_fma_f64 %3, %3, %16, -%17\n v_fma_f64 %4, |%4|, %16, %17\n v_fma_f64 %5, %5, %16, %17\n v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17")
        if constexpr (KIND == 21) {  // select with the mask in an SGPR pair instead of VCC
            unsigned long long msk = 0x5555555555555555ull + ia;
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia), "s"(msk));)
        }
        if constexpr (KIND == 22) {  // compare into VCC, then the two halves of a 64-bit select (what `c ? x : y` on doubles compiles to)
            REP8(asm volatile("v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f64 vcc, %1, %8\n v_cndmask_b32 %6, %6, %9, vcc\n"
                              "v_cndmask_b32 %7, %7, %9, vcc\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a), "v"(ia) : "vcc");)
        }
        if constexpr (KIND == 23) {  // the same with the compare result in an SGPR pair
            REP8(asm volatile("v_cmp_gt_f64_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cmp_gt_f64_e64 s[22:23], %1, %8\n"
                              "v_cndmask_b32_e64 %6, %6, %9, s[22:23]\n v_cndmask_b32_e64 %7, %7, %9, s[22:23]\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a), "v"(ia) : "s20", "s21", "s22", "s23");)
        }
        if constexpr (KIND == 24) {  // a lane-divergent branch skeleton: compare, save exec, (one instruction inside), restore -- no jump taken
            REP8(asm volatile("v_cmp_gt_f64 vcc, %0, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f64 %1, %1, %4\n s_or_b64 exec, exec, s[20:21]\n"
                              "v_cmp_gt_f64 vcc, %2, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f64 %3, %3, %4\n s_or_b64 exec, exec, s[20:21]"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc", "s20", "s21");)
        }
        if constexpr (KIND == 25) {  // v_max_f64 x, x, x + v_max 0 (the compiler's max(z, 0) with NaN quieting)
            REP8(asm volatile("v_max_f64 %0, %0, %0\n v_max_f64 %0, %0, 0\n v_max_f64 %1, %1, %1\n v_max_f64 %1, %1, 0\n v_max_f64 %2, %2, %2\n v_max_f64 %2, %2, 0\n v_max_f64 %3, %3, %3\n v_max_f64 %3, %3, 0"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        }
        if constexpr (KIND == 26) {  // v_readfirstlane + s_cmp + s_cselect: scalarising a uniform decision
            REP8(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s21, %3\n"
                              "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s21, %3"
                              : : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "s20", "s21");)
        }
        if constexpr (KIND == 27) {  // v_fmac_f64 + v_mov_b64 (the compiler's Horner step) vs one v_fma_f64 with a scalar addend
            REP8(asm volatile("v_mov_b64 %4, %6\n v_fmac_f64 %4, %0, %7\n v_mov_b64 %5, %6\n v_fmac_f64 %5, %1, %7\n v_mov_b64 %4, %6\n v_fmac_f64 %4, %2, %7\n v_mov_b64 %5, %6\n v_fmac_f64 %5, %3, %7"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));)
        }
        if constexpr (KIND == 28) {
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "s"(b));)
        }
        if constexpr (KIND == 19) {  // ds_read_b128 throughput: 8 reads in flight, then wait
            REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                              : "=v"(*(double2*)&x0), "=v"(*(double2*)&x2), "=v"(*(double2*)&x4), "=v"(*(double2*)&x6) : "v"(la) : "memory");)
        }
        if constexpr (KIND == 20) {  // ds_read_b64
            REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                              : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(la) : "memory");)
        }
        if constexpr (KIND == 29) {  // ds_add_f64 without return, all lanes on different words
            REP8(asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %2 offset:2048\n ds_add_f64 %0, %3 offset:4096\n ds_add_f64 %0, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                              : : "v"(la8), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");)
        }
        if constexpr (KIND == 30) {  // ds_add_f64, half of the lanes masked off
            REP8(asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0x00000000ffffffff\n ds_add_f64 %0, %1\n ds_add_f64 %0, %2 offset:2048\n ds_add_f64 %0, %3 offset:4096\n ds_add_f64 %0, %4 offset:6144\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, s[20:21]"
                              : : "v"(la8), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory", "s20", "s21");)
        }
        if constexpr (KIND == 31) {  // ds_read2_b64 of two neighbouring words per lane (lanes overlap their neighbours' words)
            REP8(asm volatile("ds_read2_b64 %0, %4 offset1:1\n ds_read2_b64 %1, %4 offset0:2 offset1:3\n ds_read2_b64 %2, %4 offset0:4 offset1:5\n ds_read2_b64 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)"
                              : "=v"(*(double2*)&x0), "=v"(*(double2*)&x2), "=v"(*(double2*)&x4), "=v"(*(double2*)&x6) : "v"(la8) : "memory");)
        }
        if constexpr (KIND == 32) {  // ds_add_f64 with eight lanes per word (same-address serialisation)
            REP8(asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %2 offset:2048\n ds_add_f64 %0, %3 offset:4096\n ds_add_f64 %0, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                              : : "v"(la_dup), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
}

template <int KIND>
static void run(const char* name, int per_iter) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4;  // 4 blocks of 4 waves per CU = 4 waves per SIMD
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * blocks);
    kern<KIND><<<blocks, 256>>>(d, 10, 0.999, 1e-3, 3);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[2];
    const int iters[2] = {2000, 22000};  // the difference of two lengths: launch overhead and clock ramp drop out
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        kern<KIND><<<blocks, 256>>>(d, iters[r], 0.999, 1e-3, 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[r], e0, e1);
    }
    const double n_per_simd = (double)(iters[1] - iters[0]) * per_iter * 4;  // wave-instructions per SIMD (4 waves each)
    const double clk = p.clockRate * 1e3;                                    // Hz (nominal; the chip may run lower under load)
    printf("%-22s %7.3f / %7.3f ms  %6.2f SIMD cycles per wave64 instruction at the nominal %.0f MHz\n", name, ms[0], ms[1],
           (ms[1] - ms[0]) * 1e-3 * clk / n_per_simd, clk / 1e6);
    hipFree(d);
}

int main() {
    run<0>("v_fma_f64", 64);
    run<18>("v_fma_f64 |x| -x", 64);
    run<1>("v_add_f64", 64);
    run<2>("v_mul_f64", 64);
    run<9>("v_max_f64", 64);
    run<3>("v_rndne_f64", 64);
    run<4>("v_ldexp_f64", 64);
    run<5>("v_cvt_i32_f64", 64);
    run<6>("v_cvt_f64_i32", 64);
    run<7>("v_cmp_gt_f64", 64);
    run<14>("v_cmp_class_f64", 64);
    run<15>("v_rcp_f64", 64);
    run<13>("v_mov_b64", 64);
    run<8>("v_cndmask_b32", 64);
    run<10>("v_add_u32", 64);
    run<17>("v_min_u32", 64);
    run<11>("v_lshl_add_u32", 64);
    run<12>("v_mad_u32_u24", 64);
    run<16>("v_fma_f32", 64);
    run<21>("v_cndmask_b32 sgpr mask", 64);
    run<22>("cmp+2cndmask vcc (x2)+2add", 64);
    run<23>("cmp+2cndmask sgpr (x2)+2add", 64);
    run<24>("cmp+saveexec+add+restore", 64);
    run<25>("v_max_f64 pair", 64);
    run<26>("v_readfirstlane", 64);
    run<27>("v_mov_b64+v_fmac_f64", 64);
    run<28>("v_fma_f64 sgpr addend", 64);
    run<19>("ds_read_b128", 32);
    run<20>("ds_read_b64", 32);
    run<31>("ds_read2_b64 (q, q+1)", 32);
    run<29>("ds_add_f64", 32);
    run<30>("ds_add_f64 half exec", 32);
    run<32>("ds_add_f64 8 lanes/word", 32);
    return 0;
}
