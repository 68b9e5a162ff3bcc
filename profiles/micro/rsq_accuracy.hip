// Microbenchmark: accuracy of v_rsq_f64 and of the short square-root forms built on it (gfx950).
// hipcc --offload-arch=gfx950 -O3 rsq_accuracy.hip -o rsq_accuracy && ./rsq_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* raw, double* s4, double* s9, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double y = __builtin_amdgcn_rsq(v);
    raw[i] = y;
    {   // 4 operations: s0 = x y, s1 = s0 + (y / 2)(x - s0^2)
        const double s0 = v * y;
        s4[i] = fma(fma(-s0, s0, v), 0.5 * y, s0);
    }
    {   // 9 operations: Newton on y first (vag_device.h: sqrt_fast)
        double yy = y * fma(-0.5 * v * y, y, 1.5);
        double s = v * yy;
        s9[i] = fma(fma(-s, s, v), 0.5 * yy, s);
    }
}

int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), raw(n), s4(n), s9(n);
    for (int i = 0; i < n; ++i) h[i] = std::exp2(-300.0 + 600.0 * (i + 0.37) / n) * (1.0 + (i % 997) / 997.0);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(raw.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(s4.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(s9.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e4 = 0, e9 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = sqrtl((long double)h[i]);
        e0 = std::fmax(e0, (double)fabsl(raw[i] * t - 1));
        e4 = std::fmax(e4, (double)fabsl(s4[i] / t - 1));
        e9 = std::fmax(e9, (double)fabsl(s9[i] / t - 1));
    }
    printf("v_rsq_f64 max rel err %.3e; sqrt in 4 ops %.3e; sqrt in 9 ops %.3e\n", e0, e4, e9);
    return 0;
}
