// Microbenchmark: relative error of v_rcp_f64 / v_rsq_f64 estimates after 0, 1, 2 Newton steps (gfx950).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    out[i] = r;
    r = fma(r, fma(-v, r, 1.0), r);
    out[n + i] = r;
    r = fma(r, fma(-v, r, 1.0), r);
    out[2 * n + i] = r;
    double y = __builtin_amdgcn_rsq(v);
    out[3 * n + i] = v * y;
    y = y * fma(-0.5 * v * y, y, 1.5);
    double s = v * y;
    out[4 * n + i] = s;
    s = fma(fma(-s, s, v), 0.5 * y, s);
    out[5 * n + i] = s;
    s = fma(fma(-s, s, v), 0.5 * y, s);
    out[6 * n + i] = s;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), o(7 * (size_t)n);
    for (int i = 0; i < n; ++i) h[i] = std::ldexp(1.0 + (i + 0.5) / n, (i % 61) - 30);
    double *dx, *dout;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dout, 7 * (size_t)n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 7 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* names[7] = {"rcp", "rcp+1N", "rcp+2N", "x*rsq", "rsq+1N", "rsq+1N+1C", "rsq+1N+2C"};
    for (int q = 0; q < 7; ++q) {
        double e = 0;
        for (int i = 0; i < n; ++i) {
            const long double want = q < 3 ? 1.0L / h[i] : sqrtl((long double)h[i]);
            e = std::fmax(e, (double)fabsl((o[(size_t)q * n + i] - want) / want));
        }
        printf("%-10s max rel err %.3e\n", names[q], e);
    }
    return 0;
}
