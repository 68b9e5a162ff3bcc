// Microbenchmark: when does a store from a RUNNING kernel into pinned host memory become visible to the host (gfx950)?
// Kernel 1 publishes a flag and returns; kernel 2 (queued behind it) spins for ~2 ms.  Prints how long after the launch the host
// saw the flag, for coherent and default pinned allocations, and for a flag written from inside a long kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void publish(int* flag, int v, long long spin) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
    }
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {
    }
}
static void run(unsigned flags, const char* name, long long spin1) {
    int* h;
    hipHostMalloc(&h, 64, flags);
    *h = 0;
    int* d;
    hipHostGetDevicePointer((void**)&d, h, 0);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 1; rep <= 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        publish<<<1, 64, 0, st>>>(d, rep, spin1);
        publish<<<1, 64, 0, st>>>(d + 8, rep, 4000000);  // ~2 ms behind it
        while (*(volatile int*)h != rep) {
        }
        const double seen = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        hipStreamSynchronize(st);
        const double all = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%-28s flag seen after %8.1f us, queue drained after %8.1f us\n", name, seen, all);
    }
    hipHostFree(h);
}
int main() {
    run(hipHostMallocMapped | hipHostMallocCoherent, "coherent, short kernel", 0);
    run(hipHostMallocDefault, "default, short kernel", 0);
    run(hipHostMallocMapped | hipHostMallocCoherent, "coherent, inside long kernel", 2000000);
    return 0;
}
