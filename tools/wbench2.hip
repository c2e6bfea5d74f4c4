// tools/wbench2.hip -- which store flavour / launch shape gives the highest pure-write bandwidth on MI355X? (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_flat(u4 *p, size_t n16) {
    u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 0) p[i] = v;
        if (MODE == 1) __builtin_nontemporal_store(v, p + i);
        if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
        if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
        if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p + i), "v"(v) : "memory");
        if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off nt sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
    }
}
// each block streams its own contiguous slab (block-contiguous instead of grid-strided)
__global__ void __launch_bounds__(256) k_slab(u4 *p, size_t n16) {
    u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    const size_t per = n16 / gridDim.x;
    u4 *b = p + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) b[i] = v;
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const size_t total = (size_t)22 << 30;
    u4 *p; CK(hipMalloc(&p, total));
    const char *names[] = {"plain", "nt", "sc0 sc1", "sc1", "sc0", "nt sc0 sc1"};
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_flat<0>, dim3(blocks), dim3(256), 0, 0, p, total / 16); }, 3);
        printf("flat plain  blocks=%5d : %.3f ms  %.1f GB/s\n", blocks, ms, total / ms / 1e6);
    }
#define RUN(M) { float ms = timeit([&] { hipLaunchKernelGGL(k_flat<M>, dim3(2048), dim3(256), 0, 0, p, total / 16); }, 3); \
                 printf("flat %-10s blocks= 2048 : %.3f ms  %.1f GB/s\n", names[M], ms, total / ms / 1e6); }
    RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    for (int blocks : {1024, 4096, 16384}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_slab, dim3(blocks), dim3(256), 0, 0, p, total / 16); }, 3);
        printf("slab plain  blocks=%5d : %.3f ms  %.1f GB/s\n", blocks, ms, total / ms / 1e6);
    }
    for (size_t gb : {1, 4, 8}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_flat<0>, dim3(2048), dim3(256), 0, 0, p, (gb << 30) / 16); }, 5);
        printf("flat plain  %zu GiB        : %.3f ms  %.1f GB/s\n", gb, ms, (double)(gb << 30) / ms / 1e6);
    }
    float ms = timeit([&] { CK(hipMemsetAsync(p, 0, total, 0)); }, 3);
    printf("hipMemset                 : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
    return 0;
}
