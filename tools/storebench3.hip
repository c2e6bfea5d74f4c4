// tools/storebench3.hip -- does the memory type of the destination change the write rate?  The same streaming-store kernel
// (16 waves per CU, 1 KiB per wave and step) into hipMalloc memory, fine-grained and uncached device memory (hipExtMallocWithFlags).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_store(char *base, size_t bytesPerWave, int steps) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    char *p = base + wave * bytesPerWave + (size_t)lane * 16;
    unsigned v = (unsigned)wave * 2654435761u + lane;
    for (int t = 0; t < steps; t++) { v = v * 1664525u + 1013904223u; u32x4 w = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<u32x4 *>(p) = w; p += 1024; }
}
static void run(const char *name, char *buf, size_t total) {
    const size_t waves = 256 * 16 * 4; const int steps = (int)(total / waves / 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_store, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, buf, (size_t)steps * 1024, steps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep) best = ms < best ? ms : best;
    }
    printf("%-14s %8.3f ms  %6.2f TB/s\n", name, best, (double)waves * steps * 1024 / best / 1e9);
}
int main() {
    const size_t total = (size_t)16 << 30;
    char *a = nullptr, *b = nullptr, *c = nullptr;
    if (hipMalloc(&a, total) == hipSuccess) { run("hipMalloc", a, total); (void)hipFree(a); }
    if (hipExtMallocWithFlags((void **)&b, total, hipDeviceMallocFinegrained) == hipSuccess) { run("fine-grained", b, total); (void)hipFree(b); } else printf("fine-grained: alloc failed\n");
    if (hipExtMallocWithFlags((void **)&c, total, hipDeviceMallocUncached) == hipSuccess) { run("uncached", c, total); (void)hipFree(c); } else printf("uncached: alloc failed\n");
    return 0;
}
