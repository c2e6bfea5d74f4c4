// tools/storebench2.hip -- write bandwidth by stream shape (development aid): `waves` persistent waves, each streaming through its
// own contiguous region with NS consecutive 1-KiB global_store_dwordx4 per step (NS KiB contiguous per wave and step).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/storebench2 tools/storebench2.hip && tools/bin/storebench2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NS>
__global__ void __launch_bounds__(64) k_store(char *base, size_t bytesPerWave, int steps, int spin) {
    const int lane = threadIdx.x & 63;
    const size_t wave = blockIdx.x;
    char *p = base + wave * bytesPerWave + (size_t)lane * 16;
    unsigned v = (unsigned)wave * 2654435761u + lane;
    for (int t = 0; t < steps; t++) {
        for (int k = 0; k < spin; k++) v = v * 1664525u + 1013904223u; /* stand-in for the arithmetic between two stores */
#pragma unroll
        for (int s = 0; s < NS; s++) { u32x4 w = {v, v + 1, v + 2, v + (unsigned)s}; *reinterpret_cast<u32x4 *>(p + s * 1024) = w; }
        p += NS * 1024;
    }
}

template <int NS>
void run(char *buf, size_t total, int wavesPerCU, int spin) {
    const size_t waves = 256 * (size_t)wavesPerCU;
    const int steps = (int)(total / (waves * NS * 1024));
    const size_t bytesPerWave = (size_t)steps * NS * 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_store<NS>), dim3((unsigned)waves), dim3(64), 0, 0, buf, bytesPerWave, steps, spin);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    const double bytes = (double)waves * bytesPerWave;
    printf("%2d waves/CU  %d KiB/step  spin %3d  %8.3f ms  %6.2f TB/s\n", wavesPerCU, NS, spin, best, bytes / best / 1e9);
}

int main() {
    const size_t total = (size_t)12 << 30;
    char *buf;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, total);
    for (int spin : {0, 40}) {
        for (int w : {2, 4, 8, 16, 32}) { run<1>(buf, total, w, spin); run<2>(buf, total, w, spin); run<3>(buf, total, w, spin); run<4>(buf, total, w, spin); run<8>(buf, total, w, spin); }
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); (void)hipMemsetAsync(buf, 1, total, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemset %8.3f ms  %6.2f TB/s\n", ms, total / ms / 1e9);
    return 0;
}
