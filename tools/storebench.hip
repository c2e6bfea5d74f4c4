// tools/storebench.hip -- what a store instruction costs on gfx950, by width and by active lanes (development aid).
// Every wave streams through its own region with one store instruction per step, 16 waves per CU; variants:
//   x4/64   global_store_dwordx4, all 64 lanes (1 KiB per instruction)          x4/48, x4/32: lanes >= 48 / 32 masked off (whole lines)
//   x2/64   global_store_dwordx2 (512 B per instruction)                        x1/64: global_store_dword (256 B)
// Prints time, bytes, TB/s and cycles per store instruction and CU (assuming 2.3 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/storebench tools/storebench.hip && tools/bin/storebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int WIDTH, int LANES>
__global__ void __launch_bounds__(256) k_store(char *base, size_t bytesPerWave, int steps) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    char *p = base + wave * bytesPerWave + (size_t)lane * (4 * WIDTH);
    unsigned v = (unsigned)wave * 2654435761u + lane;
    for (int t = 0; t < steps; t++) {
        v = v * 1664525u + 1013904223u;
        if (lane < LANES) {
            if constexpr (WIDTH == 4) { u32x4 w = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<u32x4 *>(p) = w; }
            else if constexpr (WIDTH == 2) { u32x2 w = {v, v + 1}; *reinterpret_cast<u32x2 *>(p) = w; }
            else *reinterpret_cast<unsigned *>(p) = v;
        }
        p += 64 * 4 * WIDTH;
    }
}

template <int WIDTH, int LANES>
void run(const char *name, char *buf, size_t waves, int steps) {
    const size_t bytesPerWave = (size_t)steps * 64 * 4 * WIDTH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_store<WIDTH, LANES>), dim3((unsigned)(waves / 4)), dim3(256), 0, 0, buf, bytesPerWave, steps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    const double bytes = (double)waves * steps * LANES * 4 * WIDTH, instr = (double)waves * steps;
    printf("%-6s %8.3f ms  %7.2f GB  %6.2f TB/s  %7.1f cycles per store instruction and CU\n", name, best, bytes / 1e9, bytes / best / 1e9,
           best * 1e-3 * 2.3e9 / (instr / 256.0));
}

int main() {
    const size_t waves = 256 * 16 * 4; // four rounds of 16 waves per CU
    const int steps = 1024;
    char *buf;
    if (hipMalloc(&buf, waves * (size_t)steps * 1024) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<4, 64>("x4/64", buf, waves, steps);
    run<4, 48>("x4/48", buf, waves, steps);
    run<4, 32>("x4/32", buf, waves, steps);
    run<4, 8>("x4/8", buf, waves, steps);
    run<2, 64>("x2/64", buf, waves, steps);
    run<1, 64>("x1/64", buf, waves, steps);
    run<4, 64>("x4/64", buf, waves, steps);
    return 0;
}
