// tools/wbench.hip -- write-bandwidth microbenchmarks that bracket the fill kernel's store pattern (development aid).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wbench tools/wbench.hip && /tmp/wbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// (a) flat grid-stride streaming store: the classic ceiling
__global__ void k_flat(uint4 *p, size_t n16, uint4 v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// (b) one wave per region, consecutive CHUNK-byte chunks (CHUNK = 16*64*K): the fill kernel's pattern, no compute
template <int K, bool NT>
__global__ void __launch_bounds__(256) k_region(uint4 *p, size_t regionBytes, int regions, int spin) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= regions) return;
    uint4 *base = p + (size_t)r * (regionBytes / 16);
    const size_t steps = regionBytes / (1024 * K);
    uint4 v = make_uint4(lane, r, 0, 0);
    for (size_t s = 0; s < steps; s++) {
        // `spin` dependent VALU ops per step emulate the DP arithmetic between stores
        for (int q = 0; q < spin; q++) v.x = v.x * 3u + v.y;
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint4 *dst = base + (s * K + k) * 64 + lane;
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            if (NT) __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4 *>(dst)); else *dst = v;
        }
    }
}
// (c) like (b) but each lane writes K consecutive 16-B pieces (lane-contiguous K*16 B): the "step group" layout
template <int K>
__global__ void __launch_bounds__(256) k_region_lanegroup(uint4 *p, size_t regionBytes, int regions, int spin) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= regions) return;
    uint4 *base = p + (size_t)r * (regionBytes / 16);
    const size_t steps = regionBytes / (1024 * K);
    uint4 v = make_uint4(lane, r, 0, 0);
    for (size_t s = 0; s < steps; s++) {
        for (int q = 0; q < spin; q++) v.x = v.x * 3u + v.y;
#pragma unroll
        for (int k = 0; k < K; k++) base[(s * 64 + lane) * K + k] = v;
    }
}

// (d) persistent waves: P waves in total, wave w streams regions w, w+P, w+2P, ... (1 KiB chunks) -- how does the number of
// concurrently open write streams change the achievable bandwidth?
__global__ void __launch_bounds__(256) k_region_persistent(uint4 *p, size_t regionBytes, int regions) {
    const int lane = threadIdx.x & 63;
    const int P = gridDim.x * 4;
    uint4 v = make_uint4(lane, 1, 0, 0);
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < regions; r += P) {
        uint4 *base = p + (size_t)r * (regionBytes / 16);
        const size_t steps = regionBytes / 1024;
        for (size_t s = 0; s < steps; s++) base[s * 64 + lane] = v;
    }
}

// (e) group-interleaved layout: G consecutive waves share a block of G regions; chunk s of wave g sits at (s*G + g) KiB, so
// the G waves together write one compact, forward-moving window (few DRAM pages / TLB entries open at a time)
__global__ void __launch_bounds__(256) k_region_interleaved(uint4 *p, size_t regionBytes, int regions, int G) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (regions / G) * G) return; /* full groups only: a partial last group would index past the allocation */
    const int grp = r / G, g = r % G;
    uint4 *base = p + (size_t)grp * G * (regionBytes / 16);
    const size_t steps = regionBytes / 1024;
    uint4 v = make_uint4(lane, r, 0, 0);
    for (size_t s = 0; s < steps; s++) base[(s * G + g) * 64 + lane] = v;
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
    const int regions = 10000; const size_t regionBytes = 2228224; // 2 stripes x 1088 steps x 1 KiB, like LSW 1024^2 R=8
    const size_t total = (size_t)regions * regionBytes;
    uint4 *p; CK(hipMalloc(&p, total));
    uint4 v = make_uint4(1, 2, 3, 4);
    float ms = timeit([&] { hipLaunchKernelGGL(k_flat, dim3(256 * 8), dim3(256), 0, 0, p, total / 16, v); }, 5);
    printf("flat store                      : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
    const dim3 grid((regions + 3) / 4), blk(256);
    for (int rep = 0; rep < 2; rep++) {
        ms = timeit([&] { hipLaunchKernelGGL((k_region<1, false>), grid, blk, 0, 0, p, regionBytes, regions, 0); }, 3);
        printf("region  1 KiB/step (64 x 16 B)          : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_region<2, false>), grid, blk, 0, 0, p, regionBytes, regions, 0); }, 3);
        printf("region  2 KiB/step (2 x 1 KiB rows)     : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_region<4, false>), grid, blk, 0, 0, p, regionBytes, regions, 0); }, 3);
        printf("region  4 KiB/step (4 x 1 KiB rows)     : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_region<8, false>), grid, blk, 0, 0, p, regionBytes, regions, 0); }, 3);
        printf("region  8 KiB/step (8 x 1 KiB rows)     : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_region_lanegroup<2>), grid, blk, 0, 0, p, regionBytes, regions, 0); }, 3);
        printf("lanegroup 2 x 16 B per lane (2 KiB)     : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_region_lanegroup<4>), grid, blk, 0, 0, p, regionBytes, regions, 0); }, 3);
        printf("lanegroup 4 x 16 B per lane (4 KiB)     : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_flat, dim3(256), dim3(256), 0, 0, p, total / 16, v); }, 3);
        printf("flat store, 256 blocks                  : %.3f ms  %.1f GB/s\n", ms, total / ms / 1e6);
        for (int G : {4, 16, 64, 256, 1000}) {
            ms = timeit([&] { hipLaunchKernelGGL(k_region_interleaved, grid, blk, 0, 0, p, regionBytes, regions, G); }, 3);
            const double bytes = (double)(regions / G) * G * regionBytes;
            printf("interleaved regions, G = %4d           : %.3f ms  %.1f GB/s\n", G, ms, bytes / ms / 1e6);
        }
        for (int blocks : {256, 1792}) {
            ms = timeit([&] { hipLaunchKernelGGL(k_region_persistent, dim3(blocks), blk, 0, 0, p, regionBytes, regions); }, 3);
            printf("persistent regions, %4d waves          : %.3f ms  %.1f GB/s\n", blocks * 4, ms, total / ms / 1e6);
        }
    }
    CK(hipFree(p));
    return 0;
}
