// tools/storebench4.hip -- round 4: why does hipMemset write the matrix pool faster (7.2 TB/s) than any store kernel of rounds 2-3 (5.6-5.85)
// and than the headline fill (6.5)?  hipMemset's kernel (__amd_rocclr_fillBufferAligned) is a GRID-STRIDE loop: every lane writes 16 bytes,
// then advances by the size of the whole grid, so all resident waves together write ONE compact window that moves through the buffer.
// The store kernels of rounds 2-3 gave every wave a private stream.  This tool writes the same 22 GB chunked virtual range (256-MiB
// chunks, as the engine's pool) with: hipMemset; the grid-stride pattern at several grid sizes; private streams; the engine's
// group-interleaved layout (64 waves share a block, chunk T of wave g at (T * 64 + g) * chunkBytes) with 1, 2 and 4 KiB per wave and
// step; and the grid-stride pattern with the nt / sc1 cache-policy bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int POLICY> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
    if constexpr (POLICY == 0) *p = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
// grid-stride: lane i of the grid writes 16 B at i*16, then advances by gridBytes (hipMemset's pattern)
template <int POLICY> __global__ void __launch_bounds__(256) k_gridstride(char *base, size_t total, size_t gridBytes) {
    size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    unsigned v = (unsigned)off;
    for (; off < total; off += gridBytes) { u32x4 w = {v, v + 1, v + 2, v + 3}; st<POLICY>(reinterpret_cast<u32x4 *>(base + off), w); v += 7; }
}
// private streams: wave w writes [w * bytesPerWave, (w+1) * bytesPerWave) in steps of KIB KiB (KIB stores of 1 KiB each)
template <int KIB> __global__ void __launch_bounds__(256) k_private(char *base, size_t bytesPerWave, int steps) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    char *p = base + wave * bytesPerWave + (size_t)lane * 16;
    unsigned v = (unsigned)wave;
    for (int t = 0; t < steps; t++) {
#pragma unroll
        for (int k = 0; k < KIB; k++) { u32x4 w = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<u32x4 *>(p + k * 1024) = w; }
        p += KIB * 1024; v += 7;
    }
}
// the engine's layout: groups of G waves share a block; chunk T (KIB KiB) of member g at groupBase + (T * G + g) * KIB KiB
template <int KIB> __global__ void __launch_bounds__(256) k_grouped(char *base, int G, int steps) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t grp = wave / G, g = wave % G;
    char *p = base + grp * ((size_t)steps * G * KIB * 1024) + g * (KIB * 1024) + (size_t)lane * 16;
    unsigned v = (unsigned)wave;
    for (int t = 0; t < steps; t++) {
#pragma unroll
        for (int k = 0; k < KIB; k++) { u32x4 w = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<u32x4 *>(p + k * 1024) = w; }
        p += (size_t)G * KIB * 1024; v += 7;
    }
}
// the headline's shape: waves of n + 63 steps whose skew ramps store only the lanes that are on a cell (rounded out to whole 128-byte
// lines, as the engine's rampLines), WORK dependent packed-int16 operations per step between the stores (the fill's recurrence is ~175
// VALU instructions per step on the headline), groups of 64 waves, 4 KiB per step
template <int WORK, bool RAMP> __global__ void __launch_bounds__(256) k_fill_like(char *base, int G, int n) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t grp = wave / G, g = wave % G;
    const int steps = n + 63;
    char *p = base + grp * ((size_t)steps * G * 4096) + g * 4096 + (size_t)lane * 16;
    unsigned v = (unsigned)wave * 2654435761u + lane, a = v ^ 0x5bd1e995u, b = v + 77u, c = v * 3u;
    const int lo = lane & ~7, hi = lane | 7; /* the line's lanes */
    for (int t = 0; t < steps; t++) {
#pragma unroll
        for (int k = 0; k < WORK / 4; k++) { /* four dependent VOP3P ops per trip, two chains */
            asm volatile("v_pk_add_u16 %0, %0, %2\n\tv_pk_max_i16 %1, %1, %0\n\tv_pk_add_u16 %0, %0, %1\n\tv_pk_max_i16 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(v));
        }
        const bool on = !RAMP || (t - lo >= 0 && t - hi < n); /* some lane of this lane's line is on a cell */
        if (on) {
#pragma unroll
            for (int k = 0; k < 4; k++) { u32x4 w = {a, b, a + k, b + k}; *reinterpret_cast<u32x4 *>(p + k * 1024) = w; }
        }
        p += (size_t)G * 4096;
    }
}
static hipEvent_t e0, e1;
template <class F> static void timeit(const char *name, size_t bytes, F launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep) best = ms < best ? ms : best;
    }
    printf("%-58s %8.3f ms  %6.2f TB/s\n", name, best, (double)bytes / best / 1e9); fflush(stdout);
}
int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atol(argv[1]) : 21) << 30, chunk = (size_t)256 << 20;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // the engine's pool: a reserved range backed by 256-MiB chunks
    void *res = nullptr; CK(hipMemAddressReserve(&res, total + chunk, 0, nullptr, 0));
    char *buf = (char *)(((size_t)res + chunk - 1) / chunk * chunk);
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop); prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t off = 0; off < total; off += chunk) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap(buf + off, chunk, 0, h, 0)); hs.push_back(h); }
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc); acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(buf, total, &acc, 1));
    for (int round = 0; round < 2; round++) {
        timeit("hipMemsetAsync", total, [&] { (void)hipMemsetAsync(buf, 0, total, 0); });
        for (int wgPerCu : {2, 4, 8, 16}) {
            char name[96]; snprintf(name, sizeof name, "grid-stride, %d workgroups of 256 per CU", wgPerCu);
            const unsigned grid = 256u * wgPerCu;
            timeit(name, total, [&] { hipLaunchKernelGGL(k_gridstride<0>, dim3(grid), dim3(256), 0, 0, buf, total, (size_t)grid * 4096); });
        }
        timeit("grid-stride 8 wg/CU, nt", total, [&] { hipLaunchKernelGGL(k_gridstride<1>, dim3(2048), dim3(256), 0, 0, buf, total, (size_t)2048 * 4096); });
        timeit("grid-stride 8 wg/CU, sc1", total, [&] { hipLaunchKernelGGL(k_gridstride<2>, dim3(2048), dim3(256), 0, 0, buf, total, (size_t)2048 * 4096); });
        timeit("grid-stride 8 wg/CU, sc0 sc1", total, [&] { hipLaunchKernelGGL(k_gridstride<3>, dim3(2048), dim3(256), 0, 0, buf, total, (size_t)2048 * 4096); });
        { // private streams, 16 waves per CU, all resident
            const size_t waves = 4096; 
            { const int steps = (int)(total / waves / 1024); timeit("private streams, 4096 waves, 1 KiB per step", (size_t)steps * waves * 1024, [&] { hipLaunchKernelGGL(k_private<1>, dim3(waves / 4), dim3(256), 0, 0, buf, (size_t)steps * 1024, steps); }); }
            { const int steps = (int)(total / waves / 4096); timeit("private streams, 4096 waves, 4 KiB per step", (size_t)steps * waves * 4096, [&] { hipLaunchKernelGGL(k_private<4>, dim3(waves / 4), dim3(256), 0, 0, buf, (size_t)steps * 4096, steps); }); }
        }
        for (int G : {64, 512, 4096}) { // the engine's groups (5000 waves of 1087 steps each write 4 KiB per step on the headline)
            const size_t waves = 4096;
            char name[96];
            { const int steps = (int)(total / waves / 1024); snprintf(name, sizeof name, "groups of %d waves, 4096 waves, 1 KiB per step", G);
              timeit(name, (size_t)steps * waves * 1024, [&] { hipLaunchKernelGGL(k_grouped<1>, dim3(waves / 4), dim3(256), 0, 0, buf, G, steps); }); }
            { const int steps = (int)(total / waves / 4096); snprintf(name, sizeof name, "groups of %d waves, 4096 waves, 4 KiB per step", G);
              timeit(name, (size_t)steps * waves * 4096, [&] { hipLaunchKernelGGL(k_grouped<4>, dim3(waves / 4), dim3(256), 0, 0, buf, G, steps); }); }
        }
        { // the headline's shape: 5000 waves x (1024 + 63) steps x 4 KiB; bytes = what is actually stored
            const int n = 1024; const size_t waves = 5000;
            const size_t full = waves * (size_t)(n + 63) * 4096, ramp = waves * ((size_t)n * 4096 + (size_t)0); /* ramp variant: ~n*4096 + line rounding */
            timeit("fill-like, whole chunks on the ramps, no work", full, [&] { hipLaunchKernelGGL((k_fill_like<0, false>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
            timeit("fill-like, ramp lines only, no work (bytes = n*4 KiB/wave)", ramp, [&] { hipLaunchKernelGGL((k_fill_like<0, true>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
            timeit("fill-like, ramp lines, 64 VALU per step", ramp, [&] { hipLaunchKernelGGL((k_fill_like<64, true>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
            timeit("fill-like, ramp lines, 128 VALU per step", ramp, [&] { hipLaunchKernelGGL((k_fill_like<128, true>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
            timeit("fill-like, ramp lines, 176 VALU per step", ramp, [&] { hipLaunchKernelGGL((k_fill_like<176, true>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
            timeit("fill-like, ramp lines, 240 VALU per step", ramp, [&] { hipLaunchKernelGGL((k_fill_like<240, true>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
            timeit("fill-like, whole chunks, 176 VALU per step", full, [&] { hipLaunchKernelGGL((k_fill_like<176, false>), dim3(waves / 4), dim3(256), 0, 0, buf, 64, n); });
        }
        { // like the headline: more waves than wave slots (5120 waves of ~1000 steps, 4 KiB per step, groups of 64)
            const size_t waves = 5120; const int steps = (int)(total / waves / 4096);
            timeit("groups of 64 waves, 5120 waves (> residency), 4 KiB per step", (size_t)steps * waves * 4096, [&] { hipLaunchKernelGGL(k_grouped<4>, dim3(waves / 4), dim3(256), 0, 0, buf, 64, steps); });
        }
    }
    return 0;
}
