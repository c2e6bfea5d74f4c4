// tools/gatherbench.hip -- round 4: how long does a wave wait for a SCATTERED window fetch (what the wave traceback does per window: every
// lane reads 16 bytes from its own chunk, `stride` bytes from its neighbour's)?  waves x 20 dependent rounds of 3 loads per lane;
// prints microseconds per round for strides of 2 KiB ... 132 KiB, with 1 wave and with 1712 waves in flight, on hipMalloc memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k_gather(const char *base, size_t stride, size_t waveBytes, int rounds, unsigned *out, unsigned long long *ticks) {
    const int lane = threadIdx.x;
    const char *p = base + (size_t)blockIdx.x * waveBytes + (size_t)lane * stride;
    unsigned acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; r++) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(p), b = *reinterpret_cast<const u32x4 *>(p + 1024), c = *reinterpret_cast<const u32x4 *>(p + stride * 64);
        acc += a.x + b.y + c.z;
        p += (size_t)(16 + (acc & 1)) * 16; /* the next round's addresses depend on this round's data: rounds are serialised like window loads */
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) { out[blockIdx.x] = acc; ticks[blockIdx.x] = t1 - t0; }
}
int main() {
    const size_t total = (size_t)8 << 30;
    char *buf = nullptr; unsigned *out = nullptr; unsigned long long *ticks = nullptr;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess || hipMalloc(&ticks, 1 << 20) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, total);
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0); /* kHz */
    static unsigned long long h[4096];
    for (int waves : {1, 256, 1712}) {
        for (size_t stride : {(size_t)2048, (size_t)4096, (size_t)(128 << 10), (size_t)(132 << 10)}) {
            const int rounds = 20;
            const size_t waveBytes = stride * 130 < ((size_t)4 << 20) ? ((size_t)4 << 20) : stride * 130; /* every wave its own region */
            if (waveBytes * waves > total) { printf("waves %5d stride %7zu: does not fit\n", waves, stride); continue; }
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_gather, dim3(waves), dim3(64), 0, 0, buf, stride, waveBytes, rounds, out, ticks);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, ticks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < waves; i++) s += (double)h[i];
            printf("waves %5d stride %7zu B: %.2f us per round of 3 scattered 16-byte loads per lane (wall clock %d kHz)\n", waves, stride, s / waves / rounds / (rate / 1e3), rate);
        }
    }
    return 0;
}
