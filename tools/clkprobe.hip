// tools/clkprobe.hip -- what clock does this box's GPU hold under a VALU load / under a store load?  (development aid, round 3:
// the same fill binary measures 3.17 ms on one box and 3.63 ms on another while hipMemset of the pool is equally fast on both)
// in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_valu(unsigned long long *out, int iters, unsigned seed) {
    unsigned a = threadIdx.x ^ seed, b = blockIdx.x + 1u, c = 0x9e3779b9u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { a = a * 1664525u + b; b = (b ^ a) + c; c = max(c + a, b); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a + b + c == 12345u) out[0] = 1; // keep the loop
}
__global__ void __launch_bounds__(256) k_store(u4 *p, size_t n16, unsigned long long *out) {
    u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
static double median_mhz(const std::vector<unsigned long long> &h, int blocks) {
    std::vector<double> f;
    for (int i = 0; i < blocks; i++) if (h[2 * i + 1]) f.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    std::sort(f.begin(), f.end());
    return f.empty() ? 0.0 : f[f.size() / 2];
}
int main() {
    const int blocks = 4096;
    unsigned long long *d; CK(hipMalloc(&d, blocks * 16));
    std::vector<unsigned long long> h(blocks * 2);
    u4 *buf; const size_t bytes = (size_t)8 << 30; CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, 2000, (unsigned)w); // ~0.3 s of load first
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, 2000, 77u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost));
        const double fv = median_mhz(h, blocks);
        for (int w = 0; w < 30; w++) hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, d);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, d);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        CK(hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost));
        printf("clkprobe rep %d: VALU loop %.2f ms at %.0f MHz | store kernel %.2f ms (%.2f TB/s) at %.0f MHz\n", rep, ms, fv, ms2, bytes / ms2 / 1e9, median_mhz(h, blocks));
    }
    return 0;
}
