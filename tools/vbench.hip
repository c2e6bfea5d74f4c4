// tools/vbench.hip -- integer VALU issue-rate microbenchmark (development aid): cycles per wave64 instruction per SIMD
// for the instruction mix of the DP cell update, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(int *out, int iters, int a0, int b0) {
    int x0 = threadIdx.x, x1 = a0, x2 = b0, x3 = a0 ^ b0, x4 = 7, x5 = 9, x6 = 11, x7 = 13;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { x0 += x1; x2 += x3; x4 += x5; x6 += x7; x1 += x0; x3 += x2; x5 += x4; x7 += x6; }                 // v_add_u32 x8
            if (KIND == 1) { x0 = max(x0, x1); x2 = max(x2, x3) + 1; x4 = max(x4, x5); x6 = max(x6, x7) + 1; x1 = max(x1, x0) + 1; x3 = max(x3, x2); x5 = max(x5, x4) + 1; x7 = max(x7, x6); } // max+add mix
            if (KIND == 2) { x0 = max(max(x0, x1), x2) + 1; x3 = max(max(x3, x4), x5) + 1; x6 = max(max(x6, x7), x0) + 1; x1 = max(max(x1, x2), x3) + 1; }                   // v_max3 + add
            if (KIND == 3) { x0 = (x1 == x2) ? x3 : x4; x1 += x0; x2 = (x3 == x4) ? x5 : x6; x3 += x2; x4 = (x5 == x6) ? x7 : x0; x5 += x4; }                              // cmp+cndmask+add
            if (KIND == 4) { x0 = __builtin_amdgcn_update_dpp(x1, x0, 0x138, 0xf, 0xf, false) + 1; x2 = __builtin_amdgcn_update_dpp(x3, x2, 0x138, 0xf, 0xf, false) + 1; }   // dpp mov + add
            if (KIND == 5) { x0 = __builtin_amdgcn_perm(x0, x1, 0x05040100) + 1; x2 = __builtin_amdgcn_perm(x2, x3, 0x05040100) + 1; x4 = __builtin_amdgcn_perm(x4, x5, 0x05040100) + 1; } // perm + add
            if (KIND == 6) { x0 = max((unsigned)x0, ((unsigned)x1 << 16) | (unsigned)x2); x3 = max((unsigned)x3, ((unsigned)x4 << 16) | (unsigned)x5); x1 += 1; x4 += 1; }   // lshl_or + max_u32
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND> void run(const char *name, int instrPerUnroll, int *out) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps; // 256 CUs x (4 waves per block) x wps blocks per CU = wps waves per SIMD
        const int iters = 4000;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 3, 5); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 3, 5);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)iters * 16 * instrPerUnroll * wps; // per SIMD
        printf("%-22s waves/SIMD=%d : %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wps, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    }
}

int main() {
    int *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_add_u32 x8", 8, out);
    run<1>("max/add mix x12", 12, out);
    run<2>("max3+add x8", 8, out);
    run<3>("cmp+cndmask+add x9", 9, out);
    run<4>("dpp mov+add x4", 4, out);
    run<5>("perm+add x6", 6, out);
    run<6>("lshl_or+max_u32+add x6", 6, out);
    return 0;
}
