/*
 * dpx_prims.hpp -- the DPX primitive set on CDNA4 (gfx950).
 *
 * The reference models NVIDIA's DPX intrinsics on the CPU (c++/FakeDPX.hpp:19-126, FakeDPX.cpp) and uses the
 * real ones in its CUDA kernels (__vibmax_s32 / __vibmax_s16x2, cuda/LNW/LinearNeedlemanWunschV19.cu:16-24).
 * gfx950 has no DPX unit; the same primitives collapse onto ordinary VALU instructions:
 *
 *   __vimax3_s32 / __vimin3_s32          -> v_max3_i32 / v_min3_i32        (one instruction)
 *   __vimax3_u32 / __vimin3_u32          -> v_max3_u32 / v_min3_u32
 *   *_relu                               -> the third v_max3 operand is the inline constant 0
 *   __viaddmax_s32(a,b,c)                -> v_add_u32 + v_max_i32  (v_add3_u32 when two addends share a cell)
 *   __vibmax_s32(a,b,&pred)              -> v_max_i32 + v_cmp_ge_i32 (pred only materialised where it is used;
 *                                           the fill kernels never need it -- directions are recomputed
 *                                           from the stored scores by the traceback with the same >= rule)
 *   *_s16x2 / *_u16x2                    -> v_pk_max_i16 / v_pk_min_i16 / v_pk_max_u16 / v_pk_add_i16 (VOP3P)
 *
 * Everything is plain C++ on int / 2 x int16 vectors; hipcc selects the instructions named above (checked in
 * the .s).  The packed forms return mathematically correct halves (the reference's
 * __vimax3_s16x2 forgets to mask a negative low half, c++/FakeDPX.cpp:28 -- not reproduced).
 */
#ifndef DPX_PRIMS_HPP
#define DPX_PRIMS_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpx {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

/* element-wise packed min/max: clang lowers these to v_pk_max_i16 / v_pk_min_i16 / v_pk_max_u16 / v_pk_min_u16 */
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ u16x2 pk_max(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 pk_min(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }

/* ---- scalar 32-bit ---- */
__device__ __forceinline__ int vimax3_s32(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int vimin3_s32(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t vimax3_u32(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }
__device__ __forceinline__ uint32_t vimin3_u32(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ int vimax_s32_relu(int a, int b) { return max(max(a, b), 0); }
__device__ __forceinline__ int vimin_s32_relu(int a, int b) { return max(min(a, b), 0); }
__device__ __forceinline__ int vimax3_s32_relu(int a, int b, int c) { return max(vimax3_s32(a, b, c), 0); }
__device__ __forceinline__ int vimin3_s32_relu(int a, int b, int c) { return max(vimin3_s32(a, b, c), 0); }
__device__ __forceinline__ int vibmax_s32(int a, int b, bool *pred) { *pred = a >= b; return max(a, b); }
__device__ __forceinline__ uint32_t vibmax_u32(uint32_t a, uint32_t b, bool *pred) { *pred = a >= b; return max(a, b); }
__device__ __forceinline__ int vibmin_s32(int a, int b, bool *pred) { *pred = a <= b; return min(a, b); }
__device__ __forceinline__ uint32_t vibmin_u32(uint32_t a, uint32_t b, bool *pred) { *pred = a <= b; return min(a, b); }
__device__ __forceinline__ int viaddmax_s32(int a, int b, int c) { return max((int)((uint32_t)a + (uint32_t)b), c); }
__device__ __forceinline__ uint32_t viaddmax_u32(uint32_t a, uint32_t b, uint32_t c) { return max(a + b, c); }
__device__ __forceinline__ int viaddmin_s32(int a, int b, int c) { return min((int)((uint32_t)a + (uint32_t)b), c); }
__device__ __forceinline__ uint32_t viaddmin_u32(uint32_t a, uint32_t b, uint32_t c) { return min(a + b, c); }
__device__ __forceinline__ int viaddmax_s32_relu(int a, int b, int c) { return max(viaddmax_s32(a, b, c), 0); }
__device__ __forceinline__ int viaddmin_s32_relu(int a, int b, int c) { return max(viaddmin_s32(a, b, c), 0); }

/* ---- packed 2 x 16-bit ---- */
__device__ __forceinline__ uint32_t vimax3_s16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_max(pk_max(as_s16x2(a), as_s16x2(b)), as_s16x2(c))); }
__device__ __forceinline__ uint32_t vimin3_s16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_min(pk_min(as_s16x2(a), as_s16x2(b)), as_s16x2(c))); }
__device__ __forceinline__ uint32_t vimax3_u16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_max(pk_max(as_u16x2(a), as_u16x2(b)), as_u16x2(c))); }
__device__ __forceinline__ uint32_t vimin3_u16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_min(pk_min(as_u16x2(a), as_u16x2(b)), as_u16x2(c))); }
__device__ __forceinline__ uint32_t vimax_s16x2_relu(uint32_t a, uint32_t b) { return vimax3_s16x2(a, b, 0u); }
__device__ __forceinline__ uint32_t vimin_s16x2_relu(uint32_t a, uint32_t b) { return as_u32(pk_max(pk_min(as_s16x2(a), as_s16x2(b)), as_s16x2(0u))); }
__device__ __forceinline__ uint32_t vimax3_s16x2_relu(uint32_t a, uint32_t b, uint32_t c) { return vimax_s16x2_relu(vimax_s16x2_relu(a, b), c); }
/* reference nests two ReLU'd mins (c++/FakeDPX.cpp:137) */
__device__ __forceinline__ uint32_t vimin3_s16x2_relu(uint32_t a, uint32_t b, uint32_t c) { return vimin_s16x2_relu(vimin_s16x2_relu(a, b), c); }

__device__ __forceinline__ uint32_t vibmax_s16x2(uint32_t a, uint32_t b, bool *ph, bool *pl) {
    s16x2 x = as_s16x2(a), y = as_s16x2(b);
    *ph = x.y >= y.y; *pl = x.x >= y.x;   /* .y = high half, .x = low half */
    return as_u32(pk_max(x, y));
}
__device__ __forceinline__ uint32_t vibmax_u16x2(uint32_t a, uint32_t b, bool *ph, bool *pl) {
    u16x2 x = as_u16x2(a), y = as_u16x2(b);
    *ph = x.y >= y.y; *pl = x.x >= y.x;
    return as_u32(pk_max(x, y));
}
__device__ __forceinline__ uint32_t vibmin_s16x2(uint32_t a, uint32_t b, bool *ph, bool *pl) {
    s16x2 x = as_s16x2(a), y = as_s16x2(b);
    *ph = x.y <= y.y; *pl = x.x <= y.x;
    return as_u32(pk_min(x, y));
}
__device__ __forceinline__ uint32_t vibmin_u16x2(uint32_t a, uint32_t b, bool *ph, bool *pl) {
    u16x2 x = as_u16x2(a), y = as_u16x2(b);
    *ph = x.y <= y.y; *pl = x.x <= y.x;
    return as_u32(pk_min(x, y));
}
/* 16-bit adds wrap, exactly like the reference's `short` arithmetic (c++/FakeDPX.cpp:304-316) */
__device__ __forceinline__ uint32_t viaddmax_s16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_max((s16x2)(as_s16x2(a) + as_s16x2(b)), as_s16x2(c))); }
__device__ __forceinline__ uint32_t viaddmax_u16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_max((u16x2)(as_u16x2(a) + as_u16x2(b)), as_u16x2(c))); }
__device__ __forceinline__ uint32_t viaddmin_s16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_min((s16x2)(as_s16x2(a) + as_s16x2(b)), as_s16x2(c))); }
__device__ __forceinline__ uint32_t viaddmin_u16x2(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_min((u16x2)(as_u16x2(a) + as_u16x2(b)), as_u16x2(c))); }
__device__ __forceinline__ uint32_t viaddmax_s16x2_relu(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_max(as_s16x2(viaddmax_s16x2(a, b, c)), as_s16x2(0u))); }
__device__ __forceinline__ uint32_t viaddmin_s16x2_relu(uint32_t a, uint32_t b, uint32_t c) { return as_u32(pk_max(as_s16x2(viaddmin_s16x2(a, b, c)), as_s16x2(0u))); }

/* ---- cross-lane: previous lane's value (the reference's __shfl_up_sync(mask, v, 1), V12.cu:149) as a single
 * DPP move, `v_mov_b32_dpp ... wave_shr:1`.  Lane 0 has no source lane and keeps `lane0` (the stripe-edge
 * value).  wave_shl1 is the mirror image (lane 63 keeps `lane63`). ---- */
__device__ __forceinline__ int wave_shr1(int v, int lane0) { return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shl1(int v, int lane63) { return __builtin_amdgcn_update_dpp(lane63, v, 0x130, 0xf, 0xf, false); }

/* hipcc lowers a packed `min(x,1)*k+c` into per-half v_cmp/v_cndmask/v_perm (5 instructions); these two keep it on the
 * VOP3P pipe.  Register-only single instructions: no memory, no wait states needed around them. */
__device__ __forceinline__ uint32_t pk_min_u16_raw(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad_i16_raw(uint32_t a, uint32_t b, uint32_t c) { /* per half: a*b + c (wraps) */
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t pk_sub_u16_raw(uint32_t a, uint32_t b) { /* per half: a - b (wraps) */
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16_raw(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t bfi_b32(uint32_t mask, uint32_t a, uint32_t b) { /* (mask & a) | (~mask & b) */
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

/* pack the low halves of two ints into one dword: one v_perm_b32 */
__device__ __forceinline__ uint32_t pack_lo16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

} // namespace dpx
#endif
