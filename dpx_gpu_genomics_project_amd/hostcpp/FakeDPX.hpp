// FakeDPX.hpp -- the reference's DPX-intrinsic model (c++/FakeDPX.hpp:19-126), here backed by the REAL device
// mappings: every call evaluates the primitive on the GPU with the CDNA4 instruction the fill kernels use
// (csrc/dpx_prims.hpp: v_max3_i32, v_pk_max_i16, v_pk_add_i16 ...) through dpx_prim_eval().  Same 36 static entry
// points and signatures, so the reference's own c++/testFakeDPX.cpp compiles against this header and becomes a
// known-answer test of the engine's primitive layer.  (One element per call: this is a test/diagnostic surface, the
// kernels use the device functions directly.)
#pragma once

class FakeDPX {
  public:
    static int __vimax3_s32(const int a, const int b, const int c);
    static unsigned int __vimax3_s16x2(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __vimax3_u32(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __vimax3_u16x2(const unsigned int a, const unsigned int b, const unsigned int c);
    static int __vimin3_s32(const int a, const int b, const int c);
    static unsigned int __vimin3_s16x2(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __vimin3_u32(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __vimin3_u16x2(const unsigned int a, const unsigned int b, const unsigned int c);

    static int __vimax_s32_relu(const int a, const int b);
    static unsigned int __vimax_s16x2_relu(const unsigned int a, const unsigned int b);
    static int __vimin_s32_relu(const int a, const int b);
    static unsigned int __vimin_s16x2_relu(const unsigned int a, const unsigned int b);

    static int __vimax3_s32_relu(const int a, const int b, const int c);
    static unsigned int __vimax3_s16x2_relu(const unsigned int a, const unsigned int b, const unsigned int c);
    static int __vimin3_s32_relu(const int a, const int b, const int c);
    static unsigned int __vimin3_s16x2_relu(const unsigned int a, const unsigned int b, const unsigned int c);

    static int __vibmax_s32(const int a, const int b, bool *const pred);
    static unsigned int __vibmax_u32(const unsigned int a, const unsigned int b, bool *const pred);
    static int __vibmin_s32(const int a, const int b, bool *const pred);
    static unsigned int __vibmin_u32(const unsigned int a, const unsigned int b, bool *const pred);
    static unsigned int __vibmax_s16x2(const unsigned int a, const unsigned int b, bool *const pred_hi, bool *const pred_lo);
    static unsigned int __vibmax_u16x2(const unsigned int a, const unsigned int b, bool *const pred_hi, bool *const pred_lo);
    static unsigned int __vibmin_s16x2(const unsigned int a, const unsigned int b, bool *const pred_hi, bool *const pred_lo);
    static unsigned int __vibmin_u16x2(const unsigned int a, const unsigned int b, bool *const pred_hi, bool *const pred_lo);

    static int __viaddmax_s32(const int a, const int b, const int c);
    static unsigned int __viaddmax_u32(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __viaddmax_s16x2(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __viaddmax_u16x2(const unsigned int a, const unsigned int b, const unsigned int c);
    static int __viaddmin_s32(const int a, const int b, const int c);
    static unsigned int __viaddmin_u32(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __viaddmin_s16x2(const unsigned int a, const unsigned int b, const unsigned int c);
    static unsigned int __viaddmin_u16x2(const unsigned int a, const unsigned int b, const unsigned int c);

    static int __viaddmax_s32_relu(const int a, const int b, const int c);
    static unsigned int __viaddmax_s16x2_relu(const unsigned int a, const unsigned int b, const unsigned int c);
    static int __viaddmin_s32_relu(const int a, const int b, const int c);
    static unsigned int __viaddmin_s16x2_relu(const unsigned int a, const unsigned int b, const unsigned int c);
};
