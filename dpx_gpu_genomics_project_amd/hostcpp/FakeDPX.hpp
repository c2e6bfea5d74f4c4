// FakeDPX.hpp -- the reference's DPX-intrinsic model (c++/FakeDPX.hpp:19-126), here backed by the REAL device
// mappings: every call evaluates the primitive on the GPU with the CDNA4 instruction the fill kernels use
// (csrc/dpx_prims.hpp: v_max3_i32, v_pk_max_i16, v_pk_add_i16 ...) through dpx_prim_eval().  Same 36 static entry
// points and signatures, so the reference's own c++/testFakeDPX.cpp compiles against this header and becomes a
// known-answer test of the engine's primitive layer.  (One element per call: this is a test/diagnostic surface, the
// kernels use the device functions directly.)
#pragma once

using dpx_u32 = unsigned int; // two int16 / uint16 halves for the *16x2 forms (pair A high, pair B low)

class FakeDPX {
  public:
    static int __vimax3_s32(int x, int y, int z);
    static dpx_u32 __vimax3_s16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __vimax3_u32(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __vimax3_u16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static int __vimin3_s32(int x, int y, int z);
    static dpx_u32 __vimin3_s16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __vimin3_u32(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __vimin3_u16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);

    static int __vimax_s32_relu(int x, int y);
    static dpx_u32 __vimax_s16x2_relu(dpx_u32 x, dpx_u32 y);
    static int __vimin_s32_relu(int x, int y);
    static dpx_u32 __vimin_s16x2_relu(dpx_u32 x, dpx_u32 y);

    static int __vimax3_s32_relu(int x, int y, int z);
    static dpx_u32 __vimax3_s16x2_relu(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static int __vimin3_s32_relu(int x, int y, int z);
    static dpx_u32 __vimin3_s16x2_relu(dpx_u32 x, dpx_u32 y, dpx_u32 z);

    static int __vibmax_s32(int x, int y, bool *firstWins);
    static dpx_u32 __vibmax_u32(dpx_u32 x, dpx_u32 y, bool *firstWins);
    static int __vibmin_s32(int x, int y, bool *firstWins);
    static dpx_u32 __vibmin_u32(dpx_u32 x, dpx_u32 y, bool *firstWins);
    static dpx_u32 __vibmax_s16x2(dpx_u32 x, dpx_u32 y, bool *hiWins, bool *loWins);
    static dpx_u32 __vibmax_u16x2(dpx_u32 x, dpx_u32 y, bool *hiWins, bool *loWins);
    static dpx_u32 __vibmin_s16x2(dpx_u32 x, dpx_u32 y, bool *hiWins, bool *loWins);
    static dpx_u32 __vibmin_u16x2(dpx_u32 x, dpx_u32 y, bool *hiWins, bool *loWins);

    static int __viaddmax_s32(int x, int y, int z);
    static dpx_u32 __viaddmax_u32(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __viaddmax_s16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __viaddmax_u16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static int __viaddmin_s32(int x, int y, int z);
    static dpx_u32 __viaddmin_u32(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __viaddmin_s16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static dpx_u32 __viaddmin_u16x2(dpx_u32 x, dpx_u32 y, dpx_u32 z);

    static int __viaddmax_s32_relu(int x, int y, int z);
    static dpx_u32 __viaddmax_s16x2_relu(dpx_u32 x, dpx_u32 y, dpx_u32 z);
    static int __viaddmin_s32_relu(int x, int y, int z);
    static dpx_u32 __viaddmin_s16x2_relu(dpx_u32 x, dpx_u32 y, dpx_u32 z);
};
