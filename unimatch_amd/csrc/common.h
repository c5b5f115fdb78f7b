// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.  gfx950 only: no other-arch paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unimatch_hip.h"

// A/B switches for same-box timing runs exist ONLY in diagnostic builds (-DUM_DEBUG_SWITCHES, `python -m unimatch_amd.build
// --variant NAME -DUM_DEBUG_SWITCHES ...` -> unimatch_amd/_variants/libNAME.so, loaded through UM_LIB): the shipped library never
// reads the environment, its behaviour is a pure function of the arguments of each call.
#ifdef UM_DEBUG_SWITCHES
#include <stdlib.h>
static inline const char* um_debug_env(const char* name) { return getenv(name); }
#else
static inline const char* um_debug_env(const char*) { return nullptr; }
#endif

// CUs of the calling thread's CURRENT device (the launch plans that need "every workgroup resident at once" are sized by it):
// looked up per device and remembered; 256 when there is no device (host-side plan queries on a GPU-less box).
static inline int um_num_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    if (dev < 64 && cache[dev] > 0) return cache[dev];
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    if (dev < 64) cache[dev] = v;      // racing writers store the same value
    return v;
}

#define UM_CHANNELS 128          // feature channels of every UniMatch variant (unimatch/unimatch.py:19)
#define UM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define UM_LOG2E 1.4426950408889634f
#define UM_NEG_INIT (-1.0e30f)   // running-max start value
#define UM_NEG_MASK (-2.0e30f)   // excluded score: strictly below UM_NEG_INIT so exp2(s - m) == 0 even
                                 // for a lane that has not met a valid key yet

// ---- 16-bit element traits -------------------------------------------------------------------
// Fp16: used with hi+lo split operands ("exact" mode).  Bf16: single operand ("fast" mode).
struct Fp16 {
    static __device__ __forceinline__ unsigned short down(float x) {
        _Float16 h = (_Float16)x;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ float up(unsigned short b) {
        return (float)__builtin_bit_cast(_Float16, b);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        f16x2 h = __builtin_convertvector(v, f16x2);
        return __builtin_bit_cast(unsigned, h);
    }
    static __device__ __forceinline__ f32x2 unpack2(unsigned u) {
        f16x2 h = __builtin_bit_cast(f16x2, u);
        return __builtin_convertvector(h, f32x2);
    }
    // lo plane of a pair: rn16(p - hi), hi = pack2(p0, p1).  `neg1` must be um_opaque_neg1(): fma(float(hi), -1, p) with a -1
    // the compiler cannot see selects v_fma_mixlo_f16 + v_fma_mixhi_f16 (the fp16 -> fp32 extension rides inside the
    // instruction), 2 VALU per pair; the literal form folds to a subtraction and costs 2 v_cvt_f32_f16 + 2 v_sub (or one
    // v_pk_add_f32) + v_cvt_pk_f16_f32.  Same bits either way: p - hi is exact in fp32.  Needs -fno-slp-vectorize (build.py).
    static __device__ __forceinline__ unsigned lo2(float p0, float p1, unsigned hi, float neg1) {
        const f16x2 h = __builtin_bit_cast(f16x2, hi);
        const f16x2 l = {(_Float16)__builtin_fmaf((float)h[0], neg1, p0), (_Float16)__builtin_fmaf((float)h[1], neg1, p1)};
        return __builtin_bit_cast(unsigned, l);
    }
    static __device__ __forceinline__ f32x16 mfma(i16x8 a, i16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

struct Bf16 {
    static __device__ __forceinline__ unsigned short down(float x) {
        __bf16 h = (__bf16)x;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ float up(unsigned short b) {
        return __builtin_bit_cast(float, ((unsigned)b) << 16);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        bf16x2 h = __builtin_convertvector(v, bf16x2);
        return __builtin_bit_cast(unsigned, h);
    }
    static __device__ __forceinline__ f32x2 unpack2(unsigned u) {
        f32x2 r = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
        return r;
    }
    static __device__ __forceinline__ unsigned lo2(float p0, float p1, unsigned hi, float neg1) {   // (unused: one bf16 plane)
        const f32x2 hh = unpack2(hi);
        return pack2(__builtin_fmaf(hh[0], neg1, p0), __builtin_fmaf(hh[1], neg1, p1));
    }
    static __device__ __forceinline__ f32x16 mfma(i16x8 a, i16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// ---- operand-range note (um_range_flags, include/unimatch_hip.h): `mx` = the largest |value| this lane turned into an fp16
// operand; at 65504 the hi plane is inf.  One compare per lane at the end of a section; the atomic only ever runs on overflow.
template <class T> struct UmRange;
template <> struct UmRange<Fp16> { static constexpr float limit = 65504.0f; };
template <> struct UmRange<Bf16> { static constexpr float limit = 3.0e38f; };
template <class T>
__device__ __forceinline__ void um_range_note(unsigned* flag, float mx, unsigned bit) {
#ifndef UM_NO_RANGE_NOTE          // (diagnostic builds: what the tracking costs -- the maxima are dead code without the note)
    if (flag != nullptr && !(mx < UmRange<T>::limit)) __hip_atomic_fetch_or(flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
unsigned* um_range_flag_dev();          // capi.hip: device address of the sticky flag word (nullptr: unavailable)

// -1.0f in an SGPR whose value the compiler cannot see (see Fp16::lo2)
__device__ __forceinline__ float um_opaque_neg1() {
    float v = -1.0f;
    asm volatile("" : "+s"(v));
    return v;
}

// ---- MFMA 32x32x16 fragment conventions (wave64) ---------------------------------------------
//   A (32 x 16): lane l holds A[m = l & 31][k = 8 * (l >> 5) + j], j = 0..7   (16 contiguous bytes)
//   B (16 x 32): lane l holds B[k = 8 * (l >> 5) + j][n = l & 31]
//   D (32 x 32): lane l, reg r holds D[m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][n = l & 31]
// All kernels compute "swapped" products D = K . Q^T so that n = query: every lane owns one query
// row and the softmax reductions are in-lane plus one exchange with lane ^ 32.
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ i16x8 ld_global_16B(const unsigned short* p) {
    return *reinterpret_cast<const i16x8*>(p);
}

// Exchange between the two half-waves without LDS: returns, on every lane, {own value, value of lane ^ 32} in
// some order.  v_permlane32_swap swaps the upper half of one register with the lower half of another.
// Two hipcc (ROCm 7.2) pitfalls are handled here, both found the hard way:
//   * symmetric uses of the two results -- fmaxf(r0, r1), r0 + r1 -- are folded as if r0 == r1 (the combine
//     looks at the producing node, not at the result index): each result is laundered through an empty asm;
//   * hipcc's hazard recognizer does not look through an inline-asm statement, so the 2 wait states the swap
//     needs after a VALU write of its operands are supplied explicitly.
__device__ __forceinline__ void half_wave_pair(float v, float& a, float& b) {
    unsigned x = __builtin_bit_cast(unsigned, v), y = x;
    asm volatile("s_nop 1" : "+v"(y), "+v"(x));
    const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    unsigned r0 = sw[0], r1 = sw[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}

// 16-byte agent-scope accesses for workgroup -> workgroup hand-offs through memory (MI355X_MICROARCH.md, "inter-workgroup
// visibility": sc1 write-through stores + sc1 loads need no L2 write-back / invalidate; 16-byte sc1 accesses run at the plain
// rate, dword ones at a sixth of it).  The stores must be drained (s_waitcnt vmcnt(0)) before the flag is raised; the load
// helper waits for its own four loads.
__device__ __forceinline__ void st_agent_16B(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld_agent_16Bx4(const float* p0, const float* p1, const float* p2, const float* p3, f32x4& a, f32x4& b,
                                               f32x4& c, f32x4& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}

__device__ __forceinline__ f32x4 ld_agent_16B(const float* p) {
    f32x4 a;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
    return a;
}

// XCD-aware remap of a linear workgroup id: the dispatcher places consecutive ids on consecutive
// XCDs (id % 8); this gives every XCD a contiguous range of logical ids so that workgroups sharing
// K/V (query tiles of one window) hit the same L2.  Bijective for any total.  Speed only.
__device__ __forceinline__ int xcd_remap(int id, int total) {
    const int nx = 8;
    int q = total / nx, r = total % nx;
    int xcd = id % nx, k = id / nx;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}
