// Diagnostic only: what the matrix pipes sustain on THIS part with no memory traffic at all -- a loop of independent
// v_mfma_f32_32x32x16_f16 on four accumulators per wave, 2 waves per SIMD on every CU.  The roofline fractions in DESIGN.md
// are quoted against the 2.5 PFLOP/s data-sheet peak; this number says how much of that the chip keeps under its power limit.
#include "common.h"

__global__ __launch_bounds__(512, 2) void mfma_peak_kernel(float* sink, int iters, float seed) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // seed < 0: pseudo-random operands that change from one MFMA to the next (realistic data toggling, hence power);
    // seed >= 0: one constant value everywhere (the friendliest case)
    i16x8 a, b;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x = x * 1664525u + 1013904223u;
        const float va = seed < 0.f ? ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) : seed;
        x = x * 1664525u + 1013904223u;
        const float vb = seed < 0.f ? ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) : seed;
        a[j] = (short)Fp16::down(va);
        b[j] = (short)Fp16::down(vb);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = Fp16::mfma(a, b, acc[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = Fp16::mfma(b, a, acc[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;            // never true: keeps the loop alive
}

extern "C" int um_debug_mfma_peak(float* sink, int iters, int random_operands, void* stream) {
    // 256 CUs x 1 workgroup of 8 waves; every wave issues 8 * iters MFMAs of 32*32*16*2 flop
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(256 * 4), dim3(512), 0, (hipStream_t)stream, sink, iters, random_operands ? -1.0f : 0.5f);
    return (int)hipGetLastError();
}
