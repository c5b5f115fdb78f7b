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

// Calibration of s_memtime against the matrix pipe: the same loop, one workgroup per CU of WAVES waves (4 = one wave per SIMD, 8 =
// two), every wave stamping s_memtime before and after its MFMAs.  With the kernel's wall time (host events) this gives, in one
// measurement, the tick rate of s_memtime under this load and the ticks one wave spends per MFMA -- the numbers the section stamps of
// the attention kernel (tools/trace_attn.py) have to be read against.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void mfma_ticks_kernel(unsigned long long* ticks, float* sink, int iters, float seed, int chain) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
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
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (chain) {                                   // all MFMAs on ONE accumulator (the QK^T chain of the attention kernel)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0] = Fp16::mfma(a, b, acc[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0] = Fp16::mfma(b, a, acc[0]);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = Fp16::mfma(a, b, acc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = Fp16::mfma(b, a, acc[i]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int um_debug_mfma_ticks(unsigned long long* ticks, float* sink, int iters, int random_operands, int waves, int chain, void* stream) {
    const float seed = random_operands ? -1.0f : 0.5f;
    if (waves == 8)
        hipLaunchKernelGGL(mfma_ticks_kernel<8>, dim3(256), dim3(512), 0, (hipStream_t)stream, ticks, sink, iters, seed, chain);
    else
        hipLaunchKernelGGL(mfma_ticks_kernel<4>, dim3(256), dim3(256), 0, (hipStream_t)stream, ticks, sink, iters, seed, chain);
    return (int)hipGetLastError();
}

// The attention kernel's QK^T pattern in isolation: per k-step two A fragments come from LDS (ds_read_b128, issued two k-steps ahead)
// and feed three MFMAs against register-resident B fragments, 24 MFMAs per "tile" on one accumulator.  mode 0: as described; mode 1:
// the same instruction stream but the MFMAs use loop-invariant A registers (the LDS reads still happen): separates "operands arrive
// from LDS" from "LDS instructions sit in the stream".
template <int MODE>
__global__ __launch_bounds__(256, 1) void mfma_lds_kernel(unsigned long long* ticks, float* sink, int tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003800u + (unsigned)(i * 2654435761u >> 20);
    __syncthreads();
    i16x8 q[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) q[p][ks] = *reinterpret_cast<const i16x8*>(lds + ((lane * 16 + ks * 1024 + p * 8192) & 65535));
    int koff[8];
    const int r = lane & 31, half = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = r * 256 + (((2 * ks + half) ^ (r & 15)) << 4);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const i16x8 fixh = q[0][0], fixl = q[1][0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const unsigned char* kb = lds + (t & 1) * 16384;
        i16x8 fh[3], fl[3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            fh[ks] = *reinterpret_cast<const i16x8*>(kb + koff[ks]);
            fl[ks] = *reinterpret_cast<const i16x8*>(kb + 8192 + koff[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 2 < 8) {
                fh[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(kb + koff[ks + 2]);
                fl[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(kb + 8192 + koff[ks + 2]);
            }
            if (MODE == 0) {
                acc = Fp16::mfma(fl[ks % 3], q[0][ks], acc);
                acc = Fp16::mfma(fh[ks % 3], q[1][ks], acc);
                acc = Fp16::mfma(fh[ks % 3], q[0][ks], acc);
            } else {
                acc = Fp16::mfma(fixl, q[0][ks], acc);
                acc = Fp16::mfma(fixh, q[1][ks], acc);
                acc = Fp16::mfma(fixh, q[0][ks], acc);
                asm volatile("" : : "v"(fh[ks % 3]), "v"(fl[ks % 3]));        // the reads stay
            }
        }
        constexpr int RD = 2, MF = 3;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 0);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int um_debug_mfma_lds(unsigned long long* ticks, float* sink, int tiles, int mode, void* stream) {
    if (mode == 0)
        hipLaunchKernelGGL(mfma_lds_kernel<0>, dim3(256), dim3(256), 0, (hipStream_t)stream, ticks, sink, tiles);
    else
        hipLaunchKernelGGL(mfma_lds_kernel<1>, dim3(256), dim3(256), 0, (hipStream_t)stream, ticks, sink, tiles);
    return (int)hipGetLastError();
}
