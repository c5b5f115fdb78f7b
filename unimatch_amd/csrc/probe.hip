// Box probes (round 6; MEASUREMENT ABI, shipped): three tiny kernels bench.py times inside its own run so that a bench line describes
// the box it was measured on.  MI355X boxes of one pool differ by +-5 % in what they sustain (power / clock); without these
// numbers two lines from two boxes cannot be told apart from a code change.
//   um_probe_mfma   a memory-free loop of independent v_mfma_f32_32x32x16_f16 with pseudo-random operands on every SIMD of every CU
//                   (two waves per SIMD): what the matrix pipes sustain under the part's power limit on realistic data toggling
//                   (constant operands reach the data-sheet 2.5 PFLOP/s; random ones ~1.7)
//   um_probe_copy   a float4 grid-stride copy: HBM read + write bandwidth
//   um_probe_chase  one wave chasing a pointer ring through memory: dependent-load latency (what the latency-bound glue kernels --
//                   statistics merges, ticket hand-offs -- are made of)
// Nothing here is on the product path; nothing on the product path depends on it.
#include "common.h"

extern void um_set_error(const char* fmt, ...);

__global__ __launch_bounds__(512, 2) void probe_mfma_kernel(float* sink, int iters) {
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
        a[j] = (short)Fp16::down((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f);
        x = x * 1664525u + 1013904223u;
        b[j] = (short)Fp16::down((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f);
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

// FLOPs one um_probe_mfma(iters) launch executes: 2 workgroups per CU x 8 waves x 8 MFMAs per iteration x 32*32*16*2
extern "C" double um_probe_mfma_flops(int iters) {
    return (double)um_num_cus() * 2.0 * 8.0 * 8.0 * (double)iters * (32.0 * 32.0 * 16.0 * 2.0);
}

extern "C" int um_probe_mfma(float* sink, int iters, void* stream) {
    if (!sink || iters <= 0) {
        um_set_error("um_probe_mfma: null sink or non-positive iteration count");
        return UM_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(um_num_cus() * 2), dim3(512), 0, (hipStream_t)stream, sink, iters);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int um_probe_copy(const void* src, void* dst, size_t bytes, void* stream) {
    if (!src || !dst || bytes < 16 || (bytes & 15) || ((size_t)src & 15) || ((size_t)dst & 15)) {
        um_set_error("um_probe_copy: null / unaligned buffers or a byte count that is not a positive multiple of 16");
        return UM_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(probe_copy_kernel, dim3(um_num_cus() * 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst,
                       bytes / 16);
    return (int)hipGetLastError();
}

// ring[i] = index of the next element (built by the caller: a permutation with one cycle, stride far beyond a cache line); lane 0 of
// one wave follows it for `hops` steps
__global__ __launch_bounds__(64) void probe_chase_kernel(const unsigned* __restrict__ ring, unsigned* out, int hops) {
    if (threadIdx.x != 0) return;
    unsigned i = 0;
    for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(ring + i);
    out[0] = i;
}

extern "C" int um_probe_chase(const unsigned* ring, unsigned* out, int hops, void* stream) {
    if (!ring || !out || hops <= 0) {
        um_set_error("um_probe_chase: null pointer or non-positive hop count");
        return UM_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(probe_chase_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ring, out, hops);
    return (int)hipGetLastError();
}
