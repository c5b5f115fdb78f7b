// fp32 -> 16-bit operand planes (the MFMA operand format of the attention / correlation kernels).
//   exact mode: 2 planes of fp16,  x*scale = hi + lo  (hi = rn16(x*scale), lo = rn16(x*scale - hi))
//   fast  mode: 1 plane of bf16
// Plane layout: [NS][rows][128] 16-bit, plane stride = rows*128 elements.
#pragma once
#include "common.h"
#include "timing.h"

template <class T, int NS>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src,
                                                           unsigned short* __restrict__ dst,
                                                           long n8, float scale, long plane_stride, unsigned* range_flag) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    float mx = 0.f;
    for (; i < n8; i += step) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + 2 * i;
        f32x4 a = s4[0], b = s4[1];
        float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        float y[8];
        u32x4 hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            y[j] = x[j] * scale;
            mx = fmaxf(mx, __builtin_fabsf(y[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) hi[j] = T::pack2(y[2 * j], y[2 * j + 1]);
        *reinterpret_cast<u32x4*>(dst + 8 * i) = hi;
        if (NS == 2) {
            u32x4 lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 h = T::unpack2(hi[j]);
                lo[j] = T::pack2(y[2 * j] - h[0], y[2 * j + 1] - h[1]);
            }
            *reinterpret_cast<u32x4*>(dst + plane_stride + 8 * i) = lo;
        }
    }
    um_range_note<T>(range_flag, mx, UM_RANGE_PLANES);
}

// n_elems fp32 values (a multiple of 8) -> planes at dst, plane stride n_elems.  Returns hipError_t of the launch.
static inline hipError_t launch_split_elems(const float* src, unsigned short* dst, long n_elems, float scale,
                                            int mode, hipStream_t stream);

// rows*128 fp32 values -> planes at dst.
static inline hipError_t launch_split_planes(const float* src, unsigned short* dst, long rows, float scale,
                                             int mode, hipStream_t stream) {
    return launch_split_elems(src, dst, rows * UM_CHANNELS, scale, mode, stream);
}

static inline hipError_t launch_split_elems(const float* src, unsigned short* dst, long n_elems, float scale,
                                            int mode, hipStream_t stream) {
    const long n8 = n_elems / 8;
    long blocks = (n8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    const long plane_stride = n_elems;
    ScopedKernelTimer timer(UM_K_SPLIT_PLANES, stream);
    if (mode == 0)
        hipLaunchKernelGGL((split_planes_kernel<Fp16, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n8,
                           scale, plane_stride, um_range_flag_dev());
    else
        hipLaunchKernelGGL((split_planes_kernel<Bf16, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n8,
                           scale, plane_stride, (unsigned*)nullptr);
    return hipGetLastError();
}

static inline size_t planes_bytes(long rows, int mode) {
    return (size_t)rows * UM_CHANNELS * 2 * (mode == 0 ? 2 : 1);
}
