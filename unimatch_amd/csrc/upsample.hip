// RAFT-style convex upsampling, fused.                                                      gfx950 / wave64
//
//   up[b, c, f*y + fy, f*x + fx] = sum_k softmax_k( mask[b, k*f*f + fy*f + fx, y, x] ) * mult * flow[b, c, y + k/3 - 1, x + k%3 - 1]
//
// Replaces upsample_flow_with_mask (unimatch/utils.py:134-152): the reference materialises the softmaxed mask
// [B,1,9,f,f,h,w], the unfolded flow [B,c,9,1,1,h,w], their product and a permuted copy; here one thread owns one
// (low-resolution pixel, output sub-row), keeps the 3x3 flow neighbourhood (zero padded) in registers and streams
// its 9*f mask logits once (coalesced along x), writing f contiguous outputs.  HBM-bound: mask read + output write.
// SURVEY section 8(f) rank 2 ("next" row).
#include "common.h"
#include "timing.h"

template <int F, int V>
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ up, int batch, int h, int w,
                                                              float mult, long mask_cs, long mask_ps) {
    // one thread = one low-resolution pixel x one output sub-row fy: index = ((b*h + y)*F + fy)*w + x, so that a
    // wave reads 64 consecutive x of one mask channel (coalesced) and writes 64 x F consecutive outputs of one row
    const long total = (long)batch * h * F * w;
    const long tix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= total) return;
    int x, fy, y, b;
    if (mask_cs == 1) {
        // channels-last mask: the 9 F^2 logits of a pixel are contiguous, so consecutive lanes take consecutive sub-rows
        // fy of one pixel (8 lanes read 256 contiguous bytes per tap)
        fy = (int)(tix % F);
        const long t2 = tix / F;
        x = (int)(t2 % w);
        const long t3 = t2 / w;
        y = (int)(t3 % h);
        b = (int)(t3 / h);
    } else {
        x = (int)(tix % w);
        const long t2 = tix / w;
        fy = (int)(t2 % F);
        const long t3 = t2 / F;
        y = (int)(t3 % h);
        b = (int)(t3 / h);
    }
    const int L = h * w, p = y * w + x;
    float nb[V][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
        for (int c = 0; c < V; ++c) nb[c][k] = ok ? mult * flow[((long)b * V + c) * L + yy * w + xx] : 0.f;
    }
    // mask element (b, channel, pixel) at b * 9 F^2 L + channel * mask_cs + pixel * mask_ps  (NCHW: L, 1; NHWC: 1, 9 F^2)
    const float* mb = mask + (long)b * 9 * F * F * L + (long)fy * F * mask_cs + (long)p * mask_ps;
    float out[V][F];
#pragma unroll
    for (int fx = 0; fx < F; ++fx) {
        float lg[9];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            lg[k] = mb[(long)(k * F * F + fx) * mask_cs];
            mx = fmaxf(mx, lg[k]);
        }
        float den = 0.f, num[V];
#pragma unroll
        for (int c = 0; c < V; ++c) num[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float e = __expf(lg[k] - mx);
            den += e;
#pragma unroll
            for (int c = 0; c < V; ++c) num[c] = __builtin_fmaf(e, nb[c][k], num[c]);
        }
#pragma unroll
        for (int c = 0; c < V; ++c) out[c][fx] = num[c] / den;
    }
    const int W = F * w;
#pragma unroll
    for (int c = 0; c < V; ++c) {
        float* dst = up + (((long)b * V + c) * (F * h) + (long)F * y + fy) * W + (long)F * x;
#pragma unroll
        for (int fx = 0; fx < F; fx += 4)
            *reinterpret_cast<f32x4*>(dst + fx) = f32x4{out[c][fx], out[c][fx + 1], out[c][fx + 2], out[c][fx + 3]};
    }
}

extern void um_set_error(const char* fmt, ...);

extern "C" int um_convex_upsample(const float* flow, const float* mask, float* up, int batch, int channels, int h, int w,
                                  int factor, int is_depth, int mask_nhwc, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!flow || !mask || !up || batch <= 0 || h <= 0 || w <= 0) {
        um_set_error("um_convex_upsample: null pointer or non-positive size");
        return -1;
    }
    if ((factor != 4 && factor != 8) || (channels != 1 && channels != 2)) {
        um_set_error("um_convex_upsample: factor=%d channels=%d unsupported (factor 4|8, channels 1|2)", factor, channels);
        return -4;
    }
    const long total = (long)batch * h * w * factor;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    const float mult = is_depth ? 1.f : (float)factor;
    const long cs = mask_nhwc ? 1 : (long)h * w, ps = mask_nhwc ? 9L * factor * factor : 1;
    ScopedKernelTimer timer(UM_K_CONVEX_UPSAMPLE, stream);
    if (factor == 8 && channels == 2)
        hipLaunchKernelGGL((convex_upsample_kernel<8, 2>), grid, block, 0, stream, flow, mask, up, batch, h, w, mult, cs, ps);
    else if (factor == 8)
        hipLaunchKernelGGL((convex_upsample_kernel<8, 1>), grid, block, 0, stream, flow, mask, up, batch, h, w, mult, cs, ps);
    else if (channels == 2)
        hipLaunchKernelGGL((convex_upsample_kernel<4, 2>), grid, block, 0, stream, flow, mask, up, batch, h, w, mult, cs, ps);
    else
        hipLaunchKernelGGL((convex_upsample_kernel<4, 1>), grid, block, 0, stream, flow, mask, up, batch, h, w, mult, cs, ps);
    return (int)hipGetLastError();
}
