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

template <int F, int V, bool CL = false>
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
    if (CL || mask_cs == 1) {
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
    // CL (channels-last, the layout the mask head's convolution writes): the F logits of (tap, sub-row) are contiguous -- F / 4
    // 16-byte loads per tap, 8 lanes = the F sub-rows of a pixel cover 256 contiguous bytes (the stride-generic form below reads
    // them one float at a time: the compiler cannot prove mask_cs == 1)
    f32x4 lv[9][F / 4];
    if (CL) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int q = 0; q < F / 4; ++q) lv[k][q] = *reinterpret_cast<const f32x4*>(mb + k * F * F + 4 * q);
    }
#pragma unroll
    for (int fx = 0; fx < F; ++fx) {
        float lg[9];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            lg[k] = CL ? lv[k][fx >> 2][fx & 3] : mb[(long)(k * F * F + fx) * mask_cs];
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

// ---- flow_warp (unimatch/geometry.py:41-72): out[b, p, :] = bilinear sample of the token-major feature at p + flow[b, :, p],
// zeros outside, align_corners.  One thread = 4 channels of one pixel (32 threads cover a 128-channel token row, so every tap
// is one coalesced 512-byte read).  The coordinate arithmetic repeats the reference's normalise / un-normalise round trip
// (2 x / (w - 1) - 1, then ((g + 1) / 2) (w - 1) inside grid_sample) so that the fp32 rounding is the same.
__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ feat, const float* __restrict__ flow,
                                                        float* __restrict__ out, int batch, int h, int w, int c4n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int L = h * w;
    if (idx >= (long)batch * L * c4n) return;
    const int cq = (int)(idx % c4n);
    const long pid = idx / c4n;
    const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
    const int y = p / w, x = p - y * w;
    const float px = (float)x + flow[((long)b * 2) * L + p], py = (float)y + flow[((long)b * 2 + 1) * L + p];
    const float gx = 2.0f * px / (float)(w - 1) - 1.0f, gy = 2.0f * py / (float)(h - 1) - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    const float wnw = (fx1 - ix) * (fy1 - iy), wne = (ix - fx0) * (fy1 - iy);
    const float wsw = (fx1 - ix) * (iy - fy0), wse = (ix - fx0) * (iy - fy0);
    const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
    const float* fb = feat + ((long)b * L) * (4 * c4n) + 4 * cq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto tap = [&](int yy, int xx, float wt) {
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(fb + (long)(yy * w + xx) * (4 * c4n));
            acc += v * wt;
        }
    };
    tap(y0, x0, wnw);
    tap(y0, x0 + 1, wne);
    tap(y0 + 1, x0, wsw);
    tap(y0 + 1, x0 + 1, wse);
    *reinterpret_cast<f32x4*>(out + pid * (4 * c4n) + 4 * cq) = acc;
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
#define UM_UPS(F_, V_, CL_) hipLaunchKernelGGL((convex_upsample_kernel<F_, V_, CL_>), grid, block, 0, stream, flow, mask, up, batch, h, w, mult, cs, ps)
    if (mask_nhwc) {
        if (factor == 8 && channels == 2) UM_UPS(8, 2, true);
        else if (factor == 8) UM_UPS(8, 1, true);
        else if (channels == 2) UM_UPS(4, 2, true);
        else UM_UPS(4, 1, true);
    } else {
        if (factor == 8 && channels == 2) UM_UPS(8, 2, false);
        else if (factor == 8) UM_UPS(8, 1, false);
        else if (channels == 2) UM_UPS(4, 2, false);
        else UM_UPS(4, 1, false);
    }
#undef UM_UPS
    return (int)hipGetLastError();
}

extern "C" int um_flow_warp(const float* feature_tokens, const float* flow, float* out_tokens, int batch, int h, int w,
                            int channels, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!feature_tokens || !flow || !out_tokens || batch <= 0 || h < 2 || w < 2 || channels <= 0 || channels % 4 != 0) {
        um_set_error("um_flow_warp: bad argument (batch=%d h=%d w=%d channels=%d)", batch, h, w, channels);
        return -1;
    }
    const long total = (long)batch * h * w * (channels / 4);
    ScopedKernelTimer timer(UM_K_CONVEX_UPSAMPLE, stream);
    hipLaunchKernelGGL(flow_warp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, feature_tokens, flow,
                       out_tokens, batch, h, w, channels / 4);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// The small glue ops of the per-scale loop that were still torch calls ("small torch ops on the path").
//
// flow_upsample2x: flow <- 2 * bilinear_up2(flow), align_corners = True (unimatch/unimatch.py:162-163): out[y2, x2] with
// source position y2 * (h-1)/(2h-1), in ATen's operation order (h0 (w0 a + w1 b) + h1 (w0 c + w1 d)), times `mult`.
__global__ __launch_bounds__(256) void flow_upsample2x_kernel(const float* __restrict__ flow, float* __restrict__ out, long planes, int h,
                                                              int w, float rh, float rw, float mult) {
    const int ho = 2 * h, wo = 2 * w;
    const long total = planes * ho * wo;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x2 = (int)(i % wo);
    const long t = i / wo;
    const int y2 = (int)(t % ho);
    const long pl = t / ho;
    const float sy = rh * (float)y2, sx = rw * (float)x2;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* p = flow + pl * h * w + (long)y0 * w + x0;
    const float v = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[yp * w] + lx1 * p[yp * w + xp]);
    out[i] = v * mult;
}


extern "C" int um_flow_upsample2x(const float* flow, float* out, int batch, int channels, int h, int w, float mult, void* stream) {
    if (!flow || !out || batch <= 0 || channels <= 0 || h <= 0 || w <= 0) {
        um_set_error("um_flow_upsample2x: null pointer or non-positive size");
        return -1;
    }
    const long planes = (long)batch * channels, total = planes * 4 * h * w;
    const float rh = h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f, rw = w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    hipLaunchKernelGGL(flow_upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flow, out, planes,
                       h, w, rh, rw, mult);
    return (int)hipGetLastError();
}

// depth_cam_pack: the 30 floats per sample the depth kernels take -- Kinv[9] | R[9] | t[3] | K[9], row major -- from the caller's
// intrinsics [B,3,3] (rows 0-1 divided by `stride_div`, unimatch/unimatch.py:147-150) and relative pose [B,4,4]; entries B .. 2B-1
// (when `bidir`) hold the inverse pose (matching.py:226-233, unimatch.py:296-300).  Inverses in closed form (3x3 adjugate; a pose
// is affine: [A t; 0 0 0 1]^-1 = [A^-1, -A^-1 t]) instead of torch.inverse, which synchronises the device and keeps the depth
// path out of HIP graphs.
// Adjugate / determinant evaluated in fp64 and rounded once (torch.inverse runs an fp32 LU: this is at least as accurate; inputs
// given in double are rounded to fp32 by the caller first, as the reference's fp32 tensors are).  A singular matrix cannot raise
// from a kernel the way torch.inverse does (that would need the device synchronisation this kernel exists to avoid): the
// inverse is all NaN, so every prediction of that sample is NaN and no other sample is touched (tested).
__device__ __forceinline__ void inv3(const float* mf, float* o) {
    double m[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) m[j] = (double)mf[j];
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = det != 0.0 ? 1.0 / det : (double)__builtin_nanf("");
    o[0] = (float)(c00 * id);
    o[1] = (float)((m[2] * m[7] - m[1] * m[8]) * id);
    o[2] = (float)((m[1] * m[5] - m[2] * m[4]) * id);
    o[3] = (float)(c01 * id);
    o[4] = (float)((m[0] * m[8] - m[2] * m[6]) * id);
    o[5] = (float)((m[2] * m[3] - m[0] * m[5]) * id);
    o[6] = (float)(c02 * id);
    o[7] = (float)((m[1] * m[6] - m[0] * m[7]) * id);
    o[8] = (float)((m[0] * m[4] - m[1] * m[3]) * id);
}

__global__ void depth_cam_pack_kernel(const float* __restrict__ intr, const float* __restrict__ pose, float* __restrict__ cam, int batch,
                                      float stride_div, int bidir) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = bidir ? 2 * batch : batch;
    if (i >= n) return;
    const int b = i % batch;
    float k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = j < 6 ? intr[b * 9 + j] / stride_div : intr[b * 9 + j];
    float* c = cam + (long)i * 30;
    inv3(k, c);
    const float* p = pose + b * 16;
    float r[9] = {p[0], p[1], p[2], p[4], p[5], p[6], p[8], p[9], p[10]}, t[3] = {p[3], p[7], p[11]};
    if (i >= batch) {
        float ri[9];
        inv3(r, ri);
        const float t0 = -(ri[0] * t[0] + ri[1] * t[1] + ri[2] * t[2]), t1 = -(ri[3] * t[0] + ri[4] * t[1] + ri[5] * t[2]),
                    t2 = -(ri[6] * t[0] + ri[7] * t[1] + ri[8] * t[2]);
#pragma unroll
        for (int j = 0; j < 9; ++j) r[j] = ri[j];
        t[0] = t0;
        t[1] = t1;
        t[2] = t2;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) c[9 + j] = r[j];
    c[18] = t[0];
    c[19] = t[1];
    c[20] = t[2];
#pragma unroll
    for (int j = 0; j < 9; ++j) c[21 + j] = k[j];
}

extern "C" int um_depth_cam_pack(const float* intrinsics, const float* pose, float* cam, int batch, float stride_div, int bidir, void* stream) {
    if (!intrinsics || !pose || !cam || batch <= 0 || !(stride_div > 0.f)) {
        um_set_error("um_depth_cam_pack: null pointer, non-positive batch or stride divisor");
        return -1;
    }
    const int n = bidir ? 2 * batch : batch;
    hipLaunchKernelGGL(depth_cam_pack_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, intrinsics, pose, cam, batch, stride_div, bidir);
    return (int)hipGetLastError();
}

// rigid_flow: the flow a depth map and a relative pose induce (unimatch/geometry.py:99-195, used by the depth refinement,
// unimatch.py:295-305): X = R (Kinv [x y 1]^T) z + t, u = K X, flow = u_xy / max(u_z, 1e-3) - (x, y); z = 1 / inv_depth.
__global__ __launch_bounds__(256) void rigid_flow_kernel(const float* __restrict__ inv_depth, const float* __restrict__ cam,
                                                         float* __restrict__ flow, int batch, int h, int w) {
    const int L = h * w;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * L) return;
    const int b = (int)(i / L), p = (int)(i - (long)b * L);
    const int y = p / w, x = p - y * w;
    const float* cm = cam + (long)b * 30;
    const float gx = (float)x, gy = (float)y;
    const float z = 1.0f / inv_depth[i];
    const float r0 = (cm[0] * gx + cm[1] * gy + cm[2]) * z, r1 = (cm[3] * gx + cm[4] * gy + cm[5]) * z, r2 = (cm[6] * gx + cm[7] * gy + cm[8]) * z;
    const float X = cm[9] * r0 + cm[10] * r1 + cm[11] * r2 + cm[18];
    const float Y = cm[12] * r0 + cm[13] * r1 + cm[14] * r2 + cm[19];
    const float Z = cm[15] * r0 + cm[16] * r1 + cm[17] * r2 + cm[20];
    const float u = cm[21] * X + cm[22] * Y + cm[23] * Z, v = cm[24] * X + cm[25] * Y + cm[26] * Z;
    const float zz = fmaxf(cm[27] * X + cm[28] * Y + cm[29] * Z, 1e-3f);
    flow[((long)b * 2) * L + p] = u / zz - gx;
    flow[((long)b * 2 + 1) * L + p] = v / zz - gy;
}

extern "C" int um_rigid_flow(const float* inv_depth, const float* cam, float* flow, int batch, int h, int w, void* stream) {
    if (!inv_depth || !cam || !flow || batch <= 0 || h <= 0 || w <= 0) {
        um_set_error("um_rigid_flow: null pointer or non-positive size");
        return -1;
    }
    const long total = (long)batch * h * w;
    hipLaunchKernelGGL(rigid_flow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inv_depth, cam, flow, batch, h, w);
    return (int)hipGetLastError();
}
