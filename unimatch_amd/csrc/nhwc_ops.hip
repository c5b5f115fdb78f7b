// NHWC glue around conv.hip: InstanceNorm (+ ReLU, + shortcut + ReLU) whose output is the NEXT convolution's operand
// planes, and the NCHW -> NHWC hand-over from a MIOpen convolution.                                   gfx950 / wave64
//
// Replaces nn.InstanceNorm2d + ReLU (+ residual add + ReLU) of unimatch/backbone.py:7-36 in channels-last layout.
// All three kernels are HBM-bound: every activation is read once for the statistics, once for the apply pass, and
// written once per requested output format (operand planes [NS][rows + 1][C] with the all-zero padding row conv.hip
// expects, and / or fp32 [rows][C] for a later shortcut).
//
// Statistics are deterministic and cancellation-safe: every workgroup reduces a chunk of pixels with a per-channel
// shift (the chunk's first pixel), writing (shifted sum, shifted sum of squares); the finalize kernel merges the chunks
// with the parallel-variance formula in fp64 and emits (mean, 1/sqrt(var + eps)) per (image, channel).
#include "common.h"
#include "planes.h"

#define NHWC_CHUNK_ROWS 256

// ---- pass 1: per-chunk shifted sums.  grid (chunks, B), block 256; thread = (row lane, channel quad) --------------
__global__ __launch_bounds__(256) void nhwc_stats_kernel(const float* x, float* partial, int P, int C, int nchunk) {
    __shared__ float red[2][256 * 4];
    const int b = blockIdx.y, ch = blockIdx.x;
    const int quads = C >> 2, rl_n = 256 / quads;
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    const int r0 = ch * NHWC_CHUNK_ROWS, r1 = min(P, r0 + NHWC_CHUNK_ROWS);
    const float* xb = x + ((long)b * P) * C + 4 * q;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (rl < rl_n) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(xb + (long)r0 * C);
        for (int r = r0 + rl; r < r1; r += rl_n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)r * C) - k;
            s1 += v;
            s2 += v * v;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[0][threadIdx.x * 4 + i] = s1[i];
        red[1][threadIdx.x * 4 + i] = s2[i];
    }
    __syncthreads();
    // thread c < C sums channel c over the row lanes (fixed order: deterministic)
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x, cq = c >> 2, ci = c & 3;
        float a1 = 0.f, a2 = 0.f;
        for (int l = 0; l < rl_n; ++l) {
            a1 += red[0][(l * quads + cq) * 4 + ci];
            a2 += red[1][(l * quads + cq) * 4 + ci];
        }
        float* pr = partial + (((long)b * nchunk + ch) * 3) * C;
        pr[c] = x[((long)b * P + r0) * C + c];                    // the shift
        pr[C + c] = a1;
        pr[2 * C + c] = a2;
    }
}

// ---- pass 2: merge the chunks (fp64, Chan et al.) -> stats[b][0][c] = mean, stats[b][1][c] = rstd ---------------------
// partial[b][part][3][C], either (shift k, sum(x - k), sum((x - k)^2)) over `rpp` pixels per part (the last one may be short),
// written by nhwc_stats_kernel (rpp = NHWC_CHUNK_ROWS), or -- rpp == 0 -- (mean, pixel count, sum of squared deviations) as
// written by the convolutions' epilogues (parts of <= 128 pixels in the order of the kernel's tiles; empty parts have count 0).
// grid (C / 8, B), block 256 = 8 channels x 32 lanes.  All in fp64 and in a fixed order (deterministic): the mean from the
// sums, then sum(M2_i + n_i (mean_i - mean)^2) -- the parallel-variance formula with the global mean known, so there is no
// division inside the loops.  The parts were written by other XCDs (a microsecond away): up to 1024 parts per image every
// lane loads ALL its parts in one batch and keeps them in registers for the second sum, so the kernel costs about one
// memory round trip (it is launched 15 times per forward); larger images take two batched passes over memory.
__global__ __launch_bounds__(256) void nhwc_stats_finalize_kernel(const float* partial, float* stats, int P, int C, int nparts,
                                                                  int rpp, float eps) {
    __shared__ double red[256];
    __shared__ double bc[8];
    const int b = blockIdx.y;
    const int ch = threadIdx.x & 7, c = blockIdx.x * 8 + ch, ln = threadIdx.x >> 3;
    const bool counted = rpp == 0;                                // parts carry their pixel count
    const int tail = counted ? 1 : P - (nparts - 1) * rpp;        // pixels of the last part
    const double inv_full = counted ? 0.0 : 1.0 / (double)rpp, inv_tail = 1.0 / (double)tail;
    const float* base = partial + ((long)b * nparts * 3) * C + c;
    constexpr int NB = 32;                                        // parts per lane held in registers
    const bool in_regs = nparts <= 32 * NB;
    float rk[NB], r1[NB], r2[NB];
    double sx = 0.0;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int pt = ln + 32 * j;
            rk[j] = r1[j] = r2[j] = 0.f;
            if (pt < nparts) {
                const float* pr = base + (long)pt * 3 * C;
                rk[j] = pr[0];
                r1[j] = pr[C];
                r2[j] = pr[2 * C];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int pt = ln + 32 * j;
            if (pt < nparts)
                sx += counted ? (double)r1[j] * (double)rk[j]
                              : (pt == nparts - 1 ? (double)tail : (double)rpp) * (double)rk[j] + (double)r1[j];
        }
    } else {
#pragma unroll 8
        for (int pt = ln; pt < nparts; pt += 32) {
            const float* pr = base + (long)pt * 3 * C;
            sx += counted ? (double)pr[C] * (double)pr[0]
                          : (pt == nparts - 1 ? (double)tail : (double)rpp) * (double)pr[0] + (double)pr[C];   // sum of x = n k + sum(x - k)
        }
    }
    red[threadIdx.x] = sx;
    __syncthreads();
    if (ln == 0) {
        double t = 0.0;
        for (int l = 0; l < 32; ++l) t += red[l * 8 + ch];
        bc[ch] = t / (double)P;
    }
    __syncthreads();
    const double mean = bc[ch];
    double m2 = 0.0;
    auto add_part = [&](int pt, double k, double s1, double s2) {
        if (counted) {                                            // (mean, n, M2)
            const double d = k - mean;
            m2 += s2 + s1 * d * d;
            return;
        }
        const bool last = pt == nparts - 1;
        const double n = last ? (double)tail : (double)rpp, inv = last ? inv_tail : inv_full;
        const double d = k + s1 * inv - mean;
        m2 += (s2 - s1 * s1 * inv) + n * d * d;
    };
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int pt = ln + 32 * j;
            if (pt < nparts) add_part(pt, rk[j], r1[j], r2[j]);
        }
    } else {
#pragma unroll 8
        for (int pt = ln; pt < nparts; pt += 32) {
            const float* pr = base + (long)pt * 3 * C;
            add_part(pt, pr[0], pr[C], pr[2 * C]);
        }
    }
    red[threadIdx.x] = m2;
    __syncthreads();
    if (ln == 0) {
        double t = 0.0;
        for (int l = 0; l < 32; ++l) t += red[l * 8 + ch];
        const double var = t / (double)P;                         // biased, as nn.InstanceNorm2d
        stats[((long)b * 2) * C + c] = (float)mean;
        stats[((long)b * 2 + 1) * C + c] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
}

// ---- pass 3: apply.  One thread = 8 channels of one pixel (16 bytes of every output plane).  ----------------------
//   y = x                          (stats == null: plain format conversion)
//   y = (x - mean) * rstd          then ReLU if relu
//   y = relu(shortcut + y)         if shortcut (fp32) or shortcut_planes (operand planes of the same shape, hi + lo: what the
//                                  block's first convolution read -- the identity shortcut then needs no fp32 copy in HBM)
template <typename T, int NS>
__global__ __launch_bounds__(256) void nhwc_apply_kernel(const float* x, const float* stats, const float* shortcut,
                                                         const unsigned short* shortcut_planes, unsigned short* planes,
                                                         float* outf, long rows, int P, int C, int relu) {
    const int c8n = C >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (rows + (planes ? 1 : 0)) * c8n;          // + the zero row of the planes
    if (idx >= total) return;
    const long row = idx / c8n;
    const int c = (int)(idx - row * c8n) * 8;
    const long plane_stride = (rows + 1) * C;
    if (row == rows) {                                            // the padding row conv.hip reads for out-of-image taps
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) *reinterpret_cast<u32x4*>(planes + pl * plane_stride + row * C + c) = z;
        return;
    }
    float v[8];
    {
        // x is read exactly once: streaming loads (the output planes are re-read by the next convolution: ordinary stores)
        const f32x4 a0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * C + c));
        const f32x4 a1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * C + c + 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = a0[i];
            v[4 + i] = a1[i];
        }
    }
    if (stats) {
        const int b = (int)(row / P);
        const float* mu = stats + ((long)b * 2) * C + c;
        const float* rs = mu + C;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (v[i] - mu[i]) * rs[i];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (shortcut) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(shortcut + row * C + c);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(shortcut + row * C + c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(v[i] + s0[i], 0.f);
            v[4 + i] = fmaxf(v[4 + i] + s1[i], 0.f);
        }
    } else if (shortcut_planes) {
        const u32x4 sh = *reinterpret_cast<const u32x4*>(shortcut_planes + row * C + c);
        u32x4 sl = {0u, 0u, 0u, 0u};
        if (NS == 2) sl = *reinterpret_cast<const u32x4*>(shortcut_planes + plane_stride + row * C + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 uh = T::unpack2(sh[i]);
            f32x2 ul = {0.f, 0.f};
            if (NS == 2) ul = T::unpack2(sl[i]);
            v[2 * i] = fmaxf(v[2 * i] + (uh[0] + ul[0]), 0.f);
            v[2 * i + 1] = fmaxf(v[2 * i + 1] + (uh[1] + ul[1]), 0.f);
        }
    }
    if (outf) {
        *reinterpret_cast<f32x4*>(outf + row * C + c) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(outf + row * C + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (planes) {
        const u32x4 h = {T::pack2(v[0], v[1]), T::pack2(v[2], v[3]), T::pack2(v[4], v[5]), T::pack2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(planes + row * C + c) = h;
        if (NS == 2) {
            u32x4 l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 u = T::unpack2(h[i]);
                l[i] = T::pack2(v[2 * i] - u[0], v[2 * i + 1] - u[1]);
            }
            *reinterpret_cast<u32x4*>(planes + plane_stride + row * C + c) = l;
        }
    }
}

// ---- NCHW fp32 -> NHWC planes (+ fp32): 64 pixels x C channels through LDS.  grid (P / 64, B), block 256 ----------------
template <typename T, int NS>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, unsigned short* planes, float* outf, int P, int C,
                                                           long rows) {
    __shared__ float tile[256 * 65];
    const int b = blockIdx.y, p0 = blockIdx.x * 64;
    const int px = threadIdx.x & 63;
    for (int c = threadIdx.x >> 6; c < C; c += 4) {
        const int p = min(p0 + px, P - 1);
        tile[c * 65 + px] = x[((long)b * C + c) * P + p];
    }
    __syncthreads();
    const int c8n = C >> 3;
    const long plane_stride = (rows + 1) * C;
    for (int it = threadIdx.x; it < 64 * c8n; it += 256) {
        const int pl_px = it / c8n, c = (it - pl_px * c8n) * 8;
        if (p0 + pl_px >= P) continue;
        const long row = (long)b * P + p0 + pl_px;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = tile[(c + i) * 65 + pl_px];
        if (outf) {
            *reinterpret_cast<f32x4*>(outf + row * C + c) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(outf + row * C + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (planes) {
            const u32x4 h = {T::pack2(v[0], v[1]), T::pack2(v[2], v[3]), T::pack2(v[4], v[5]), T::pack2(v[6], v[7])};
            *reinterpret_cast<u32x4*>(planes + row * C + c) = h;
            if (NS == 2) {
                u32x4 l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 u = T::unpack2(h[i]);
                    l[i] = T::pack2(v[2 * i] - u[0], v[2 * i + 1] - u[1]);
                }
                *reinterpret_cast<u32x4*>(planes + plane_stride + row * C + c) = l;
            }
        }
    }
    if (planes && blockIdx.x == 0 && b == 0) {                   // the zero row
        for (int c = threadIdx.x; c < C; c += 256)
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) planes[pl * plane_stride + rows * C + c] = 0;
    }
}

// ---- small channels-last helpers of the refinement block (SepConvGRU gate arithmetic, reg_refine.py:55-76) ------------------
// One thread = 4 channels of one pixel.  mode 0: planes[coff + c] = src[c]                      (column scatter)
//                                         mode 1: planes[coff + c] = zr[C + c] * h[c]             (r * h)
//                                         mode 2: h[c] = (1 - zr[c]) * h[c] + zr[c] * q[c]; planes[coff + c] = h[c]
__global__ __launch_bounds__(256) void nhwc_gate_kernel(int mode, const float* src, int src_ld, const float* zr, float* hbuf,
                                                        unsigned short* planes, long plane_stride, int ld, int coff, long rows,
                                                        int C) {
    const int c4n = (C + 3) >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * c4n) return;
    const long row = idx / c4n;
    const int c = (int)(idx - row * c4n) * 4;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = c + i;
        float x = 0.f;
        if (ci < C) {
            if (mode == 0) x = src[row * src_ld + ci];
            else if (mode == 1) x = zr[row * (2 * C) + C + ci] * hbuf[row * C + ci];
            else {
                const float z = zr[row * (2 * C) + ci];
                x = (1.0f - z) * hbuf[row * C + ci] + z * src[row * src_ld + ci];
                hbuf[row * C + ci] = x;
            }
        }
        v[i] = x;
    }
    if (!planes) return;
    const unsigned h0 = Fp16::pack2(v[0], v[1]), h1 = Fp16::pack2(v[2], v[3]);
    const f32x2 u0 = Fp16::unpack2(h0), u1 = Fp16::unpack2(h1);
    const unsigned l0 = Fp16::pack2(v[0] - u0[0], v[1] - u0[1]), l1 = Fp16::pack2(v[2] - u1[0], v[3] - u1[1]);
    unsigned short* d = planes + row * ld + coff + c;
    if (c + 4 <= C) {
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + plane_stride) = u32x2{l0, l1};
    } else {                                                      // ragged tail (C = 1, 2, 3 ...): element stores
        const unsigned hh[2] = {h0, h1}, ll[2] = {l0, l1};
        for (int i = 0; c + i < C; ++i) {
            d[i] = (unsigned short)(hh[i >> 1] >> (16 * (i & 1)));
            d[plane_stride + i] = (unsigned short)(ll[i >> 1] >> (16 * (i & 1)));
        }
    }
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

extern "C" size_t um_nhwc_norm_workspace_bytes(int batch, int pixels, int channels) {
    if (batch <= 0 || pixels <= 0 || channels <= 0) return 0;
    const long nchunk = (pixels + NHWC_CHUNK_ROWS - 1) / NHWC_CHUNK_ROWS;
    return (size_t)((long)batch * nchunk * 3 * channels + (long)batch * 2 * channels) * sizeof(float);
}

extern "C" size_t um_conv_stats_bytes(int batch, int parts, int channels) {
    if (batch <= 0 || parts <= 0 || channels <= 0) return 0;
    return (size_t)((long)batch * parts * 3 * channels) * sizeof(float);
}

static bool nhwc_channels_ok(int c) { return c > 0 && c % 8 == 0 && c <= 256; }

extern "C" int um_nhwc_instance_norm(const float* x, const float* shortcut, const void* shortcut_planes, void* planes_out, float* f32_out, int batch,
                                     int pixels, int channels, float eps, int normalize, int relu, const float* conv_stats, int conv_stats_parts,
                                     void* workspace, size_t workspace_bytes, int mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || (!planes_out && !f32_out) || batch <= 0 || pixels <= 0 || !nhwc_channels_ok(channels) || (mode != 0 && mode != 1)) {
        um_set_error("um_nhwc_instance_norm: bad argument (batch=%d pixels=%d channels=%d: channels must be a multiple of 8, <= 256)",
                     batch, pixels, channels);
        return -1;
    }
    const long rows = (long)batch * pixels;
    const int nchunk = (pixels + NHWC_CHUNK_ROWS - 1) / NHWC_CHUNK_ROWS;
    float* partial = (float*)workspace;
    float* stats = nullptr;
    ScopedKernelTimer timer(UM_K_INSTANCE_NORM, stream);
    if (normalize) {
        if (!workspace || workspace_bytes < um_nhwc_norm_workspace_bytes(batch, pixels, channels)) {
            um_set_error("um_nhwc_instance_norm: workspace too small");
            return -3;
        }
        stats = partial + (long)batch * nchunk * 3 * channels;
        if (conv_stats) {                                          // per-tile statistics from the producing convolution
            if (conv_stats_parts <= 0) {
                um_set_error("um_nhwc_instance_norm: conv_stats needs conv_stats_parts = um_conv_stats_parts() of the convolution");
                return -1;
            }
            hipLaunchKernelGGL(nhwc_stats_finalize_kernel, dim3(channels / 8, batch), dim3(256), 0, stream, conv_stats, stats, pixels, channels,
                               conv_stats_parts, 0, eps);
        } else {
            hipLaunchKernelGGL(nhwc_stats_kernel, dim3(nchunk, batch), dim3(256), 0, stream, x, partial, pixels, channels, nchunk);
            hipLaunchKernelGGL(nhwc_stats_finalize_kernel, dim3(channels / 8, batch), dim3(256), 0, stream, partial, stats, pixels, channels,
                               nchunk, NHWC_CHUNK_ROWS, eps);
        }
    }
    const long total = (rows + (planes_out ? 1 : 0)) * (channels / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (mode == 0)
        hipLaunchKernelGGL((nhwc_apply_kernel<Fp16, 2>), grid, dim3(256), 0, stream, x, stats, shortcut,
                           (const unsigned short*)shortcut_planes, (unsigned short*)planes_out, f32_out, rows, pixels, channels, relu);
    else
        hipLaunchKernelGGL((nhwc_apply_kernel<Bf16, 1>), grid, dim3(256), 0, stream, x, stats, shortcut,
                           (const unsigned short*)shortcut_planes, (unsigned short*)planes_out, f32_out, rows, pixels, channels, relu);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        um_set_error("um_nhwc_instance_norm: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int um_nchw_to_nhwc(const float* x, void* planes_out, float* f32_out, int batch, int channels, int pixels, int mode,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || (!planes_out && !f32_out) || batch <= 0 || pixels <= 0 || channels <= 0 || channels % 8 != 0 || channels > 256 ||
        (mode != 0 && mode != 1)) {
        um_set_error("um_nchw_to_nhwc: bad argument (batch=%d channels=%d pixels=%d: channels must be a multiple of 8, <= 256)",
                     batch, channels, pixels);
        return -1;
    }
    const long rows = (long)batch * pixels;
    const dim3 grid((pixels + 63) / 64, batch);
    ScopedKernelTimer timer(UM_K_INSTANCE_NORM, stream);
    if (mode == 0)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<Fp16, 2>), grid, dim3(256), 0, stream, x, (unsigned short*)planes_out, f32_out,
                           pixels, channels, rows);
    else
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<Bf16, 1>), grid, dim3(256), 0, stream, x, (unsigned short*)planes_out, f32_out,
                           pixels, channels, rows);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        um_set_error("um_nchw_to_nhwc: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int um_nhwc_gate(int mode, const float* src, int src_ld, const float* zr, float* hbuf, void* planes_out, int ld, int coff,
                            long plane_rows, long rows, int channels, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool ok = rows > 0 && channels > 0 && mode >= 0 && mode <= 2 && (mode == 1 || src) && (mode == 0 || (zr && hbuf)) &&
                    (mode == 2 || planes_out) &&
                    (!planes_out || (ld >= coff + channels && plane_rows >= rows && (channels % 4 != 0 || coff % 4 == 0)));
    if (!ok) {
        um_set_error("um_nhwc_gate: bad argument (mode=%d rows=%ld channels=%d ld=%d coff=%d)", mode, rows, channels, ld, coff);
        return -1;
    }
    const long total = rows * ((channels + 3) / 4);
    ScopedKernelTimer timer(UM_K_INSTANCE_NORM, stream);
    hipLaunchKernelGGL(nhwc_gate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mode, src, src_ld, zr, hbuf,
                       (unsigned short*)planes_out, plane_rows * ld, ld, coff, rows, channels);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        um_set_error("um_nhwc_gate: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
