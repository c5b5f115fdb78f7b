// Local-window kernels: short fp32 dot products against a handful of neighbouring tokens.   gfx950 / wave64
//
//   um_local_corr_softmax     unimatch/matching.py:39-83, 154-200   (2r+1)^2 / (2r+1) taps, softmax, E[offset]
//   um_local_corr_with_flow   unimatch/matching.py:86-123           (2r+1)^2 bilinear taps at p + d + flow(p)
//   um_prop_local_attn        unimatch/attention.py:217-253         (2r+1)^2 zero-padded self-attention taps
//   um_depth_corr_softmax     unimatch/matching.py:203-282          D plane-sweep candidates, soft-argmin
//
// These are gather + short-reduction kernels (arithmetic intensity of a few flop/byte): they are bounded by
// L2/HBM bandwidth and VALU issue, not by the matrix cores, so they are NOT reshaped into GEMMs.  The
// reference materialises a [B, L, C, taps] gathered tensor for each of them (1.0-1.3 GB per pair per call
// for the cost volume); here nothing but the [taps] logits of one pixel ever exists, in registers.
//
// Common structure: one wavefront per pixel, lanes = 16 tap slots x 4 channel quarters.  A lane holds 32
// channels of f0(p) in registers (interleaved in groups of 4: see load32), reads the same channels of the
// neighbour token in 8 loads, each of which covers 64 contiguous bytes per slot, does 32 FMAs and the 4
// quarter sums are combined with two DPP-class shuffles.  Taps are processed 16 per round.  Softmax statistics are reduced across the 16 slots with
// four more shuffles.  Token-major feature layout [B, L, 128] fp32.
#include "common.h"
#include "timing.h"

#define LOCAL_MAX_ROUNDS 8      // up to 128 taps / depth candidates

// A 128-channel fp32 row (512 B) is shared by the 4 lanes of a tap slot.  Lane quarter q owns channels 16 i + 4 q .. + 3
// (i = 0..7): callers pass `row + 4 * quarter`, and load i of the four lanes covers 64 CONTIGUOUS bytes -- one request to
// the L1 per slot and load.  (Round 1 gave each lane 32 consecutive channels: four 16-byte pieces 128 B apart per slot and
// load, i.e. 64 cache lines touched by every wave instruction; these gather kernels were bound by exactly that.)
__device__ __forceinline__ void load32(f32x4 (&r)[8], const float* p) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = reinterpret_cast<const f32x4*>(p)[4 * i];
}

__device__ __forceinline__ float dot32(const f32x4 (&a)[8], const float* p) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 b = reinterpret_cast<const f32x4*>(p)[4 * i];
        acc = __builtin_fmaf(a[i][0], b[0], acc);
        acc = __builtin_fmaf(a[i][1], b[1], acc);
        acc = __builtin_fmaf(a[i][2], b[2], acc);
        acc = __builtin_fmaf(a[i][3], b[3], acc);
    }
    return acc;
}

__device__ __forceinline__ float quad_sum(float v) {      // sum over the 4 channel quarters (lanes 4k..4k+3)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}
__device__ __forceinline__ float slot_max(float v) {      // reduce over the 16 tap slots (lane bits 2..5)
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float slot_sum(float v) {
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// ------------------------------------------------------------------------------------------------------
// K3: local correlation + softmax + expected offset
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void local_corr_softmax_kernel(const float* __restrict__ f0,
                                                                 const float* __restrict__ f1,
                                                                 float* __restrict__ out, int batch, int h, int w,
                                                                 int radius, int one_d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane >> 2, quarter = lane & 3;
    const int L = h * w;
    const long total = (long)batch * L;
    const int kw = 2 * radius + 1;
    const int ntaps = one_d ? kw : kw * kw;
    const int rounds = (ntaps + 15) >> 4;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    for (long pid = (long)blockIdx.x * 4 + wave; pid < total; pid += (long)gridDim.x * 4) {
        const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
        const int y = p / w, x = p - y * w;
        f32x4 a[8];
        load32(a, f0 + pid * UM_CHANNELS + 4 * quarter);
        float logit[LOCAL_MAX_ROUNDS];
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < LOCAL_MAX_ROUNDS; ++r) {
            logit[r] = -3.0e38f;
            if (r < rounds) {
                const int t = r * 16 + slot;
                const int dy = one_d ? 0 : t / kw - radius;
                const int dx = (one_d ? t : t % kw) - radius;
                const int yy = y + dy, xx = x + dx;
                const bool tap = t < ntaps;
                const bool ok = tap && yy >= 0 && yy < h && xx >= 0 && xx < w;
                float d = 0.f;
                if (ok) d = dot32(a, f1 + ((long)b * L + yy * w + xx) * UM_CHANNELS + 4 * quarter);
                d = quad_sum(d);
                // out-of-image taps take part with logit -1e9 (matching.py:73); slots beyond ntaps do not exist
                logit[r] = tap ? (ok ? d * scale : -1.0e9f) : -3.0e38f;
                mx = fmaxf(mx, logit[r]);
            }
        }
        mx = slot_max(mx);
        float sp = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int r = 0; r < LOCAL_MAX_ROUNDS; ++r) {
            if (r < rounds) {
                const int t = r * 16 + slot;
                const float e = t < ntaps ? __expf(logit[r] - mx) : 0.f;
                const int dy = one_d ? 0 : t / kw - radius;
                const int dx = (one_d ? t : t % kw) - radius;
                sp += e;
                sx += e * (float)dx;
                sy += e * (float)dy;
            }
        }
        sp = slot_sum(sp);
        sx = slot_sum(sx);
        sy = slot_sum(sy);
        if (lane == 0) {
            if (one_d) {
                out[pid] = -(sx / sp);                       // disparity residual = -flow_x (matching.py:198)
            } else {
                out[((long)b * 2 + 0) * L + p] = sx / sp;
                out[((long)b * 2 + 1) * L + p] = sy / sp;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// K4: local cost volume at flow-displaced positions (input of the regression refinement)
// All (2r+1)^2 taps of a pixel share one fractional offset and bilinear sampling is linear in f1, so the
// kernel computes the (2r+2)^2 integer-position dot products around floor(p + flow) once and blends 4 of
// them per tap: 100 dots instead of 81 x 4 gathers for r = 4.  Out-of-image corners contribute 0 (zeros pad).
// A wave walks 16 consecutive pixels and stages its [taps][16] outputs in LDS so that the [B, taps, h, w]
// volume is written in 64-byte runs.
// ------------------------------------------------------------------------------------------------------
#define K4_PIX 16
#define K4_MAX_TAPS 81
#define K4_MAX_DOTS 100
__global__ __launch_bounds__(256) void local_corr_with_flow_kernel(const float* __restrict__ f0,
                                                                   const float* __restrict__ f1,
                                                                   const float* __restrict__ flow,
                                                                   float* __restrict__ cost, int batch, int h, int w,
                                                                   int radius, unsigned short* __restrict__ planes,
                                                                   long plane_stride, int ld) {
    __shared__ float dots_s[4][K4_MAX_DOTS + 4];
    __shared__ float dots4_s[4][4][K4_MAX_DOTS + 4];
    __shared__ float tile_s[4][K4_MAX_TAPS][K4_PIX + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane >> 2, quarter = lane & 3;
    const int L = h * w;
    const long total = (long)batch * L;
    const int kw = 2 * radius + 1, n1 = kw + 1;
    const int ntaps = kw * kw, ndots = n1 * n1;
    const int rounds = (ndots + 15) >> 4;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    const long nblocks = (total + K4_PIX - 1) / K4_PIX;
    for (long pb = (long)blockIdx.x * 4 + wave; pb < nblocks; pb += (long)gridDim.x * 4) {
        const long p0 = pb * K4_PIX;
        // Four consecutive pixels at a time: their (2r+2)^2 integer neighbourhoods overlap almost completely when the flow
        // is locally smooth, and the kernel is bound by the L2 -> CU traffic of those neighbour rows (51 KB per pixel), so
        // every neighbour row inside the group's bounding box is fetched ONCE and dotted with all four f0 vectors it
        // belongs to (~3x fewer bytes).  The dot products themselves are unchanged (same order of operations).  Groups
        // whose windows do not overlap enough (motion boundaries, row ends) take the one-pixel-at-a-time path.
        for (int j0 = 0; j0 < K4_PIX; j0 += 4) {
            if (p0 + j0 >= total) break;
            int gb[4], gbx[4], gby[4];
            float gwx[4], gwy[4];
            bool gok[4];
            int ux0 = 1 << 30, ux1 = -(1 << 30), uy0 = 1 << 30, uy1 = -(1 << 30);
            bool same_image = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long pid = p0 + j0 + i;
                gok[i] = pid < total;
                const long pc = gok[i] ? pid : p0 + j0;
                const int b = (int)(pc / L), p = (int)(pc - (long)b * L);
                const int y = p / w, x = p - y * w;
                const float px = (float)x + flow[((long)b * 2 + 0) * L + p];
                const float py = (float)y + flow[((long)b * 2 + 1) * L + p];
                const float fbx = floorf(px), fby = floorf(py);
                gwx[i] = px - fbx;
                gwy[i] = py - fby;
                // clamp far-away bases so the int conversion is defined; such samples are all-zero anyway
                gbx[i] = (int)fminf(fmaxf(fbx, -32768.f), 32768.f);
                gby[i] = (int)fminf(fmaxf(fby, -32768.f), 32768.f);
                gb[i] = b;
                if (gok[i]) {
                    ux0 = min(ux0, gbx[i] - radius);
                    ux1 = max(ux1, gbx[i] - radius + n1 - 1);
                    uy0 = min(uy0, gby[i] - radius);
                    uy1 = max(uy1, gby[i] - radius + n1 - 1);
                    same_image = same_image && b == gb[0];
                }
            }
            const int nw = ux1 - ux0 + 1, nh = uy1 - uy0 + 1;
            const bool blocked = same_image && nw > 0 && nh > 0 && (long)nw * nh <= 2 * ndots;   // <= half the separate loads
            if (blocked) {
                f32x4 a4[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) load32(a4[i], f0 + (gok[i] ? p0 + j0 + i : p0 + j0) * UM_CHANNELS + 4 * quarter);
                const int ncand = nw * nh;
                const float* f1b = f1 + ((long)gb[0] * L) * UM_CHANNELS + 4 * quarter;
                for (int r = 0; r * 16 < ncand; ++r) {
                    const int t = r * 16 + slot;
                    const int cy = t / nw, cx = t - cy * nw;
                    const int yy = uy0 + cy, xx = ux0 + cx;
                    const bool ok = t < ncand && yy >= 0 && yy < h && xx >= 0 && xx < w;
                    f32x4 bv[8];
                    if (ok) load32(bv, f1b + (long)(yy * w + xx) * UM_CHANNELS);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ix = xx - (gbx[i] - radius), iy = yy - (gby[i] - radius);
                        const bool mine = t < ncand && gok[i] && (unsigned)ix < (unsigned)n1 && (unsigned)iy < (unsigned)n1;
                        float d = 0.f;
                        if (ok && mine) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                d = __builtin_fmaf(a4[i][q][0], bv[q][0], d);
                                d = __builtin_fmaf(a4[i][q][1], bv[q][1], d);
                                d = __builtin_fmaf(a4[i][q][2], bv[q][2], d);
                                d = __builtin_fmaf(a4[i][q][3], bv[q][3], d);
                            }
                        }
                        d = quad_sum(d);
                        if (quarter == 0 && mine) dots4_s[wave][i][iy * n1 + ix] = d;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!gok[i]) continue;
                    const float wx = gwx[i], wy = gwy[i];
                    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy);
                    const float w10 = (1.f - wx) * wy, w11 = wx * wy;
                    for (int k = lane; k < ntaps; k += 64) {
                        const int ty = k / kw, tx = k - ty * kw;
                        const float* dd = &dots4_s[wave][i][ty * n1 + tx];
                        const float v = w00 * dd[0] + w01 * dd[1] + w10 * dd[n1] + w11 * dd[n1 + 1];
                        tile_s[wave][k][j0 + i] = v * scale;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        for (int j = j0; j < j0 + 4; ++j) {
            const long pid = p0 + j;
            if (pid >= total) break;
            const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
            const int y = p / w, x = p - y * w;
            const float px = (float)x + flow[((long)b * 2 + 0) * L + p];
            const float py = (float)y + flow[((long)b * 2 + 1) * L + p];
            const float fbx = floorf(px), fby = floorf(py);
            const float wx = px - fbx, wy = py - fby;
            // clamp far-away bases so the int conversion is defined; such samples are all-zero anyway
            const int bx = (int)fminf(fmaxf(fbx, -32768.f), 32768.f);
            const int by = (int)fminf(fmaxf(fby, -32768.f), 32768.f);
            f32x4 a[8];
            load32(a, f0 + pid * UM_CHANNELS + 4 * quarter);
            for (int r = 0; r < rounds; ++r) {
                const int t = r * 16 + slot;
                const int iy = t / n1, ix = t - iy * n1;
                const int yy = by + iy - radius, xx = bx + ix - radius;
                const bool ok = t < ndots && yy >= 0 && yy < h && xx >= 0 && xx < w;
                float d = 0.f;
                if (ok) d = dot32(a, f1 + ((long)b * L + yy * w + xx) * UM_CHANNELS + 4 * quarter);
                d = quad_sum(d);
                if (quarter == 0 && t < ndots) dots_s[wave][t] = d;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy);
            const float w10 = (1.f - wx) * wy, w11 = wx * wy;
            for (int k = lane; k < ntaps; k += 64) {
                const int ty = k / kw, tx = k - ty * kw;
                const float* dd = &dots_s[wave][ty * n1 + tx];
                const float v = w00 * dd[0] + w01 * dd[1] + w10 * dd[n1] + w11 * dd[n1 + 1];
                tile_s[wave][k][j] = v * scale;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        }   // groups of four pixels
        if (planes) {
            // channels-last operand planes for um_conv2d_ex (the motion encoder's 1x1 convolution reads them directly: the
            // [B, taps, h, w] fp32 volume never exists): pixel row = ld channels, taps first, zeros up to ld
            const int pairs = ld >> 1;
            for (int idx = lane; idx < pairs * K4_PIX; idx += 64) {
                const int j = idx / pairs, c = (idx - j * pairs) * 2;
                const long pid = p0 + j;
                if (pid < total) {
                    const float v0 = c < ntaps ? tile_s[wave][c][j] : 0.f, v1 = c + 1 < ntaps ? tile_s[wave][c + 1][j] : 0.f;
                    const unsigned hh = Fp16::pack2(v0, v1);
                    const f32x2 u = Fp16::unpack2(hh);
                    *reinterpret_cast<unsigned*>(planes + pid * ld + c) = hh;
                    *reinterpret_cast<unsigned*>(planes + plane_stride + pid * ld + c) = Fp16::pack2(v0 - u[0], v1 - u[1]);
                }
            }
        } else
        // write the [taps][16] tile: 16 consecutive pixels of one tap are contiguous in the volume
        for (int idx = lane; idx < ntaps * K4_PIX; idx += 64) {
            const int k = idx / K4_PIX, j = idx - k * K4_PIX;
            const long pid = p0 + j;
            if (pid < total) {
                const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
                cost[((long)b * ntaps + k) * L + p] = tile_s[wave][k][j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------
// K6: local self-attention propagation.  Out-of-image neighbours have key 0 -> logit 0 and value 0, and
// they DO take part in the softmax (zero padding of F.unfold, attention.py:234-246).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prop_local_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ value,
                                                              float* __restrict__ out, int batch, int h, int w,
                                                              int vch, int radius) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane >> 2, quarter = lane & 3;
    const int L = h * w;
    const long total = (long)batch * L;
    const int kw = 2 * radius + 1;
    const int ntaps = kw * kw;
    const int rounds = (ntaps + 15) >> 4;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    for (long pid = (long)blockIdx.x * 4 + wave; pid < total; pid += (long)gridDim.x * 4) {
        const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
        const int y = p / w, x = p - y * w;
        f32x4 a[8];
        load32(a, q + pid * UM_CHANNELS + 4 * quarter);
        float logit[LOCAL_MAX_ROUNDS], v0[LOCAL_MAX_ROUNDS], v1[LOCAL_MAX_ROUNDS];
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < LOCAL_MAX_ROUNDS; ++r) {
            logit[r] = -3.0e38f;
            v0[r] = v1[r] = 0.f;
            if (r < rounds) {
                const int t = r * 16 + slot;
                const int dy = t / kw - radius, dx = t % kw - radius;
                const int yy = y + dy, xx = x + dx;
                const bool tap = t < ntaps;
                const bool ok = tap && yy >= 0 && yy < h && xx >= 0 && xx < w;
                float d = 0.f;
                if (ok) {
                    d = dot32(a, k + ((long)b * L + yy * w + xx) * UM_CHANNELS + 4 * quarter);
                    v0[r] = value[((long)b * vch) * L + yy * w + xx];
                    if (vch == 2) v1[r] = value[((long)b * vch + 1) * L + yy * w + xx];
                }
                d = quad_sum(d);
                logit[r] = tap ? d * scale : -3.0e38f;     // d == 0 for padded neighbours
                mx = fmaxf(mx, logit[r]);
            }
        }
        mx = slot_max(mx);
        float sp = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < LOCAL_MAX_ROUNDS; ++r) {
            if (r < rounds) {
                const float e = (r * 16 + slot) < ntaps ? __expf(logit[r] - mx) : 0.f;
                sp += e;
                s0 += e * v0[r];
                s1 += e * v1[r];
            }
        }
        sp = slot_sum(sp);
        s0 = slot_sum(s0);
        s1 = slot_sum(s1);
        if (lane == 0) {
            out[((long)b * vch) * L + p] = s0 / sp;
            if (vch == 2) out[((long)b * vch + 1) * L + p] = s1 / sp;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// K7: plane-sweep depth correlation.  For every inverse-depth candidate the pixel is back-projected,
// moved by the relative pose, re-projected (same operation order as matching.py:259-270), feature1 is
// sampled bilinearly with zero padding and correlated with feature0; softmax over candidates.
// cam: per sample 30 floats = Kinv[9] | R[9] | t[3] | K[9], row major.
// The [B, C, D, h, w] warped volume of the reference (2.5 GB at B=16, 480x640) never exists.
// ------------------------------------------------------------------------------------------------------
// One pixel by plain gathers (the wave's lanes = 16 candidate slots x 4 channel quarters, four bilinear corners of every candidate
// fetched from f1: 4 x 512 B per candidate).  Serves launches with more than 64 candidates and the pixels whose candidates' corner
// set does not fit the box table of depth_corr_softmax_box_kernel.  Returns the pixel's result on every lane.
__device__ __forceinline__ float depth_pixel_gather(const float* __restrict__ f0, const float* __restrict__ f1, const float* cm,
                                                    const float* __restrict__ cand, long pid, int b, int h, int w, int nd,
                                                    int from_argmax, float q0, float q1, float q2, int lane) {
    const int slot = lane >> 2, quarter = lane & 3;
    const int L = h * w;
    const int rounds = (nd + 15) >> 4;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    f32x4 a[8];
    load32(a, f0 + pid * UM_CHANNELS + 4 * quarter);
    float logit[LOCAL_MAX_ROUNDS], cv[LOCAL_MAX_ROUNDS];
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < LOCAL_MAX_ROUNDS; ++r) {
        logit[r] = -3.0e38f;
        cv[r] = 0.f;
        if (r < rounds) {
            const int t = r * 16 + slot;
            const bool tap = t < nd;
            const float ci = cand[tap ? t : 0];
            cv[r] = ci;
            const float depth = 1.0f / ci;
            const float X = q0 * depth + cm[18], Y = q1 * depth + cm[19], Z = q2 * depth + cm[20];
            const float u = cm[21] * X + cm[22] * Y + cm[23] * Z;
            const float v = cm[24] * X + cm[25] * Y + cm[26] * Z;
            const float zz = fmaxf(cm[27] * X + cm[28] * Y + cm[29] * Z, 1e-3f);
            const float sx = fminf(fmaxf(u / zz, -1.0e6f), 1.0e6f);
            const float sy = fminf(fmaxf(v / zz, -1.0e6f), 1.0e6f);
            const float fx0 = floorf(sx), fy0 = floorf(sy);
            const float wx = sx - fx0, wy = sy - fy0;
            const int x0 = (int)fx0, y0 = (int)fy0;
            float d = 0.f;
            if (tap) {
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const int yy = y0 + cy, xx = x0 + cx;
                        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                            const float wt = (cx ? wx : 1.f - wx) * (cy ? wy : 1.f - wy);
                            d += wt * dot32(a, f1 + ((long)b * L + yy * w + xx) * UM_CHANNELS + 4 * quarter);
                        }
                    }
            }
            d = quad_sum(d);
            logit[r] = tap ? d * scale : -3.0e38f;
            mx = fmaxf(mx, logit[r]);
        }
    }
    mx = slot_max(mx);
    float sp = 0.f, sc = 0.f;
    float best = -3.0e38f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < LOCAL_MAX_ROUNDS; ++r) {
        if (r < rounds) {
            const int t = r * 16 + slot;
            const float e = t < nd ? __expf(logit[r] - mx) : 0.f;
            sp += e;
            sc += e * cv[r];
            if (t < nd && logit[r] > best) { best = logit[r]; besti = t; }
        }
    }
    sp = slot_sum(sp);
    sc = slot_sum(sc);
    float res = sc / sp;
    if (from_argmax) {   // first index attaining the maximum
        const int cand_i = (best == mx) ? besti : 0x7fffffff;
        int mi = cand_i;
        mi = min(mi, __shfl_xor(mi, 4));
        mi = min(mi, __shfl_xor(mi, 8));
        mi = min(mi, __shfl_xor(mi, 16));
        mi = min(mi, __shfl_xor(mi, 32));
        res = cand[mi];
    }
    return res;
}

__global__ __launch_bounds__(256) void depth_corr_softmax_kernel(const float* __restrict__ f0,
                                                                 const float* __restrict__ f1,
                                                                 const float* __restrict__ cam,
                                                                 const float* __restrict__ cand,
                                                                 float* __restrict__ out, int batch, int h, int w,
                                                                 int nd, int from_argmax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = h * w;
    const long total = (long)batch * L;
    for (long pid = (long)blockIdx.x * 4 + wave; pid < total; pid += (long)gridDim.x * 4) {
        const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
        const int y = p / w, x = p - y * w;
        const float* cm = cam + (long)b * 30;
        const float gx = (float)x, gy = (float)y;
        // ray = R (Kinv [x y 1]^T)
        const float r0 = cm[0] * gx + cm[1] * gy + cm[2];
        const float r1 = cm[3] * gx + cm[4] * gy + cm[5];
        const float r2 = cm[6] * gx + cm[7] * gy + cm[8];
        const float q0 = cm[9] * r0 + cm[10] * r1 + cm[11] * r2;
        const float q1 = cm[12] * r0 + cm[13] * r1 + cm[14] * r2;
        const float q2 = cm[15] * r0 + cm[16] * r1 + cm[17] * r2;
        const float res = depth_pixel_gather(f0, f1, cm, cand, pid, b, h, w, nd, from_argmax, q0, q1, q2, lane);
        if (lane == 0) out[pid] = res;
    }
}

// Round 4: the plane sweep by its integer neighbourhood (as the cost volume, matching.py:86-123, has been since round 1).  The
// candidates of one pixel walk along its epipolar line in sub-pixel steps (config 5: 64 inverse depths over ~16 feature cells), so
// their 4 x 64 bilinear corners are a few dozen DISTINCT f1 rows -- the gather kernel fetched every corner of every candidate: 128 KB
// per pixel from L2, 9.8 GB per launch at config 5 (17 TB/s: the kernel sat on the L2 -> CU bandwidth; 314 MB of it missed to HBM,
// profiles/r04_gather_rooflines.txt).  Here: lane = candidate (<= 64); the wave takes the bounding box of the corners that can
// touch the image, dots f0(p) with every f1 row of the box ONCE (16 rows per round, 4 lanes per row as before), keeps the dots in a
// wave-private LDS table, and every candidate blends its four corners from the table (a dot product is linear in f1, so the blend of
// the dots is the dot with the blended row; summation order differs from the gather form by fp32 rounding only).  A box of more
// than DEPTH_BOX rows (a camera move with a long diagonal epipolar segment) takes the gather path for that pixel.
#define DEPTH_BOX 160
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void depth_corr_softmax_box_kernel(const float* __restrict__ f0,
                                                                     const float* __restrict__ f1,
                                                                     const float* __restrict__ cam,
                                                                     const float* __restrict__ cand,
                                                                     float* __restrict__ out, int batch, int h, int w,
                                                                     int nd, int from_argmax) {
    __shared__ float tab_all[4][DEPTH_BOX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane >> 2, quarter = lane & 3;
    float* tab = tab_all[wave];
    const int L = h * w;
    const long total = (long)batch * L;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    for (long pid = (long)blockIdx.x * 4 + wave; pid < total; pid += (long)gridDim.x * 4) {
        const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
        const int y = p / w, x = p - y * w;
        const float* cm = cam + (long)b * 30;
        const float gx = (float)x, gy = (float)y;
        const float r0 = cm[0] * gx + cm[1] * gy + cm[2];
        const float r1 = cm[3] * gx + cm[4] * gy + cm[5];
        const float r2 = cm[6] * gx + cm[7] * gy + cm[8];
        const float q0 = cm[9] * r0 + cm[10] * r1 + cm[11] * r2;
        const float q1 = cm[12] * r0 + cm[13] * r1 + cm[14] * r2;
        const float q2 = cm[15] * r0 + cm[16] * r1 + cm[17] * r2;
        // this lane's candidate: the same projection arithmetic, in the same order, as the gather form
        const bool tap = lane < nd;
        const float ci = cand[tap ? lane : 0];
        const float depth = 1.0f / ci;
        const float X = q0 * depth + cm[18], Y = q1 * depth + cm[19], Z = q2 * depth + cm[20];
        const float u = cm[21] * X + cm[22] * Y + cm[23] * Z;
        const float v = cm[24] * X + cm[25] * Y + cm[26] * Z;
        const float zz = fmaxf(cm[27] * X + cm[28] * Y + cm[29] * Z, 1e-3f);
        const float sx = fminf(fmaxf(u / zz, -1.0e6f), 1.0e6f);
        const float sy = fminf(fmaxf(v / zz, -1.0e6f), 1.0e6f);
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const float wx = sx - fx0, wy = sy - fy0;
        const int x0 = (int)fx0, y0 = (int)fy0;
        // bounding box of the corners that can lie inside the image: a candidate whose x0 is outside [-1, w-1] has no corner in it
        const int cx0 = min(max(x0, -1), w - 1), cy0 = min(max(y0, -1), h - 1);
        const int bx0 = wave_min_i(tap ? cx0 : 0x7fffffff), bx1 = wave_max_i(tap ? cx0 + 1 : -0x7fffffff);
        const int by0 = wave_min_i(tap ? cy0 : 0x7fffffff), by1 = wave_max_i(tap ? cy0 + 1 : -0x7fffffff);
        const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
        const int npos = bw * bh;
        float res;
        if (npos > DEPTH_BOX) {                              // wave-uniform
            res = depth_pixel_gather(f0, f1, cm, cand, pid, b, h, w, nd, from_argmax, q0, q1, q2, lane);
        } else {
            f32x4 a[8];
            load32(a, f0 + pid * UM_CHANNELS + 4 * quarter);
            const float rbw = 1.0f / (float)bw;
            for (int i0 = 0; i0 < npos; i0 += 16) {
                const int i = i0 + slot;
                int ry = (int)((float)i * rbw);               // i / bw for 0 <= i < 160, corrected below
                ry += ((ry + 1) * bw <= i) ? 1 : 0;
                ry -= (ry * bw > i) ? 1 : 0;
                const int py = by0 + ry, px = bx0 + (i - ry * bw);
                float d = 0.f;
                if (i < npos && py >= 0 && py < h && px >= 0 && px < w)
                    d = dot32(a, f1 + ((long)b * L + py * w + px) * UM_CHANNELS + 4 * quarter);
                d = quad_sum(d);
                if (quarter == 0 && i < npos) tab[i] = d;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is complete (same wave: LDS executes in order)
            __builtin_amdgcn_wave_barrier();
            float d = 0.f;
            if (tap) {
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const int yy = y0 + cy, xx = x0 + cx;
                        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                            const float wt = (cx ? wx : 1.f - wx) * (cy ? wy : 1.f - wy);
                            d += wt * tab[(yy - by0) * bw + (xx - bx0)];
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every lane has read before the next pixel overwrites the table
            __builtin_amdgcn_wave_barrier();
            const float logit = tap ? d * scale : -3.0e38f;
            const float mx = wave_max_f(logit);
            const float e = tap ? __expf(logit - mx) : 0.f;
            const float sp = wave_sum_f(e), sc = wave_sum_f(e * ci);
            res = sc / sp;
            if (from_argmax) {                                   // first index attaining the maximum
                const int mi = wave_min_i((tap && logit == mx) ? lane : 0x7fffffff);
                res = cand[mi];
            }
        }
        if (lane == 0) out[pid] = res;
    }
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

static int local_check(const void* a, const void* b, const void* c, int batch, int h, int w, int channels) {
    if (!a || !b || !c || batch <= 0 || h <= 0 || w <= 0) {
        um_set_error("null pointer or non-positive size (batch=%d h=%d w=%d)", batch, h, w);
        return -1;
    }
    if (channels != UM_CHANNELS) {
        um_set_error("channels=%d unsupported (the library is built for %d)", channels, UM_CHANNELS);
        return -1;
    }
    return 0;
}

static unsigned pixel_grid_blocks(long pixels) {
    long blocks = (pixels + 3) / 4;
    const long cap = 256 * 16;           // 16 workgroups per CU, grid-stride beyond
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int um_local_corr_softmax(const float* f0, const float* f1, float* out, int batch, int h, int w,
                                     int channels, int radius, int one_d, void* stream) {
    if (int e = local_check(f0, f1, out, batch, h, w, channels)) return e;
    const int kw = 2 * radius + 1;
    if (radius < 1 || (one_d ? kw : kw * kw) > 16 * LOCAL_MAX_ROUNDS) {
        um_set_error("radius=%d unsupported (at most %d taps)", radius, 16 * LOCAL_MAX_ROUNDS);
        return -4;
    }
    ScopedKernelTimer timer(UM_K_LOCAL_CORR, (hipStream_t)stream);
    um_census_hit(UM_V_K3_VALU);
    hipLaunchKernelGGL(local_corr_softmax_kernel, dim3(pixel_grid_blocks((long)batch * h * w)), dim3(256), 0,
                       (hipStream_t)stream, f0, f1, out, batch, h, w, radius, one_d);
    return (int)hipGetLastError();
}

extern "C" int um_local_corr_with_flow(const float* f0, const float* f1, const float* flow, float* cost, int batch,
                                       int h, int w, int channels, int radius, void* stream) {
    if (int e = local_check(f0, f1, cost, batch, h, w, channels)) return e;
    if (!flow) {
        um_set_error("flow is null");
        return -1;
    }
    if (radius < 1 || (2 * radius + 1) * (2 * radius + 1) > K4_MAX_TAPS) {
        um_set_error("radius=%d unsupported (at most %d taps)", radius, K4_MAX_TAPS);
        return -4;
    }
    const long nblocks = ((long)batch * h * w + K4_PIX - 1) / K4_PIX;
    ScopedKernelTimer timer(UM_K_COST_VOLUME, (hipStream_t)stream);
    um_census_hit(UM_V_K4_VALU);
    hipLaunchKernelGGL(local_corr_with_flow_kernel, dim3(pixel_grid_blocks(nblocks)), dim3(256), 0,
                       (hipStream_t)stream, f0, f1, flow, cost, batch, h, w, radius, (unsigned short*)nullptr, 0L, 0);
    return (int)hipGetLastError();
}

extern "C" int um_local_corr_with_flow_planes(const float* f0, const float* f1, const float* flow, void* planes_out, int ld,
                                              long plane_rows, int batch, int h, int w, int channels, int radius, void* stream) {
    if (int e = local_check(f0, f1, planes_out, batch, h, w, channels)) return e;
    const int ntaps = (2 * radius + 1) * (2 * radius + 1);
    if (!flow || radius < 1 || ntaps > K4_MAX_TAPS || ld < ntaps || ld % 8 != 0 || plane_rows < (long)batch * h * w + 1) {
        um_set_error("um_local_corr_with_flow_planes: bad argument (radius=%d ld=%d rows=%ld)", radius, ld, plane_rows);
        return -1;
    }
    const long nblocks = ((long)batch * h * w + K4_PIX - 1) / K4_PIX;
    ScopedKernelTimer timer(UM_K_COST_VOLUME, (hipStream_t)stream);
    um_census_hit(UM_V_K4_VALU);
    hipLaunchKernelGGL(local_corr_with_flow_kernel, dim3(pixel_grid_blocks(nblocks)), dim3(256), 0, (hipStream_t)stream, f0, f1,
                       flow, (float*)nullptr, batch, h, w, radius, (unsigned short*)planes_out, plane_rows * ld, ld);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// local_correlation_with_flow with a tap DILATION (unimatch/matching.py:86-123, `dilation` argument; the reference's callers
// all pass 1, which the kernels above serve): taps at p + dilation * d_k + flow(p).  With dilation > 1 the taps of a pixel no
// longer share their bilinear corners, so every tap is sampled for itself (4 corner dots); wave per pixel, lanes = 16 tap
// slots x 4 channel quarters as everywhere in this file.  A correctness path for the reference's full signature, not a hot one.
__global__ __launch_bounds__(256) void local_corr_with_flow_dilated_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                                           const float* __restrict__ flow, float* __restrict__ cost,
                                                                           int batch, int h, int w, int radius, int dilation) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane >> 2, quarter = lane & 3;
    const int L = h * w, k = 2 * radius + 1, taps = k * k;
    const long total = (long)batch * L;
    const int rounds = (taps + 15) >> 4;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    for (long pid = (long)blockIdx.x * 4 + wave; pid < total; pid += (long)gridDim.x * 4) {
        const int b = (int)(pid / L), p = (int)(pid - (long)b * L);
        const int y = p / w, x = p - y * w;
        const float px = (float)x + flow[((long)b * 2) * L + p], py = (float)y + flow[((long)b * 2 + 1) * L + p];
        f32x4 a[8];
        load32(a, f0 + pid * UM_CHANNELS + 4 * quarter);
        for (int r = 0; r < rounds; ++r) {
            const int t = r * 16 + slot;
            const bool tap = t < taps;
            const int dy = (tap ? t : 0) / k - radius, dx = (tap ? t : 0) % k - radius;
            const float sx = px + (float)(dx * dilation), sy = py + (float)(dy * dilation);
            const float fx0 = floorf(sx), fy0 = floorf(sy);
            const float wx = sx - fx0, wy = sy - fy0;
            const int x0 = (int)fminf(fmaxf(fx0, -1.0e6f), 1.0e6f), y0 = (int)fminf(fmaxf(fy0, -1.0e6f), 1.0e6f);
            float d = 0.f;
            if (tap) {
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const int yy = y0 + cy, xx = x0 + cx;
                        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                            const float wt = (cx ? wx : 1.f - wx) * (cy ? wy : 1.f - wy);
                            d += wt * dot32(a, f1 + ((long)b * L + yy * w + xx) * UM_CHANNELS + 4 * quarter);
                        }
                    }
            }
            d = quad_sum(d);
            if (tap && quarter == 0) cost[((long)b * taps + t) * L + p] = d * scale;
        }
    }
}

extern "C" int um_local_corr_with_flow_dilated(const float* f0, const float* f1, const float* flow, float* cost, int batch, int h, int w,
                                               int channels, int radius, int dilation, void* stream) {
    if (dilation == 1) return um_local_corr_with_flow(f0, f1, flow, cost, batch, h, w, channels, radius, stream);
    if (int e = local_check(f0, f1, cost, batch, h, w, channels)) return e;
    if (!flow || dilation < 1) {
        um_set_error("um_local_corr_with_flow_dilated: null flow or dilation=%d < 1", dilation);
        return -1;
    }
    if (radius < 1 || (2 * radius + 1) * (2 * radius + 1) > K4_MAX_TAPS) {
        um_set_error("radius=%d unsupported (at most %d taps)", radius, K4_MAX_TAPS);
        return -4;
    }
    ScopedKernelTimer timer(UM_K_COST_VOLUME, (hipStream_t)stream);
    um_census_hit(UM_V_K4_VALU);
    hipLaunchKernelGGL(local_corr_with_flow_dilated_kernel, dim3(pixel_grid_blocks((long)batch * h * w)), dim3(256), 0, (hipStream_t)stream,
                       f0, f1, flow, cost, batch, h, w, radius, dilation);
    return (int)hipGetLastError();
}

extern "C" int um_prop_local_attn(const float* q, const float* k, const float* value, float* out, int batch, int h,
                                  int w, int channels, int value_channels, int radius, void* stream) {
    if (int e = local_check(q, k, out, batch, h, w, channels)) return e;
    if (!value || (value_channels != 1 && value_channels != 2)) {
        um_set_error("value_channels=%d: the reference propagates flow (2) or disparity/depth (1)", value_channels);
        return -1;
    }
    if (radius < 1 || (2 * radius + 1) * (2 * radius + 1) > 16 * LOCAL_MAX_ROUNDS) {
        um_set_error("radius=%d unsupported (at most %d taps)", radius, 16 * LOCAL_MAX_ROUNDS);
        return -4;
    }
    ScopedKernelTimer timer(UM_K_PROP_LOCAL, (hipStream_t)stream);
    hipLaunchKernelGGL(prop_local_attn_kernel, dim3(pixel_grid_blocks((long)batch * h * w)), dim3(256), 0,
                       (hipStream_t)stream, q, k, value, out, batch, h, w, value_channels, radius);
    return (int)hipGetLastError();
}

extern "C" int um_depth_corr_softmax(const float* f0, const float* f1, const float* cam, const float* candidates,
                                     float* out, int batch, int h, int w, int channels, int num_candidates,
                                     int from_argmax, void* stream) {
    if (int e = local_check(f0, f1, out, batch, h, w, channels)) return e;
    if (!cam || !candidates || num_candidates < 1 || num_candidates > 16 * LOCAL_MAX_ROUNDS) {
        um_set_error("num_candidates=%d unsupported (1..%d)", num_candidates, 16 * LOCAL_MAX_ROUNDS);
        return -4;
    }
    ScopedKernelTimer timer(UM_K_DEPTH_CORR, (hipStream_t)stream);
    static const bool no_box = um_debug_env("UM_DEPTH_NO_BOX") != nullptr;     // A/B switch (diagnostic builds)
    if (num_candidates <= 64 && !no_box)                       // lane = candidate: the integer-neighbourhood form
        hipLaunchKernelGGL(depth_corr_softmax_box_kernel, dim3(pixel_grid_blocks((long)batch * h * w)), dim3(256), 0,
                           (hipStream_t)stream, f0, f1, cam, candidates, out, batch, h, w, num_candidates, from_argmax);
    else
        hipLaunchKernelGGL(depth_corr_softmax_kernel, dim3(pixel_grid_blocks((long)batch * h * w)), dim3(256), 0,
                           (hipStream_t)stream, f0, f1, cam, candidates, out, batch, h, w, num_candidates, from_argmax);
    return (int)hipGetLastError();
}
