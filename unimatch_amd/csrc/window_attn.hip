// Windowed single-head attention, flash style.                                gfx950 / wave64 / MFMA
//
//   out = softmax( q k^T / sqrt(C) + shift_mask ) v        inside (optionally cyclically shifted) windows
//
// Replaces single_head_split_window_attention / _1d / full_attention / full_attention_1d
// (unimatch/attention.py:8-163) including torch.roll, split_feature / merge_splits and the additive -100
// shifted-window masks (unimatch/utils.py:84-108,199-216).  None of those tensors exist here:
//   * a window is a set of ROLLED coordinates; the token that sits at rolled (ry, rx) is the original token
//     ((ry + shift_h) % h, (rx + shift_w) % w)  -> pure index arithmetic on the gather/scatter addresses;
//   * the mask is a comparison of 3x3 region labels computed from the rolled coordinates;
//   * scores and probabilities never leave registers (the reference materialises [2B*K^2, n, n] fp32 scores
//     plus a repeated mask: 2 x 604 MB per layer at B=8, 512x768).
//
// Decomposition: workgroup = 4 waves = 128 query tokens of one window; wave = 32 queries; K/V tiles of 64
// window tokens staged through LDS.  Two workgroups share a CU (76 KB LDS each in exact mode) so one
// workgroup's staging overlaps the other's MFMAs.  Everything is computed transposed so that the MFMA
// column index n is the query: lane l owns query (l & 31) in BOTH products,
//       S^T = K . Q^T      (A = K rows from LDS via ds_read_b128,            B = Q^T held in registers)
//       O^T = V^T . P^T    (A = V^T via ds_read_b64_tr_b16 transpose reads,  B = P^T built in registers)
// which makes the softmax a per-lane affair (one exchange with lane^32 per tile for the row max) and lets
// the probabilities go from the S^T accumulators to the P^T operand with v_permlane32_swap only.
//
// Precision (mode): exact = fp16 hi+lo operands, 3 MFMA products per contraction (lo*hi, hi*lo, hi*hi),
// probabilities scaled by 2^14 before the fp16 split so that small p keep 22 bits; fast = bf16, 1 product.
#include "common.h"
#include "planes.h"

struct WattnArgs {
    const unsigned short* qp;    // planes [NS][S][L][128]
    const unsigned short* kp;
    const unsigned short* vp;
    long plane_stride;           // S * L * 128
    float* out;                  // [S][L][128]
    int h, w, win_h, win_w, shift_h, shift_w;
    int nwx, nwin;               // windows per row, windows per stream
    int n;                       // tokens per window
    int nqt;                     // 128-query tiles per window
    int total;                   // workgroups
    float scale_log2;            // log2(e) / sqrt(C)
    float mask_raw;              // -100 * sqrt(C): the shifted-window mask in raw q.k units
};

// window-local token -> global token index (and its shifted-window region label)
__device__ __forceinline__ int window_token(const WattnArgs& a, int wy, int wx, int t, int& label) {
    const int ly = t / a.win_w, lx = t - ly * a.win_w;
    const int ry = wy * a.win_h + ly, rx = wx * a.win_w + lx;
    int oy = ry + a.shift_h, ox = rx + a.shift_w;
    oy = oy >= a.h ? oy - a.h : oy;
    ox = ox >= a.w ? ox - a.w : ox;
    const int rl = a.shift_h > 0 ? (int)(ry >= a.h - a.win_h) + (int)(ry >= a.h - a.shift_h) : 0;
    const int cl = a.shift_w > 0 ? (int)(rx >= a.w - a.win_w) + (int)(rx >= a.w - a.shift_w) : 0;
    label = 3 * rl + cl;
    return oy * a.w + ox;
}

template <class T, int NS>
__global__ __launch_bounds__(256, 2) void window_attn_kernel(WattnArgs a) {
    constexpr int KROW = 272;                 // K rows: 256 B + 16 B pad  (ds_read_b128 conflict free)
    constexpr int VROW = 320;                 // V rows: 256 B + 64 B pad  (4 consecutive keys -> 4 bank quarters)
    constexpr int KPLANE = 64 * KROW, VPLANE = 64 * VROW;
    constexpr int PSHIFT = (NS == 2) ? 14 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * (KPLANE + VPLANE) + 64];
    unsigned char* ldsV = lds + NS * KPLANE;
    unsigned char* klab = lds + NS * (KPLANE + VPLANE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, a.total);
    const int qt = wg % a.nqt;
    const int win = (wg / a.nqt) % a.nwin;
    const int s = wg / (a.nqt * a.nwin);
    const int wy = win / a.nwx, wx = win - wy * a.nwx;
    const bool has_mask = (a.shift_h > 0 && (wy + 1) * a.win_h == a.h) || (a.shift_w > 0 && (wx + 1) * a.win_w == a.w);
    const float c = a.scale_log2;
    const long sbase = (long)s * a.h * a.w;

    // ---- this lane's query -----------------------------------------------------------------------
    const int tq = qt * 128 + wave * 32 + (lane & 31);
    int labq;
    const int tokq = window_token(a, wy, wx, min(tq, a.n - 1), labq);
    i16x8 qf[NS][8];
    {
        const unsigned short* qb = a.qp + (sbase + tokq) * UM_CHANNELS + 8 * half;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[pl][ks] = ld_global_16B(qb + pl * a.plane_stride + 16 * ks);
    }

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = UM_NEG_INIT, M = -ceilf(UM_NEG_INIT * c), l = 0.f;

    const int ntiles = (a.n + 63) >> 6;
    // per-lane LDS bases
    const unsigned char* kb = lds + (lane & 31) * KROW + half * 16;
    const int li = lane & 15, lg = (lane >> 4) & 1;
    const unsigned char* vb = ldsV + (8 * half + (li >> 2)) * VROW + (16 * lg + 4 * (li & 3)) * 2;

    for (int t = 0; t < ntiles; ++t) {
        const int t0 = t * 64;
        __syncthreads();                       // everyone is done reading the previous tile
        // ---- stage K and V tiles: 64 window tokens x 256 B per plane, gathered by token index ---------
        {
            long off[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = (tid >> 4) + 16 * i;
                int lab;
                const int tk = t0 + key;
                const int tok = window_token(a, wy, wx, min(tk, a.n - 1), lab);
                off[i] = (sbase + tok) * UM_CHANNELS + (tid & 15) * 8;
                if ((tid & 15) == 0) klab[key] = tk < a.n ? (unsigned char)lab : (unsigned char)255;
            }
            i16x8 st[NS][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) st[pl][i] = ld_global_16B(a.kp + pl * a.plane_stride + off[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    *reinterpret_cast<i16x8*>(lds + pl * KPLANE + ((tid >> 4) + 16 * i) * KROW + (tid & 15) * 16) = st[pl][i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) st[pl][i] = ld_global_16B(a.vp + pl * a.plane_stride + off[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    *reinterpret_cast<i16x8*>(ldsV + pl * VPLANE + ((tid >> 4) + 16 * i) * VROW + (tid & 15) * 16) = st[pl][i];
        }
        __syncthreads();

        // ---- S^T = K . Q^T for the two 32-key sub-tiles ------------------------------------------------
        f32x16 sc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[0][r] = sc[1][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const i16x8 a0h = *reinterpret_cast<const i16x8*>(kb + ks * 32);
            const i16x8 a1h = *reinterpret_cast<const i16x8*>(kb + 32 * KROW + ks * 32);
            if (NS == 2) {
                const i16x8 a0l = *reinterpret_cast<const i16x8*>(kb + KPLANE + ks * 32);
                const i16x8 a1l = *reinterpret_cast<const i16x8*>(kb + KPLANE + 32 * KROW + ks * 32);
                sc[0] = T::mfma(a0l, qf[0][ks], sc[0]);
                sc[1] = T::mfma(a1l, qf[0][ks], sc[1]);
                sc[0] = T::mfma(a0h, qf[NS - 1][ks], sc[0]);
                sc[1] = T::mfma(a1h, qf[NS - 1][ks], sc[1]);
            }
            sc[0] = T::mfma(a0h, qf[0][ks], sc[0]);
            sc[1] = T::mfma(a1h, qf[0][ks], sc[1]);
        }

        // ---- mask: region labels (shifted windows) and the ragged tail of the window ---------------------
        if (has_mask || t0 + 64 > a.n) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned labs = *reinterpret_cast<const unsigned*>(klab + 32 * sub + 8 * g + 4 * half);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int lab = (labs >> (8 * i)) & 255;
                        float v = sc[sub][4 * g + i];
                        v = (has_mask && lab != labq) ? v + a.mask_raw : v;
                        sc[sub][4 * g + i] = lab == 255 ? UM_NEG_MASK : v;
                    }
                }
        }

        // ---- online softmax (row max shared by the lane pair l, l^32) ------------------------------------
        float mx = sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        m = fmaxf(m, mx);
        const float Mn = -ceilf(m * c);
        if (Mn != M) {                          // exact power-of-two rescale (see global_match.hip)
            const float resc = fast_exp2(Mn - M);
            M = Mn;
            l *= resc;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= resc;
        }
        const float mc = M + (float)PSHIFT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(__builtin_fmaf(sc[sub][r], c, mc));
                sc[sub][r] = p;
                l += p;
            }

        // ---- P^T operand fragments: cvt + v_permlane32_swap, no LDS ---------------------------------------
        // k-step ks (16 keys) uses regs 8*(ks&1)..+7 of sub-tile ks>>1; after the swaps a lane holds
        // keys 8*half .. 8*half+7 of the step for its own query (B operand layout).
        i16x8 pf[NS][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sub = ks >> 1, r0 = 8 * (ks & 1);
            unsigned wh[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = sc[sub][r0 + 2 * j], p1 = sc[sub][r0 + 2 * j + 1];
                wh[j] = T::pack2(p0, p1);
                if (NS == 2) {
                    const f32x2 hh = T::unpack2(wh[j]);
                    wl[j] = T::pack2(p0 - hh[0], p1 - hh[1]);
                }
            }
            {
                const auto x = __builtin_amdgcn_permlane32_swap(wh[0], wh[2], false, false);
                const auto y = __builtin_amdgcn_permlane32_swap(wh[1], wh[3], false, false);
                const u32x4 f = {x[0], y[0], x[1], y[1]};
                pf[0][ks] = __builtin_bit_cast(i16x8, f);
            }
            if (NS == 2) {
                const auto x = __builtin_amdgcn_permlane32_swap(wl[0], wl[2], false, false);
                const auto y = __builtin_amdgcn_permlane32_swap(wl[1], wl[3], false, false);
                const u32x4 f = {x[0], y[0], x[1], y[1]};
                pf[NS - 1][ks] = __builtin_bit_cast(i16x8, f);
            }
        }

        // ---- O^T += V^T . P^T ----------------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* va = vb + ks * 16 * VROW + dt * 64;
                i16x8 vh, vl;
                {
                    const i16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va));
                    const i16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va + 4 * VROW));
                    vh = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (NS == 2) {
                    const i16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va + VPLANE));
                    const i16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va + VPLANE + 4 * VROW));
                    vl = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = T::mfma(vl, pf[0][ks], o[dt]);
                    o[dt] = T::mfma(vh, pf[NS - 1][ks], o[dt]);
                }
                o[dt] = T::mfma(vh, pf[0][ks], o[dt]);
            }
        }
    }

    // ---- normalise and scatter back to the original token positions -----------------------------------------
    const float lt = l + __shfl_xor(l, 32);
    const float inv = 1.0f / lt;
    if (tq < a.n) {
        float* ob = a.out + (sbase + tokq) * UM_CHANNELS + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(ob + 32 * dt + 8 * g) = v;
            }
    }
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

static size_t align256w(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t um_window_attn_workspace_bytes(int streams, int tokens, int channels, int mode) {
    if (streams <= 0 || tokens <= 0 || channels != UM_CHANNELS || (mode != 0 && mode != 1)) return 0;
    return 3 * align256w(planes_bytes((long)streams * tokens, mode));
}

extern "C" int um_window_attn_fwd(const float* q, const float* k, const float* v, float* out, int streams, int h,
                                  int w, int channels, int win_h, int win_w, int shift_h, int shift_w, int mode,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !k || !v || !out || streams <= 0 || h <= 0 || w <= 0) {
        um_set_error("null pointer or non-positive size (streams=%d h=%d w=%d)", streams, h, w);
        return -1;
    }
    if (channels != UM_CHANNELS) {
        um_set_error("channels=%d unsupported (the library is built for %d)", channels, UM_CHANNELS);
        return -1;
    }
    if (mode != 0 && mode != 1) {
        um_set_error("mode=%d is neither UM_MODE_EXACT nor UM_MODE_FAST", mode);
        return -1;
    }
    if (win_h <= 0 || win_w <= 0 || h % win_h != 0 || w % win_w != 0) {
        um_set_error("window %dx%d does not tile the %dx%d map", win_h, win_w, h, w);
        return -2;
    }
    const bool shift_h_ok = shift_h == 0 || (shift_h > 0 && shift_h < win_h && win_h < h);
    const bool shift_w_ok = shift_w == 0 || (shift_w > 0 && shift_w < win_w && win_w < w);
    if (!shift_h_ok || !shift_w_ok) {
        um_set_error("shift (%d,%d) invalid for window %dx%d on a %dx%d map", shift_h, shift_w, win_h, win_w, h, w);
        return -2;
    }
    const long L = (long)h * w;
    const size_t need = um_window_attn_workspace_bytes(streams, (int)L, channels, mode);
    if (!workspace || workspace_bytes < need) {
        um_set_error("workspace too small: %zu bytes given, %zu needed", workspace_bytes, need);
        return -3;
    }
    unsigned char* ws = (unsigned char*)workspace;
    const size_t pb = align256w(planes_bytes(streams * L, mode));
    unsigned short* pq = (unsigned short*)ws;
    unsigned short* pk = (unsigned short*)(ws + pb);
    unsigned short* pv = (unsigned short*)(ws + 2 * pb);
    hipError_t e;
    if ((e = launch_split_planes(q, pq, streams * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(k, pk, streams * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(v, pv, streams * L, 1.f, mode, stream)) != hipSuccess) return (int)e;

    WattnArgs a;
    a.qp = pq;
    a.kp = pk;
    a.vp = pv;
    a.plane_stride = streams * L * UM_CHANNELS;
    a.out = out;
    a.h = h;
    a.w = w;
    a.win_h = win_h;
    a.win_w = win_w;
    a.shift_h = shift_h;
    a.shift_w = shift_w;
    a.nwx = w / win_w;
    a.nwin = (h / win_h) * a.nwx;
    a.n = win_h * win_w;
    a.nqt = (a.n + 127) / 128;
    a.total = a.nqt * a.nwin * streams;
    a.scale_log2 = UM_LOG2E / sqrtf((float)channels);
    a.mask_raw = -100.0f * sqrtf((float)channels);
    ScopedKernelTimer timer(UM_K_WINDOW_ATTN, stream);
    if (mode == 0)
        hipLaunchKernelGGL((window_attn_kernel<Fp16, 2>), dim3(a.total), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((window_attn_kernel<Bf16, 1>), dim3(a.total), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
