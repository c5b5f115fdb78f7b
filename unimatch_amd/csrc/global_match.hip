// Global (all-pairs) correlation + softmax + expected value, fused.            gfx950 / wave64 / MFMA
//
//   out[b, c, i] = alpha * sum_j softmax_j( q_i . k_j / sqrt(C) )[j] * v[b, c, j]  +  beta * v[b, c, i]
//
// One kernel serves three reference functions (see include/unimatch_hip.h):
//   global_correlation_softmax          q=f0, k=f1, v=pixel grid (x,y),   alpha=1,  beta=-1  (flow)
//   global_correlation_softmax_stereo   per scanline, keys x' <= x only,  v=x,  alpha=-1, beta=1
//   SelfAttnPropagation (global)        q, k projected features, v=flow,  alpha=1,  beta=0
// The L x L correlation / probability matrices (151 MB each per pair at 512x768 in the reference,
// unimatch/matching.py:15,29) are never formed: scores live in MFMA accumulators, the softmax is
// evaluated blockwise with a running (max, sum, sum*v) per query, and because v has only 1-2 channels
// the "P.V" product is a few fp32 FMAs per score instead of a second matmul.
//
// Work decomposition: workgroup = 4 waves = 128 queries; wave = 32 queries; key tiles of 64.
// Scores are computed "swapped", S^T = K . Q^T with v_mfma_f32_32x32x16, so lane l owns query (l & 31)
// and holds 16 of the 32 key scores of a sub-tile; lanes l and l^32 split the keys of a query between
// them and keep INDEPENDENT running maxima, merged once at the end (no cross-lane traffic per tile).
// K tiles are staged through LDS (rows padded to 272 B: conflict-free ds_read_b128 A-fragments) with
// the next tile's global loads in flight during the current tile's MFMAs.
#include "common.h"
#include "planes.h"

struct GsvArgs {
    const unsigned short* qp;   // [NS][nbatch][Lq][128]
    const unsigned short* kp;   // [NS][nbatch][Lk][128]
    long q_plane_stride, k_plane_stride;
    const float* v;             // v[b * v_batch_stride + c * v_chan_stride + key]
    long v_batch_stride, v_chan_stride;
    float* out;                 // [nbatch][NV][Lq]
    int Lq, Lk;
    float scale_log2;           // log2(e) / sqrt(C)
    float alpha, beta;
};

template <class T, int NS, int NV, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void gsv_kernel(GsvArgs a) {
    constexpr int KROW = 272;                 // 128 * 2 B + 16 B pad
    constexpr int KPLANE = 64 * KROW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * KPLANE + NV * 64 * 4];
    float* vt = reinterpret_cast<float*>(lds + NS * KPLANE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    const int b = blockIdx.y;
    const int qwg = blockIdx.x * 128;
    const int qi = qwg + wave * 32 + (lane & 31);
    const float c = a.scale_log2;

    // ---- Q fragments: B operand of S^T = K . Q^T -------------------------------------------
    i16x8 qf[NS][8];
    {
        const int qr = min(qi, a.Lq - 1);
        const unsigned short* qb = a.qp + ((long)b * a.Lq + qr) * UM_CHANNELS + 8 * half;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[pl][ks] = ld_global_16B(qb + pl * a.q_plane_stride + 16 * ks);
    }

    int ntiles = (a.Lk + 63) >> 6;
    if (CAUSAL) ntiles = min(ntiles, (min(qwg + 127, a.Lq - 1) >> 6) + 1);

    // ---- staging: next K tile (and its values) travel HBM -> registers during the MFMAs --------
    i16x8 st[NS][4];
    float sv[NV];
    const unsigned short* kbase = a.kp + (long)b * a.Lk * UM_CHANNELS;
    const float* vbase = a.v + (long)b * a.v_batch_stride;

    auto issue = [&](int t) {
        const int t0 = t * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i;
            const int kr = min(t0 + (id >> 4), a.Lk - 1);
            const unsigned short* src = kbase + (long)kr * UM_CHANNELS + (id & 15) * 8;
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) st[pl][i] = ld_global_16B(src + pl * a.k_plane_stride);
        }
        if (tid < 64) {
            const int kr = min(t0 + tid, a.Lk - 1);
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) sv[ch] = vbase[ch * a.v_chan_stride + kr];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i;
            unsigned char* dst = lds + (id >> 4) * KROW + (id & 15) * 16;
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) *reinterpret_cast<i16x8*>(dst + pl * KPLANE) = st[pl][i];
        }
        if (tid < 64) {
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) vt[ch * 64 + tid] = sv[ch];
        }
    };

    // Running softmax state.  m = running max of the RAW scores; the exponent offset actually used is
    // the integer M = -ceil(m * c), so every rescale factor exp2(M_new - M_old) is an exact power of two
    // and l / acc are rescaled without rounding (p of the running max lies in (1/2, 1]).
    float m = UM_NEG_INIT, M = -ceilf(UM_NEG_INIT * c), l = 0.f;
    float acc[NV];
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) acc[ch] = 0.f;

    // online-softmax update with the 16 scores this lane holds of a 32-key sub-tile
    auto update = [&](f32x16 s, int key0 /* first key of the sub-tile */, int ldsk0 /* its slot in the tile */) {
        const int kl = key0 + 4 * half;
        if (CAUSAL || key0 + 32 > a.Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kl + (r & 3) + 8 * (r >> 2);
                const bool ok = (key < a.Lk) && (!CAUSAL || key <= qi);
                s[r] = ok ? s[r] : UM_NEG_MASK;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        m = fmaxf(m, mx);
        const float Mn = -ceilf(m * c);
        const float resc = fast_exp2(Mn - M);
        M = Mn;
        l *= resc;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[ch] *= resc;
        const float mc = M;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 vv[NV];
#pragma unroll
            for (int ch = 0; ch < NV; ++ch)
                vv[ch] = *reinterpret_cast<const f32x4*>(vt + ch * 64 + ldsk0 + 8 * g + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = fast_exp2(__builtin_fmaf(s[4 * g + i], c, mc));
                l += p;
#pragma unroll
                for (int ch = 0; ch < NV; ++ch) acc[ch] = __builtin_fmaf(p, vv[ch][i], acc[ch]);
            }
        }
    };

    issue(0);
    for (int t = 0; t < ntiles; ++t) {
        commit();
        __syncthreads();
        if (t + 1 < ntiles) issue(t + 1);

        f32x16 s0 = {0}, s1 = {0};
        const unsigned char* kb = lds + (lane & 31) * KROW + half * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const i16x8 a0h = *reinterpret_cast<const i16x8*>(kb + ks * 32);
            const i16x8 a1h = *reinterpret_cast<const i16x8*>(kb + 32 * KROW + ks * 32);
            if (NS == 2) {
                const i16x8 a0l = *reinterpret_cast<const i16x8*>(kb + KPLANE + ks * 32);
                const i16x8 a1l = *reinterpret_cast<const i16x8*>(kb + KPLANE + 32 * KROW + ks * 32);
                s0 = T::mfma(a0l, qf[0][ks], s0);
                s1 = T::mfma(a1l, qf[0][ks], s1);
                s0 = T::mfma(a0h, qf[NS - 1][ks], s0);
                s1 = T::mfma(a1h, qf[NS - 1][ks], s1);
            }
            s0 = T::mfma(a0h, qf[0][ks], s0);
            s1 = T::mfma(a1h, qf[0][ks], s1);
        }
        update(s0, t * 64, 0);
        update(s1, t * 64 + 32, 32);
        __syncthreads();
    }

    // ---- merge the two half-waves' partial softmaxes and write ------------------------------------
    const float M2 = __shfl_xor(M, 32);
    const float l2 = __shfl_xor(l, 32);
    const float MM = fminf(M, M2);
    const float f1 = fast_exp2(MM - M), f2 = fast_exp2(MM - M2);
    const float lt = l * f1 + l2 * f2;
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) {
        const float a2 = __shfl_xor(acc[ch], 32);
        const float at = acc[ch] * f1 + a2 * f2;
        if (half == 0 && qi < a.Lq) {
            float r = a.alpha * (at / lt);
            if (a.beta != 0.f) r += a.beta * vbase[ch * a.v_chan_stride + qi];
            a.out[((long)b * NV + ch) * a.Lq + qi] = r;
        }
    }
}

// pixel grid table: g[0][i] = x = i % w, g[1][i] = y = i / w   (unimatch/geometry.py:5-21)
__global__ void fill_grid_kernel(float* g, int h, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = h * w;
    if (i < L) {
        g[i] = (float)(i % w);
        g[L + i] = (float)(i / w);
    }
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

template <int NV, bool CAUSAL>
static hipError_t launch_gsv(const GsvArgs& a, int nbatch, int mode, hipStream_t stream) {
    dim3 grid((a.Lq + 127) / 128, nbatch), block(256);
    ScopedKernelTimer timer(UM_K_GLOBAL_SOFTMAX, stream);
    if (mode == 0)
        hipLaunchKernelGGL((gsv_kernel<Fp16, 2, NV, CAUSAL>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((gsv_kernel<Bf16, 1, NV, CAUSAL>), grid, block, 0, stream, a);
    return hipGetLastError();
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t um_global_corr_workspace_bytes(int batch, int tokens, int channels, int mode) {
    if (batch <= 0 || tokens <= 0 || channels != UM_CHANNELS || (mode != 0 && mode != 1)) return 0;
    return 2 * align256(planes_bytes((long)batch * tokens, mode)) + align256((size_t)tokens * 2 * sizeof(float));
}

static int check_common(const void* p0, const void* p1, const void* p2, int batch, int h, int w, int channels,
                        int mode, const void* ws, size_t ws_bytes, size_t need) {
    if (!p0 || !p1 || !p2 || batch <= 0 || h <= 0 || w <= 0) {
        um_set_error("null pointer or non-positive size (batch=%d h=%d w=%d)", batch, h, w);
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
    if (!ws || ws_bytes < need) {
        um_set_error("workspace too small: %zu bytes given, %zu needed", ws_bytes, need);
        return -3;
    }
    return 0;
}

extern "C" int um_global_corr_softmax_flow(const float* f0, const float* f1, float* flow, int batch, int h, int w,
                                           int channels, int bidir, int mode, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long L = (long)h * w;
    const size_t need = um_global_corr_workspace_bytes(batch, (int)L, channels, mode);
    if (int e = check_common(f0, f1, flow, batch, h, w, channels, mode, workspace, workspace_bytes, need)) return e;
    unsigned char* ws = (unsigned char*)workspace;
    const size_t pb = align256(planes_bytes(batch * L, mode));
    unsigned short* p0 = (unsigned short*)ws;
    unsigned short* p1 = (unsigned short*)(ws + pb);
    float* grid = (float*)(ws + 2 * pb);
    hipError_t e;
    if ((e = launch_split_planes(f0, p0, batch * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(f1, p1, batch * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fill_grid_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, stream, grid, h, w);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;

    GsvArgs a;
    a.q_plane_stride = a.k_plane_stride = batch * L * UM_CHANNELS;
    a.v = grid;
    a.v_batch_stride = 0;
    a.v_chan_stride = L;
    a.Lq = a.Lk = (int)L;
    a.scale_log2 = UM_LOG2E / sqrtf((float)channels);
    a.alpha = 1.f;
    a.beta = -1.f;
    a.qp = p0;
    a.kp = p1;
    a.out = flow;
    if ((e = launch_gsv<2, false>(a, batch, mode, stream)) != hipSuccess) return (int)e;
    if (bidir) {   // backward flow: softmax over the other axis of the same correlation (matching.py:23-27)
        a.qp = p1;
        a.kp = p0;
        a.out = flow + (long)batch * 2 * L;
        if ((e = launch_gsv<2, false>(a, batch, mode, stream)) != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" int um_global_corr_softmax_stereo(const float* f0, const float* f1, float* disp, int batch, int h, int w,
                                             int channels, int mode, void* workspace, size_t workspace_bytes,
                                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long L = (long)h * w;
    const size_t need = um_global_corr_workspace_bytes(batch, (int)L, channels, mode);
    if (int e = check_common(f0, f1, disp, batch, h, w, channels, mode, workspace, workspace_bytes, need)) return e;
    unsigned char* ws = (unsigned char*)workspace;
    const size_t pb = align256(planes_bytes(batch * L, mode));
    unsigned short* p0 = (unsigned short*)ws;
    unsigned short* p1 = (unsigned short*)(ws + pb);
    float* grid = (float*)(ws + 2 * pb);
    hipError_t e;
    if ((e = launch_split_planes(f0, p0, batch * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(f1, p1, batch * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    // x table = first w entries of a 1 x w pixel grid
    hipLaunchKernelGGL(fill_grid_kernel, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, stream, grid, 1, w);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    GsvArgs a;
    a.qp = p0;
    a.kp = p1;
    a.q_plane_stride = a.k_plane_stride = batch * L * UM_CHANNELS;
    a.v = grid;
    a.v_batch_stride = 0;
    a.v_chan_stride = 0;
    a.out = disp;
    a.Lq = a.Lk = w;                    // every scanline is an independent "batch" entry
    a.scale_log2 = UM_LOG2E / sqrtf((float)channels);
    a.alpha = -1.f;                     // disparity = x - E[x']   (matching.py:147-149)
    a.beta = 1.f;
    if ((e = launch_gsv<1, true>(a, batch * h, mode, stream)) != hipSuccess) return (int)e;
    return 0;
}

extern "C" int um_prop_global_attn(const float* q, const float* k, const float* value, float* out, int batch, int h,
                                   int w, int channels, int value_channels, int mode, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long L = (long)h * w;
    const size_t need = um_global_corr_workspace_bytes(batch, (int)L, channels, mode);
    if (int e = check_common(q, k, out, batch, h, w, channels, mode, workspace, workspace_bytes, need)) return e;
    if (!value || (value_channels != 1 && value_channels != 2)) {
        um_set_error("value_channels=%d: the reference propagates flow (2) or disparity/depth (1)", value_channels);
        return -1;
    }
    unsigned char* ws = (unsigned char*)workspace;
    const size_t pb = align256(planes_bytes(batch * L, mode));
    unsigned short* pq = (unsigned short*)ws;
    unsigned short* pk = (unsigned short*)(ws + pb);
    hipError_t e;
    if ((e = launch_split_planes(q, pq, batch * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(k, pk, batch * L, 1.f, mode, stream)) != hipSuccess) return (int)e;
    GsvArgs a;
    a.qp = pq;
    a.kp = pk;
    a.q_plane_stride = a.k_plane_stride = batch * L * UM_CHANNELS;
    a.v = value;
    a.v_batch_stride = value_channels * L;
    a.v_chan_stride = L;
    a.out = out;
    a.Lq = a.Lk = (int)L;
    a.scale_log2 = UM_LOG2E / sqrtf((float)channels);
    a.alpha = 1.f;
    a.beta = 0.f;
    if (value_channels == 2)
        e = launch_gsv<2, false>(a, batch, mode, stream);
    else
        e = launch_gsv<1, false>(a, batch, mode, stream);
    return (int)e;
}
