// Transformer-layer linears on the matrix cores with fused prologues / epilogues.      gfx950 / wave64 / MFMA
//
//   C[M, N] = A[M, K] . W[N, K]^T          (nn.Linear without bias, unimatch/transformer.py:22-39)
//
// replaces, per Transformer layer, the hipBLASLt fp32 GEMMs and the element-wise kernels between them:
//   q/k/v projections (transformer.py:58-60)          -> EPI_PLANES: the result is written straight in the
//                                                         attention kernel's operand format (no fp32 round trip)
//   merge + LayerNorm (+ residual)  (:137-138, :144)  -> EPI_LN
//   FFN: cat[source, message] -> 1024, GELU (:141)    -> A_CONCAT prologue (no concatenated tensor), EPI_GELU_PLANES
//   FFN: 1024 -> 128, LayerNorm, + source (:141-144)  -> A_PLANES prologue, EPI_LN with residual
//
// Arithmetic is the same "exact" scheme as the attention kernels: fp16 hi + lo split operands, three MFMA
// products (lo.hi, hi.lo, hi.hi), fp32 accumulation; weights are pre-scaled by 2^wshift before the split so
// that their lo parts stay in fp16's normal range, and the accumulator is scaled back by the exact 2^-wshift.
// Fast mode: bf16 operands, one product.
//
// Decomposition: workgroup = 4 waves = 128 tokens x 128 output features; wave = 32 tokens x 128 features,
// computed transposed (D^T = W . A^T) so that lane = token: LayerNorm statistics are in-lane sums plus one
// exchange with lane^32.  K is walked in stages of 32 through a double-buffered LDS ring: weight (and plane)
// tiles arrive by LDS-DMA, fp32 activations are loaded one stage ahead into registers, split, and written
// with the same XOR swizzle (conflict-free ds_read_b128 fragments from 64-byte rows).
#include <type_traits>
#include "common.h"
#include "planes.h"

enum { A_F32 = 0, A_CONCAT = 1, A_PLANES = 2 };
enum { EPI_PLANES = 0, EPI_LN = 1, EPI_GELU_PLANES = 2, EPI_F32 = 3 };

struct LinArgs {
    const float* a0;              // A_F32: [M, K];  A_CONCAT: [M, K/2] (first half of K)
    const float* a1;              // A_CONCAT: [M, K/2] (second half of K)
    const unsigned short* ap;     // A_PLANES: [NS][M][K]
    long a_plane_stride;
    const unsigned short* wp;     // [NS][N][K], pre-scaled by 2^wshift
    long w_plane_stride;
    int M, N, K;
    float out_scale;              // 2^-wshift
    unsigned short* outp;         // EPI_PLANES / EPI_GELU_PLANES: [NS][M][N]
    long out_plane_stride;
    float* outf;                  // EPI_LN: [M, N]
    const float* gamma;
    const float* beta;
    const float* residual;        // EPI_LN: optional [M, N]
    float eps;
    const float* bias;            // optional [N]: C = A.W^T * out_scale + bias * bias_scale  (nn.Linear with bias)
    float bias_scale;
    unsigned* range_flag;         // um_range_flags: sticky operand-range word (device address; nullptr: none)
};

__device__ __forceinline__ void lin_dma16(const void* base, unsigned byte_off, const unsigned char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(dst)
                 : "memory");
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <class T, int NS, int ASRC, int EPI>
__global__ __launch_bounds__(256, 2) void linear_kernel(LinArgs a) {
    constexpr int BK = 32;                       // K per stage (two MFMA k-steps)
    constexpr int TILE = 128 * 64;               // one 128-row x 64-byte operand tile (one plane, one stage)
    constexpr int STAGE = 2 * NS * TILE;         // A planes then W planes
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const float neg1 = um_opaque_neg1();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    // 1-D XCD-aware grid: the n tiles of one token tile are neighbours on ONE XCD, so the second reads the activation tile
    // from that XCD's L2 (a 2-D grid put them on neighbouring XCDs: the k | v projection fetched its input twice from HBM,
    // 96 MB per launch for a 50 MB tensor -- profiles/r02_pmc_fetch_v3.json)
    const int ntn = a.N / 128;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (wgid % ntn) * 128, m0 = (wgid / ntn) * 128;
    const int nstage = a.K / BK;

    // ---- staging --------------------------------------------------------------------------------------------
    // 16-byte chunk cp of row r holds source chunk cp ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128 lane group
    // then cover all 16 slots of a 256-byte bank line.
    // DMA (planes): one instruction = 16 rows x 64 B; wave w moves rows 32w .. 32w+31 of a tile (2 instructions).
    const int drow = 32 * wave + (lane >> 2);                   // + 16 * i
    const int dcp = lane & 3;
    auto dma_tile = [&](const unsigned short* base, long plane_stride, int row0, int nrows_total, int ld, int k0,
                        unsigned char* dst) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = drow + 16 * i;
            const int grow = min(row0 + r, nrows_total - 1);
            const int sc = dcp ^ ((r >> 2) & 3);
            const unsigned off = (unsigned)(((long)grow * ld + k0 + 8 * sc) * 2);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) lin_dma16(base + pl * plane_stride, off, dst + pl * TILE + (32 * wave + 16 * i) * 64);
        }
    };
    // fp32 activations: 8 consecutive lanes read one row's 128 contiguous bytes (32 floats); 4 passes of 32 rows
    f32x4 areg[4];
    auto load_a = [&](int s) {
        const int k0 = s * BK;
        const float* src = a.a0;
        int kk = k0, ld = a.K;
        if (ASRC == A_CONCAT) {
            ld = a.K / 2;
            if (k0 >= ld) { src = a.a1; kk = k0 - ld; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * i + (tid >> 3);
            const int grow = min(m0 + r, a.M - 1);
            areg[i] = *reinterpret_cast<const f32x4*>(src + (long)grow * ld + kk + 4 * (tid & 7));
        }
    };
    float rmx = 0.f;                 // largest magnitude turned into an fp16 operand (fp32 inputs, plane outputs; um_range_flags)
    auto store_a = [&](unsigned char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * i + (tid >> 3), f4 = tid & 7;
            rmx = fmaxf(fmaxf(rmx, fmaxf(__builtin_fabsf(areg[i][0]), __builtin_fabsf(areg[i][1]))),
                        fmaxf(__builtin_fabsf(areg[i][2]), __builtin_fabsf(areg[i][3])));
            const unsigned h0 = T::pack2(areg[i][0], areg[i][1]), h1 = T::pack2(areg[i][2], areg[i][3]);
            unsigned char* p = dst + r * 64 + (((f4 >> 1) ^ ((r >> 2) & 3)) << 4) + (f4 & 1) * 8;
            *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
            if (NS == 2) {
                const unsigned l0 = T::lo2(areg[i][0], areg[i][1], h0, neg1), l1 = T::lo2(areg[i][2], areg[i][3], h1, neg1);
                *reinterpret_cast<u32x2*>(p + TILE) = u32x2{l0, l1};
            }
        }
    };
    auto stage_async = [&](int s, unsigned char* buf) {          // everything of stage s that goes by DMA
        if (ASRC == A_PLANES) dma_tile(a.ap, a.a_plane_stride, m0, a.M, a.K, s * BK, buf);
        dma_tile(a.wp, a.w_plane_stride, n0, a.N, a.K, s * BK, buf + NS * TILE);
    };

    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // fragment offsets inside a tile: row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4), chunk = 2 * kstep + half
    const int fr = lane & 31;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = fr * 64 + (((2 * ks + half) ^ ((fr >> 2) & 3)) << 4);

    // ---- prologue: stage 0 ------------------------------------------------------------------------------------
    if (ASRC != A_PLANES) load_a(0);
    stage_async(0, lds);
    if (ASRC != A_PLANES) store_a(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < nstage; ++s) {
        unsigned char* cur = lds + (s & 1) * STAGE;
        unsigned char* nxt = lds + ((s & 1) ^ 1) * STAGE;
        const bool more = s + 1 < nstage;
        if (more) {
            stage_async(s + 1, nxt);
            if (ASRC != A_PLANES) load_a(s + 1);
        }
        const unsigned char* at = cur + 32 * wave * 64;           // this wave's 32 token rows
        const unsigned char* wt = cur + NS * TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const i16x8 bh = *reinterpret_cast<const i16x8*>(at + foff[ks]);
            i16x8 bl;
            if (NS == 2) bl = *reinterpret_cast<const i16x8*>(at + TILE + foff[ks]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const i16x8 wh = *reinterpret_cast<const i16x8*>(wt + nt * 32 * 64 + foff[ks]);
                if (NS == 2) {
                    const i16x8 wl = *reinterpret_cast<const i16x8*>(wt + TILE + nt * 32 * 64 + foff[ks]);
                    acc[nt] = T::mfma(wl, bh, acc[nt]);
                    acc[nt] = T::mfma(wh, bl, acc[nt]);
                }
                acc[nt] = T::mfma(wh, bh, acc[nt]);
            }
        }
        if (more && ASRC != A_PLANES) store_a(nxt);              // (compiler waits for its own loads here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    // lane holds, for token m0 + 32*wave + (lane & 31), features n0 + 32*nt + 8*g + 4*half + i  (reg r = 4*g + i)
    const int tok = m0 + 32 * wave + (lane & 31);
    const bool valid = tok < a.M;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] *= a.out_scale;
    if (a.bias) {                                                 // wave-uniform; accumulator row = output feature
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + n0 + 32 * nt + 8 * g + 4 * half);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nt][4 * g + i] = __builtin_fmaf(bb[i], a.bias_scale, acc[nt][4 * g + i]);
            }
    }

    if (EPI == EPI_F32) {                                         // plain fp32 result (the propagation layer's local q / k)
        if (valid) {
            float* ob = a.outf + (long)tok * a.N + n0 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = acc[nt][4 * g + i];
                    *reinterpret_cast<f32x4*>(ob + 32 * nt + 8 * g) = y;
                }
        }
    } else if (EPI == EPI_PLANES || EPI == EPI_GELU_PLANES) {
        // The accumulator layout gives every lane 8-byte pieces of 32 different token rows: written directly that is
        // 32 partial cache lines per store instruction (measured: the stores cost as much as the whole main loop).
        // The tile goes through the idle staging ring instead -- every wave transposes its own 32 x 128 block, one
        // 8 KB region per plane, 16-byte chunk c of row r at chunk c ^ (r & 15) -- and leaves as full 256-byte rows.
        unsigned char* stg = lds + wave * 8192;                   // plane pl at + pl * 32768
        const int tl = lane & 31;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[nt][4 * g + i];
                    if (EPI == EPI_GELU_PLANES) v[i] = gelu_erf(v[i]);
                }
                unsigned char* p = stg + tl * 256 + (((4 * nt + g) ^ (tl & 15)) << 4) + 8 * half;
                rmx = fmaxf(fmaxf(rmx, fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
                const unsigned h0 = T::pack2(v[0], v[1]), h1 = T::pack2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
                if (NS == 2) {
                    const unsigned l0 = T::lo2(v[0], v[1], h0, neg1), l1 = T::lo2(v[2], v[3], h1, neg1);
                    *reinterpret_cast<u32x2*>(p + 32768) = u32x2{l0, l1};
                }
            }
        um_range_note<T>(a.range_flag, rmx, UM_RANGE_LINEAR);
        __builtin_amdgcn_wave_barrier();                          // same wave: LDS executes its accesses in order
        const int row0 = m0 + 32 * wave;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = 4 * it + (lane >> 4), cch = lane & 15;
                const u32x4 d = *reinterpret_cast<const u32x4*>(stg + pl * 32768 + r * 256 + ((cch ^ (r & 15)) << 4));
                if (row0 + r < a.M)
                    *reinterpret_cast<u32x4*>(a.outp + pl * a.out_plane_stride + (long)(row0 + r) * a.N + n0 + 8 * cch) = d;
            }
    } else {
        um_range_note<T>(a.range_flag, rmx, UM_RANGE_LINEAR);       // (fp32 inputs of this launch)
        // LayerNorm over the N = 128 features of the token (two-pass in registers), optional residual
        float s1 = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[nt][r];
        float u, v2;
        half_wave_pair(s1, u, v2);
        const float mean = (u + v2) * (1.0f / 128.0f);
        float s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[nt][r] - mean;
                s2 = __builtin_fmaf(d, d, s2);
            }
        half_wave_pair(s2, u, v2);
        const float rstd = 1.0f / sqrtf((u + v2) * (1.0f / 128.0f) + a.eps);
        if (valid) {
            float* ob = a.outf + (long)tok * a.N + 4 * half;
            const float* rb = a.residual ? a.residual + (long)tok * a.N + 4 * half : nullptr;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * nt + 8 * g;
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + n + 4 * half);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + n + 4 * half);
                    f32x4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = (acc[nt][4 * g + i] - mean) * rstd * gm[i] + bt[i];
                    if (rb) {
                        const f32x4 rr = *reinterpret_cast<const f32x4*>(rb + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] += rr[i];
                    }
                    *reinterpret_cast<f32x4*>(ob + n) = y;
                }
        }
    }
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

template <int ASRC, int EPI>
static hipError_t launch_linear(const LinArgs& a, int mode, hipStream_t stream) {
    dim3 grid((a.N / 128) * ((a.M + 127) / 128)), block(256);
    ScopedKernelTimer timer(UM_K_LINEAR, stream);
    if (mode == 0)
        hipLaunchKernelGGL((linear_kernel<Fp16, 2, ASRC, EPI>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((linear_kernel<Bf16, 1, ASRC, EPI>), grid, block, 0, stream, a);
    return hipGetLastError();
}

extern "C" size_t um_planes_bytes(long rows, int cols, int mode) {
    if (rows <= 0 || cols <= 0 || (mode != 0 && mode != 1)) return 0;
    return (size_t)rows * cols * 2 * (mode == 0 ? 2 : 1);
}

extern "C" int um_weight_planes(const float* w, void* planes, int n, int k, int wshift, int mode, void* stream) {
    if (!w || !planes || n <= 0 || k <= 0 || ((long)n * k) % 8 != 0 || (mode != 0 && mode != 1) || wshift < 0 || wshift > 14) {
        um_set_error("um_weight_planes: bad argument (n=%d k=%d wshift=%d mode=%d)", n, k, wshift, mode);
        return -1;
    }
    return (int)launch_split_elems(w, (unsigned short*)planes, (long)n * k, ldexpf(1.f, wshift), mode, (hipStream_t)stream);
}

extern "C" int um_linear_fwd(const float* a0, const float* a1, const void* a_planes, const void* w_planes, int m,
                             int n, int k, int wshift, int epilogue, void* out, const float* gamma, const float* beta,
                             const float* residual, float eps, int mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (m <= 0 || (m + 127) / 128 > 65535 || n <= 0 || k <= 0 || n % 128 != 0 || k % 32 != 0 || !w_planes || !out || (mode != 0 && mode != 1)) {
        um_set_error("um_linear_fwd: bad argument (m=%d n=%d k=%d: n must be a multiple of 128, k of 32)", m, n, k);
        return -1;
    }
    const int nsrc = (a0 != nullptr) + (a_planes != nullptr);
    if (nsrc != 1 || (a1 && !a0)) {
        um_set_error("um_linear_fwd: give exactly one of a0 (fp32, optionally with a1 for a K-concatenation) or a_planes");
        return -1;
    }
    if (a1 && (k % 64 != 0)) {
        um_set_error("um_linear_fwd: concatenated input needs k %% 64 == 0");
        return -1;
    }
    if (epilogue == EPI_LN && (n != 128 || !gamma || !beta)) {
        um_set_error("um_linear_fwd: the LayerNorm epilogue needs n == 128 and gamma/beta");
        return -1;
    }
    if (epilogue < 0 || epilogue > 2 || wshift < 0 || wshift > 14) {
        um_set_error("um_linear_fwd: unknown epilogue %d or wshift %d", epilogue, wshift);
        return -1;
    }
    if ((long)m * k * 2 >= (1L << 32) || (long)n * k * 2 >= (1L << 32)) {
        um_set_error("um_linear_fwd: operand planes beyond 4 GiB are not addressable by this kernel");
        return -4;
    }
    LinArgs a;
    a.a0 = a0;
    a.a1 = a1;
    a.ap = (const unsigned short*)a_planes;
    a.a_plane_stride = (long)m * k;
    a.wp = (const unsigned short*)w_planes;
    a.w_plane_stride = (long)n * k;
    a.M = m;
    a.N = n;
    a.K = k;
    a.out_scale = ldexpf(1.f, -wshift);
    a.outp = (unsigned short*)out;
    a.out_plane_stride = (long)m * n;
    a.outf = (float*)out;
    a.gamma = gamma;
    a.beta = beta;
    a.residual = residual;
    a.eps = eps;
    a.bias = nullptr;
    a.bias_scale = 0.f;
    a.range_flag = (mode == 0) ? um_range_flag_dev() : nullptr;
    hipError_t e;
    if (a_planes) {
        if (epilogue == EPI_LN) e = launch_linear<A_PLANES, EPI_LN>(a, mode, stream);
        else if (epilogue == EPI_PLANES) e = launch_linear<A_PLANES, EPI_PLANES>(a, mode, stream);
        else e = launch_linear<A_PLANES, EPI_GELU_PLANES>(a, mode, stream);
    } else if (a1) {
        if (epilogue == EPI_LN) e = launch_linear<A_CONCAT, EPI_LN>(a, mode, stream);
        else if (epilogue == EPI_PLANES) e = launch_linear<A_CONCAT, EPI_PLANES>(a, mode, stream);
        else e = launch_linear<A_CONCAT, EPI_GELU_PLANES>(a, mode, stream);
    } else {
        if (epilogue == EPI_LN) e = launch_linear<A_F32, EPI_LN>(a, mode, stream);
        else if (epilogue == EPI_PLANES) e = launch_linear<A_F32, EPI_PLANES>(a, mode, stream);
        else e = launch_linear<A_F32, EPI_GELU_PLANES>(a, mode, stream);
    }
    return (int)e;
}

// nn.Linear WITH bias (the propagation layer's projections, unimatch/attention.py:196-205, 229-232):
//     C = (A . W^T) * out_mul + bias * bias_mul      ->  operand planes [NS][M][N]  or  fp32 [M, N]
// out_mul / bias_mul let the caller hand the result to um_prop_global_attn_planes() already carrying the plane factor.
extern "C" int um_linear_bias_fwd(const float* a0, const void* a_planes, const void* w_planes, const float* bias, int m, int n,
                                  int k, int wshift, float out_mul, float bias_mul, void* out_planes, float* out_f32, int mode,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (m <= 0 || (m + 127) / 128 > 65535 || n <= 0 || k <= 0 || n % 128 != 0 || k % 32 != 0 || !w_planes || !bias ||
        (mode != 0 && mode != 1) || wshift < 0 || wshift > 14) {
        um_set_error("um_linear_bias_fwd: bad argument (m=%d n=%d k=%d wshift=%d: n must be a multiple of 128, k of 32)", m, n, k, wshift);
        return -1;
    }
    if (((a0 != nullptr) + (a_planes != nullptr)) != 1 || ((out_planes != nullptr) + (out_f32 != nullptr)) != 1) {
        um_set_error("um_linear_bias_fwd: give exactly one input (a0 | a_planes) and exactly one output (out_planes | out_f32)");
        return -1;
    }
    if (((uintptr_t)bias & 15) != 0) {
        um_set_error("um_linear_bias_fwd: bias must be 16-byte aligned");
        return -1;
    }
    if ((long)m * k * 2 >= (1L << 32) || (long)n * k * 2 >= (1L << 32)) {
        um_set_error("um_linear_bias_fwd: operand planes beyond 4 GiB are not addressable by this kernel");
        return -4;
    }
    LinArgs a;
    a.a0 = a0;
    a.a1 = nullptr;
    a.ap = (const unsigned short*)a_planes;
    a.a_plane_stride = (long)m * k;
    a.wp = (const unsigned short*)w_planes;
    a.w_plane_stride = (long)n * k;
    a.M = m;
    a.N = n;
    a.K = k;
    a.out_scale = ldexpf(1.f, -wshift) * out_mul;
    a.outp = (unsigned short*)out_planes;
    a.out_plane_stride = (long)m * n;
    a.outf = out_f32;
    a.gamma = a.beta = a.residual = nullptr;
    a.eps = 0.f;
    a.bias = bias;
    a.bias_scale = bias_mul;
    a.range_flag = (mode == 0) ? um_range_flag_dev() : nullptr;
    hipError_t e;
    if (a_planes) e = out_planes ? launch_linear<A_PLANES, EPI_PLANES>(a, mode, stream) : launch_linear<A_PLANES, EPI_F32>(a, mode, stream);
    else e = out_planes ? launch_linear<A_F32, EPI_PLANES>(a, mode, stream) : launch_linear<A_F32, EPI_F32>(a, mode, stream);
    return (int)e;
}
