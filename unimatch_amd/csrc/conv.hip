// 2-D convolution as an implicit GEMM on the matrix cores, NHWC.                        gfx950 / wave64 / MFMA
//
//   out[p, co] = bias[co] + sum_{ky,kx,ci} in[b, y*s - ph + ky, x*s - pw + kx, ci] * W[co, ky, kx, ci]      p = (b, y, x)
//
// for the convolutions that sit either side of the matching path: the refinement block (SURVEY.md 8(f) rank 3:
// unimatch/reg_refine.py:6-119 -- 3x3, 1x1, 1x5 / 5x1, 7x7) and the residual encoder (unimatch/backbone.py:7-133 -- 3x3
// at stride 1 / 2, 1x1 projections).  MIOpen runs these in fp32 Winograd at 105-116 TFLOP/s effective; this kernel uses
// the same "exact" arithmetic as the rest of the library (fp16 hi + lo split operands, three MFMA products, fp32
// accumulation: fp32-equivalent results) and the same tile machinery as linear.hip.
//
// Operands: activations are ALREADY operand planes [NS][rows + 1][Cin] in NHWC order (written by the normalisation
// kernel that precedes every convolution), with one extra all-zero row: a tap that falls outside the image reads that
// row, so zero padding costs nothing in the main loop.  Weights are planes [NS][Cout][KH*KW*Cin] (tap-major, channel
// minor), pre-scaled by 2^wshift.  The GEMM's K axis is walked tap by tap in chunks of 32 channels; for every chunk a
// workgroup moves a [128 pixels x 32 channels] activation tile and a [32 NT outputs x 32] weight tile into a 2-slot
// LDS ring by LDS-DMA -- the only per-tap work is two row indices per lane.
// Decomposition: workgroup = 4 waves = 128 output pixels x 32 NT output channels (NT = 2, 3, 4), wave = 32 pixels,
// computed transposed (D^T = W . A^T) so that lane = pixel.  Output: fp32 NHWC and / or the next convolution's operand
// planes (+ bias, activation, SepConvGRU gates, InstanceNorm statistics), see conv_epilogue.
//
// Three kernels share the operand format and the epilogue; conv_pick() chooses:
//   conv_kernel        any geometry (strides, 1x1, 5x1, 7x7 via um_conv7_fwd): one staged tile per tap and 32-channel chunk
//   conv_rows_kernel   same-size stride-1 rows of 3 / 5 taps: one 256-pixel window per kernel row serves its KW taps
//   conv_patch_kernel  3x3 / stride 1 / pad 1: an 8 x 32 pixel tile whose halo patch is staged once per 16-channel chunk for
//                      all nine taps (the default for every 3x3 layer of the encoder and the refinement block)
#include <cstdlib>
#include <cstring>
// -DUM_CONV_2P=1|2 (diagnostic builds: profiles/r04_precision_budget_conv.txt): two MFMA products instead of three -- 1 drops
// W_lo . A_hi (weights effectively rounded to one fp16 plane), 2 drops W_hi . A_lo (activations in one plane).
#ifndef UM_CONV_2P
#define UM_CONV_2P 0
#endif
#include "common.h"
#include "planes.h"

struct ConvArgs {
    const unsigned short* ap;     // [NS][rows_in + 1][Cin], row rows_in = zeros
    long a_plane_stride;
    unsigned row_stride;          // bytes between consecutive input "rows" (pixels): Cin * 2, or less when the Cin
                                  // contiguous elements of a tap span several packed pixels (the stem, um_stem_conv_fwd)
    unsigned zero_row;            // row index of the all-zero row
    const unsigned short* wp;     // [NS][Cout][taps * Cin]
    long w_plane_stride;
    const float* bias;            // [Cout] or null
    float* out;                   // optional fp32 [M][out_ld], written at column offset out_coff
    unsigned short* outp;         // optional operand planes [NS][...][outp_ld] at column offset outp_coff (the next
    long outp_plane_stride;       //   convolution's input: no fp32 round trip, channel concatenation by offset)
    int out_ld, out_coff, outp_ld, outp_coff;
    int act;                      // 0 none, 1 ReLU, 2 sigmoid, 3 tanh
    // SepConvGRU gate arithmetic in the epilogue (unimatch/reg_refine.py:66-74), after the activation:
    //   gate 1 (z | r convolution, Cout = 2C): columns < C (z) go to `out` as usual; columns >= C (r) are multiplied by
    //           gate_h[row][col - C] and written ONLY as planes at outp_coff + col - C          (r * h)
    //   gate 2 (q convolution, Cout = C): v <- (1 - z) h + z v with z = gate_z[row][col], h = gate_h[row][col]; gate_h is
    //           updated in place and v is written as planes (and to `out` if given)            (the new hidden state)
    int gate, gate_c, gate_zld;
    float* gate_h;
    const float* gate_z;
    float* gate_hout;             // gate 2: where the new hidden state goes (gate_h itself: in place), row stride gate_hout_ld
    int gate_hout_ld;
    float* stats;                 // optional [M / 128][3][Cout]: per 128-pixel tile (mean, 0, sum of squared deviations)
    // optional per-pixel addend, fp32 [M][addend_ld]: added to the scaled accumulator (+ bias) BEFORE the activation.  A convolution is
    // linear in its input channels, so the contribution of input channels that do not change between calls (the refinement loop's
    // context features and initial hidden state, unimatch.py:315-331) is computed once and comes in here (round 4).
    const float* addend;
    int addend_ld;
    int B, Hi, Wi, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w;
    int M;                        // B * Ho * Wo
    float out_scale;              // 2^-wshift
    int xcd;                      // XCD-aware workgroup order (0: plain; A/B switch UM_CONV_NO_XCD)
};

__device__ __forceinline__ void conv_dma16(const void* base, unsigned byte_off, const unsigned char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(dst)
                 : "memory");
}

// ---- epilogue shared by all kernels: lane holds, for pixel rloc0 + (lane & 31) of image bt (rloc0 = the wave's first pixel,
// nvalid <= 32 of its pixels exist; part0 = statistics part of the workgroup's first four waves), outputs
// n0 + 32*nt + 8*g + 4*half + i (reg 4*g + i).  Output tiles never straddle images (the last tile of an image is ragged).  `scratch` = LDS beyond the (now
// idle) staging ring: 2 * 32 NT floats per wave for the statistics.  One pass handles the n-tiles [NT0, NT0 + NTP) of the
// wave's NT (the transposed tile of a pass must fit the ring: conv_epilogue below picks the pass width).
template <int V>
struct IntC {
    static constexpr int value = V;
};

template <typename T, int NS, int NT, int NT0, int NTP, int WSTRIDE>   // WSTRIDE: bytes of a wave's private staging block
__device__ __forceinline__ void conv_epilogue_pass(const ConvArgs& a, f32x16 (&acc)[NT], unsigned char* lds, unsigned char* scratch,
                                                   int bt, int rloc0, int nvalid, int part0, bool dense_parts, int n0, int tid,
                                                   int wave, int lane, int half) {
    const int P = a.Ho * a.Wo;                                    // pixels per image
    // The tile goes through the idle staging ring (every wave transposes its own 32 x 32NTP block; 16-byte chunk c of
    // row r at chunk c ^ (r & 7)) and leaves as full rows: direct stores from this layout hit 32 partial lines each.
    // The wave-uniform switches (activation, gate mode) are taken ONCE around straight-line loops: with K = 576 .. 1152 the
    // encoder's convolutions spend as many issue cycles here as in their main loop.
    constexpr int ROWB = 32 * NTP * 4;                            // bytes per pixel row of the pass's tile
    const int nb = n0 + 32 * NT0;                                 // first output channel of the pass
    unsigned char* stg = lds + wave * WSTRIDE;                    // same block in every pass: waves are not synchronised
    const int tl = lane & 31;
    // ---- side tensors of the pass's rows (addend, gate z / h): REQUESTED before they are needed.  h is updated in place and nothing
    // here is __restrict__, so a load written after a store waits for it: one exposed round trip to HBM per iteration and wave with
    // the matrix pipes idle (the hoisted GRU gates of config 4 spent a fifth of their launches like that).  Batch 0 is requested before
    // the transpose below; with 128-wide tiles (one workgroup of <= 256 registers per lane) batch b + 1 is requested before batch b is used.
    constexpr int CPR = 8 * NTP;                                  // 16-byte chunks per row
    constexpr int ITER = 32 * CPR / 64;
    constexpr int BATCH = ITER < 4 ? ITER : 4, NBATCH = ITER / BATCH, NSIDE = NT == 4 ? 2 : 1;
    static_assert(ITER % BATCH == 0, "batches cover the iterations");
    const long rowbase = (long)bt * P + rloc0;                    // first output row of this wave
    const int nrows = nvalid;                                     // valid rows (0: none)
    const int gate = a.gate;
    float* hbase = gate ? a.gate_h + rowbase * a.gate_c : nullptr;
    const float* zbase = gate == 2 ? a.gate_z + rowbase * a.gate_zld : nullptr;
    const float* abase = a.addend ? a.addend + rowbase * a.addend_ld : nullptr;
    float* hobase = gate == 2 ? a.gate_hout + rowbase * a.gate_hout_ld : nullptr;
    f32x4 ad[NSIDE][BATCH], gz[NSIDE][BATCH], gh[NSIDE][BATCH];
    auto preload = [&](int b, int slot) {                         // both compile-time constants once the callers' loops are unrolled
        if (!abase && !gate) return;
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int idx = (b * BATCH + j) * 64 + lane;
            const int r = idx / CPR, c = idx - r * CPR;
            const int col = nb + 4 * c;
            if (r < nrows && col < a.Cout) {
                if (abase) ad[slot][j] = *reinterpret_cast<const f32x4*>(abase + (unsigned)(r * a.addend_ld + col));
                if (gate == 1 && col >= a.gate_c) gh[slot][j] = *reinterpret_cast<const f32x4*>(hbase + (unsigned)(r * a.gate_c + col - a.gate_c));
                if (gate == 2) {
                    gz[slot][j] = *reinterpret_cast<const f32x4*>(zbase + (unsigned)(r * a.gate_zld + col));
                    gh[slot][j] = *reinterpret_cast<const f32x4*>(hbase + (unsigned)(r * a.gate_c + col));
                }
            }
        }
    };
    preload(0, 0);
    auto to_lds = [&](auto act_tag, auto bias_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool BIAS = decltype(bias_tag)::value != 0;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 32 * nt + 8 * g + 4 * half;
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[NT0 + nt][4 * g + i] * a.out_scale;
                if (BIAS) v = v + *reinterpret_cast<const f32x4*>(a.bias + min(n, a.Cout - 4));   // columns >= Cout are dropped below
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (ACT == 1) v[i] = fmaxf(v[i], 0.f);
                    else if (ACT == 2) v[i] = 1.0f / (1.0f + __expf(-v[i]));
                    else if (ACT == 3) v[i] = tanhf(v[i]);
                }
                const int c = 8 * nt + 2 * g + half;             // 16-byte chunk index inside the row
                *reinterpret_cast<f32x4*>(stg + tl * ROWB + ((c ^ (tl & 7)) << 4)) = v;
            }
    };
    auto to_lds_b = [&](auto act_tag) {
        if (a.bias) to_lds(act_tag, IntC<1>{});
        else to_lds(act_tag, IntC<0>{});
    };
    switch (a.addend ? 0 : a.act) {                               // with an addend the activation waits for it (store_rows)
        case 0: to_lds_b(IntC<0>{}); break;
        case 1: to_lds_b(IntC<1>{}); break;
        case 2: to_lds_b(IntC<2>{}); break;
        default: to_lds_b(IntC<3>{}); break;
    }
    __builtin_amdgcn_wave_barrier();
    if (a.stats) {
        // InstanceNorm statistics of the NEXT layer for free: column sums of the tile that is sitting in LDS anyway.
        // Per wave: its valid pixels (32 except in the ragged last tile of an image), shifted by the first one (no
        // cancellation); lane = (chunk of 4 channels, group of rows); the row groups are folded with shuffles; the four waves of
        // 128 pixels are merged with the parallel-variance formula; um_nhwc_instance_norm merges the tiles (in fp64).
        constexpr int RG = 64 / CPR >= 8 ? 8 : 64 / CPR >= 4 ? 4 : 2;       // row groups (lanes beyond RG * CPR idle)
        constexpr int RPG = 32 / RG;
        const int nv = nvalid;
        const int c = lane % CPR, rg = lane / CPR;
        const f32x4 k = *reinterpret_cast<const f32x4*>(stg + (c << 4));                     // row 0 (chunk c ^ 0)
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (rg < RG) {
            if (nv == 32) {
#pragma unroll
                for (int j = 0; j < RPG; ++j) {
                    const int r = rg * RPG + j;
                    const f32x4 d = *reinterpret_cast<const f32x4*>(stg + r * ROWB + ((c ^ (r & 7)) << 4)) - k;
                    s1 = s1 + d;
                    s2 = s2 + d * d;
                }
            } else {
#pragma unroll
                for (int j = 0; j < RPG; ++j) {
                    const int r = rg * RPG + j;
                    f32x4 d = *reinterpret_cast<const f32x4*>(stg + r * ROWB + ((c ^ (r & 7)) << 4)) - k;
                    if (r >= nv) d = f32x4{0.f, 0.f, 0.f, 0.f};
                    s1 = s1 + d;
                    s2 = s2 + d * d;
                }
            }
        }
#pragma unroll
        for (int step = CPR * RG / 2; step >= CPR; step >>= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s1[i] += __shfl_down(s1[i], step, 64);
                s2[i] += __shfl_down(s2[i], step, 64);
            }
        float* ws = reinterpret_cast<float*>(scratch) + wave * (2 * 32 * NT) + 32 * NT0;
        float* cnt = reinterpret_cast<float*>(scratch) + 8 * (2 * 32 * NT);             // valid pixels per wave
        if (lane == 0) cnt[wave] = (float)nv;
        if (lane < CPR) {
            const float inv = nv > 0 ? 1.0f / (float)nv : 0.f;
            *reinterpret_cast<f32x4*>(ws + 4 * c) = k + s1 * inv;                            // mean of the wave's valid pixels
            *reinterpret_cast<f32x4*>(ws + 32 * NT + 4 * c) = s2 - s1 * s1 * inv;            // sum of squared deviations
        }
        __syncthreads();
        // threads 0 .. 32 NTP - 1 of every group of four waves merge that group's 128 pixels
        const int grp = tid >> 8, gt = tid & 255;
        // a group's first wave has pixels if any has; an empty group writes nothing unless the kernel numbers its parts densely
        if (gt < 32 * NTP && nb + gt < a.Cout && (cnt[4 * grp] > 0.f || dense_parts)) {
            const float* w0 = reinterpret_cast<const float*>(scratch) + grp * 4 * (2 * 32 * NT) + 32 * NT0;
            float mean = w0[gt], m2 = w0[32 * NT + gt];
            float n = cnt[4 * grp];
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const float nw = cnt[4 * grp + wv];
                if (nw > 0.f) {
                    const float mw = w0[wv * (2 * 32 * NT) + gt], m2w = w0[wv * (2 * 32 * NT) + 32 * NT + gt];
                    const float delta = mw - mean, nn = n + nw;
                    mean += delta * (nw / nn);
                    m2 += m2w + delta * delta * (n * nw / nn);
                    n = nn;
                }
            }
            float* pr = a.stats + ((long)(part0 + grp) * 3) * a.Cout + nb + gt;
            pr[0] = n > 0.f ? mean : 0.f;                                   // part = (mean, pixel count, sum of squared deviations)
            pr[a.Cout] = n;
            pr[2 * a.Cout] = n > 0.f ? m2 : 0.f;
        }
    }
    // ---- full-row stores: wave-uniform 64-bit bases, 32-bit lane offsets
    auto store_rows = [&](auto gate_tag, auto add_tag) {
        constexpr int GATE = decltype(gate_tag)::value;
        constexpr bool ADD = decltype(add_tag)::value != 0;
        float* obase = a.out ? a.out + rowbase * a.out_ld + a.out_coff : nullptr;
        unsigned short* pbase = a.outp ? a.outp + rowbase * a.outp_ld + a.outp_coff : nullptr;
#pragma unroll
        for (int b = 0; b < NBATCH; ++b) {
            const int slot = NSIDE == 2 ? (b & 1) : 0;
            if ((ADD || GATE != 0) && NSIDE == 2 && b + 1 < NBATCH) preload(b + 1, (b + 1) & 1);     // plain stores: one straight-line block
            if ((ADD || GATE != 0) && NSIDE == 1 && b > 0) preload(b, 0);
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int idx = (b * BATCH + j) * 64 + lane;
                const int r = idx / CPR, c = idx - r * CPR;
                f32x4 d = *reinterpret_cast<const f32x4*>(stg + r * ROWB + ((c ^ (r & 7)) << 4));
                int col = nb + 4 * c;
                if (r < nrows && col < a.Cout) {
                    if (ADD) {                                    // full-row reads, then the activation that waited for them
                        d = d + ad[slot][j];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (a.act == 1) d[i] = fmaxf(d[i], 0.f);
                            else if (a.act == 2) d[i] = 1.0f / (1.0f + __expf(-d[i]));
                            else if (a.act == 3) d[i] = tanhf(d[i]);
                        }
                    }
                    bool to_out = obase != nullptr, to_planes = pbase != nullptr;
                    if (GATE == 1) {
                        if (col >= a.gate_c) {
                            col -= a.gate_c;
                            d = d * gh[slot][j];
                            to_out = false;
                        } else {
                            to_planes = false;
                        }
                    } else if (GATE == 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) d[i] = (1.0f - gz[slot][j][i]) * gh[slot][j][i] + gz[slot][j][i] * d[i];
                        *reinterpret_cast<f32x4*>(hobase + (unsigned)(r * a.gate_hout_ld + col)) = d;
                    }
                    // streaming stores: the tile is next touched by another kernel; as ordinary stores these lines evicted the
                    // activation rows the neighbouring workgroups are about to re-read (+1.5 % end to end, same-box ABAB)
                    if (to_out) __builtin_nontemporal_store(d, reinterpret_cast<f32x4*>(obase + (unsigned)(r * a.out_ld + col)));
                    if (to_planes) {
                        unsigned short* dst = pbase + (unsigned)(r * a.outp_ld + col);
                        const unsigned h0 = T::pack2(d[0], d[1]), h1 = T::pack2(d[2], d[3]);
                        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};     // planes feed the next convolution: ordinary stores
                        if (NS == 2) {
                            const f32x2 u0 = T::unpack2(h0), u1 = T::unpack2(h1);
                            const u32x2 lo = u32x2{T::pack2(d[0] - u0[0], d[1] - u0[1]), T::pack2(d[2] - u1[0], d[3] - u1[1])};
                            *reinterpret_cast<u32x2*>(dst + a.outp_plane_stride) = lo;
                        }
                    }
                }
            }
        }
    };
    auto store_rows_g = [&](auto gate_tag) {
        if (a.addend) store_rows(gate_tag, IntC<1>{});
        else store_rows(gate_tag, IntC<0>{});
    };
    if (a.gate == 0) store_rows_g(IntC<0>{});
    else if (a.gate == 1) store_rows_g(IntC<1>{});
    else store_rows_g(IntC<2>{});
}

// NTE = n-tiles per epilogue pass (NT: the whole tile at once).
template <typename T, int NS, int NT, int NTE = NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NT], unsigned char* lds, unsigned char* scratch,
                                              int bt, int rloc0, int nvalid, int part0, bool dense_parts, int n0, int tid, int wave,
                                              int lane, int half) {
    constexpr int NP = NTE < NT ? NTE : NT;
    constexpr int WSTRIDE = 32 * (32 * NP * 4);
    conv_epilogue_pass<T, NS, NT, 0, NP, WSTRIDE>(a, acc, lds, scratch, bt, rloc0, nvalid, part0, dense_parts, n0, tid, wave, lane, half);
    if constexpr (NTE < NT)          // the wave's staging block is private and DS operations of a wave execute in order
        conv_epilogue_pass<T, NS, NT, NTE, NT - NTE, WSTRIDE>(a, acc, lds, scratch, bt, rloc0, nvalid, part0, dense_parts, n0, tid, wave,
                                                              lane, half);
}

template <typename T, int NS, int NT>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvArgs a) {
    constexpr int TILE = 128 * 64;               // one 128-row x 64-byte operand tile (one plane, one stage)
    constexpr int WTILE = 32 * NT * 64;          // one weight plane of a stage: 32 NT output rows x 64 bytes
    constexpr int STAGE = NS * (TILE + WTILE);   // activation planes then weight planes (NT = 2: 48 KB ring -> 3 WG / CU)
    constexpr int EPI = 4 * 32 * (32 * NT * 4);   // the epilogue's transposed tile
    constexpr int RING = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
    // + per-wave (mean, M2) columns at the stride of 8 waves + per-wave pixel counts (the epilogue's scratch layout)
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING + 8 * 2 * 32 * NT * 4 + 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int P = a.Ho * a.Wo, tpi = (P + 127) / 128;            // pixels / 128-pixel tiles per image
    // 1-D grid, XCD-aware: every XCD (own L2) gets a contiguous range of (pixel tile, channel tile) pairs, channel tile
    // fastest -- the workgroups that share an activation tile and the neighbours that share its halo rows hit the same L2
    const int ny = (a.Cout + 32 * NT - 1) / (32 * NT);
    const int wg = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tile = wg / ny;
    const int bt = tile / tpi, pl0 = (tile - bt * tpi) * 128;
    const int n0 = (wg - tile * ny) * (32 * NT);
    const int cpt = a.Cin >> 5;                  // 32-channel chunks per tap
    const int nstage = a.KH * a.KW * cpt;
    const int ktot = a.KH * a.KW * a.Cin;

    // ---- the two pixels whose rows this lane moves (DMA: one instruction = 16 rows x 64 B; wave w moves rows 32w..) ----
    const int dcp = lane & 3;
    int py[2], px[2], pbase[2];                  // input-space origin (y*s - ph, x*s - pw) and image base row
    bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = pl0 + 32 * wave + 16 * i + (lane >> 2);
        pok[i] = p < P;
        const int rem = pok[i] ? p : 0;
        const int y = rem / a.Wo, x = rem - y * a.Wo;
        py[i] = y * a.stride - a.pad_h;
        px[i] = x * a.stride - a.pad_w;
        pbase[i] = bt * a.Hi * a.Wi;
    }
    const unsigned zero_row = a.zero_row;
    unsigned rowoff[2];                          // byte offset of the source row of the tap being staged
    auto set_tap = [&](int ky, int kx) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = py[i] + ky, ix = px[i] + kx;
            const bool ok = pok[i] && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const unsigned row = ok ? (unsigned)(pbase[i] + iy * a.Wi + ix) : zero_row;
            rowoff[i] = row * a.row_stride;
        }
    };
    int ky = 0, kx = 0, cc = 0;                  // position of the NEXT stage to be issued
    // 16-byte chunk cp of row r holds source chunk cp ^ ((r >> 2) & 3) (conflict-free ds_read_b128 fragments)
    auto stage_async = [&](int c0, int kglob, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * wave + 16 * i + (lane >> 2);
            const int sc = dcp ^ ((r >> 2) & 3);
            const unsigned off = rowoff[i] + (unsigned)((c0 + 8 * sc) * 2);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) conv_dma16(a.ap + pl * a.a_plane_stride, off, buf + pl * TILE + (32 * wave + 16 * i) * 64);
        }
        // the weight tile is 2 NT blocks of 16 rows; block j goes to wave j mod 4 (every wave issues its share)
#pragma unroll
        for (int i = 0; i < (2 * NT + 3) / 4; ++i) {
            const int j = wave + 4 * i;
            if (j < 2 * NT) {
                const int r = 16 * j + (lane >> 2);
                const int sc = dcp ^ ((r >> 2) & 3);
                const int n = min(n0 + r, a.Cout - 1);
                const unsigned off = (unsigned)(((long)n * ktot + kglob + 8 * sc) * 2);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    conv_dma16(a.wp + pl * a.w_plane_stride, off, buf + NS * TILE + pl * WTILE + (16 * j) * 64);
            }
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // fragment offsets inside a tile: row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4), chunk = 2 * kstep + half
    const int fr = lane & 31;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = fr * 64 + (((2 * ks + half) ^ ((fr >> 2) & 3)) << 4);

    // ---- prologue: stage 0 ------------------------------------------------------------------------------------
    set_tap(0, 0);
    stage_async(0, 0, lds);
    auto advance = [&]() {                       // move (ky, kx, cc) one stage on; recompute the row offsets on a new tap
        if (++cc == cpt) {
            cc = 0;
            if (++kx == a.KW) {
                kx = 0;
                ++ky;
            }
            set_tap(ky, kx);
        }
    };
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < nstage; ++s) {
        unsigned char* cur = lds + (s & 1) * STAGE;
        unsigned char* nxt = lds + ((s & 1) ^ 1) * STAGE;
        if (s + 1 < nstage) {
            stage_async(cc * 32, (s + 1) * 32, nxt);
            advance();
        }
        const unsigned char* at = cur + 32 * wave * 64;           // this wave's 32 pixel rows
        const unsigned char* wt = cur + NS * TILE;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const i16x8 bh = *reinterpret_cast<const i16x8*>(at + foff[ks]);
            i16x8 bl;
            if (NS == 2) bl = *reinterpret_cast<const i16x8*>(at + TILE + foff[ks]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const i16x8 wh = *reinterpret_cast<const i16x8*>(wt + nt * 32 * 64 + foff[ks]);
                if (NS == 2) {
                    const i16x8 wl = *reinterpret_cast<const i16x8*>(wt + WTILE + nt * 32 * 64 + foff[ks]);
                    if (UM_CONV_2P != 1) acc[nt] = T::mfma(wl, bh, acc[nt]);
                    if (UM_CONV_2P != 2) acc[nt] = T::mfma(wh, bl, acc[nt]);
                }
                acc[nt] = T::mfma(wh, bh, acc[nt]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    {
        const int rloc0 = pl0 + 32 * wave;
        conv_epilogue<T, NS, NT>(a, acc, lds, lds + RING, bt, rloc0, min(32, max(0, P - rloc0)), bt * ((P + 127) / 128) + pl0 / 128, false,
                                 n0, tid, wave, lane, half);
    }
}

// ---- row-window variant: same-size stride-1 convolutions with KW = 3 / 5 horizontal taps -------------------------------------
// In the generic kernel every tap stages its own [128 x 32] activation tile, although the KW horizontal taps of one kernel row
// read the SAME input pixels shifted by one: with flat pixel indices (output and input images have the same shape) tap
// (ky, kx) of output pixel p is input pixel p + (ky - ph) W + (kx - pw).  Here a workgroup of 8 waves owns 256 consecutive
// output pixels of one image and, per stage = (ky, 16 channels), stages ONE window of 256 + KW - 1 input rows that serves all
// KW taps (fragment reads at row offset kx) next to the KW weight tiles: KW x fewer activation DMAs and barriers per MFMA than
// the generic kernel.  Taps that cross the left / right image border are zeroed per lane at fragment level; rows above /
// below the image are staged from the zero row.
// 16-channel stages (rows of 32 bytes) keep the ring small: at NT = 2 / 3 TWO workgroups share a CU (<= 80 KB each, <= 128
// VGPRs).  The encoder's convolutions have only K = 576 .. 1152, so a lone workgroup's prologue and epilogue (transpose,
// statistics, 64 .. 96 KB of stores, all CUs in phase) ran with the matrix pipes idle.  At NT = 4 (refinement block: 1x5 GRU
// gates, 3x3 -> 128 / 256) the small stage is what fits the window + KW weight tiles into LDS at all.  Layout: 16-byte chunk
// c (0 / 1) of row r sits at chunk c ^ ((r >> 3) & 1), which makes the ds_read_b128 lane groups of a 32-row fragment
// conflict-free; one DMA instruction moves 32 rows.
// Measured and dropped (round 1, same-box ABAB of bench.py): 32-channel stages with one workgroup per CU (equal within
// 1 %); NSLOT = 3 .. 5 ring slots, i.e. 2 .. 4 stages in flight with one workgroup per CU: -5 % end to end -- these
// kernels are not DMA-latency-bound, the second resident workgroup is worth more than the deeper ring.
template <int NS, int NT, int KW, int NSLOT>
struct ConvRowsLds {
    static constexpr int NTE = NT < 2 ? NT : 2;                   // epilogue pass width
    static constexpr int ATILE = 288 * 32;                        // 9 DMA blocks of 32 rows >= 256 + KW - 1
    static constexpr int WTILE = 32 * NT * 32;
    static constexpr int STAGE = NS * (ATILE + KW * WTILE);
    static constexpr int EPI = 8 * 32 * (32 * NTE * 4);
    static constexpr int RING = (NSLOT * STAGE > EPI) ? NSLOT * STAGE : EPI;
    static constexpr int TOTAL = RING + 8 * 2 * 32 * NT * 4 + 64;   // + statistics scratch: (mean, M2) columns and pixel counts per wave
};

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate)
__device__ __forceinline__ void conv_wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // always safe
    }
}

// NSLOT = depth of the staging ring: stage s + NSLOT - 1 is in flight while stage s is multiplied (2 everywhere, see above).
template <typename T, int NS, int NT, int KW, int NSLOT>
__global__ __launch_bounds__(512, (NSLOT == 2 && NT < 4 ? 2 : 1)) void conv_rows_kernel(ConvArgs a) {
    using L = ConvRowsLds<NS, NT, KW, NSLOT>;
    constexpr int ATILE = L::ATILE, WTILE = L::WTILE, STAGE = L::STAGE, RING = L::RING;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, fr = lane & 31;
    const int P = a.Ho * a.Wo, tpi = (P + 255) / 256;
    const int ny = (a.Cout + 32 * NT - 1) / (32 * NT);           // 1-D XCD-aware grid, see conv_kernel
    const int wg = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tile = wg / ny;
    const int bt = tile / tpi, pl0 = (tile - bt * tpi) * 256;
    const int n0 = (wg - tile * ny) * (32 * NT);
    const int cpt = a.Cin >> 4;                  // 16-channel chunks per tap
    const int nstage = a.KH * cpt;
    const int ktot = a.KH * KW * a.Cin;

    // ---- window rows this lane stages: block jr (32 rows), row j = 32 jr + (lane >> 1); blocks 0..7 by wave jr, block 8 by
    // wave 7 (which has no weight block at KW NT <= 7)
    const int dcp = lane & 1;
    int wy[2];
    long wflat[2];
    bool wok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int jr = i == 0 ? wave : 8;
        const int j = 32 * jr + (lane >> 1);
        const int p = pl0 + j - a.pad_w;
        wok[i] = (i == 0 || wave == 7) && p >= 0 && p < P;
        const int rem = wok[i] ? p : 0;
        wy[i] = rem / a.Wo - a.pad_h;
        wflat[i] = (long)bt * P + rem - (long)a.pad_h * a.Wi;
    }
    unsigned rowoff[2];
    auto set_ky = [&](int ky) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = wy[i] + ky;
            const bool ok = wok[i] && (unsigned)iy < (unsigned)a.Hi;
            const unsigned row = ok ? (unsigned)(wflat[i] + (long)ky * a.Wi) : a.zero_row;
            rowoff[i] = row * a.row_stride;
        }
    };
    // staging pieces of a wave: 0 / 1 = window blocks, 2.. = weight blocks (tap kx, 32 output rows jb)
    constexpr int NPIECE = 2 + (KW * NT + 7) / 8;
    auto stage_piece = [&](int k, int ky, int cc, unsigned char* buf) {
        if (k < 2) {
            const int jr = k == 0 ? wave : 8;
            if (k == 0 || wave == 7) {
                const int r = 32 * jr + (lane >> 1);
                const int sc = dcp ^ ((r >> 3) & 1);
                const unsigned off = rowoff[k] + (unsigned)((cc * 16 + 8 * sc) * 2);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) conv_dma16(a.ap + pl * a.a_plane_stride, off, buf + pl * ATILE + (32 * jr) * 32);
            }
        } else {
            const int q = wave + 8 * (k - 2);
            if (q < KW * NT) {
                const int kx = q / NT, jb = q - kx * NT;
                const int r = 32 * jb + (lane >> 1);
                const int sc = dcp ^ ((r >> 3) & 1);
                const int n = min(n0 + r, a.Cout - 1);
                const unsigned off = (unsigned)(((long)n * ktot + (ky * KW + kx) * a.Cin + cc * 16 + 8 * sc) * 2);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    conv_dma16(a.wp + pl * a.w_plane_stride, off, buf + NS * ATILE + (kx * NS + pl) * WTILE + (32 * jb) * 32);
            }
        }
    };

    bool tap_ok[KW];
    bool any_masked = false;
    {
        const int p = pl0 + 32 * wave + fr;
        const bool pok = p < P;
        const int x = (pok ? p : 0) % a.Wo;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            tap_ok[kx] = pok && (unsigned)(x + kx - a.pad_w) < (unsigned)a.Wi;
            any_masked |= !tap_ok[kx];
        }
    }
    const bool wave_masked = __any(any_masked);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // fragment offsets (one k-step of 16 channels per stage): weights row fr; activations window row 32 wave + fr + kx
    const int foffw = fr * 32 + ((half ^ ((fr >> 3) & 1)) << 4);
    int foffa[KW];
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
        const int r = 32 * wave + fr + kx;
        foffa[kx] = r * 32 + ((half ^ ((r >> 3) & 1)) << 4);
    }

    // DMA instructions this wave issues per stage (wave-uniform): what s_waitcnt may leave outstanding
    int cnt = NS * (1 + (wave == 7 ? 1 : 0));
#pragma unroll
    for (int k = 2; k < NPIECE; ++k) cnt += (wave + 8 * (k - 2) < KW * NT) ? NS : 0;
    constexpr int D = NSLOT - 1;                 // prefetch distance in stages

    int ky = 0, cc = 0;                          // position of the NEXT stage to be issued
    set_ky(0);
    auto advance = [&]() {
        if (++cc == cpt) {
            cc = 0;
            ++ky;
            set_ky(ky);
        }
    };
    // ---- prologue: stages 0 .. D - 1
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nstage) {
#pragma unroll
            for (int k = 0; k < NPIECE; ++k) stage_piece(k, ky, cc, lds + d * STAGE);
            advance();
        }
    conv_wait_vm(nstage >= D ? cnt * (D - 1) : 0);
    __syncthreads();

    int cur_off = 0, nxt_off = D * STAGE;        // ring offsets of the stage being multiplied / being issued
    for (int s = 0; s < nstage; ++s) {
        unsigned char* cur = lds + cur_off;
        unsigned char* nxt = lds + nxt_off;
        const bool more = s + D < nstage;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const unsigned char* wt = cur + NS * ATILE + kx * NS * WTILE;
            if (more) {                                           // the pieces of stage s + D, spread over the first KW - 1 taps
                constexpr int GROUPS = KW - 1;
#pragma unroll
                for (int k = 0; k < NPIECE; ++k)
                    if (k % GROUPS == kx && kx < GROUPS) stage_piece(k, ky, cc, nxt);
            }
            i16x8 bh = *reinterpret_cast<const i16x8*>(cur + foffa[kx]);
            i16x8 bl;
            if (NS == 2) bl = *reinterpret_cast<const i16x8*>(cur + ATILE + foffa[kx]);
            if (wave_masked && !tap_ok[kx]) {
                const i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                bh = z;
                bl = z;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const i16x8 wh = *reinterpret_cast<const i16x8*>(wt + nt * 32 * 32 + foffw);
                if (NS == 2) {
                    const i16x8 wl = *reinterpret_cast<const i16x8*>(wt + WTILE + nt * 32 * 32 + foffw);
                    if (UM_CONV_2P != 1) acc[nt] = T::mfma(wl, bh, acc[nt]);
                    if (UM_CONV_2P != 2) acc[nt] = T::mfma(wh, bl, acc[nt]);
                }
                acc[nt] = T::mfma(wh, bh, acc[nt]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) advance();
        conv_wait_vm(more ? cnt * (D - 1) : 0);                   // stage s + 1 has landed (the tail drains everything)
        __syncthreads();
        cur_off = cur_off + STAGE == NSLOT * STAGE ? 0 : cur_off + STAGE;
        nxt_off = nxt_off + STAGE == NSLOT * STAGE ? 0 : nxt_off + STAGE;
    }
    {
        const int rloc0 = pl0 + 32 * wave;
        conv_epilogue<T, NS, NT, L::NTE>(a, acc, lds, lds + RING, bt, rloc0, min(32, max(0, P - rloc0)), bt * ((P + 127) / 128) + pl0 / 128,
                                         false, n0, tid, wave, lane, half);
    }
}


// ---- 2-D patch variant: 3x3 / stride 1 / pad 1 ------------------------------------------------------------------------------
// The row-window kernel still stages a tile's activations once per kernel ROW: three windows, image-row-length apart, that a
// 4 MB L2 shared by 64 resident tiles does not keep (1.3 GB fetched per launch for 0.4 GB of input at 256 x 384 x 16).  Here
// the tile is 8 rows x 32 columns and its (8 + 2) x (32 + 2) halo patch is staged ONCE per 16-channel chunk and serves all
// nine taps: wave w = tile row w, lane = column, tap (ky, kx) reads patch pixel (w + ky) * 34 + kx + lane -- 32 consecutive
// 32-byte rows at any offset, conflict-free with the chunk c ^ ((r >> 3) & 1) layout.  2.5x fewer activation bytes through
// L2 / LDS-DMA, no tap masks, no per-row index arithmetic (out-of-image patch pixels point at the zero row once).  Stages are
// (chunk, kernel row): the patch is double-buffered per chunk, the three weight tiles of a kernel row per stage; <= 80 KB at
// NT = 2 / 3 so two workgroups share a CU.  Maps that are not whole tiles have ragged right / bottom tiles (waves with fewer
// than 32, or no, valid pixels); the dispatcher uses this kernel when at least 3/4 of the tiled area is image.  A statistics
// part is a group of four tile rows, numbered densely (2 per tile), with its pixel count.
template <int NS, int NT>
struct ConvPatchLds {
    static constexpr int NTE = NT < 2 ? NT : 2;
    static constexpr int PW = 34, PPIX = 10 * PW;                 // patch: 10 x 34 pixels
    static constexpr int PROWS = 352;                             // 11 DMA blocks of 32 patch pixels
    static constexpr int ATILE = PROWS * 32;                      // one plane of the patch (16 channels)
    static constexpr int WTILE = 32 * NT * 32;                    // one plane of one tap's weight tile
    static constexpr int ASLOT = NS * ATILE, WSLOT = NS * 3 * WTILE;
    static constexpr int WBASE = 2 * ASLOT;
    static constexpr int STAGES = 2 * ASLOT + 2 * WSLOT;
    static constexpr int EPI = 8 * 32 * (32 * NTE * 4);
    static constexpr int SCRATCH = 8 * 2 * 32 * NT * 4 + 64;      // the epilogue's statistics scratch sits behind its transposed
    static constexpr int TOTAL = (STAGES > EPI + SCRATCH) ? STAGES : EPI + SCRATCH;   // tile, inside the then idle staging area:
};                                                                // NT = 3 is exactly 80 KB, two workgroups fill the CU's LDS

template <typename T, int NS, int NT>
__global__ __launch_bounds__(512, (NT < 4 ? 2 : 1)) void conv_patch_kernel(ConvArgs a) {
    using L = ConvPatchLds<NS, NT>;
    constexpr int ATILE = L::ATILE, WTILE = L::WTILE, ASLOT = L::ASLOT, WSLOT = L::WSLOT, WBASE = L::WBASE;
    constexpr int PW = L::PW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, fr = lane & 31;
    const int twn = (a.Wo + 31) >> 5, tpi = twn * ((a.Ho + 7) >> 3);             // tiles per row / per image
    const int ny = (a.Cout + 32 * NT - 1) / (32 * NT);           // 1-D XCD-aware grid, see conv_kernel
    const int wg = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tile = wg / ny;
    const int bt = tile / tpi, tt = tile - bt * tpi;
    const int ty = tt / twn, tx = tt - ty * twn;
    const int y0 = ty * 8, x0 = tx * 32;
    const int n0 = (wg - tile * ny) * (32 * NT);
    const int cpt = a.Cin >> 4;                  // 16-channel chunks
    const int ktot = 9 * a.Cin;

    // ---- patch pixels this lane stages: DMA block blk = wave (and 8 + wave for waves 0..2), pixel j = 32 blk + (lane >> 1)
    const int dcp = lane & 1;
    unsigned rowoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = 32 * (wave + 8 * i) + (lane >> 1);
        const int py = j / PW, px = j - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = j < L::PPIX && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
        const unsigned row = ok ? (unsigned)((bt * a.Hi + iy) * a.Wi + ix) : a.zero_row;
        rowoff[i] = row * a.row_stride;
    }
    auto stage_patch = [&](int cc, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (i == 0 || wave < 3) {
                const int blk = wave + 8 * i;
                const int r = 32 * blk + (lane >> 1);
                const int sc = dcp ^ ((r >> 3) & 1);
                const unsigned off = rowoff[i] + (unsigned)((cc * 16 + 8 * sc) * 2);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) conv_dma16(a.ap + pl * a.a_plane_stride, off, buf + pl * ATILE + (32 * blk) * 32);
            }
    };
    // the three weight tiles of kernel row ky: 3 NT blocks of 32 output rows, block q = (tap kx, rows 32 jb ..)
    auto stage_w = [&](int cc, int ky, unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < (3 * NT + 7) / 8; ++k) {
            const int q = wave + 8 * k;
            if (q < 3 * NT) {
                const int kx = q / NT, jb = q - kx * NT;
                const int r = 32 * jb + (lane >> 1);
                const int sc = dcp ^ ((r >> 3) & 1);
                const int n = min(n0 + r, a.Cout - 1);
                const unsigned off = (unsigned)(((long)n * ktot + (ky * 3 + kx) * a.Cin + cc * 16 + 8 * sc) * 2);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    conv_dma16(a.wp + pl * a.w_plane_stride, off, buf + (kx * NS + pl) * WTILE + (32 * jb) * 32);
            }
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // fragment offsets: weights row fr; activations patch pixel (wave + ky) * 34 + kx + fr
    const int foffw = fr * 32 + ((half ^ ((fr >> 3) & 1)) << 4);
    int foffa[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = (wave + ky) * PW + kx + fr;
            foffa[ky][kx] = r * 32 + ((half ^ ((r >> 3) & 1)) << 4);
        }

    stage_patch(0, lds);
    stage_w(0, 0, lds + WBASE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int cc = 0; cc < cpt; ++cc) {
        const int a_cur = (cc & 1) * ASLOT, a_nxt = ASLOT - a_cur;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int w_par = (cc + ky) & 1;                      // stage index 3 cc + ky has the parity of cc + ky
            const int w_cur = WBASE + w_par * WSLOT, w_nxt = WBASE + (w_par ^ 1) * WSLOT;
            const bool last = ky == 2 && cc + 1 == cpt;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (kx == 0 && !last) stage_w(ky == 2 ? cc + 1 : cc, ky == 2 ? 0 : ky + 1, lds + w_nxt);
                if (kx == 1 && ky == 0 && cc + 1 < cpt) stage_patch(cc + 1, lds + a_nxt);     // used from stage (cc + 1, 0) on
                const i16x8 bh = *reinterpret_cast<const i16x8*>(lds + a_cur + foffa[ky][kx]);
                i16x8 bl;
                if (NS == 2) bl = *reinterpret_cast<const i16x8*>(lds + a_cur + ATILE + foffa[ky][kx]);
                const unsigned char* wt = lds + w_cur + kx * NS * WTILE;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const i16x8 wh = *reinterpret_cast<const i16x8*>(wt + nt * 32 * 32 + foffw);
                    if (NS == 2) {
                        const i16x8 wl = *reinterpret_cast<const i16x8*>(wt + WTILE + nt * 32 * 32 + foffw);
                        if (UM_CONV_2P != 1) acc[nt] = T::mfma(wl, bh, acc[nt]);
                        if (UM_CONV_2P != 2) acc[nt] = T::mfma(wh, bl, acc[nt]);
                    }
                    acc[nt] = T::mfma(wh, bh, acc[nt]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const int nvalid = y0 + wave < a.Ho ? min(32, a.Wo - x0) : 0;               // this wave's pixels inside the image
    conv_epilogue<T, NS, NT, L::NTE>(a, acc, lds, lds + L::EPI, bt, (y0 + wave) * a.Wo + x0, nvalid, (bt * tpi + tt) * 2, true, n0, tid,
                                     wave, lane, half);
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

static int conv_xcd_enabled() {
    static const int on = um_debug_env("UM_CONV_NO_XCD") == nullptr;     // A/B switch (tools/ab_bench.py), read once
    return on;
}

template <int NT>
static hipError_t launch_conv(const ConvArgs& a, int mode, hipStream_t stream) {
    dim3 grid(a.B * ((a.Ho * a.Wo + 127) / 128) * ((a.Cout + 32 * NT - 1) / (32 * NT))), block(256);
    ScopedKernelTimer timer(UM_K_CONV, stream);
    um_census_hit(UM_V_CONV_GENERIC);
    if (mode == 0)
        hipLaunchKernelGGL((conv_kernel<Fp16, 2, NT>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((conv_kernel<Bf16, 1, NT>), grid, block, 0, stream, a);
    return hipGetLastError();
}

template <int NT, int KW, int NSLOT>
static hipError_t launch_conv_rows(const ConvArgs& a, int mode, hipStream_t stream) {
    static bool configured[2] = {false, false};    // opt in to > 64 KB of LDS once per instantiation
    dim3 grid(a.B * ((a.Ho * a.Wo + 255) / 256) * ((a.Cout + 32 * NT - 1) / (32 * NT))), block(512);
    constexpr int LDS2 = ConvRowsLds<2, NT, KW, NSLOT>::TOTAL, LDS1 = ConvRowsLds<1, NT, KW, NSLOT>::TOTAL;
    static_assert(LDS2 <= 160 * 1024 && LDS1 <= 160 * 1024, "ring beyond the CU's LDS");
    ScopedKernelTimer timer(UM_K_CONV, stream);
    um_census_hit(UM_V_CONV_ROWS);
    if (mode == 0) {
        if (!configured[0]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_kernel<Fp16, 2, NT, KW, NSLOT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
            if (e != hipSuccess) return e;
            configured[0] = true;
        }
        hipLaunchKernelGGL((conv_rows_kernel<Fp16, 2, NT, KW, NSLOT>), grid, block, LDS2, stream, a);
    } else {
        if (!configured[1]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_kernel<Bf16, 1, NT, KW, NSLOT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
            if (e != hipSuccess) return e;
            configured[1] = true;
        }
        hipLaunchKernelGGL((conv_rows_kernel<Bf16, 1, NT, KW, NSLOT>), grid, block, LDS1, stream, a);
    }
    return hipGetLastError();
}

template <int NT>
static hipError_t launch_conv_patch(const ConvArgs& a, int mode, hipStream_t stream) {
    static bool configured[2] = {false, false};    // opt in to > 64 KB of LDS once per instantiation
    dim3 grid(a.B * ((a.Ho + 7) / 8) * ((a.Wo + 31) / 32) * ((a.Cout + 32 * NT - 1) / (32 * NT))), block(512);
    constexpr int LDS2 = ConvPatchLds<2, NT>::TOTAL, LDS1 = ConvPatchLds<1, NT>::TOTAL;
    static_assert(LDS2 <= 160 * 1024 && LDS1 <= 160 * 1024, "ring beyond the CU's LDS");
    ScopedKernelTimer timer(UM_K_CONV, stream);
    um_census_hit(UM_V_CONV_PATCH);
    if (mode == 0) {
        if (!configured[0]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<Fp16, 2, NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
            if (e != hipSuccess) return e;
            configured[0] = true;
        }
        hipLaunchKernelGGL((conv_patch_kernel<Fp16, 2, NT>), grid, block, LDS2, stream, a);
    } else {
        if (!configured[1]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<Bf16, 1, NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
            if (e != hipSuccess) return e;
            configured[1] = true;
        }
        hipLaunchKernelGGL((conv_patch_kernel<Bf16, 1, NT>), grid, block, LDS1, stream, a);
    }
    return hipGetLastError();
}

// ---- which kernel serves a geometry (one place: um_conv_stats_parts() must agree with the launch)
enum ConvKind { CONV_GENERIC = 0, CONV_ROWS = 1, CONV_PATCH = 2 };

static ConvKind conv_pick(int hi, int wi, int ho, int wo, int cout, int kh, int kw, int stride, int pad_h, int pad_w, int* nt_out) {
    // widest output tile that does not waste more than a third of its columns
    const int nt = (cout % 128 == 0 || cout > 192) ? 4 : (cout % 96 == 0) ? 3 : (cout <= 64 || cout % 64 == 0) ? 2 : 4;
    *nt_out = nt;
    // A/B switches (tools/ab_bench.py), read once: UM_CONV_NO_ROWS = generic kernel only; UM_CONV_PATCH = the tile widths
    // (digits of NT) the 2-D patch kernel may serve
    static const bool rows_enabled = um_debug_env("UM_CONV_NO_ROWS") == nullptr;
    static const char* patch_env = um_debug_env("UM_CONV_PATCH");
    static const char* patch_nts = patch_env ? patch_env : "234";
    // same-size stride-1 rows of 3 taps (any tile width) or 5 taps (128-wide tiles: the GRU's 1x5 gates): row-window kernel
    const bool same = stride == 1 && ho == hi && wo == wi && (long)ho * wo >= 256 && rows_enabled;
    const bool rows3 = same && kw == 3 && pad_w == 1, rows5 = same && kw == 5 && pad_w == 2 && nt == 4;
    // 3x3: the 2-D patch kernel when at least 3/4 of the 8 x 32 tiles' area is image
    const long tiled = (long)((ho + 7) / 8 * 8) * ((wo + 31) / 32 * 32);
    if (rows3 && kh == 3 && pad_h == 1 && 4L * ho * wo >= 3 * tiled && strchr(patch_nts, '0' + nt) != nullptr) {
        // heads with a handful of outputs (the flow head's second convolution: 256 -> 2): a 32-wide tile, half the products of a 64-wide one
        if (cout <= 32) *nt_out = 1;
        return CONV_PATCH;
    }
    return rows3 || rows5 ? CONV_ROWS : CONV_GENERIC;
}

extern "C" int um_conv_stats_parts(int hi, int wi, int cout, int kh, int kw, int stride, int pad_h, int pad_w) {
    if (hi <= 0 || wi <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad_h < 0 || pad_w < 0) return -1;
    const int ho = (hi + 2 * pad_h - kh) / stride + 1, wo = (wi + 2 * pad_w - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return -1;
    int nt;
    if (conv_pick(hi, wi, ho, wo, cout, kh, kw, stride, pad_h, pad_w, &nt) == CONV_PATCH) return 2 * ((ho + 7) / 8) * ((wo + 31) / 32);
    return (int)(((long)ho * wo + 127) / 128);
}

static int conv2d_impl(const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes, const float* bias,
                       float* out, int out_ld, int out_coff, void* out_planes, int outp_ld, int outp_coff, long outp_rows,
                       float* stats_out, int batch, int hi, int wi, int cin, int cout, int kh, int kw, int stride,
                       int pad_h, int pad_w, int act, int wshift, int mode, void* stream_, int gate, float* gate_h,
                       const float* gate_z, int gate_zld, const float* addend = nullptr, int addend_ld = 0, float* gate_hout = nullptr,
                       int gate_hout_ld = 0) {
    if (gate_hout && (gate != 2 || gate_hout_ld < cout || gate_hout_ld % 4 != 0 || ((unsigned long)gate_hout & 15) != 0)) {
        um_set_error("um_conv2d: a separate new-state tensor goes with gate 2 only: 16-byte aligned fp32 [M][ld >= channels, ld %% 4 == 0]");
        return -1;
    }
    if (addend && (addend_ld < cout || addend_ld % 4 != 0 || ((unsigned long)addend & 15) != 0 || stats_out)) {
        um_set_error("um_conv2d: the addend must be 16-byte aligned fp32 [M][ld >= cout, ld %% 4 == 0] (and excludes fused statistics)");
        return -1;
    }
    if (!a_planes || !w_planes || (!out && !out_planes) || batch <= 0 || hi <= 0 || wi <= 0 || cin <= 0 || cin % 32 != 0 ||
        cout <= 0 || cout % 4 != 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad_h < 0 || pad_w < 0 || (mode != 0 && mode != 1) ||
        wshift < 0 || wshift > 14 || act < 0 || act > 3) {
        um_set_error("um_conv2d: bad argument (batch=%d hi=%d wi=%d cin=%d cout=%d k=%dx%d stride=%d pad=%d,%d act=%d; cin must "
                     "be a multiple of 32, cout of 4)", batch, hi, wi, cin, cout, kh, kw, stride, pad_h, pad_w, act);
        return -1;
    }
    const int ho = (hi + 2 * pad_h - kh) / stride + 1, wo = (wi + 2 * pad_w - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) {
        um_set_error("um_conv2d: empty output (%d x %d)", ho, wo);
        return -1;
    }
    const long rows_in = (long)batch * hi * wi, m = (long)batch * ho * wo;
    if (a_ld < a_coff + cin || a_ld % 8 != 0 || a_coff % 8 != 0 || a_rows < rows_in + 1 ||
        (out && (out_ld < out_coff + cout || out_ld % 4 != 0 || out_coff % 4 != 0)) ||
        (out_planes && (outp_ld < outp_coff + (gate == 1 ? cout / 2 : cout) || outp_ld % 4 != 0 || outp_coff % 4 != 0 ||
                        outp_rows < m))) {
        um_set_error("um_conv2d: inconsistent leading dimensions / offsets / row counts");
        return -1;
    }
    if (stats_out && !out) {
        um_set_error("um_conv2d: fused statistics need the fp32 output");
        return -1;
    }
    if (bias && ((unsigned long)bias & 15) != 0) {
        um_set_error("um_conv2d: bias must be 16-byte aligned");
        return -1;
    }
    if (a_rows * a_ld * 2 >= (1L << 32) || (long)cout * kh * kw * cin * 2 >= (1L << 32) || m >= (1L << 31)) {
        um_set_error("um_conv2d: operand planes beyond 4 GiB are not addressable by this kernel");
        return -4;
    }
    ConvArgs a;
    a.ap = (const unsigned short*)a_planes + a_coff;
    a.a_plane_stride = a_rows * a_ld;
    a.row_stride = (unsigned)a_ld * 2;
    a.zero_row = (unsigned)(a_rows - 1);
    a.wp = (const unsigned short*)w_planes;
    a.w_plane_stride = (long)cout * kh * kw * cin;
    a.bias = bias;
    a.out = out;
    a.out_ld = out_ld;
    a.out_coff = out_coff;
    a.outp = (unsigned short*)out_planes;
    a.outp_ld = outp_ld;
    a.outp_coff = outp_coff;
    a.outp_plane_stride = outp_rows * outp_ld;
    a.stats = stats_out;
    a.addend = addend;
    a.addend_ld = addend_ld;
    a.B = batch;
    a.Hi = hi;
    a.Wi = wi;
    a.Cin = cin;
    a.Ho = ho;
    a.Wo = wo;
    a.Cout = cout;
    a.KH = kh;
    a.KW = kw;
    a.stride = stride;
    a.pad_h = pad_h;
    a.pad_w = pad_w;
    a.M = (int)m;
    a.act = act;
    a.gate = gate;
    a.gate_c = gate == 1 ? cout / 2 : cout;
    a.gate_h = gate_h;
    a.gate_z = gate_z;
    a.gate_zld = gate_zld;
    a.gate_hout = gate_hout ? gate_hout : gate_h;
    a.gate_hout_ld = gate_hout ? gate_hout_ld : a.gate_c;
    a.out_scale = ldexpf(1.f, -wshift);
    a.xcd = conv_xcd_enabled();
    hipError_t e;
    int nt;
    const ConvKind kind = conv_pick(hi, wi, ho, wo, cout, kh, kw, stride, pad_h, pad_w, &nt);
    hipStream_t st = (hipStream_t)stream_;
    if (kind == CONV_PATCH)
        e = nt == 1 ? launch_conv_patch<1>(a, mode, st) : nt == 2 ? launch_conv_patch<2>(a, mode, st) : nt == 3 ? launch_conv_patch<3>(a, mode, st)
                                                                                                          : launch_conv_patch<4>(a, mode, st);
    else if (kind == CONV_ROWS && kw == 5) e = launch_conv_rows<4, 5, 2>(a, mode, st);
    else if (kind == CONV_ROWS) e = nt == 2 ? launch_conv_rows<2, 3, 2>(a, mode, st) : nt == 3 ? launch_conv_rows<3, 3, 2>(a, mode, st) : launch_conv_rows<4, 3, 2>(a, mode, st);
    else e = nt == 2 ? launch_conv<2>(a, mode, st) : nt == 3 ? launch_conv<3>(a, mode, st) : launch_conv<4>(a, mode, st);
    if (e != hipSuccess) {
        um_set_error("um_conv2d: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int um_conv2d_ex(const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes, const float* bias,
                            float* out, int out_ld, int out_coff, void* out_planes, int outp_ld, int outp_coff, long outp_rows,
                            float* stats_out, int batch, int hi, int wi, int cin, int cout, int kh, int kw, int stride,
                            int pad_h, int pad_w, int act, int wshift, int mode, void* stream_) {
    return conv2d_impl(a_planes, a_ld, a_coff, a_rows, w_planes, bias, out, out_ld, out_coff, out_planes, outp_ld, outp_coff,
                       outp_rows, stats_out, batch, hi, wi, cin, cout, kh, kw, stride, pad_h, pad_w, act, wshift, mode, stream_, 0,
                       nullptr, nullptr, 0);
}

extern "C" int um_conv2d_gru_fwd(int gate, const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes,
                                 const float* bias, float* hidden, const float* z, int z_ld, float* z_out, int z_out_ld,
                                 void* out_planes, int outp_ld, int outp_coff, long outp_rows, int batch, int hi, int wi, int cin,
                                 int channels, int kh, int kw, int pad_h, int pad_w, int wshift, int mode, void* stream_) {
    if ((gate != 1 && gate != 2) || !hidden || !out_planes || channels <= 0 || channels % 4 != 0 || (gate == 1 && !z_out) ||
        (gate == 2 && !z)) {
        um_set_error("um_conv2d_gru_fwd: bad argument (gate=%d channels=%d)", gate, channels);
        return -1;
    }
    if (gate == 1)       // z | r: sigmoid; z -> z_out[.][z_out_ld], r * hidden -> planes
        return conv2d_impl(a_planes, a_ld, a_coff, a_rows, w_planes, bias, z_out, z_out_ld, 0, out_planes, outp_ld, outp_coff,
                           outp_rows, nullptr, batch, hi, wi, cin, 2 * channels, kh, kw, 1, pad_h, pad_w, 2, wshift, mode, stream_,
                           1, hidden, nullptr, 0);
    // q: tanh; hidden <- (1 - z) hidden + z q, also written as planes
    return conv2d_impl(a_planes, a_ld, a_coff, a_rows, w_planes, bias, nullptr, 0, 0, out_planes, outp_ld, outp_coff, outp_rows,
                       nullptr, batch, hi, wi, cin, channels, kh, kw, 1, pad_h, pad_w, 3, wshift, mode, stream_, 2, hidden, z, z_ld);
}

// um_conv2d_gru_fwd with a per-pixel addend (fp32 [rows][addend_ld], column = output channel of the convolution: 2 * channels for
// gate 1, channels for gate 2) added before the gate's activation: the contribution of the input channels that are the same in every
// refinement iteration, computed once per scale with um_conv2d_ex (unimatch_amd/refine_nhwc.py).
extern "C" int um_conv2d_gru_add_fwd(int gate, const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes,
                                     const float* bias, const float* addend, int addend_ld, float* hidden, const float* z, int z_ld,
                                     float* z_out, int z_out_ld, void* out_planes, int outp_ld, int outp_coff, long outp_rows,
                                     int batch, int hi, int wi, int cin, int channels, int kh, int kw, int pad_h, int pad_w,
                                     int wshift, int mode, void* stream_) {
    if ((gate != 1 && gate != 2) || !hidden || !out_planes || !addend || channels <= 0 || channels % 4 != 0 || (gate == 1 && !z_out) ||
        (gate == 2 && !z)) {
        um_set_error("um_conv2d_gru_add_fwd: bad argument (gate=%d channels=%d)", gate, channels);
        return -1;
    }
    if (gate == 1)
        return conv2d_impl(a_planes, a_ld, a_coff, a_rows, w_planes, bias, z_out, z_out_ld, 0, out_planes, outp_ld, outp_coff,
                           outp_rows, nullptr, batch, hi, wi, cin, 2 * channels, kh, kw, 1, pad_h, pad_w, 2, wshift, mode, stream_,
                           1, hidden, nullptr, 0, addend, addend_ld);
    // gate 2: z_out, when given, receives the new hidden state and `hidden` is only read (the refinement loop restarts from the same
    // net0 in every iteration, unimatch.py:322-331: no copy of it per iteration)
    return conv2d_impl(a_planes, a_ld, a_coff, a_rows, w_planes, bias, nullptr, 0, 0, out_planes, outp_ld, outp_coff, outp_rows,
                       nullptr, batch, hi, wi, cin, channels, kh, kw, 1, pad_h, pad_w, 3, wshift, mode, stream_, 2, hidden, z, z_ld,
                       addend, addend_ld, z_out, z_out_ld);
}

extern "C" int um_conv2d_fwd(const void* a_planes, const void* w_planes, const float* bias, float* out, float* stats_out,
                             int batch, int hi, int wi, int cin, int cout, int kh, int kw, int stride, int pad_h, int pad_w,
                             int relu, int wshift, int mode, void* stream_) {
    const long rows_in = (long)batch * hi * wi;
    return um_conv2d_ex(a_planes, cin, 0, rows_in + 1, w_planes, bias, out, cout, 0, nullptr, 0, 0, 0, stats_out, batch, hi, wi,
                        cin, cout, kh, kw, stride, pad_h, pad_w, relu ? 1 : 0, wshift, mode, stream_);
}

// ---- 7x7 convolutions with very few input channels on the same kernel ----------------------------------------------------
// (the encoder stem, unimatch/backbone.py:49: 3 -> 64, stride 2; the motion encoder's flow branch, reg_refine.py:13: 1|2 -> 128,
// stride 1).  The image is packed once into zero-bordered NHWC planes with CPP channels per pixel (4 for stride 2, 8 for
// stride 1; missing channels = 0), [NS][B][H + 6][Wp][CPP].  For output pixel (y, x) and kernel row ky the 7 taps x C
// channels are then 8 CPP CONTIGUOUS elements starting at packed pixel (s y + ky, s x): pixels s x .. s x + 7, where the 8th
// pixel and the missing channels meet zero weights.  So the layer is a "7 x 1 convolution with 8 CPP input channels" whose
// rows advance by one packed pixel: no im2col, no border logic, 16-byte aligned LDS-DMA (K = 224 / 448 instead of 49 C).
template <int CPP>
__global__ __launch_bounds__(256) void pack7_kernel(const float* img, unsigned short* planes, long plane_stride, int B, int C,
                                                    int H, int W, int Hp, int Wp, int normalize, float m0, float m1, float m2,
                                                    float s0, float s1, float s2) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * Hp * Wp;
    if (idx >= total) return;
    const int xp = (int)(idx % Wp);
    const long t = idx / Wp;
    const int yp = (int)(t % Hp), b = (int)(t / Hp);
    const int y = yp - 3, x = xp - 3;
    float v[CPP];
#pragma unroll
    for (int c = 0; c < CPP; ++c) v[c] = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
        const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            if (c < C) {
                float p = img[(((long)b * C + c) * H + y) * W + x];
                if (normalize && c < 3) p = (p / 255.0f - mean[c]) / sd[c];    // the reference's operation order (unimatch.py:122-124)
                v[c] = p;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPP; c += 4) {
        const unsigned h0 = Fp16::pack2(v[c], v[c + 1]), h1 = Fp16::pack2(v[c + 2], v[c + 3]);
        *reinterpret_cast<u32x2*>(planes + idx * CPP + c) = u32x2{h0, h1};
        const f32x2 u0 = Fp16::unpack2(h0), u1 = Fp16::unpack2(h1);
        *reinterpret_cast<u32x2*>(planes + plane_stride + idx * CPP + c) =
            u32x2{Fp16::pack2(v[c] - u0[0], v[c + 1] - u0[1]), Fp16::pack2(v[c + 2] - u1[0], v[c + 3] - u1[1])};
    }
}

static int conv7_wp(int w) { return (w + 8 + 1) & ~1; }

extern "C" size_t um_conv7_planes_bytes(int batch, int h, int w, int stride) {
    if (batch <= 0 || h <= 0 || w <= 0 || (stride != 1 && stride != 2)) return 0;
    const long cpp = stride == 2 ? 4 : 8;
    return (size_t)(2 * ((long)batch * (h + 6) * conv7_wp(w) + 8) * cpp * 2);    // two fp16 planes, 8 pixels of slack
}

extern "C" int um_conv7_fwd(const float* image, int channels, int normalize, const float* mean3, const float* std3,
                            void* image_planes, const void* w_planes, const float* bias, float* out, int out_ld, int out_coff,
                            void* out_planes, int outp_ld, int outp_coff, long outp_rows, float* stats_out, int batch, int h,
                            int w, int cout, int stride, int act, int wshift, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int cpp = stride == 2 ? 4 : 8;
    if (!image || !image_planes || !w_planes || (!out && !out_planes) || batch <= 0 || h < 7 || w < 7 || cout <= 0 || cout % 4 != 0 ||
        wshift < 0 || wshift > 14 || (stride != 1 && stride != 2) || channels <= 0 || channels > (stride == 2 ? 3 : 8) ||
        (normalize && (!mean3 || !std3)) || act < 0 || act > 3) {
        um_set_error("um_conv7_fwd: bad argument (batch=%d channels=%d h=%d w=%d cout=%d stride=%d)", batch, channels, h, w, cout, stride);
        return -1;
    }
    const int hp = h + 6, wp = conv7_wp(w);
    const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;       // (h + 2*3 - 7) / stride + 1
    const long rows = (long)batch * hp * wp, m = (long)batch * ho * wo;
    if ((out && (out_ld < out_coff + cout || out_ld % 4 != 0 || out_coff % 4 != 0)) ||
        (out_planes && (outp_ld < outp_coff + cout || outp_ld % 4 != 0 || outp_coff % 4 != 0 || outp_rows < m))) {
        um_set_error("um_conv7_fwd: inconsistent leading dimensions / offsets / row counts");
        return -1;
    }
    if (stats_out && !out) {
        um_set_error("um_conv7_fwd: fused statistics need the fp32 output");
        return -1;
    }
    if (bias && ((unsigned long)bias & 15) != 0) {
        um_set_error("um_conv7_fwd: bias must be 16-byte aligned");
        return -1;
    }
    if ((rows + 8) * cpp * 2 >= (1L << 32) || m >= (1L << 31)) {
        um_set_error("um_conv7_fwd: image batch beyond the 32-bit addressing of this kernel");
        return -4;
    }
    const long plane_stride = (rows + 8) * cpp;
    {
        ScopedKernelTimer timer(UM_K_CONV, stream);
        const float mm[3] = {normalize ? mean3[0] : 0.f, normalize ? mean3[1] : 0.f, normalize ? mean3[2] : 0.f};
        const float ss[3] = {normalize ? std3[0] : 1.f, normalize ? std3[1] : 1.f, normalize ? std3[2] : 1.f};
        const dim3 grid((unsigned)((rows + 255) / 256));
        if (cpp == 4)
            hipLaunchKernelGGL((pack7_kernel<4>), grid, dim3(256), 0, stream, image, (unsigned short*)image_planes, plane_stride,
                               batch, channels, h, w, hp, wp, normalize, mm[0], mm[1], mm[2], ss[0], ss[1], ss[2]);
        else
            hipLaunchKernelGGL((pack7_kernel<8>), grid, dim3(256), 0, stream, image, (unsigned short*)image_planes, plane_stride,
                               batch, channels, h, w, hp, wp, normalize, mm[0], mm[1], mm[2], ss[0], ss[1], ss[2]);
    }
    ConvArgs a;
    a.ap = (const unsigned short*)image_planes;
    a.a_plane_stride = plane_stride;
    a.row_stride = (unsigned)cpp * 2;                             // one packed pixel
    a.zero_row = 0;                                               // never used: every tap is inside the padded image
    a.wp = (const unsigned short*)w_planes;
    a.w_plane_stride = (long)cout * 7 * 8 * cpp;
    a.bias = bias;
    a.out = out;
    a.out_ld = out_ld;
    a.out_coff = out_coff;
    a.outp = (unsigned short*)out_planes;
    a.outp_ld = outp_ld;
    a.outp_coff = outp_coff;
    a.outp_plane_stride = outp_rows * outp_ld;
    a.stats = stats_out;
    a.addend = nullptr;
    a.addend_ld = 0;
    a.B = batch;
    a.Hi = hp;
    a.Wi = wp;
    a.Cin = 8 * cpp;
    a.Ho = ho;
    a.Wo = wo;
    a.Cout = cout;
    a.KH = 7;
    a.KW = 1;
    a.stride = stride;
    a.pad_h = 0;
    a.pad_w = 0;
    a.M = (int)m;
    a.act = act;
    a.gate = 0;
    a.gate_c = a.gate_zld = 0;
    a.gate_h = nullptr;
    a.gate_z = nullptr;
    a.out_scale = ldexpf(1.f, -wshift);
    a.xcd = conv_xcd_enabled();
    hipError_t e;
    if (cout % 128 == 0 || cout > 192) e = launch_conv<4>(a, 0, stream);
    else if (cout % 96 == 0) e = launch_conv<3>(a, 0, stream);
    else e = launch_conv<2>(a, 0, stream);
    if (e != hipSuccess) {
        um_set_error("um_conv7_fwd: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" size_t um_stem_planes_bytes(int batch, int h, int w) { return um_conv7_planes_bytes(batch, h, w, 2); }

extern "C" int um_stem_conv_fwd(const float* image, int normalize, const float* mean3, const float* std3, void* image_planes,
                                const void* w_planes, float* out, float* stats_out, int batch, int h, int w, int cout,
                                int wshift, void* stream_) {
    return um_conv7_fwd(image, 3, normalize, mean3, std3, image_planes, w_planes, nullptr, out, cout, 0, nullptr, 0, 0, 0, stats_out,
                        batch, h, w, cout, 2, 0, wshift, stream_);
}
