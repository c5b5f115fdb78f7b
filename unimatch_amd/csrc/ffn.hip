// Fused Transformer FFN:  out = x + LayerNorm( W2 . gelu( W1 . [x | y] ) )                gfx950 / wave64 / MFMA
//
// replaces `source + norm2(mlp(cat([source, message])))` (unimatch/transformer.py:141-144; mlp = Linear(2C, 8C,
// bias=False), GELU, Linear(8C, C, bias=False), :44-50).  The two-kernel version (linear.hip) writes the
// [M, 8C] hidden activations to HBM as operand planes and reads them back: 2 x 403 MB per call at config 2 -- the
// largest single HBM round trip of the model.  Here the hidden activations only ever
// exist as MFMA accumulators and operand fragments of one wave.
//
// Decomposition ("flash" over the hidden dimension, same swapped products as the attention kernel):
//   workgroup = 8 waves = 128 tokens; wave pair p (waves p and p + 4) owns tokens 32p .. 32p+31, lane = token.
//   The hidden dimension is walked in slices of 32 units.  For slice j:
//     phase A   S^T[32 hid][32 tok] = W1_j[32][256] . X^T          -- the pair splits K: role 0 multiplies the
//               `x` half (k 0..127), role 1 the `y` half (k 128..255); this is what makes the token operand fit
//               in registers (64 VGPRs per wave instead of 128) and it also makes the concatenation free;
//     exchange  each wave keeps accumulator rows 0..15 and hands rows 16..31 to its partner through LDS (role 1
//               reads W1 rows permuted by ^16, so that "rows 0..15" are hidden units 16..31 for it);
//     GELU      on the wave's own 16 hidden units x 32 tokens, split into fp16 hi + lo, turned into B-operand
//               fragments with v_permlane32_swap (no LDS);
//     phase B   O^T[128 out][32 tok] += W2[:, own 16 hid] . H^T    -- the pair splits K again, each wave carries
//               a partial O of all 128 outputs.
//   Epilogue: role 1 hands its partial O to role 0 through LDS; role 0 applies LayerNorm + residual and stores.
//
// Weight slices stream through LDS by LDS-DMA (W1 two slices ahead, W2 one ahead), one workgroup barrier per
// slice, which also orders the accumulator exchange.  Arithmetic: fp16 hi + lo split operands, three products,
// fp32 accumulation (exact mode); bf16, one product (fast mode).  Weights are pre-scaled by 2^wshift.
#include <type_traits>
#include <stdlib.h>
// -DUM_FFN_H1=1 (diagnostic builds: profiles/r03_precision_budget.txt): the second GEMM with gelu(H) in ONE fp16 plane
// (W2_lo.H + W2_hi.H: 2 products instead of 3, no lo split of the hidden activations).
#ifndef UM_FFN_H1
#define UM_FFN_H1 0
#endif
#include "common.h"
#include "planes.h"

// -DUM_FFN_TRACE (diagnostic builds; tools/trace_ffn.py): lane 0 of waves 0 and 4 (the two roles of pair 0) of every 37th workgroup
// stamps s_memtime at the section boundaries of its first 24 slices into the buffer given to um_debug_set_ffn_trace().
#ifdef UM_FFN_TRACE
__device__ unsigned long long* g_um_ffn_trace = nullptr;
#define UM_FSTAMP(slot) do { if (tracing && i < 24) trace_buf[(i) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UM_FSTAMP(slot) do { } while (0)
#endif

struct Kv4Args {
    const float* x;              // stand-alone kernel only: [M][128] fp32 tokens
    const unsigned short* w;     // planes [NS][256][256] of Wc, pre-scaled by 2^wshift
    long w_plane_stride;
    unsigned short* out;         // planes [NS][4][M][128]
    long out_plane_stride;       // 4 * M * 128
    int M;
    float out_scale;             // 2^-wshift
    unsigned* range_flag;        // um_range_flags: sticky operand-range word (device address; nullptr: none)
};

struct FfnArgs {
    const float* x;               // [M, 128] source; also the residual
    const float* y;               // [M, 128] message
    const unsigned short* w1;     // [NS][hid][256], pre-scaled by 2^wshift
    long w1_plane_stride;
    const unsigned short* w2;     // [NS][128][hid], pre-scaled by 2^wshift
    long w2_plane_stride;
    const float* gamma;
    const float* beta;
    float* out;                   // [M, 128]
    int M, hid;
    float out_scale;              // 2^-wshift
    float eps;
    // HSPLIT variant (small launches): `split` workgroups per 128-token tile, each on 1 / split of the hidden slices
    int split;
    float* hs_part;               // [tiles][split][16][256][4] fp32: every part's partial O^T (already scaled)
    unsigned* hs_flag;            // [tiles] arrival counters, zero between launches
    // KV4 variant: the NEXT block's key / value projections of the tile this workgroup has just finished (kv4_project)
    Kv4Args kv;
    unsigned* range_flag;         // um_range_flags (as Kv4Args)
};

// the same with the LDS destination as an ADDRESS (an SGPR value): a flat pointer into the LDS array costs an address-space cast
// with its null check (s_add_u32 / s_addc_u32 / s_cmp_lg_u64 / s_cselect_b32) per statement; base address + constant costs one s_add
__device__ __forceinline__ void ffn_dma16_at(const void* base, unsigned byte_off, unsigned lds_addr) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);         // (uniform already; folds away when the compiler can prove it)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void ffn_dma16(const void* base, unsigned byte_off, const unsigned char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(dst)
                 : "memory");
}

// Main-loop order (round 4).  1: phase A of slice i+1 FIRST, then phase B of slice i-1; the LDS-DMA statements of the next weight
// slices ride in the gaps of the phase-A MFMAs instead of standing between the barrier and the first MFMA, the exchange read is
// issued right behind the barrier, and the accumulator hand-over (add, send) happens in the gaps of phase B -- so the stretch of a
// slice in which BOTH waves of a SIMD (they run in lockstep: one workgroup barrier per slice) leave the matrix pipe idle shrinks
// from [tail + barrier + DMA issue + exchange + first fragments] to [barrier + first fragments].  0: the round-1..3 order (B then
// A, DMA right after the barrier, hand-over after the stream).  Same arithmetic in the same order per accumulator: bitwise equal.
#ifndef UM_FFN_ORDER
#define UM_FFN_ORDER 1
#endif
// Issue priority of the pair's two waves inside the MFMA stream (UM_FFN_ORDER 1).  The partners w and w + 4 share a SIMD and run
// in lockstep (one workgroup barrier per slice); with equal priority the arbiter serves the OLDER wave (role 0) first, its stream
// ends ~750 cycles before the partner's (section stamps: 2210 against 2980 cycles per slice), it idles at the barrier and the
// partner finishes alone with the matrix pipe half used.  1: role 0 favoured in the first half of the stream, role 1 in the second;
// 2 / 4 / 5: the favoured role alternates every 6 / 3 / 12 MFMAs; 3: role 1 favoured throughout (control); 0: equal priority
// (rounds 1-3).
#ifndef UM_FFN_PRIO
#define UM_FFN_PRIO 0
#endif
// Who requests the weight slices (UM_FFN_ORDER 1).  0: every wave its sixth of the slice (6 LDS-DMA statements per wave and slice).
// 1: the ROLE-0 waves request everything (12 statements each), the role-1 waves none: role 0 is the older wave of its SIMD, the
// arbiter serves it first and it finishes its stream ~750 cycles before its partner (section stamps) -- the requests cost issue
// slots, not matrix-pipe time, so moving them to the wave that has the slack shortens the partner's stream, which is the one the
// slice waits for.
#ifndef UM_FFN_DMA_ASYM
#define UM_FFN_DMA_ASYM 0
#endif
// Diagnostic builds: -DUM_FFN_NT=1 non-temporal loads of the token rows (x | y, residual), 2: ... and non-temporal stores of the output --
// whether keeping the streamed rows out of the XCD's L2 saves the weight slices' re-fetches (profiles/r05_ffn_overfetch.txt)
#ifndef UM_FFN_NT
#define UM_FFN_NT 0
#endif
#if UM_FFN_NT >= 1
#define UM_FFN_LD(p) __builtin_nontemporal_load(p)
#else
#define UM_FFN_LD(p) (*(p))
#endif
#if UM_FFN_NT >= 2
#define UM_FFN_ST(p, v) __builtin_nontemporal_store(v, p)
#else
#define UM_FFN_ST(p, v) (*(p) = (v))
#endif
#ifndef UM_FFN_ROWSTORE
#define UM_FFN_ROWSTORE 1
#endif
#ifndef UM_FFN_ABL
#define UM_FFN_ABL 0     // timing ablations (results are wrong when non-zero): 1 gelu, 2 dma, 4 phase A, 8 phase B, 16 exchange, 32 residual read
#endif
// erf-GELU on two values at once, branch free and in ONE piece (ocml's erff is two divergent branches per element, ~40 VALU
// slots and a scheduling wall between every element; the two-piece minimax erff this kernel used in round 1 evaluated both
// pieces, 23 slots + v_exp_f32).  With u = |x|, h = x / 2:
//     gelu(x) = h (1 + erf(x / sqrt 2)) = (h + |h|) - |h| erfc(u / sqrt 2),        erfc(u / sqrt 2) = 2^(u g(u))
// g = degree-7 weighted-minimax fit of log2(erfc(u / sqrt 2)) / u on [0, 6.2] (weight = the error it causes in gelu; leading
// coefficient negative, so 2^(u g(u)) -> 0 beyond).  erf's RELATIVE accuracy near 0 -- what the second piece of an erff buys
// -- is not needed: gelu only sees 1 + erf.  h + |h| is exact and the final FMA rounds once, so for x < 0 the result carries
// only v_exp_f32's relative error.  12 VALU slots per value; max error 1.03 ulp(x) against fp64 over [-12, 12] and N(0, 2)
// samples (the two-piece form: 1.02), |error| <= 7.4e-8 |x|   (fit + evaluation: tools/gelu_fit.py).
struct Gelu2 {
    float x[2], r[2];            // after the last stage: x holds gelu(x)
};
#define UM_GELU_STAGES 4
// One of four stages of 3 instructions per value (the main loop pins stages behind MFMAs).  Scalar fp32 on purpose:
// packed fp32 VALU issues slower beside MFMAs than the two scalar instructions it replaces (MI355X_MICROARCH.md).
__device__ __forceinline__ void ffn_gelu_stage(Gelu2& g, int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (UM_FFN_ABL & 1) {
            if (stage == UM_GELU_STAGES - 1) g.x[i] = g.x[i] * 0.5f;
            continue;
        }
        const float u = __builtin_fabsf(g.x[i]);            // a source modifier, not an instruction
        switch (stage) {
        case 0:
            g.r[i] = __builtin_fmaf(-2.572117637100746e-06f, u, 3.6567180359270424e-05f);
            g.r[i] = __builtin_fmaf(g.r[i], u, -1.7448408470954746e-04f);
            g.r[i] = __builtin_fmaf(g.r[i], u, -1.6111040895339102e-04f);
            break;
        case 1:
            g.r[i] = __builtin_fmaf(g.r[i], u, 7.089670747518539e-03f);
            g.r[i] = __builtin_fmaf(g.r[i], u, -5.2510716021060944e-02f);
            g.r[i] = __builtin_fmaf(g.r[i], u, -4.592045545578003e-01f);
            break;
        case 2:
            g.r[i] = __builtin_fmaf(g.r[i], u, -1.151105523109436f);
            g.r[i] = g.r[i] * u;
            g.r[i] = fast_exp2(g.r[i]);
            break;
        default: {
            const float hx = g.x[i] * 0.5f;
            const float sum = hx + __builtin_fabsf(hx);
            g.x[i] = __builtin_fmaf(-__builtin_fabsf(hx), g.r[i], sum);
            break;
        }
        }
    }
}

template <int NS>
struct FfnLds {
    static constexpr int W1P = 32 * 512;         // one plane of a W1 slice: 32 hidden rows x 256 k x 2 B
    static constexpr int W1S = NS * W1P;
    static constexpr int W2P = 128 * 64;         // one plane of a W2 slice: 128 output rows x 32 hidden x 2 B
    static constexpr int W2S = NS * W2P;
    static constexpr int W1_OFF = 0, W2_OFF = 2 * W1S, XB_OFF = W2_OFF + 2 * W2S;
    static constexpr int XB = 2048;              // one wave's outgoing accumulator half
    static constexpr int RING = XB_OFF + 2 * 8 * XB;
    static constexpr int TOTAL = RING > 65536 ? RING : 65536;    // the epilogue overlays 4 x 16 KB of partial O
};

// =================================================================================================================
// k | v projections of BOTH layers of a Transformer block in one pass (unimatch/transformer.py:58-60, four bias-free
// 128 x 128 Linears on the token stream as it enters the block):  out[j] = X . W_j^T,  j = k_self, v_self, k_cross, v_cross,
// written as the attention kernel's operand planes in BLOCKED form [NS][4][M][128] (every projection its own [M][128] tensor:
// rows of one projection are contiguous, which the attention kernel's LDS-DMA staging streams ~2 % faster than 256-byte
// pieces of 1 KB rows).
//
// Same machinery as the FFN's first GEMM ("phase A"): workgroup = 8 waves = 128 tokens, wave (pair, role) owns the 32 tokens of
// its pair; role 0 produces k_self | v_self, role 1 k_cross | v_cross.  The token operand Y^T (all 128 features, hi | lo) stays in
// registers as B fragments for the whole pass; the 512 weight rows stream through the W1 ring in 8 chunks of (32 rows of role 0's
// half | 32 rows of role 1's half) laid out exactly like a W1 slice -- so the host packs the four weights as one [256][256]
// matrix  Wc[32 c + r][0:128] = W4[32 c + r],  Wc[32 c + r][128:256] = W4[256 + 32 c + (r ^ 16)]  (role 1 reads ring rows
// permuted by ^16, see phase A) -- one barrier per chunk, LDS-DMA one chunk ahead.  Each 32 x 32 result tile is converted to
// fp16 hi | lo, transposed through a wave-private LDS scratch (double-buffered: the global stores of chunk c leave while chunk
// c + 1 multiplies) and stored as 64-byte row pieces.
//
// kv4_project() is the device function; kv4_kernel is it as a stand-alone launch on fp32 tokens (block 0, and every block while
// the FFN runs its small-launch split variant); ffn_kernel<.., KV4> calls it from its epilogue on the tile it has just
// normalised, so that the next block's keys / values leave the FFN launch directly (finished in round 4).

template <int NS>
struct Kv4Lds {
    static constexpr int RING = 2 * FfnLds<NS>::W1S;             // the W1 ring of the FFN: [0, 64 KB)
    static constexpr int SCRP = 2048;                            // one plane of one 32 x 32 fp16 tile: 32 rows x 64 B
    static constexpr int SCRW = 2 * NS * SCRP;                   // per wave: two parities x NS planes
    static constexpr int SCR_BYTES = 8 * SCRW;                   // 64 KB at NS = 2
};

// yf: this wave's token operand, B fragments (k = 16 ks + 8 half + 0..7) of planes hi (| lo).  `ring` and `scr` are LDS regions no
// other code touches during the call (ring: Kv4Lds::RING bytes, scr: SCR_BYTES); every wave of the 512-thread workgroup calls it.
// `first_landed`: chunk 0 was already requested by the caller (dma of chunk 0 into slot 0) -- the fused epilogue issues it early.
template <typename T, int NS, bool CHUNK0_ISSUED>
__device__ __forceinline__ void kv4_project(unsigned char* ring, unsigned char* scr, const Kv4Args& k, const i16x8 (&yf)[NS][8],
                                            int m0, int wave, int lane, const int (&aoff)[8], float neg1) {
    using L = FfnLds<NS>;
    const int pair = wave & 3, role = wave >> 2;
    const int half = lane >> 5, tl = lane & 31;
    auto dma = [&](int c, int slot) {                            // as dma_w1: one instruction moves two ring rows
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = 8 * i + wave;
            const int r = 2 * blk + half, cc = tl ^ r;
            const unsigned off = (unsigned)((((long)(32 * c + r)) * 256 + 8 * cc) * 2);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                ffn_dma16(k.w + pl * k.w_plane_stride, off, ring + slot * L::W1S + pl * L::W1P + blk * 1024);
        }
    };
    unsigned char* myscr = scr + wave * Kv4Lds<NS>::SCRW;
    const int sw = (tl >> 2) & 3;                                // scratch swizzle of this lane's token row
    // global store of one finished chunk from the scratch: lane = (row 16 it + lane / 4, 16-byte piece lane % 4)
    auto store_chunk = [&](int c) {
        const unsigned char* sp = myscr + (c & 1) * (NS * Kv4Lds<NS>::SCRP);
        const int j = 2 * role + (c >> 2);                       // which projection
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = 16 * it + (lane >> 2), q = lane & 3;
                const u32x4 d = *reinterpret_cast<const u32x4*>(sp + pl * Kv4Lds<NS>::SCRP + row * 64 + ((q ^ ((row >> 2) & 3)) << 4));
                const long tok = (long)m0 + 32 * pair + row;
                if (tok < k.M)
                    *reinterpret_cast<u32x4*>(k.out + pl * k.out_plane_stride + ((long)j * k.M + tok) * 128 + 32 * (c & 3) + 8 * q) = d;
            }
    };
    if (!CHUNK0_ISSUED) dma(0, 0);
    float kmx = 0.f;                                               // largest projected key / value element (um_range_flags)
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
        const int slot = c & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this thread's pieces of chunk c (requested one chunk ago) + old stores
        __syncthreads();                                           // chunk c visible; everybody is done with chunk c - 1's slot
        if (c + 1 < 8) dma(c + 1, slot ^ 1);
        if (c > 0) store_chunk(c - 1);
        const unsigned char* w1s = ring + slot * L::W1S;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            i16x8 fh[3], fl[3];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fh[ks] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks]);
                if (NS == 2) fl[ks] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks]);
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 2 < 8) {
                    fh[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks + 2]);
                    if (NS == 2) fl[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks + 2]);
                }
                if (NS == 2) {
                    acc = T::mfma(fl[ks % 3], yf[0][ks], acc);
                    acc = T::mfma(fh[ks % 3], yf[NS - 1][ks], acc);
                }
                acc = T::mfma(fh[ks % 3], yf[0][ks], acc);
            }
            constexpr int RD = NS, MF = (NS == 2) ? 3 : 1;
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 0);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        // accumulator rows (ring rows, = output columns 32 (c & 3) + 8 g + 4 half + i of projection j) -> fp16 hi | lo, token-major
        unsigned char* dp = myscr + slot * (NS * Kv4Lds<NS>::SCRP) + tl * 64 + 8 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v0 = acc[4 * g] * k.out_scale, v1 = acc[4 * g + 1] * k.out_scale, v2 = acc[4 * g + 2] * k.out_scale,
                        v3 = acc[4 * g + 3] * k.out_scale;
            kmx = fmaxf(fmaxf(kmx, fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1))), fmaxf(__builtin_fabsf(v2), __builtin_fabsf(v3)));
            const unsigned h0 = T::pack2(v0, v1), h1 = T::pack2(v2, v3);
            *reinterpret_cast<u32x2*>(dp + ((g ^ sw) << 4)) = u32x2{h0, h1};
            if (NS == 2) {
                const unsigned l0 = T::lo2(v0, v1, h0, neg1), l1 = T::lo2(v2, v3, h1, neg1);
                *reinterpret_cast<u32x2*>(dp + Kv4Lds<NS>::SCRP + ((g ^ sw) << 4)) = u32x2{l0, l1};
            }
        }
    }
    store_chunk(7);
    um_range_note<T>(k.range_flag, kmx, UM_RANGE_KV_TOKENS);
}

template <typename T, int NS>
__global__ __launch_bounds__(512, 2) void kv4_kernel(Kv4Args k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const float neg1 = um_opaque_neg1();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, role = wave >> 2;
    const int half = lane >> 5, tl = lane & 31;
    const int m0 = (int)blockIdx.x * 128;
    const int tok = m0 + 32 * pair + tl;
    i16x8 yf[NS][8];
    {
        const float* src = k.x + (long)min(tok, k.M - 1) * 128 + 8 * half;
        float xmx = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + 16 * ks);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) xmx = fmaxf(xmx, fmaxf(__builtin_fabsf(v0[j]), __builtin_fabsf(v1[j])));
            const u32x4 h = {T::pack2(v0[0], v0[1]), T::pack2(v0[2], v0[3]), T::pack2(v1[0], v1[1]), T::pack2(v1[2], v1[3])};
            yf[0][ks] = __builtin_bit_cast(i16x8, h);
            if (NS == 2) {
                const u32x4 l = {T::lo2(v0[0], v0[1], h[0], neg1), T::lo2(v0[2], v0[3], h[1], neg1),
                                 T::lo2(v1[0], v1[1], h[2], neg1), T::lo2(v1[2], v1[3], h[3], neg1)};
                yf[NS - 1][ks] = __builtin_bit_cast(i16x8, l);
            }
        }
        um_range_note<T>(k.range_flag, xmx, UM_RANGE_KV_TOKENS);
    }
    const int arow = tl ^ (16 * role);
    int aoff[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) aoff[ks] = arow * 512 + (((16 * role + 2 * ks + half) ^ arow) << 4);
    kv4_project<T, NS, false>(lds, lds + Kv4Lds<NS>::RING, k, yf, m0, wave, lane, aoff, neg1);
}

template <typename T, int NS, bool HSPLIT = false, bool KV4 = false>
__global__ __launch_bounds__(512, 2) void ffn_kernel(FfnArgs a) {
    using L = FfnLds<NS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const float neg1 = um_opaque_neg1();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, role = wave >> 2;                 // partners w, w ^ 4 sit on the same SIMD
    const int half = lane >> 5, tl = lane & 31;
    // HSPLIT: at batch 1 the launch has 35 - 96 workgroups on 256 CUs, each walking all 32 hidden slices.  `split`
    // neighbouring workgroups share a token tile; part p walks the p-th share of the hidden slices (the second GEMM is a sum over
    // hidden units, so the parts' O^T simply add); every part leaves its O^T in a memory slot and takes a ticket, the last
    // arriver adds the slots in part order and runs LayerNorm + residual (nobody waits; see the epilogue).  Slots are touched by
    // agent-scope accesses only (no agent-scope fence: those write back / invalidate the XCD's whole L2).
    const int nsplit = HSPLIT ? a.split : 1;
    const int tile_id = HSPLIT ? (int)blockIdx.x / nsplit : (int)blockIdx.x;
    const int part = HSPLIT ? (int)blockIdx.x - tile_id * nsplit : 0;
    const int m0 = tile_id * 128;
    const int tok = m0 + 32 * pair + tl;
    const int nslice = (a.hid >> 5) / nsplit;                    // slices of THIS workgroup: global slice = sbase + local
    const int sbase = part * nslice;

    // ---- weight slices by LDS-DMA; the XOR swizzles are applied on the source side ---------------------------------
    // W1 slice: 16-byte chunk c (of 32) of row r sits at chunk c ^ r; one instruction moves two rows.
    auto dma_w1 = [&](int j, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = 8 * i + wave;
            const int r = 2 * blk + half, c = tl ^ r;
            const unsigned off = (unsigned)((((long)(32 * (sbase + j) + r)) * 256 + 8 * c) * 2);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                ffn_dma16(a.w1 + pl * a.w1_plane_stride, off, lds + L::W1_OFF + slot * L::W1S + pl * L::W1P + blk * 1024);
        }
    };
    // W2 slice: rows of 64 B (4 chunks), chunk c of row r at c ^ ((r >> 2) & 3); one instruction moves 16 rows.
    auto dma_w2 = [&](int j, int slot) {
        const int r = 16 * wave + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
        const unsigned off = (unsigned)((((long)r) * a.hid + 32 * (sbase + j) + 8 * c) * 2);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
            ffn_dma16(a.w2 + pl * a.w2_plane_stride, off, lds + L::W2_OFF + slot * L::W2S + pl * L::W2P + wave * 1024);
    };

#ifdef UM_FFN_TRACE
    const bool tracing = g_um_ffn_trace != nullptr && (blockIdx.x % 37) == 0 && lane == 0 && pair == 0;
    unsigned long long* trace_buf = g_um_ffn_trace + ((size_t)(blockIdx.x / 37) * 2 + role) * (24 * 8 + 8);
    if (tracing) trace_buf[24 * 8] = __builtin_amdgcn_s_memtime();
#endif
    // the same six statements one at a time (UM_FFN_ORDER 1: they are issued in the gaps of the phase-A MFMAs)
    // (LDS destinations as addresses: base of the array + this wave's kilobyte, in an SGPR, + a compile-time constant)
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds) + (unsigned)wave * 1024u;
    auto dma_piece = [&](int k, int jw1, int jw2, int slot) {      // k = 0..3: W1 (i = k >> 1, plane k & 1), 4..5: W2 plane k - 4
        if (k < 4) {
            const int i = k >> 1, pl = k & 1;
            if (pl >= NS) return;
            const int blk = 8 * i + wave;
            const int r = 2 * blk + half, c = tl ^ r;
            const unsigned off = (unsigned)((((long)(32 * (sbase + jw1) + r)) * 256 + 8 * c) * 2);
            ffn_dma16_at(a.w1 + pl * a.w1_plane_stride, off, lds_wave + (unsigned)(L::W1_OFF + slot * L::W1S + pl * L::W1P + 8 * i * 1024));
        } else {
            const int pl = k - 4;
            if (pl >= NS) return;
            const int r = 16 * wave + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
            const unsigned off = (unsigned)((((long)r) * a.hid + 32 * (sbase + jw2) + 8 * c) * 2);
            ffn_dma16_at(a.w2 + pl * a.w2_plane_stride, off, lds_wave + (unsigned)(L::W2_OFF + slot * L::W2S + pl * L::W2P));
        }
    };

    // UM_FFN_DMA_ASYM: the role-0 wave of pair p requests a quarter of the slice: k = 0..7: W1 rows 2 (4 (k >> 1) + p) + half, plane
    // k & 1;  k = 8..11: W2 rows 16 (2 p + ((k - 8) >> 1)) + lane / 4, plane k & 1
    auto dma_piece_asym = [&](int k, int jw1, int jw2, int slot) {
        const int pl = k & 1;
        if (pl >= NS) return;
        if (k < 8) {
            const int blk = 4 * (k >> 1) + pair;
            const int r = 2 * blk + half, c = tl ^ r;
            const unsigned off = (unsigned)((((long)(32 * (sbase + jw1) + r)) * 256 + 8 * c) * 2);
            ffn_dma16(a.w1 + pl * a.w1_plane_stride, off, lds + L::W1_OFF + slot * L::W1S + pl * L::W1P + blk * 1024);
        } else {
            const int blk = 2 * pair + ((k - 8) >> 1);
            const int r = 16 * blk + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
            const unsigned off = (unsigned)((((long)r) * a.hid + 32 * (sbase + jw2) + 8 * c) * 2);
            ffn_dma16(a.w2 + pl * a.w2_plane_stride, off, lds + L::W2_OFF + slot * L::W2S + pl * L::W2P + blk * 1024);
        }
    };

    dma_w1(0, 0);

    // ---- token operand: this role's 128 of the 256 input features, as B fragments (k = 16 ks + 8 half + 0..7) ----
    i16x8 xf[NS][8];
    {
        const float* src = (role ? a.y : a.x) + (long)min(tok, a.M - 1) * 128 + 8 * half;
        float xmx = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 v0 = UM_FFN_LD(reinterpret_cast<const f32x4*>(src + 16 * ks));
            const f32x4 v1 = UM_FFN_LD(reinterpret_cast<const f32x4*>(src + 16 * ks + 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) xmx = fmaxf(xmx, fmaxf(__builtin_fabsf(v0[j]), __builtin_fabsf(v1[j])));
            const u32x4 h = {T::pack2(v0[0], v0[1]), T::pack2(v0[2], v0[3]), T::pack2(v1[0], v1[1]), T::pack2(v1[2], v1[3])};
            xf[0][ks] = __builtin_bit_cast(i16x8, h);
            if (NS == 2) {
                const u32x4 l = {T::lo2(v0[0], v0[1], h[0], neg1), T::lo2(v0[2], v0[3], h[1], neg1),
                                 T::lo2(v1[0], v1[1], h[2], neg1), T::lo2(v1[2], v1[3], h[3], neg1)};
                xf[NS - 1][ks] = __builtin_bit_cast(i16x8, l);
            }
        }
        um_range_note<T>(a.range_flag, xmx, UM_RANGE_FFN_TOKENS);
    }

    f32x16 o[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ot][r] = 0.f;

    // fragment offsets.  Phase A: W1 row (tl ^ 16 role), chunk 16 role + 2 ks + half.  Phase B: W2 row 32 ot + tl,
    // chunk 2 role + half.
    const int arow = tl ^ (16 * role);
    int aoff[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) aoff[ks] = arow * 512 + (((16 * role + 2 * ks + half) ^ arow) << 4);
    int boff[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        const int r = 32 * ot + tl;
        boff[ot] = L::W2_OFF + r * 64 + (((2 * role + half) ^ ((r >> 2) & 3)) << 4);   // (carries the ring's base: beyond the 64 KB a DS
                                                                                         // instruction's immediate offset reaches)
    }
    unsigned char* xb_out = lds + L::XB_OFF + wave * L::XB + lane * 16;
    const unsigned char* xb_in = lds + L::XB_OFF + (wave ^ 4) * L::XB + lane * 16;

    // ---- phase A of one slice: partial S^T over this role's half of K (no scheduling directives: the caller owns the
    // scheduling region, because this runs interleaved with the GELU of the previous slice)
    auto phase_a = [&](const unsigned char* w1s, f32x16& sc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        if (UM_FFN_ABL & 4) return;
        i16x8 fh[3], fl[3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            fh[ks] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks]);
            if (NS == 2) fl[ks] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 2 < 8) {
                fh[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks + 2]);
                if (NS == 2) fl[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks + 2]);
            }
            if (NS == 2) {
                sc = T::mfma(fl[ks % 3], xf[0][ks], sc);
                sc = T::mfma(fh[ks % 3], xf[NS - 1][ks], sc);
            }
            sc = T::mfma(fh[ks % 3], xf[0][ks], sc);
        }
    };
    // rows 16..31 of the accumulator go to the partner
    auto send = [&](const f32x16& sc, int parity) {
        if (UM_FFN_ABL & 16) return;
        unsigned char* p = xb_out + parity * (8 * L::XB);
        *reinterpret_cast<f32x4*>(p) = f32x4{sc[8], sc[9], sc[10], sc[11]};
        *reinterpret_cast<f32x4*>(p + 1024) = f32x4{sc[12], sc[13], sc[14], sc[15]};
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nslice > 1) dma_w1(1, 1);

    f32x16 sc;                      // rows 0..15 (regs 0..7): the wave's own hidden units of the current slice
    __builtin_amdgcn_s_setprio(1);
    phase_a(lds + L::W1_OFF, sc);
    {
        constexpr int RD = NS, MF = (NS == 2) ? 3 : 1;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 0);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    send(sc, 0);

    // H^T operand fragments of the wave's 16 hidden units (one k-step), built in four stages from the GELU outputs
    struct Frag {
        unsigned wh[4], wl[4];
    };
    float hmx = 0.f;                // largest hidden activation turned into an operand (um_range_flags)
    auto frag_stage = [&](Frag& f, const Gelu2* g, i16x8* pf, int stage) {
        switch (stage) {
        case 0:
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hmx = fmaxf(hmx, fmaxf(__builtin_fabsf(g[q].x[0]), __builtin_fabsf(g[q].x[1])));
                f.wh[q] = T::pack2(g[q].x[0], g[q].x[1]);
            }
            break;
        case 1:                                  // lo plane: 2 x (v_fma_mixlo_f16 + v_fma_mixhi_f16) per stage (Fp16::lo2)
            if (NS == 2 && !UM_FFN_H1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) f.wl[q] = T::lo2(g[q].x[0], g[q].x[1], f.wh[q], neg1);
            }
            break;
        case 2:
            if (NS == 2 && !UM_FFN_H1) {
#pragma unroll
                for (int q = 2; q < 4; ++q) f.wl[q] = T::lo2(g[q].x[0], g[q].x[1], f.wh[q], neg1);
            }
            break;
        default: {
            const auto sx = __builtin_amdgcn_permlane32_swap(f.wh[0], f.wh[2], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(f.wh[1], f.wh[3], false, false);
            const u32x4 fh = {sx[0], sy[0], sx[1], sy[1]};
            pf[0] = __builtin_bit_cast(i16x8, fh);
            if (NS == 2 && !UM_FFN_H1) {
                const auto tx = __builtin_amdgcn_permlane32_swap(f.wl[0], f.wl[2], false, false);
                const auto ty = __builtin_amdgcn_permlane32_swap(f.wl[1], f.wl[3], false, false);
                const u32x4 fl = {tx[0], ty[0], tx[1], ty[1]};
                pf[NS - 1] = __builtin_bit_cast(i16x8, fl);
            }
            break;
        }
        }
    };

    // One iteration i (after barrier i):
    //     MFMA pipe :  phase B of slice i-1 (12 MFMAs, fragments built last iteration)  then  phase A of slice i+1 (24)
    //     VALU      :  GELU of slice i (4 x 4 stages) and its H^T fragments (4 stages), pinned behind those MFMAs
    // hipcc does not interleave the two on its own (and gives up on a sched_group_barrier pipeline of this size), so the
    // order is written out: every MFMA is followed by its share of stages and a scheduling fence.
    i16x8 pf[NS];                   // H^T fragments of the previous slice
    auto iteration = [&](auto has_a_tag, auto has_b_tag, int i) {
        constexpr bool HAS_A = decltype(has_a_tag)::value && !(UM_FFN_ABL & 4);
        constexpr bool HAS_B = decltype(has_b_tag)::value && !(UM_FFN_ABL & 8);
        constexpr int MF = (NS == 2) ? 3 : 1;
        constexpr int MFB = (NS == 2 && UM_FFN_H1) ? 2 : MF;       // products of phase B (see UM_FFN_H1)
        constexpr int NMFMA = (HAS_A ? 8 * MF : 0) + (HAS_B ? 4 * MFB : 0);
        constexpr int NGELU = 4 * UM_GELU_STAGES, NSTAGE = NGELU + 4;
        const int slot = i & 1;
        const unsigned char* w1s = lds + L::W1_OFF + (slot ^ 1) * L::W1S;     // W1(i+1)
        const unsigned char* w2s = lds + (slot ^ 1) * L::W2S;                 // (+ W2_OFF inside boff)     // W2(i-1)
        UM_FSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // W1(i+1), W2(i-1): this thread's pieces have landed
        UM_FSTAMP(1);
        __syncthreads();
        UM_FSTAMP(2);
        if (!(UM_FFN_ABL & 2)) {
            if (i + 2 < nslice) dma_w1(i + 2, slot);
            dma_w2(i, slot);
        }
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        if (!(UM_FFN_ABL & 16)) {
            const unsigned char* p = xb_in + slot * (8 * L::XB);
            r0 = *reinterpret_cast<const f32x4*>(p);
            r1 = *reinterpret_cast<const f32x4*>(p + 1024);
        }
        Gelu2 g[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                g[q].x[i] = (sc[2 * q + i] + r0[2 * q + i]) * a.out_scale;
                g[2 + q].x[i] = (sc[4 + 2 * q + i] + r1[2 * q + i]) * a.out_scale;
            }
        }
        Frag fr;
        i16x8 pfn[NS];
        int done = 0;                                              // VALU stages issued so far (compile-time after unrolling)
        auto valu_share = [&](int slot_idx) {                      // stages [slot_idx, slot_idx + 1) * NSTAGE / NMFMA
            const int upto = (slot_idx + 1) * NSTAGE / (NMFMA > 0 ? NMFMA : 1);
            for (; done < upto; ++done) {
                if (done < NGELU) ffn_gelu_stage(g[done / UM_GELU_STAGES], done % UM_GELU_STAGES);
                else frag_stage(fr, g, pfn, done - NGELU);
            }
        };
        f32x16 scn, scm;                                           // phase A alternates two accumulators (see phase B)
#pragma unroll
        for (int r = 0; r < 16; ++r) scn[r] = scm[r] = 0.f;
        UM_FSTAMP(3);
        __builtin_amdgcn_s_setprio(1);
        int mslot = 0;
        i16x8 fh[3], fl[3];                                        // phase A fragments, two k-steps ahead of their MFMAs
        if (HAS_A) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fh[ks] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks]);
                if (NS == 2) fl[ks] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks]);
            }
        }
        if (HAS_B) {
            // product index outer, output tile inner: consecutive MFMAs write different accumulators (a filler between two
            // MFMAs on the same accumulator costs the forwarding path, ~43 cycles -- MI355X_MICROARCH.md)
            i16x8 vh[4], vl[4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                vh[ot] = *reinterpret_cast<const i16x8*>(w2s + boff[ot]);
                if (NS == 2) vl[ot] = *reinterpret_cast<const i16x8*>(w2s + L::W2P + boff[ot]);
            }
#pragma unroll
            for (int m = 0; m < MFB; ++m)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    const i16x8 wa = (NS == 2 && m == 0) ? vl[ot] : vh[ot];
                    const i16x8 hb = (NS == 2 && MFB == 3 && m == 1) ? pf[NS - 1] : pf[0];
                    o[ot] = T::mfma(wa, hb, o[ot]);
                    valu_share(mslot++);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x402, 16, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (HAS_A) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 2 < 8) {
                    fh[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks + 2]);
                    if (NS == 2) fl[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks + 2]);
                }
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    const i16x8 wa = (NS == 2 && m == 0) ? fl[ks % 3] : fh[ks % 3];
                    const i16x8 xb = (NS == 2 && m == 1) ? xf[NS - 1][ks] : xf[0][ks];
                    if ((ks * MF + m) & 1) scm = T::mfma(wa, xb, scm);
                    else scn = T::mfma(wa, xb, scn);
                    valu_share(mslot++);
                    if (m == 0 && ks + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, NS, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x402, 16, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        UM_FSTAMP(4);
        for (; done < NSTAGE; ++done) {                            // whatever no MFMA was left to hide
            if (done < NGELU) ffn_gelu_stage(g[done / UM_GELU_STAGES], done % UM_GELU_STAGES);
            else frag_stage(fr, g, pfn, done - NGELU);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(has_a_tag)::value) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scn[r] += scm[r];
            send(scn, slot ^ 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) sc[r] = scn[r];
        }
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) pf[pl] = pfn[pl];
        UM_FSTAMP(5);
    };

    // ---- UM_FFN_ORDER 1: one iteration i (after barrier i) --------------------------------------------------------------------
    //     MFMA pipe :  phase A of slice i+1 (24 MFMAs)  then  phase B of slice i-1 (12, fragments built last iteration)
    //     gaps      :  GELU of slice i + its H^T fragments (as before, one share per MFMA); the LDS-DMA statements of W1(i+2) and
    //                  W2(i) behind the first phase-A MFMAs; the accumulator hand-over (add, send) behind the first phase-B MFMAs
    auto iteration1 = [&](auto has_a_tag, auto has_b_tag, auto dma1_tag, auto slot_tag, int i) {
        constexpr bool HAS_A = decltype(has_a_tag)::value && !(UM_FFN_ABL & 4);
        constexpr bool HAS_B = decltype(has_b_tag)::value && !(UM_FFN_ABL & 8);
        constexpr bool DMA1 = decltype(dma1_tag)::value;           // W1(i+2) exists
        constexpr int MF = (NS == 2) ? 3 : 1;
        constexpr int MFB = (NS == 2 && UM_FFN_H1) ? 2 : MF;
        constexpr int NA = HAS_A ? 8 * MF : 0, NB = HAS_B ? 4 * MFB : 0, NMFMA = NA + NB;
        constexpr int NGELU = 4 * UM_GELU_STAGES, NSTAGE = NGELU + 4;
        // ring slot = i & 1, at COMPILE time (round 5: the loop runs two slices per trip): every fragment read, exchange access and
        // LDS-DMA destination of the slice is register + immediate -- with the slot in a register the slice carried 16 v_add_u32 and
        // ~25 scalar instructions of slot arithmetic (SQ_INSTS_SALU / SQ_INSTS_MFMA = 1.66, profiles/r05_pmc_bounds.json)
        const int slot = slot_tag;                                // (an integral_constant in the paired loop, an int around it)
        const unsigned char* w1s = lds + L::W1_OFF + (slot ^ 1) * L::W1S;     // W1(i+1)
        const unsigned char* w2s = lds + (slot ^ 1) * L::W2S;                 // (+ W2_OFF inside boff)     // W2(i-1)
        UM_FSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // W1(i+1), W2(i-1): this thread's pieces have landed
        UM_FSTAMP(1);
        __syncthreads();
        UM_FSTAMP(2);
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        if (!(UM_FFN_ABL & 16)) {                                  // the partner's half of S^T(i): first in the LDS queue
            const unsigned char* p = xb_in + slot * (8 * L::XB);
            r0 = *reinterpret_cast<const f32x4*>(p);
            r1 = *reinterpret_cast<const f32x4*>(p + 1024);
        }
        i16x8 fh[3], fl[3];                                        // phase A fragments, two k-steps ahead of their MFMAs
        if (HAS_A) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fh[ks] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks]);
                if (NS == 2) fl[ks] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks]);
            }
        }
        Gelu2 g[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                g[q].x[e] = (sc[2 * q + e] + r0[2 * q + e]) * a.out_scale;
                g[2 + q].x[e] = (sc[4 + 2 * q + e] + r1[2 * q + e]) * a.out_scale;
            }
        }
        Frag fr;
        i16x8 pfn[NS];
        int done = 0;
        auto valu_share = [&](int slot_idx) {
            const int upto = (slot_idx + 1) * NSTAGE / (NMFMA > 0 ? NMFMA : 1);
            for (; done < upto; ++done) {
                if (done < NGELU) ffn_gelu_stage(g[done / UM_GELU_STAGES], done % UM_GELU_STAGES);
                else frag_stage(fr, g, pfn, done - NGELU);
            }
        };
        // LDS-DMA pieces still to issue: W1(i+2) (4 statements at NS = 2) when it exists, W2(i) (2 statements); one per MFMA gap
        // from gap 1 on (gap 0 carries the exchange arithmetic)
        int piece = UM_FFN_DMA_ASYM ? (DMA1 ? 0 : 8) : (DMA1 ? 0 : 4);
        constexpr int NPIECE = UM_FFN_DMA_ASYM ? 12 : 6;
        auto dma_share = [&](int slot_idx) {
            if (UM_FFN_ABL & 2) return;
            if (slot_idx >= 1 && piece < NPIECE) {
                if (UM_FFN_DMA_ASYM) {
                    if (role == 0) dma_piece_asym(piece, i + 2, i, slot);
                    piece += (NS == 1) ? 2 : 1;                            // (one plane: the odd pieces do not exist)
                } else {
                    dma_piece(piece, i + 2, i, slot);
                    ++piece;
                    if (NS == 1 && (piece == 1 || piece == 3)) ++piece;    // (one plane: pieces 1, 3, 5 do not exist)
                }
            }
        };
        f32x16 scn, scm;
#pragma unroll
        for (int r = 0; r < 16; ++r) scn[r] = scm[r] = 0.f;
        UM_FSTAMP(3);
        __builtin_amdgcn_s_setprio(1);
        auto prio = [&](int ms) {                                  // ms: compile-time after unrolling; role: wave-uniform
            int fav = -1;
            if (UM_FFN_PRIO == 1 && (ms == 0 || ms == NMFMA / 2)) fav = ms == 0 ? 0 : 1;
            if (UM_FFN_PRIO == 2 && ms % 6 == 0) fav = (ms / 6) & 1;
            if (UM_FFN_PRIO == 4 && ms % 3 == 0) fav = (ms / 3) & 1;
            if (UM_FFN_PRIO == 5 && ms % 12 == 0) fav = (ms / 12) & 1;
            if (UM_FFN_PRIO == 3 && ms == 0) fav = 1;
            if (fav < 0) return;
            if (role == fav) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        };
        int mslot = 0;
        i16x8 vh[4], vl[4];                                        // phase B's W2 fragments: read in the last gaps of phase A
        if (HAS_A) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 2 < 8) {
                    fh[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + aoff[ks + 2]);
                    if (NS == 2) fl[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(w1s + L::W1P + aoff[ks + 2]);
                }
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    const i16x8 wa = (NS == 2 && m == 0) ? fl[ks % 3] : fh[ks % 3];
                    const i16x8 xb = (NS == 2 && m == 1) ? xf[NS - 1][ks] : xf[0][ks];
                    prio(mslot);
                    if ((ks * MF + m) & 1) scm = T::mfma(wa, xb, scm);
                    else scn = T::mfma(wa, xb, scn);
                    const int left = NA - 1 - (ks * MF + m);       // gaps of phase A after this one
                    const bool vread = HAS_B && left < 4;          // the last four gaps: W2 fragment ot = 3 - left
                    if (vread) {
                        vh[3 - left] = *reinterpret_cast<const i16x8*>(w2s + boff[3 - left]);
                        if (NS == 2) vl[3 - left] = *reinterpret_cast<const i16x8*>(w2s + L::W2P + boff[3 - left]);
                    }
                    dma_share(mslot);
                    valu_share(mslot++);
                    if ((m == 0 && ks + 2 < 8) || vread) __builtin_amdgcn_sched_group_barrier(0x100, NS, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x402, 16, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (HAS_B) {
            if (!HAS_A) {
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    vh[ot] = *reinterpret_cast<const i16x8*>(w2s + boff[ot]);
                    if (NS == 2) vl[ot] = *reinterpret_cast<const i16x8*>(w2s + L::W2P + boff[ot]);
                }
            }
#pragma unroll
            for (int m = 0; m < MFB; ++m)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    const i16x8 wa = (NS == 2 && m == 0) ? vl[ot] : vh[ot];
                    const i16x8 hb = (NS == 2 && MFB == 3 && m == 1) ? pf[NS - 1] : pf[0];
                    prio(mslot);
                    o[ot] = T::mfma(wa, hb, o[ot]);
                    const int bslot = m * 4 + ot;
                    dma_share(mslot);                              // (only when there was no phase A: the last slice)
                    if (decltype(has_a_tag)::value) {
                        // hand-over of S^T(i+1): the last phase-A MFMAs retire under the first phase-B MFMAs.  Gaps H0, H0 + 1, H0 + 2
                        // of phase B (2, 3, 4 with three products; 1, 2, 3 when phase B has only four MFMAs: one bf16 plane)
                        constexpr int H0 = NB >= 6 ? 2 : NB - 3;
                        static_assert(NB == 0 || H0 >= 0, "phase B too short for the hand-over");
                        if (bslot == H0) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) scn[r] += scm[r];
                        }
                        if (bslot == H0 + 1) {
#pragma unroll
                            for (int r = 8; r < 16; ++r) scn[r] += scm[r];
                        }
                        if (bslot == H0 + 2) send(scn, slot ^ 1);
                    }
                    valu_share(mslot++);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x602, 24, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        UM_FSTAMP(4);
        for (; done < NSTAGE; ++done) {
            if (done < NGELU) ffn_gelu_stage(g[done / UM_GELU_STAGES], done % UM_GELU_STAGES);
            else frag_stage(fr, g, pfn, done - NGELU);
        }
        if (!(UM_FFN_ABL & 2)) {
            while (piece < NPIECE) {                               // pieces no gap took (short streams: one plane, ablation builds)
                if (UM_FFN_DMA_ASYM) {
                    if (role == 0) dma_piece_asym(piece, i + 2, i, slot);
                    piece += (NS == 1) ? 2 : 1;
                } else {
                    dma_piece(piece, i + 2, i, slot);
                    ++piece;
                    if (NS == 1 && (piece == 1 || piece == 3)) ++piece;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(has_a_tag)::value) {
            if (!HAS_B) {                                          // first slice: no phase B to hide the hand-over in
#pragma unroll
                for (int r = 0; r < 16; ++r) scn[r] += scm[r];
                send(scn, slot ^ 1);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) sc[r] = scn[r];
        }
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) pf[pl] = pfn[pl];
        UM_FSTAMP(5);
    };

    if (UM_FFN_ORDER == 1) {
        using Y = std::true_type;
        using N = std::false_type;
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        if ((nslice & 1) == 0 && nslice >= 4) {                    // every shipped geometry: hid = 1024 -> 32 / 16 / 8 slices
            iteration1(Y{}, N{}, Y{}, S0{}, 0);
            int i = 1;
            for (; i + 4 < nslice; i += 2) {                       // slices 1 .. nslice - 4 in pairs (odd, even)
                iteration1(Y{}, Y{}, Y{}, S1{}, i);
                iteration1(Y{}, Y{}, Y{}, S0{}, i + 1);
            }
            iteration1(Y{}, Y{}, Y{}, S1{}, nslice - 3);
            iteration1(Y{}, Y{}, N{}, S0{}, nslice - 2);
            iteration1(N{}, Y{}, N{}, S1{}, nslice - 1);
        } else {                                                    // any other slice count: the slot in a register
            if (2 < nslice) iteration1(Y{}, N{}, Y{}, 0, 0);
            else iteration1(Y{}, N{}, N{}, 0, 0);
            int i = 1;
            for (; i + 2 < nslice; ++i) iteration1(Y{}, Y{}, Y{}, i & 1, i);
            for (; i + 1 < nslice; ++i) iteration1(Y{}, Y{}, N{}, i & 1, i);
            iteration1(N{}, Y{}, N{}, (nslice - 1) & 1, nslice - 1);
        }
    } else {
        iteration(std::true_type{}, std::false_type{}, 0);
        for (int i = 1; i + 1 < nslice; ++i) iteration(std::true_type{}, std::true_type{}, i);
        iteration(std::false_type{}, std::true_type{}, nslice - 1);
    }
    // The residual rows (this lane's 64 fp32 values of x) are requested HERE, a whole exchange + LayerNorm ahead of their use: the token
    // operand's 64 registers are dead since the last phase A.  (They are a second read of x: 50 MB of the launch's 182 MB of fabric reads
    // at config 2 -- the tile left the XCD's L2 ~100 us ago; profiles/r05_ffn_overfetch.txt.)
    f32x4 rr[16];
    {   // phase B of the last slice
        const unsigned char* w2s = lds + ((nslice - 1) & 1) * L::W2S;          // (+ W2_OFF inside boff)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (role == 0) {
            const float* res = a.x + (long)min(tok, a.M - 1) * 128 + 4 * half;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                rr[q] = (UM_FFN_ABL & 32) ? f32x4{0.f, 0.f, 0.f, 0.f} : UM_FFN_LD(reinterpret_cast<const f32x4*>(res + 8 * q));
        }
        if (!(UM_FFN_ABL & 8)) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                const i16x8 vh = *reinterpret_cast<const i16x8*>(w2s + boff[ot]);
                if (NS == 2) {
                    const i16x8 vl = *reinterpret_cast<const i16x8*>(w2s + L::W2P + boff[ot]);
                    o[ot] = T::mfma(vl, pf[0], o[ot]);
                    if (!UM_FFN_H1) o[ot] = T::mfma(vh, pf[NS - 1], o[ot]);
                }
                o[ot] = T::mfma(vh, pf[0], o[ot]);
            }
        }
    }

#ifdef UM_FFN_TRACE
    if (tracing) trace_buf[24 * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
    um_range_note<T>(a.range_flag, hmx, UM_RANGE_FFN_HIDDEN);
    // ---- epilogue: add the partner's partial O, LayerNorm over the 128 outputs, residual ------------------------------
    __syncthreads();                                               // every ring slot and exchange buffer is dead
    static_assert(!(KV4 && HSPLIT), "the k | v epilogue runs on whole tiles only");
    // KV4: the W1 ring [0, 64 KB) takes the packed k | v weight chunks (chunk 0 is requested now and lands under the LayerNorm);
    // the exchanges and later the transposition scratch live in [64 KB, 128 KB)
    if constexpr (KV4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = 8 * i + wave;
            const int r = 2 * blk + half, cc = tl ^ r;
            const unsigned off = (unsigned)((((long)r) * 256 + 8 * cc) * 2);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                ffn_dma16(a.kv.w + pl * a.kv.w_plane_stride, off, lds + pl * L::W1P + blk * 1024);
        }
    }
    unsigned char* ob = lds + (KV4 ? 65536 : 0) + pair * 16384 + lane * 16;
    if (role == 1) {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(ob + (4 * ot + g) * 1024) =
                    f32x4{o[ot][4 * g], o[ot][4 * g + 1], o[ot][4 * g + 2], o[ot][4 * g + 3]};
    }
    __syncthreads();
    if (!HSPLIT && !KV4 && role == 1) return;
    if (role == 0) {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(ob + (4 * ot + g) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[ot][4 * g + i] = (o[ot][4 * g + i] + r[i]) * a.out_scale;
            }
    }
    if constexpr (HSPLIT) {          // all 8 waves stay for the barriers; the role-0 waves (256 threads) carry O^T
        // Hand-off without waiting (as in window_attn.hip): every part publishes its partial O^T in its memory slot (16-byte
        // agent-scope stores), takes a ticket from the tile's arrival counter, and the LAST arriver sums all slots in part order
        // (fixed order: bitwise reproducible) and finishes the layer; the others exit.  Slot: 16 vectors x 256 threads x 16 B.
        const int t4 = 64 * pair + lane;
        constexpr int HS_SLOT = 16 * 256 * 4;                      // floats per slot
        if (role == 0) {
            float* mine = a.hs_part + ((long)tile_id * nsplit + part) * HS_SLOT + 4 * t4;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {o[ot][4 * g], o[ot][4 * g + 1], o[ot][4 * g + 2], o[ot][4 * g + 3]};
                    st_agent_16B(mine + (ot * 4 + g) * 1024, v);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // every thread's stores are complete
        unsigned* ticket = reinterpret_cast<unsigned*>(lds);       // (rings and exchange buffers are dead)
        if (tid == 0) *ticket = __hip_atomic_fetch_add(a.hs_flag + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned my_ticket = *ticket;
        __syncthreads();             // every wave holds the ticket in a register: the row-store staging below reuses this LDS word
                                     // (round-5 ADVICE: a delayed wave could have read staged data as its ticket)
        if (my_ticket != (unsigned)(nsplit - 1)) return;           // not the last arriver: done
        asm volatile("buffer_inv sc1" ::: "memory");               // acquire side of the hand-off, last arriver only (window_attn.hip)
        if (tid == 0) __hip_atomic_store(a.hs_flag + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next launch
        if (role == 1) return;
        for (int p = 0; p < nsplit; ++p) {
            const float* pr = a.hs_part + ((long)tile_id * nsplit + p) * HS_SLOT + 4 * t4;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                f32x4 w[4];
                const float* q = pr + ot * 4 * 1024;
                ld_agent_16Bx4(q, q + 1024, q + 2048, q + 3072, w[0], w[1], w[2], w[3]);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[ot][4 * g + i] = (p == 0 ? 0.f : o[ot][4 * g + i]) + w[g][i];
            }
        }
    }
    float s1 = 0.f;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1 += o[ot][r];
    float u, v2;
    half_wave_pair(s1, u, v2);
    const float mean = (u + v2) * (1.0f / 128.0f);
    float s2 = 0.f;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = o[ot][r] - mean;
            s2 = __builtin_fmaf(d, d, s2);
        }
    half_wave_pair(s2, u, v2);
    const float rstd = 1.0f / sqrtf((u + v2) * (1.0f / 128.0f) + a.eps);
#if UM_FFN_ROWSTORE
    if (role == 0) {
        // lane holds, for its token, features 32 ot + 8 g + 4 half + i  (reg 4 g + i of tile ot).  The tile leaves through LDS as
        // WHOLE ROWS (round 5): a store instruction of the row-per-lane form touches 32 separate 32-byte pieces, and the tail of 16 of
        // them per lane is store-issue-bound (MI355X_MICROARCH.md: ~9 k cycles per workgroup, exposed here -- one workgroup per CU);
        // transposed, an instruction writes two 512-byte rows.  Staging: the pair's own exchange block (16 KB, consumed above), 16-byte
        // chunk c of row r at c ^ r: the writes (8 rows per lane group) and the reads (32 chunks of a row) are conflict free.
        unsigned char* stg = lds + (KV4 ? 65536 : 0) + pair * 16384;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * ot + 8 * g;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + n + 4 * half);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + n + 4 * half);
                f32x4 yv;
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = (o[ot][4 * g + i] - mean) * rstd * gm[i] + bt[i] + rr[4 * ot + g][i];
                *reinterpret_cast<f32x4*>(stg + tl * 512 + (((8 * ot + 2 * g + half) ^ tl) << 4)) = yv;
                if constexpr (KV4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[ot][4 * g + i] = yv[i];
                }
            }
        __builtin_amdgcn_wave_barrier();                            // same wave: LDS executes its accesses in order
        const int cc = lane & 31;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = 2 * j + half;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * 512 + ((cc ^ r) << 4));
            const int trow = m0 + 32 * pair + r;
            if (trow < a.M) UM_FFN_ST(reinterpret_cast<f32x4*>(a.out + (long)trow * 128 + 4 * cc), v);
        }
    }
#else       // round 1-4: every lane stores its own token's 16-byte pieces (diagnostic builds: -DUM_FFN_ROWSTORE=0)
    if (KV4 ? role == 0 : tok < a.M) {
        // lane holds, for its token, features 32 ot + 8 g + 4 half + i  (reg 4 g + i of tile ot)
        const int tokc = min(tok, a.M - 1);                         // (KV4: rows past M are computed on a valid row and never stored)
        float* dst = a.out + (long)tokc * 128 + 4 * half;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * ot + 8 * g;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + n + 4 * half);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + n + 4 * half);
                f32x4 yv;
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = (o[ot][4 * g + i] - mean) * rstd * gm[i] + bt[i] + rr[4 * ot + g][i];
                if (tok < a.M) UM_FFN_ST(reinterpret_cast<f32x4*>(dst + n), yv);
                if constexpr (KV4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[ot][4 * g + i] = yv[i];
                }
            }
    }
#endif
#ifdef UM_FFN_TRACE
    if (tracing) trace_buf[24 * 8 + 2] = __builtin_amdgcn_s_memtime();          // end of LayerNorm + residual + stores
#endif
    if constexpr (KV4) {
        // ---- the next block's k | v projections of this tile (kv4_project): Y^T operand fragments from the normalised tile exactly as
        // the stand-alone kernel builds them from the fp32 tokens it reads back (same hi | lo split of the same fp32 values), handed to
        // the partner wave through LDS; k-step 2 ot + kk of the operand = registers 8 kk .. 8 kk + 7 of tile ot (as Q^T in window_attn.hip)
        i16x8 yf[NS][8];
        unsigned char* xch = lds + 65536 + pair * 16384 + lane * 16;
        if (role == 0) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    unsigned wh[4], wl[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float p0 = o[ot][8 * kk + 2 * j], p1 = o[ot][8 * kk + 2 * j + 1];
                        wh[j] = T::pack2(p0, p1);
                        if (NS == 2) wl[j] = T::lo2(p0, p1, wh[j], neg1);
                    }
                    {
                        const auto xx = __builtin_amdgcn_permlane32_swap(wh[0], wh[2], false, false);
                        const auto yy = __builtin_amdgcn_permlane32_swap(wh[1], wh[3], false, false);
                        const u32x4 f = {xx[0], yy[0], xx[1], yy[1]};
                        yf[0][2 * ot + kk] = __builtin_bit_cast(i16x8, f);
                    }
                    if (NS == 2) {
                        const auto xx = __builtin_amdgcn_permlane32_swap(wl[0], wl[2], false, false);
                        const auto yy = __builtin_amdgcn_permlane32_swap(wl[1], wl[3], false, false);
                        const u32x4 f = {xx[0], yy[0], xx[1], yy[1]};
                        yf[NS - 1][2 * ot + kk] = __builtin_bit_cast(i16x8, f);
                    }
                }
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) *reinterpret_cast<i16x8*>(xch + (pl * 8 + ks) * 1024) = yf[pl][ks];
        }
        __syncthreads();
        if (role == 1) {
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) yf[pl][ks] = *reinterpret_cast<const i16x8*>(xch + (pl * 8 + ks) * 1024);
        }
        kv4_project<T, NS, true>(lds, lds + 65536, a.kv, yf, m0, wave, lane, aoff, neg1);
#ifdef UM_FFN_TRACE
        if (tracing) trace_buf[24 * 8 + 3] = __builtin_amdgcn_s_memtime();      // end of the k | v projection epilogue
#endif
    }
}

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);
#ifdef UM_FFN_TRACE
extern "C" int um_debug_set_ffn_trace(void* ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_um_ffn_trace), &ptr, sizeof(ptr));
}
#endif

template <typename T, int NS, bool HSPLIT, bool KV4 = false>
static hipError_t launch_ffn_impl(const FfnArgs& a, hipStream_t stream) {
    // KV4: ring [0, 64 KB) + exchange / scratch [64 KB, 128 KB) -- the same 128 KB the main loop uses
    constexpr int LDS = FfnLds<NS>::TOTAL > 131072 ? FfnLds<NS>::TOTAL : (KV4 ? 131072 : FfnLds<NS>::TOTAL);
    static bool configured = false;          // opt in to > 64 KB of LDS once per instantiation
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_kernel<T, NS, HSPLIT, KV4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    ScopedKernelTimer timer(UM_K_FFN, stream);
    um_census_hit(HSPLIT ? UM_V_FFN_HSPLIT : UM_V_FFN_TILE);
    hipLaunchKernelGGL((ffn_kernel<T, NS, HSPLIT, KV4>), dim3(((a.M + 127) / 128) * a.split), dim3(512), LDS, stream, a);
    return hipGetLastError();
}

template <typename T, int NS>
static hipError_t launch_ffn(const FfnArgs& a, hipStream_t stream) {
    if (a.kv.w) return launch_ffn_impl<T, NS, false, true>(a, stream);       // (callers only set kv on un-split launches)
    return a.split > 1 ? launch_ffn_impl<T, NS, true>(a, stream) : launch_ffn_impl<T, NS, false>(a, stream);
}

// ---- hidden split for small launches: while tiles x parts fit the chip (one workgroup per CU: 128 KB of LDS each)
static int ffn_num_cus() { return um_num_cus(); }      // per device (common.h)

static int ffn_hidden_split(int m, int hidden) {
    static const bool off = um_debug_env("UM_FFN_NO_HSPLIT") != nullptr;        // A/B switch
    if (off) return 1;
    const int tiles = (m + 127) / 128, nslice = hidden / 32, cus = ffn_num_cus();
    int split = 1;
    while (split < 4 && tiles * (2 * split) <= cus && nslice % (2 * split) == 0 && nslice / (2 * split) >= 4) split *= 2;
    return split;
}

static size_t ffn_hs_bytes(int m, int split) {                    // arrival counters + one slot per part
    const size_t tiles = (size_t)((m + 127) / 128);
    return ((tiles * sizeof(unsigned) + 255) & ~(size_t)255) + tiles * split * (64 * 256 * sizeof(float));
}

extern "C" size_t um_ffn_split_workspace_bytes(int m, int hidden) {
    if (m <= 0 || hidden < 64 || hidden % 32 != 0) return 0;
    const int split = ffn_hidden_split(m, hidden);
    return split > 1 ? ffn_hs_bytes(m, split) : 0;
}

extern "C" int um_ffn_ws_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                             int wshift, const float* gamma, const float* beta, float eps, float* out, int mode, void* workspace,
                             size_t workspace_bytes, void* stream_);

extern "C" int um_ffn_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                          int wshift, const float* gamma, const float* beta, float eps, float* out, int mode,
                          void* stream_) {
    return um_ffn_ws_fwd(x, y, w1_planes, w2_planes, m, hidden, wshift, gamma, beta, eps, out, mode, nullptr, 0, stream_);
}

extern "C" int um_kv4_fwd(const float* x, const void* wc_planes, int m, int wshift, void* out_planes, int mode, void* stream_);
static int ffn_impl(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                    int wshift, const float* gamma, const float* beta, float eps, float* out, int mode, void* workspace,
                    size_t workspace_bytes, const void* wc_planes, void* kv_planes, void* stream_);

extern "C" int um_ffn_ws_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                             int wshift, const float* gamma, const float* beta, float eps, float* out, int mode, void* workspace,
                             size_t workspace_bytes, void* stream_) {
    return ffn_impl(x, y, w1_planes, w2_planes, m, hidden, wshift, gamma, beta, eps, out, mode, workspace, workspace_bytes, nullptr,
                    nullptr, stream_);
}

// The FFN of block i AND the key / value projections of block i + 1 (um_kv4_fwd's result on `out`) from one launch: the tile a
// workgroup has just normalised goes straight into kv4_project.  Small launches (hidden split, see um_ffn_split_workspace_bytes)
// keep the FFN's split variant and run um_kv4_fwd as a second launch: same results either way, the caller need not know.
extern "C" int um_ffn_kv_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                             int wshift, const float* gamma, const float* beta, float eps, float* out, const void* wc_planes,
                             void* kv_planes, int mode, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!wc_planes || !kv_planes) {
        um_set_error("um_ffn_kv_fwd: null k | v weight planes or output planes");
        return -1;
    }
    return ffn_impl(x, y, w1_planes, w2_planes, m, hidden, wshift, gamma, beta, eps, out, mode, workspace, workspace_bytes, wc_planes,
                    kv_planes, stream_);
}

static int ffn_impl(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                    int wshift, const float* gamma, const float* beta, float eps, float* out, int mode, void* workspace,
                    size_t workspace_bytes, const void* wc_planes, void* kv_planes, void* stream_) {
    if (!x || !y || !w1_planes || !w2_planes || !gamma || !beta || !out || m <= 0 || hidden < 64 || hidden % 32 != 0 ||
        (mode != 0 && mode != 1) || wshift < 0 || wshift > 14) {
        um_set_error("um_ffn_fwd: bad argument (m=%d hidden=%d wshift=%d mode=%d; hidden must be a multiple of 32, >= 64)",
                     m, hidden, wshift, mode);
        return -1;
    }
    if ((long)hidden * 256 * 2 >= (1L << 32)) {
        um_set_error("um_ffn_fwd: weight planes beyond 4 GiB are not addressable by this kernel");
        return -4;
    }
    FfnArgs a;
    a.x = x;
    a.y = y;
    a.w1 = (const unsigned short*)w1_planes;
    a.w1_plane_stride = (long)hidden * 256;
    a.w2 = (const unsigned short*)w2_planes;
    a.w2_plane_stride = (long)128 * hidden;
    a.gamma = gamma;
    a.beta = beta;
    a.out = out;
    a.M = m;
    a.hid = hidden;
    a.out_scale = ldexpf(1.f, -wshift);
    a.eps = eps;
    a.split = 1;
    a.hs_part = nullptr;
    a.hs_flag = nullptr;
    a.kv = Kv4Args{};
    a.range_flag = a.kv.range_flag = (mode == 0) ? um_range_flag_dev() : nullptr;
    if (workspace) {
        const int split = ffn_hidden_split(m, hidden);
        if (split > 1 && workspace_bytes >= ffn_hs_bytes(m, split)) {
            const size_t slots = (size_t)((m + 127) / 128);       // one arrival counter per tile
            a.split = split;
            a.hs_flag = (unsigned*)workspace;
            a.hs_part = (float*)((unsigned char*)workspace + ((slots * sizeof(unsigned) + 255) & ~(size_t)255));
        }
    }
    static const bool no_fuse = um_debug_env("UM_FFN_NO_KV4") != nullptr;     // A/B switch (diagnostic builds): always two launches
    const bool fuse = wc_planes && a.split == 1 && !no_fuse;
    if (fuse) {
        a.kv.x = nullptr;
        a.kv.w = (const unsigned short*)wc_planes;
        a.kv.w_plane_stride = 256L * 256;
        a.kv.out = (unsigned short*)kv_planes;
        a.kv.out_plane_stride = 4L * m * 128;
        a.kv.M = m;
        a.kv.out_scale = a.out_scale;
    }
    const hipError_t e = mode == 0 ? launch_ffn<Fp16, 2>(a, (hipStream_t)stream_) : launch_ffn<Bf16, 1>(a, (hipStream_t)stream_);
    if (e != hipSuccess) {
        um_set_error("um_ffn_fwd: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (wc_planes && !fuse) return um_kv4_fwd(out, wc_planes, m, wshift, kv_planes, mode, stream_);
    return 0;
}

// ---- stand-alone k | v projection of a Transformer block (see kv4_project): x fp32 [m][128], wc_planes = planes of the packed
// [256][256] weight (um_weight_planes), out_planes [NS][4][m][128]
extern "C" int um_kv4_fwd(const float* x, const void* wc_planes, int m, int wshift, void* out_planes, int mode, void* stream_) {
    if (!x || !wc_planes || !out_planes || m <= 0 || (mode != 0 && mode != 1) || wshift < 0 || wshift > 14) {
        um_set_error("um_kv4_fwd: bad argument (m=%d wshift=%d mode=%d)", m, wshift, mode);
        return -1;
    }
    Kv4Args k;
    k.x = x;
    k.w = (const unsigned short*)wc_planes;
    k.w_plane_stride = 256L * 256;
    k.out = (unsigned short*)out_planes;
    k.out_plane_stride = 4L * m * 128;
    k.M = m;
    k.out_scale = ldexpf(1.f, -wshift);
    k.range_flag = (mode == 0) ? um_range_flag_dev() : nullptr;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipSuccess;
    ScopedKernelTimer timer(UM_K_LINEAR, stream);
    if (mode == 0) {
        constexpr int LDS = Kv4Lds<2>::RING + Kv4Lds<2>::SCR_BYTES;
        static bool configured = false;
        if (!configured) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kv4_kernel<Fp16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return (int)e;
            configured = true;
        }
        hipLaunchKernelGGL((kv4_kernel<Fp16, 2>), dim3((m + 127) / 128), dim3(512), LDS, stream, k);
    } else {
        constexpr int LDS = Kv4Lds<1>::RING + Kv4Lds<1>::SCR_BYTES;
        static bool configured = false;
        if (!configured) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kv4_kernel<Bf16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return (int)e;
            configured = true;
        }
        hipLaunchKernelGGL((kv4_kernel<Bf16, 1>), dim3((m + 127) / 128), dim3(512), LDS, stream, k);
    }
    e = hipGetLastError();
    if (e != hipSuccess) {
        um_set_error("um_kv4_fwd: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
