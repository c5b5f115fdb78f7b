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
// Two kernels:
//   gsv4_kernel  non-causal launches whose key count is whole 64-key tiles and that fill the chip (flow, propagation at the
//                BASELINE sizes): one wave per SIMD, 64 queries per wave, Q fragments in AGPRs, stream-K over the chip;
//   gsv3_kernel  everything else -- the causal per-scanline stereo layer, ragged key counts, small launches: workgroup =
//                4 waves = 128 queries, wave = 32 queries, key tiles of 64, software-pipelined at instruction level.
// Scores are computed "swapped", S^T = K . Q^T with v_mfma_f32_32x32x16, so lane l owns query (l & 31)
// and holds 16 of the 32 key scores of a sub-tile; lanes l and l^32 split the keys of a query between
// them and keep INDEPENDENT running maxima, merged once at the end (no cross-lane traffic per tile).
// (Round 1's phase-structured gsv_kernel was retired in round 3; profiles/r02_gsv4_v1_ab.txt has its last A/B.)
#include <stdlib.h>
#include <type_traits>
#include <utility>
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
    int nsplit;                 // key range split across gridDim.z workgroups (load balance); 1 = direct output
    float* partial;             // nsplit > 1: [nsplit][nbatch][Lq][2 + NV] = (M, l, acc...) per split
    // gsv4_kernel ("stream-K" decomposition): the (batch, 256-query tile, 64-key tile) units, key tile fastest, are cut into
    // equal chunks, one per workgroup; a query tile's segments go to partial slots 0, 1, ... in chunk order
    int nbatch, qtiles, chunk;
};

// 16 (or 4) bytes per lane, global -> LDS without passing through VGPRs (LDS address = wave-uniform dst + size * lane).
// Issued through inline asm: hipcc would otherwise drain the transfer in front of the next LDS read (see window_attn.hip).
__device__ __forceinline__ void gsv_dma16(const void* base, unsigned byte_off, const unsigned char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void gsv_dma4(const void* base, unsigned byte_off, const unsigned char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(dst)
                 : "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// gsv3_kernel: software-pipelined at instruction level.
//
// What round 1's profile said about its phase-structured predecessor (profiles/r01_pmc_final.json): 6.7 VALU instructions per MFMA, every wave
// 49 % of its cycles stalled at issue and the matrix pipe busy 49 % -- a wave issued its 48 MFMAs of a tile back to back
// (32 cycles each with nothing else issued), then ~320 VALU instructions of softmax with the matrix pipe idle unless the
// co-resident wave happened to be in its MFMA phase.  Here
//   * the MFMAs of tile t+1 and the softmax of tile t form ONE instruction stream: after every MFMA come the 4-6 VALU /
//     LDS / DMA instructions that fit in its 32-cycle shadow, pinned there with scheduling fences (two accumulator sets);
//   * the softmax costs 4 VALU per score instead of 10: log2(e)/sqrt(C) is folded into the operand planes (sqrt of it on
//     each side, so one set of planes still serves both directions of a bidirectional launch), the running offset M is
//     the MFMAs' initial accumulator (srcC = a register block holding M), so p = exp2(acc) directly; the offset is
//     renormalised lazily -- only when a tile's maximum exceeds M by more than 2^40 (exact: M stays an integer, every
//     rescale factor is a power of two) -- on a separate, non-interleaved path that also serves masked (ragged / causal)
//     tiles and the first tile; that decision needs the tile maximum, 16 v_max3 at the top of the iteration.
// LDS: 2 K slots (tile t+1 is consumed while tile t+2 lands) + a 4-slot ring for the small value tiles, whose tile t is
// still being read by the softmax while tile t+2 arrives.  One workgroup barrier per tile.
template <int... Is, class F>
__device__ __forceinline__ void gsv_static_for(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

#ifndef GSV3_FAST_LIMIT
#define GSV3_FAST_LIMIT 40.f
#endif

template <class T, int NS, int NV, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void gsv3_kernel(GsvArgs a) {
    constexpr int TK = 64;
    constexpr int PLANE = TK * 256;
    constexpr int KSLOT = NS * PLANE;
    constexpr int VSLOT = NV * TK * 4;
    constexpr int VBASE = 2 * KSLOT;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * KSLOT + 4 * VSLOT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int b = blockIdx.y;
    const int qwg = blockIdx.x * 128;
    const int qi = qwg + wave * 32 + (lane & 31);

    // ---- Q fragments (B operand of S^T = K . Q^T); the planes already carry sqrt(log2e / sqrt(C)) on both sides
    i16x8 qf[NS][8];
    {
        const int qr = min(qi, a.Lq - 1);
        const unsigned short* qb = a.qp + ((long)b * a.Lq + qr) * UM_CHANNELS + 8 * half;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[pl][ks] = ld_global_16B(qb + pl * a.q_plane_stride + 16 * ks);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[pl][ks]));
    }

    int ntiles = (a.Lk + TK - 1) / TK;
    if (CAUSAL) ntiles = min(ntiles, (min(qwg + 127, a.Lq - 1) / TK) + 1);
    const int per = (ntiles + a.nsplit - 1) / a.nsplit;
    const int tbeg = blockIdx.z * per, tend = min(ntiles, tbeg + per);

    // ---- staging (as in gsv_kernel): wave w moves rows 16w .. 16w+15 of a tile, 4 rows per DMA instruction
    const unsigned kbase_bytes = (unsigned)((long)b * a.Lk * UM_CHANNELS * 2);
    const int srow = 16 * wave + (lane >> 4);
    const int scp = lane & 15;
    constexpr int NPIECE = 4 * NS;
    auto k_piece = [&](int t, int i, unsigned char* slot) {
        const int j = i / NS, pl = i % NS;
        int row = srow + 4 * j;
        // <Fp16, 2, 2, false> is the one instantiation at the register limit: keep the compiler from hoisting every piece's lane-constant
        // address arithmetic out of the tile loop (16 live registers for ~3 VALU per piece; 20 bytes of scratch per lane in round 4)
        if constexpr (NS == 2 && NV == 2 && !CAUSAL) asm volatile("" : "+v"(row));
        const int key = min(t * TK + row, a.Lk - 1);
        const unsigned off = kbase_bytes + (unsigned)key * (UM_CHANNELS * 2) + ((scp ^ (row & 15)) << 4);
        gsv_dma16(a.kp + pl * a.k_plane_stride, off, slot + pl * PLANE + (16 * wave + 4 * j) * 256);
    };
    const float* vbase = a.v + (long)b * a.v_batch_stride;
    auto v_piece = [&](int t, unsigned char* vslot) {
        if (wave < NV) {
            const int key = min(t * TK + lane, a.Lk - 1);
            gsv_dma4(vbase + wave * a.v_chan_stride, (unsigned)key * 4, vslot + wave * TK * 4);
        }
    };
    auto stage_all = [&](int t, int i) {          // tile with local index i -> K slot i & 1, value slot i & 3
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) k_piece(t, pc, lds + (i & 1) * KSLOT);
        v_piece(t, lds + VBASE + (i & 3) * VSLOT);
    };

    // Running softmax state, per lane (the two half-waves of a query keep independent states, merged at the end):
    // l = sum 2^(score + Ms), acc = sum 2^(score + Ms) v.  Ms is an integer; cinit holds it as the MFMAs' initial accumulator.
    float Ms = 0.f, l = 0.f;
    float acc[NV];
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) acc[ch] = 0.f;
    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;

    int kaddr;
    {
        const int r = lane & 31, x = r & 15;
        kaddr = r * 256 + ((x >> 1) << 5) + ((half ^ (x & 1)) << 4);
    }

    constexpr int MF = (NS == 2) ? 6 : 2;        // MFMAs per k-step (two 32-key sub-tiles)
    constexpr int NM = 8 * MF;                   // MFMAs per tile

    auto frag = [&](const unsigned char* cur, int ridx /* sub * NS + plane */, int ks) {
        return *reinterpret_cast<const i16x8*>(cur + (ridx / NS) * (32 * 256) + (ridx % NS) * PLANE + (kaddr ^ (ks << 5)));
    };
    // A fragments of a k-step: the hi planes of the two 32-key sub-tiles are double-buffered (k-step ks + 1 is read while ks
    // multiplies); the lo planes are consumed by the FIRST two MFMAs of a k-step, so ONE buffer serves them -- the next k-step's lo
    // fragments are read behind MFMAs 2 and 3, still four MFMAs ahead of their use.  24 registers instead of 32 at NS = 2: what the
    // <Fp16, 2, 2, false> instantiation was short of (round 4: 256 VGPRs + 20 bytes of scratch per lane).
    struct Frags {
        i16x8 hi[2][2];
        i16x8 lo[2];
    };
    auto frag_first = [&](const unsigned char* cur, Frags& fr) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            fr.hi[0][sub] = frag(cur, sub * NS, 0);
            if (NS == 2) fr.lo[sub] = frag(cur, sub * NS + 1, 0);
        }
    };
    // the read that rides behind MFMA (ks, j)
    auto frag_next = [&](auto kc, const unsigned char* cur, Frags& fr) {
        constexpr int K = decltype(kc)::value, ks = K / MF, j = K % MF;
        if constexpr (ks + 1 < 8) {
            if constexpr (j < 2) fr.hi[(ks + 1) & 1][j] = frag(cur, j * NS, ks + 1);
            else if constexpr (NS == 2 && j < 4) fr.lo[j - 2] = frag(cur, (j - 2) * NS + 1, ks + 1);
        }
    };
    // one MFMA of the tile: index k = ks * MF + j
    auto mfma_step = [&](auto kc, Frags& fr, f32x16& x0, f32x16& x1) {
        constexpr int K = decltype(kc)::value, ks = K / MF, j = K % MF, bq = ks & 1;
        if constexpr (NS == 2) {
            // j: 0 lo0*qh  1 lo1*qh  2 hi0*ql  3 hi1*ql  4 hi0*qh  5 hi1*qh
            constexpr int sub = j & 1, qpl = (j == 2 || j == 3) ? 1 : 0;
            f32x16& x = sub ? x1 : x0;
            const i16x8 af = (j < 2) ? fr.lo[sub] : fr.hi[bq][sub];
            if constexpr (ks == 0 && j < 2) x = T::mfma(af, qf[qpl][ks], cinit);
            else x = T::mfma(af, qf[qpl][ks], x);
        } else {
            constexpr int sub = j;
            f32x16& x = sub ? x1 : x0;
            if constexpr (ks == 0) x = T::mfma(fr.hi[bq][sub], qf[0][ks], cinit);
            else x = T::mfma(fr.hi[bq][sub], qf[0][ks], x);
        }
    };

    // ---- MFMAs of one tile with nothing interleaved (first tile, and after a slow-path softmax)
    auto mfma_plain = [&](const unsigned char* cur, f32x16& x0, f32x16& x1) {
        Frags fr;
        frag_first(cur, fr);
        gsv_static_for(std::make_integer_sequence<int, NM>{}, [&](auto kc) {
            mfma_step(kc, fr, x0, x1);
            frag_next(kc, cur, fr);
        });
    };

    // ---- slow path: masks, renormalisation of the offset, then the tile's softmax terms (not interleaved with MFMAs)
    auto slow_update = [&](f32x16& y0, f32x16& y1, int t, const float* vt) {
        const int kl = t * TK + 4 * half;
        if (CAUSAL || t * TK + TK > a.Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kl + (r & 3) + 8 * (r >> 2);
                y0[r] = ((key < a.Lk) && (!CAUSAL || key <= qi)) ? y0[r] : UM_NEG_MASK;
                y1[r] = ((key + 32 < a.Lk) && (!CAUSAL || key + 32 <= qi)) ? y1[r] : UM_NEG_MASK;
            }
        }
        float tm = fmaxf(y0[0], y1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, fmaxf(y0[r], y1[r]));
        // the accumulators hold score + Ms.  New offset: the first live tile of a lane fixes it; later only upwards.
        float d = (tm > -1.0e29f && (l == 0.f || tm > 0.f)) ? ceilf(tm) : 0.f;
        const float f = (d > 0.f) ? fast_exp2(-d) : 1.f;           // d < 0 only while the state is still empty (l == acc == 0)
        Ms -= d;
        l *= f;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[ch] *= f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 vv[NV];
#pragma unroll
                for (int ch = 0; ch < NV; ++ch)
                    vv[ch] = *reinterpret_cast<const f32x4*>(vt + ch * TK + sub * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = fast_exp2((sub ? y1[4 * g + i] : y0[4 * g + i]) - d);
                    l += p;
#pragma unroll
                    for (int ch = 0; ch < NV; ++ch) acc[ch] = __builtin_fmaf(p, vv[ch][i], acc[ch]);
                }
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = Ms;
    };

    // value group G (4 scores) of a tile: G >> 2 = sub-tile, G & 3 = register group
    auto vload = [&](const float* vt, int G, f32x4 (&vv)[NV]) {
#pragma unroll
        for (int ch = 0; ch < NV; ++ch)
            vv[ch] = *reinterpret_cast<const f32x4*>(vt + ch * TK + (G >> 2) * 32 + 8 * (G & 3) + 4 * half);
    };
    auto score = [&](auto sc, const f32x16& y0, const f32x16& y1, f32x4 (&vv)[2][NV]) {
        constexpr int S = decltype(sc)::value, sub = S >> 4, r = S & 15, G = S >> 2, i = S & 3;
        const float p = fast_exp2(sub ? y1[r] : y0[r]);
        l += p;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[ch] = __builtin_fmaf(p, vv[G & 1][ch][i], acc[ch]);
    };

    // ---- fast path: softmax terms of tile t (accumulators y) under the MFMAs of tile t+1 (accumulators x).
    // One basic block (STAGING is a compile-time flag: a branch around the DMA pieces would split it, and the softmax
    // arithmetic -- pure, needed only at the end -- would be sunk out of the MFMA shadows); after every MFMA the state is
    // passed through an empty volatile asm, which pins that gap's VALU work between two scheduling fences.
    auto fused = [&](auto staging_c, const unsigned char* cur /* K slot of t+1 */, f32x16& x0, f32x16& x1, const f32x16& y0,
                     const f32x16& y1, const float* vt, int tnext2, int inext2) {
        constexpr bool STAGING = decltype(staging_c)::value;
        Frags fr;
        f32x4 vv[2][NV];
        frag_first(cur, fr);
        vload(vt, 0, vv[0]);
        unsigned char* kdst = lds + (inext2 & 1) * KSLOT;
        unsigned char* vdst = lds + VBASE + (inext2 & 3) * VSLOT;
        gsv_static_for(std::make_integer_sequence<int, NM>{}, [&](auto kc) {
            constexpr int K = decltype(kc)::value, ks = K / MF, j = K % MF;
            mfma_step(kc, fr, x0, x1);
            // fillers in this MFMA's shadow
            frag_next(kc, cur, fr);
            constexpr int S0 = K * 32 / NM, S1 = (K + 1) * 32 / NM;
            gsv_static_for(std::make_integer_sequence<int, S1 - S0>{}, [&](auto dc) {
                constexpr int S = S0 + decltype(dc)::value;
                if constexpr ((S & 3) == 0 && (S >> 2) + 1 < 8) vload(vt, (S >> 2) + 1, vv[((S >> 2) + 1) & 1]);
                score(std::integral_constant<int, S>{}, y0, y1, vv);
            });
            if constexpr (S1 > S0) {
                if constexpr (NV == 2) asm volatile("" : "+v"(l), "+v"(acc[0]), "+v"(acc[NV - 1]));
                else asm volatile("" : "+v"(l), "+v"(acc[0]));
            }
            constexpr int STEP = NM / (NPIECE + 1);
            if constexpr (STAGING && K % STEP == 0 && K / STEP <= NPIECE) {
                if constexpr (K / STEP < NPIECE) k_piece(tnext2, K / STEP, kdst);
                else v_piece(tnext2, vdst);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto fast_update = [&](const f32x16& y0, const f32x16& y1, const float* vt) {      // last tile: nothing left to overlap
        f32x4 vv[2][NV];
        vload(vt, 0, vv[0]);
        gsv_static_for(std::make_integer_sequence<int, 32>{}, [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if constexpr ((S & 3) == 0 && (S >> 2) + 1 < 8) vload(vt, (S >> 2) + 1, vv[((S >> 2) + 1) & 1]);
            score(sc, y0, y1, vv);
        });
    };

    const int n = tend - tbeg;
    f32x16 xa0, xa1, xb0, xb1;
    if (n > 0) {
        stage_all(tbeg, 0);
        if (n > 1) stage_all(tbeg + 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mfma_plain(lds, xa0, xa1);
    }
    // iteration i: y = scores of tile i (complete), x = accumulators of tile i+1
    auto iteration = [&](int i, f32x16& y0, f32x16& y1, f32x16& x0, f32x16& x1) {
        const int t = tbeg + i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of tile i+1 has landed
        __syncthreads();                                            // ... everyone's; and everyone is done reading K slot i & 1
        const float* vt = reinterpret_cast<const float*>(lds + VBASE + (i & 3) * VSLOT);
        const unsigned char* knext = lds + ((i + 1) & 1) * KSLOT;
        const bool staging = i + 2 < n;
        float tm = fmaxf(fmaxf(y0[0], y0[1]), y0[2]);
#pragma unroll
        for (int r = 3; r < 16; ++r) tm = fmaxf(tm, y0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, y1[r]);
        const bool masked = CAUSAL || (t * TK + TK > a.Lk);
        const bool slow = masked || __builtin_amdgcn_ballot_w64(tm > GSV3_FAST_LIMIT || l == 0.f) != 0;
        if (slow) {
            if (staging) stage_all(t + 2, i + 2);
            slow_update(y0, y1, t, vt);
            if (i + 1 < n) mfma_plain(knext, x0, x1);
        } else if (staging) {
            fused(std::true_type{}, knext, x0, x1, y0, y1, vt, t + 2, i + 2);
        } else if (i + 1 < n) {
            fused(std::false_type{}, knext, x0, x1, y0, y1, vt, t + 2, i + 2);
        } else {
            fast_update(y0, y1, vt);
        }
    };
    for (int i = 0; i < n; i += 2) {
        iteration(i, xa0, xa1, xb0, xb1);
        if (i + 1 < n) iteration(i + 1, xb0, xb1, xa0, xa1);
    }

    // ---- merge the two half-waves' partial softmaxes and write (M = Ms: p = 2^(score + M)) ------------------
    // (the query index is recomputed here from an opaque copy of the lane id: kept live across the tile loop it was the one register
    // the <Fp16, 2, 2, false> instantiation spilled)
    int lane_w = lane;
    asm volatile("" : "+v"(lane_w));
    const int qi_w = qwg + wave * 32 + (lane_w & 31);
    const float M = (l == 0.f) ? 3.0e38f : Ms;                     // a lane that met no valid key must not set the common offset
    const float M2 = __shfl_xor(M, 32);
    const float l2 = __shfl_xor(l, 32);
    const float MM = fminf(M, M2);
    const float f1 = fast_exp2(MM - M), f2 = fast_exp2(MM - M2);
    const float lt = l * f1 + l2 * f2;
    if (a.nsplit > 1) {
        float* pr = a.partial + (((long)blockIdx.z * gridDim.y + b) * a.Lq + qi_w) * (2 + NV);
        if (half == 0 && qi_w < a.Lq) {
            pr[0] = MM;
            pr[1] = lt;
        }
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) {
            const float a2 = __shfl_xor(acc[ch], 32);
            if (half == 0 && qi_w < a.Lq) pr[2 + ch] = acc[ch] * f1 + a2 * f2;
        }
        return;
    }
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) {
        const float a2 = __shfl_xor(acc[ch], 32);
        const float at = acc[ch] * f1 + a2 * f2;
        if (half == 0 && qi_w < a.Lq) {
            float r = a.alpha * (at / lt);
            if (a.beta != 0.f) r += a.beta * vbase[ch * a.v_chan_stride + qi_w];
            a.out[((long)b * NV + ch) * a.Lq + qi_w] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gsv4_kernel: ONE wave per SIMD, 64 queries per wave (256 per workgroup), for key counts that are whole 64-key tiles.
//
// What the counters said about gsv3_kernel (profiles/r02_pmc_gsv.txt): its pinned inner block is right (4.5 fillers per
// MFMA), but with two 32-query waves per SIMD the matrix pipe is busy only 70 % of the time a wave is resident -- every
// wave reads the whole K tile from LDS for 48 MFMAs, the tile-maximum chain and the path selection sit unhidden at the top
// of each iteration, the last LDS-DMA piece is issued 8 MFMAs before the wait for it -- and the workgroups are resident
// only 79 % of the launch.  Here
//   * a wave owns 64 queries = two 32-query blocks: one K fragment read from LDS feeds two MFMAs per product, one tile
//     costs 96 MFMAs (exact) against 256 softmax VALU + 48 LDS reads + 9 DMA pieces: 3.6 fillers per MFMA, under the ~5 a
//     lone wave can hide in an MFMA's 32-cycle shadow (MI355X_MICROARCH.md, "one wave per SIMD");
//   * the Q fragments (128 registers in exact mode) live in the accumulator half of the register file: the MFMAs are issued
//     through inline asm with the B operand constrained to AGPRs, the score accumulators stay in VGPRs where the softmax
//     reads them.  hipcc does not see these MFMAs, so the MFMA -> VALU read distance is kept by construction: the softmax
//     of a tile starts after the third MFMA of the NEXT iteration (plus a workgroup barrier) following the last write;
//   * no tile maximum on the fast path: the offset Ms only has to keep exp2() finite, so a tile is simply evaluated and
//     the running sum checked afterwards; if it left (0, 2^90) the lane's state is restored from the pre-tile copy and the
//     tile redone on the renormalising path (the scores are still in their registers), which also adjusts the pending
//     accumulators of the next tile.  Exact as before: Ms stays an integer, every rescale is a power of two;
//   * all LDS-DMA pieces of tile t+2 are issued in the first MFMA gaps of the iteration, a whole tile before their wait.
// Grid: x = (batch, key split) fastest, so that the 24 query tiles sharing one key range run on one XCD (id % 8).
// Round 6: the MFMAs of this kernel are compiler-visible again.  Rounds 2-5 issued them through inline asm (B operand constrained to
// the accumulator file) -- LLVM's hazard recognizer does not look into an asm statement, so NOTHING kept the MAI -> VALU / VALU -> MAI wait
// states except the hand-built distances of the shipped schedule, and a variant whose register allocation moved (the two-level
// accumulation of round 5) got allocator-inserted copies / AGPR spill traffic of accumulator registers inside the hazard window:
// data-dependent wrong results.  Now: the builtin, with Q pinned into AGPRs by an empty asm ("+a") right after its loads -- srcA / srcB
// of a gfx950 MFMA may come from either file, so the value stays where it was pinned -- and `-mllvm -amdgpu-mfma-vgpr-form`
// (build.py, this file only) so that the accumulators are selected in VGPRs although the function uses AGPRs.  Same instruction
// stream as before (checked in the .s: A = VGPR fragment, B = a[..], C/D = v[..]), and the compiler inserts whatever wait states a
// future change needs.
#ifndef UM_GSV4_ASM_MFMA
template <class T> struct GsvMfmaA {
    static __device__ __forceinline__ void acc(f32x16& d, i16x8 a, i16x8 b) { d = T::mfma(a, b, d); }
    static __device__ __forceinline__ void init(f32x16& d, i16x8 a, i16x8 b, const f32x16& c) { d = T::mfma(a, b, c); }
};
#else       // diagnostic builds: rounds 2-5's hand-issued form, for the same-box A/B (profiles/r06_gsv4_builtin_ab.txt)
template <class T> struct GsvMfmaA;
template <> struct GsvMfmaA<Fp16> {
    static __device__ __forceinline__ void acc(f32x16& d, i16x8 a, i16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void init(f32x16& d, i16x8 a, i16x8 b, const f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
};
template <> struct GsvMfmaA<Bf16> {
    static __device__ __forceinline__ void acc(f32x16& d, i16x8 a, i16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void init(f32x16& d, i16x8 a, i16x8 b, const f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
};
#endif

// LDS-DMA with the destination formed in M0 by the same statement (s_add of a wave-uniform base and an immediate).  M0 is DECLARED
// clobbered (round 6; hipcc honours it -- it re-materialises M0 for its own users behind the statement -- and warns that M0 is a
// reserved register: -Wno-inline-asm in build.py); rounds 2-5 relied on "hipcc keeps nothing in M0 in this kernel".
template <int IMM>
__device__ __forceinline__ void gsv4_dma16(const void* base, unsigned byte_off, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(base), "s"(lds_base), "i"(IMM) : "memory", "scc", "m0");
}
__device__ __forceinline__ void gsv4_dma4(const void* base, unsigned byte_off, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(byte_off), "s"(base), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ unsigned gsv4_lds_addr(const unsigned char* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)p);
}

struct GsvAcc4 {
    f32x16 a[2][2];          // [32-key sub-tile][32-query block]
};

#define GSV4_L_LIMIT 1.0e27f        // ~2^90: a running sum outside (0, limit) sends the tile to the renormalising path

template <class T, int NS, int NV>
__global__ __launch_bounds__(256, 1) void gsv4_kernel(GsvArgs a) {
    constexpr int TK = 64;
    constexpr int PLANE = TK * 256;
    constexpr int KSLOT = NS * PLANE;
    constexpr int NKSLOT = 3;
    constexpr int VSLOT = NV * TK * 4;
    constexpr int VBASE = NKSLOT * KSLOT;
    constexpr int DUMP = VBASE + 4 * VSLOT;
    // Round 6, two-level accumulation: a lane's running sums (l, acc) are a chain of 32 fp32 additions per tile -- 3072 terms over
    // config 2's 96 key tiles, which with soft softmaxes (every key contributes) rounds ~8 x more often than the fp32 reference's
    // blocked GEMM + softmax (stage row match_s0 of the conditioned weights: 2.98 x the port's error, profiles/r05_stage_parity_one_scale.txt).
    // Every FLUSH tiles the lane adds its level-1 sums into level-2 sums parked in LDS (8 floats per thread: the offset they are
    // relative to, l, acc -- [field][thread], conflict free, touched by the owning lane only: no synchronisation) and restarts
    // level 1 from zero: chains of 256 + 12 instead of 3072, no extra live register in the pinned blocks.
    constexpr int L2OFF = DUMP + 256;
#ifdef UM_GSV4_ONE_LEVEL        // diagnostic builds: the single chain of rounds 2-5 (same-box A/B of what the flushes cost)
    constexpr int FLUSH = 1 << 30;
#else
    constexpr int FLUSH = 8;
#endif
    __shared__ __attribute__((aligned(16))) unsigned char lds[L2OFF + 8 * 256 * 4];   // 3 K slots, 4 value slots, a dump row, level-2 sums
    using MF = GsvMfmaA<T>;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    // ---- this workgroup's chunk of (batch, query tile, key tile) units; XCD-aware: consecutive chunks (same batch, same
    // keys) go to the workgroups of one XCD
    const int KT = a.Lk / TK;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const long ubeg = (long)wg * a.chunk;
    const long uend = min((long)a.nbatch * a.qtiles * KT, ubeg + a.chunk);
    constexpr int NPIECE = 4 * NS;
    unsigned koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * wave + (lane >> 4) + 4 * j;
        koff[j] = (unsigned)(row * 256 + (((lane & 15) ^ (row & 15)) << 4));
    }
    const long kplane_bytes = a.k_plane_stride * 2;
    const unsigned lds0 = gsv4_lds_addr(lds);
    const unsigned wrow = lds0 + 16 * wave * 256;                       // this wave's rows of K slot 0
    int kaddr;
    {
        const int r = lane & 31, x = r & 15;
        kaddr = r * 256 + ((x >> 1) << 5) + ((half ^ (x & 1)) << 4);
    }

  for (long u = ubeg; u < uend;) {                                      // one segment = this chunk's part of one query tile
    const int qtile = (int)(u / KT);
    const int tbeg = (int)(u - (long)qtile * KT);
    const int n = (int)min((long)(KT - tbeg), uend - u);
    const int b = qtile / a.qtiles;
    const int slot = wg - (int)(((long)qtile * KT) / a.chunk);         // this segment's partial slot
    const int qw = (qtile - b * a.qtiles) * 256 + wave * 64 + (lane & 31);   // query of block 0; block 1: + 32
    u += n;
    __builtin_amdgcn_s_barrier();                                       // the previous segment is out of the LDS

    // ---- Q fragments (B operands), both planes, both query blocks: AGPR residents
    i16x8 qf[2][NS][8];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = min(qw + 32 * qb, a.Lq - 1);
        const unsigned short* qp = a.qp + ((long)b * a.Lq + qr) * UM_CHANNELS + 8 * half;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[qb][pl][ks] = ld_global_16B(qp + pl * a.q_plane_stride + 16 * ks);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[qb][pl][ks]));

    // ---- staging: wave w moves rows 16w .. 16w+15 of a tile, 4 rows (64 lanes x 16 B) per DMA instruction; the per-lane
    // source offsets do not depend on the tile (whole tiles only), the tile enters through the scalar base
    const unsigned char* kbytes = reinterpret_cast<const unsigned char*>(a.kp) + (long)b * a.Lk * (UM_CHANNELS * 2);
    // piece i of the tile at tile_bytes -> the K slot whose LDS address (+ this wave's row offset) is kslot_w
    auto k_piece = [&](auto ic, const unsigned char* tile_bytes, unsigned kslot_w) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, j = i / NS, pl = i % NS;
        gsv4_dma16<pl * PLANE + 4 * j * 256>(tile_bytes + pl * kplane_bytes, koff[j], kslot_w);
    };
    const float* vbase = a.v + (long)b * a.v_batch_stride;
    // waves 0 .. NV-1 stage one value channel each; the others issue the same instruction into a dump row (no branch:
    // a branch would split the pinned block)
    const float* vsrc = vbase + ((wave < NV) ? wave : 0) * a.v_chan_stride;
    const unsigned vdst0 = (wave < NV) ? lds0 + VBASE + wave * TK * 4 : lds0 + DUMP;
    const unsigned vstep = (wave < NV) ? VSLOT : 0;
    auto v_piece = [&](int t, int i) __attribute__((always_inline)) { gsv4_dma4(vsrc, (unsigned)(t * TK + lane) * 4, vdst0 + (i & 3) * vstep); };
    auto stage_all = [&](int t, int i, int kslot) __attribute__((always_inline)) {
        const unsigned char* tb = kbytes + (long)t * (TK * 256);
        const unsigned kw = wrow + kslot * KSLOT;
        gsv_static_for(std::make_integer_sequence<int, NPIECE>{}, [&](auto ic) __attribute__((always_inline)) { k_piece(ic, tb, kw); });
        v_piece(t, i);
    };

    // ---- running softmax state per (lane, query block): l = sum 2^(score + Ms), acc = sum 2^(score + Ms) v
    float Ms[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};
    float acc[2][NV];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[qb][ch] = 0.f;
    f32x16 cinit[2];          // Ms as the MFMAs' initial accumulator (srcC; must sit in the register half of the destination)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[0][r] = cinit[1][r] = 0.f;

    constexpr int NP = (NS == 2) ? 3 : 1;        // products per (k-step, sub-tile, query block)
    constexpr int MFK = 4 * NP;                  // MFMAs per k-step
    constexpr int NM = 8 * MFK;                  // MFMAs per tile
    constexpr int HEAD = 4;                      // leading gaps of a block without accumulator reads (see header)
    constexpr int MID = NM / 2;                  // the block's workgroup barrier sits in front of MFMA number MID

    // fragment of the K slot whose per-lane base is kb = kaddr + slot * KSLOT
    auto frag = [&](int kb, int ridx /* sub * NS + plane */, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const i16x8*>(lds + (ridx / NS) * (32 * 256) + (ridx % NS) * PLANE + (kb ^ (ks << 5)));
    };
    // MFMA number K of a tile: k-step K / MFK; inside it product-major, then (sub-tile, query block): four different
    // accumulators in turn, so that no MFMA waits for its predecessor's result
    auto mfma_step = [&](auto kc, i16x8 (&fr)[2][2 * NS], GsvAcc4& x) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value, ks = K / MFK, j = K % MFK, prod = j / 4, sub = (j >> 1) & 1, qb = j & 1;
        constexpr int bq = (NS == 2) ? 0 : (ks & 1);
        constexpr int kpl = (NS == 2 && prod == 0) ? 1 : 0;      // lo_k * hi_q, hi_k * lo_q, hi_k * hi_q
        constexpr int qpl = (NS == 2 && prod == 1) ? 1 : 0;
        if constexpr (ks == 0 && prod == 0) MF::init(x.a[sub][qb], fr[bq][sub * NS + kpl], qf[qb][qpl][ks], cinit[qb]);
        else MF::acc(x.a[sub][qb], fr[bq][sub * NS + kpl], qf[qb][qpl][ks]);
    };
    // K fragments of the next k-step -- after the last k-step: the FIRST k-step of the next tile (slot base kb_next), so
    // that a block starts with its operands in registers.  Exact mode: ONE buffer, every fragment re-read in place right
    // after its last MFMA of the k-step (lo planes feed product 0 only: free after gaps 1 / 3; hi planes feed products 1
    // and 2: free after gaps 9 / 11) -- 16 registers instead of 32, and still 4+ MFMAs between a read and its first use.
    // Fast mode: two buffers.
    auto frag_refill = [&](auto kc, int kb, int kb_next, i16x8 (&fr)[2][2 * NS]) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value, ks = K / MFK, j = K % MFK, nk = (ks + 1) & 7;
        const int src = (ks + 1 < 8) ? kb : kb_next;
        if constexpr (NS == 2) {
            if constexpr (j == 1) fr[0][1] = frag(src, 1, nk);
            if constexpr (j == 3) fr[0][3] = frag(src, 3, nk);
            if constexpr (j == 9) fr[0][0] = frag(src, 0, nk);
            if constexpr (j == 11) fr[0][2] = frag(src, 2, nk);
        } else if constexpr (j < 2) {
            fr[nk & 1][j] = frag(src, j, nk);
        }
    };
    auto frag_first = [&](int kb, i16x8 (&fr)[2][2 * NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int ri = 0; ri < 2 * NS; ++ri) fr[0][ri] = frag(kb, ri, 0);
    };
    auto mfma_plain = [&](int kb, int kb_next, i16x8 (&fr)[2][2 * NS], GsvAcc4& x) __attribute__((always_inline)) {
        gsv_static_for(std::make_integer_sequence<int, NM>{}, [&](auto kc) __attribute__((always_inline)) {
            mfma_step(kc, fr, x);
            frag_refill(kc, kb, kb_next, fr);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- level-2 sums of this thread: Ms2[2] (3e38: empty), l2[2], acc2[2][NV]
    float* const lvl2 = reinterpret_cast<float*>(lds + L2OFF) + tid;
#pragma unroll
    for (int f = 0; f < 8; ++f) lvl2[f * 256] = f < 2 ? 3.0e38f : 0.f;
    // level 1 -> level 2.  The offset only ever moves towards larger maxima after the first tile (Ms decreases), so the factor that
    // brings the parked sums to the current offset is a power of two <= 1 (exact); an empty level 2 (offset 3e38) gets factor 0.
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float f = fast_exp2(Ms[qb] - lvl2[qb * 256]);
            lvl2[qb * 256] = Ms[qb];
            lvl2[(2 + qb) * 256] = __builtin_fmaf(lvl2[(2 + qb) * 256], f, l[qb]);
            l[qb] = 0.f;
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) {
                lvl2[(4 + 2 * qb + ch) * 256] = __builtin_fmaf(lvl2[(4 + 2 * qb + ch) * 256], f, acc[qb][ch]);
                acc[qb][ch] = 0.f;
            }
        }
    };

    // ---- renormalising path for one tile (first tile of a lane, or after the fast path left the safe range)
    auto slow_update = [&](auto first_c, const GsvAcc4& y, const float* vt, float (&dd)[2]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float tm = fmaxf(y.a[0][qb][0], y.a[1][qb][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) tm = fmaxf(tm, fmaxf(y.a[0][qb][r], y.a[1][qb][r]));
            // the accumulators hold score + Ms.  New offset: the first tile of a segment fixes it; later only upwards (level 1 may
            // be empty after a flush: that is not "no state")
            const float d = (FIRST || tm > 0.f) ? ceilf(tm) : 0.f;
            const float f = (d > 0.f) ? fast_exp2(-d) : 1.f;       // d < 0 only while the state is still empty
            Ms[qb] -= d;
            l[qb] *= f;
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) acc[qb][ch] *= f;
            dd[qb] = d;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 vv[NV];
#pragma unroll
                for (int ch = 0; ch < NV; ++ch)
                    vv[ch] = *reinterpret_cast<const f32x4*>(vt + ch * TK + sub * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float p = fast_exp2(y.a[sub][qb][4 * g + i] - dd[qb]);
                        l[qb] += p;
#pragma unroll
                        for (int ch = 0; ch < NV; ++ch) acc[qb][ch] = __builtin_fmaf(p, vv[ch][i], acc[qb][ch]);
                    }
            }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cinit[qb][r] = Ms[qb];
            asm volatile("s_nop 1" : "+v"(cinit[qb]));                    // VALU write -> MFMA srcC wait states, by hand
        }
    };
    // accumulators of the NEXT tile were started from the old offset: move them to the new one
    auto shift_pending = [&](GsvAcc4& x, const float (&dd)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) x.a[sub][qb][r] -= dd[qb];
    };

    // value group G (4 keys): G >> 2 = sub-tile, G & 3 = register group; shared by both query blocks
    auto vload = [&](const float* vt, int G, f32x4 (&vv)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int ch = 0; ch < NV; ++ch)
            vv[ch] = *reinterpret_cast<const f32x4*>(vt + ch * TK + (G >> 2) * 32 + 8 * (G & 3) + 4 * half);
    };
    // score S of a tile: S >> 3 = value group, (S >> 2) & 1 = query block, S & 3 = key in the group
    // the exponential is taken one score ahead of its use (a transcendental's result needs a wait state before its consumer)
    float pn = 0.f;
    auto score_exp = [&](auto sc, const GsvAcc4& y) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value, G = S >> 3, qb = (S >> 2) & 1, i = S & 3, sub = G >> 2, r = 4 * (G & 3) + i;
        pn = fast_exp2(y.a[sub][qb][r]);
    };
    auto score = [&](auto sc, const GsvAcc4& y, f32x4 (&vv)[2][NV]) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value, G = S >> 3, qb = (S >> 2) & 1, i = S & 3;
        const float p = pn;
        if constexpr (S + 1 < 64) {
            score_exp(std::integral_constant<int, S + 1>{}, y);
            __builtin_amdgcn_sched_barrier(0);          // the exponential first: three instructions between it and its consumer
        }
        l[qb] += p;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[qb][ch] = __builtin_fmaf(p, vv[G & 1][ch][i], acc[qb][ch]);
    };
    auto pin_state = [&]() __attribute__((always_inline)) {
        if constexpr (NV == 2)
            asm volatile("" : "+v"(l[0]), "+v"(l[1]), "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        else
            asm volatile("" : "+v"(l[0]), "+v"(l[1]), "+v"(acc[0][0]), "+v"(acc[1][0]));
    };

    // ---- the schedule of one block: which fillers sit in the shadow of MFMA number K.
    // Exact mode, per k-step of 12 MFMAs: gaps 1, 3, 9, 11 carry the fragment refills (+ the value loads at gap 3, + one
    // DMA piece at gaps 1 / 9 / 11 of k-steps 4 - 6, after the barrier), the other eight gaps one score each (exp, add, two
    // fma); the first block gaps read no accumulator; the two scores this leaves over go to gaps 9 / 11 of the last k-step.
    // Fast mode (4 MFMAs per k-step): scores spread evenly, DMA in the odd gaps after the barrier.
    struct Sched {
        static constexpr bool refill_gap(int j) { return j == 1 || j == 3 || j == 9 || j == 11; }
        static constexpr int scores_in(int K) {          // scores issued in gap K
            if (K < HEAD) return 0;
            if (NS == 2) {
                const int ks = K / MFK, j = K % MFK;
                if (!refill_gap(j)) return 1;
                return (ks == 7 && (j == 9 || j == 11)) ? 1 : 0;
            }
            return (K - HEAD + 1) * 64 / (NM - HEAD) - (K - HEAD) * 64 / (NM - HEAD);
        }
        static constexpr int scores_before(int K) {
            int c = 0;
            for (int k = 0; k < K; ++k) c += scores_in(k);
            return c;
        }
        static constexpr int dma_piece(int K) {          // -1: none; 0 .. NPIECE-1: K piece; NPIECE: the value piece
            if (K <= MID) return -1;
            if (NS == 2) {
                const int ks = K / MFK, j = K % MFK;
                if (ks < 4 || ks > 6) return -1;
                const int slot = (j == 1) ? 0 : (j == 9) ? 1 : (j == 11) ? 2 : -1;
                return slot < 0 ? -1 : (ks - 4) * 3 + slot;
            }
            const int d = K - MID;
            return ((d & 1) && d / 2 <= NPIECE) ? d / 2 : -1;
        }
    };
    static_assert(Sched::scores_before(NM) == 64, "every score of a tile must be scheduled");

    // ---- fast path: softmax terms of tile i (accumulators y, read only) in the shadows of the MFMAs of tile i+1 (x).
    // One basic block; every gap's work is pinned between two scheduling fences.  In front of MFMA number MID: this wave's
    // DMA of tile i+2 has landed (vmcnt) and a workgroup barrier -- after it tile i+2 is visible to everybody AND everybody
    // has finished block i-1, so the K slot of tile i (= slot of tile i+3) and the value slot of tile i-1 may be refilled.
    auto fused = [&](auto staging_c, int kb, int kb_next, i16x8 (&fr)[2][2 * NS], GsvAcc4& x, const GsvAcc4& y, const float* vt,
                     int t3, int i3, int kslot3) __attribute__((always_inline)) {
        constexpr bool STAGING = decltype(staging_c)::value;
        f32x4 vv[2][NV];
        vload(vt, 0, vv[0]);
        const unsigned char* tb = kbytes + (long)t3 * (TK * 256);
        const unsigned kw = wrow + kslot3 * KSLOT;
        __builtin_amdgcn_sched_barrier(0);
        gsv_static_for(std::make_integer_sequence<int, NM>{}, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value, ks = K / MFK, j = K % MFK;
            if constexpr (K == MID) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_step(kc, fr, x);
            frag_refill(kc, kb, kb_next, fr);
            if constexpr (STAGING && Sched::dma_piece(K) >= 0) {
                constexpr int pc = Sched::dma_piece(K);
                if constexpr (pc < NPIECE) k_piece(std::integral_constant<int, pc>{}, tb, kw);
                else v_piece(t3, i3);
            }
            if constexpr (NS == 2 && j == 3 && ks + 1 < 8) vload(vt, ks + 1, vv[(ks + 1) & 1]);
            constexpr int S0 = Sched::scores_before(K), NSC = Sched::scores_in(K);
            if constexpr (NSC > 0) {
                if constexpr (S0 == 0) score_exp(std::integral_constant<int, 0>{}, y);
                gsv_static_for(std::make_integer_sequence<int, NSC>{}, [&](auto dc) __attribute__((always_inline)) {
                    constexpr int S = S0 + decltype(dc)::value;
                    if constexpr (NS == 1 && (S & 7) == 0 && (S >> 3) + 1 < 8) vload(vt, (S >> 3) + 1, vv[((S >> 3) + 1) & 1]);
                    score(std::integral_constant<int, S>{}, y, vv);
                });
                pin_state();
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto tail_update = [&](const GsvAcc4& y, const float* vt) __attribute__((always_inline)) {      // last tile: nothing left to overlap
        f32x4 vv[2][NV];
        vload(vt, 0, vv[0]);
        score_exp(std::integral_constant<int, 0>{}, y);
        gsv_static_for(std::make_integer_sequence<int, 64>{}, [&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            if constexpr ((S & 7) == 0 && (S >> 3) + 1 < 8) vload(vt, (S >> 3) + 1, vv[((S >> 3) + 1) & 1]);
            score(sc, y, vv);
        });
    };

    // ---- pipeline.  Tile with local index i: K slot i % 3, value slot i & 3, accumulators xa (i even) / xb (i odd).
    // Block i (i >= 1): softmax of tile i, MFMAs of tile i+1, DMA of tile i+3, first fragments of tile i+2.
    GsvAcc4 xa, xb;
    i16x8 fr[2][2 * NS];
    {
        stage_all(tbeg, 0, 0);
        if (n > 1) stage_all(tbeg + 1, 1, 1);
        if (n > 2) stage_all(tbeg + 2, 2, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // tile 0 alone, then block 0: the first tile fixes the offset (not overlapped)
        frag_first(kaddr, fr);
        mfma_plain(kaddr, kaddr + KSLOT, fr, xa);
        __builtin_amdgcn_s_barrier();                                   // K slot 0 is free
        if (n > 3) stage_all(tbeg + 3, 3, 0);
        if (n > 1) mfma_plain(kaddr + KSLOT, kaddr + 2 * KSLOT, fr, xb);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float dd[2];
        slow_update(std::true_type{}, xa, reinterpret_cast<const float*>(lds + VBASE), dd);
        if (n > 1) shift_pending(xb, dd);
    }
    int s1 = 2, s2 = 0, s3 = 1;                                         // K slots of tiles i+1, i+2, i+3 at i = 1
    auto iteration = [&](int i, GsvAcc4& y, GsvAcc4& x) __attribute__((always_inline)) {
        const float* vt = reinterpret_cast<const float*>(lds + VBASE + (i & 3) * VSLOT);
        const int kb = kaddr + s1 * KSLOT, kb_next = kaddr + s2 * KSLOT;
        const float l0 = l[0], l1 = l[1];
        float a0[2][NV];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) a0[qb][ch] = acc[qb][ch];
        if (i + 3 < n) fused(std::true_type{}, kb, kb_next, fr, x, y, vt, tbeg + i + 3, i + 3, s3);
        else if (i + 1 < n) fused(std::false_type{}, kb, kb_next, fr, x, y, vt, tbeg + i + 3, i + 3, s3);
        else {
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // last asm MFMAs of the previous block -> VALU
            __builtin_amdgcn_sched_barrier(0);
            tail_update(y, vt);
        }
        // (an all-underflow tile after a flush leaves level 1 at zero: legitimate, the mass is in level 2)
        const bool bad = !(l[0] < GSV4_L_LIMIT) || !(l[1] < GSV4_L_LIMIT);
        if (__builtin_amdgcn_ballot_w64(bad) != 0) {
            float dd[2];
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // the pending MFMAs' results (asm) -> VALU
            __builtin_amdgcn_sched_barrier(0);
            l[0] = l0;
            l[1] = l1;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int ch = 0; ch < NV; ++ch) acc[qb][ch] = a0[qb][ch];
            slow_update(std::false_type{}, y, vt, dd);
            if (i + 1 < n) shift_pending(x, dd);
        }
        if ((i & (FLUSH - 1)) == 0) flush();
        const int s0 = s1;
        s1 = s2;
        s2 = s3;
        s3 = s0;
    };
    for (int i = 1; i < n; i += 2) {
        iteration(i, xb, xa);
        if (i + 1 < n) iteration(i + 1, xa, xb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flush();                                                            // the segment's totals are the level-2 sums
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        l[qb] = lvl2[(2 + qb) * 256];
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[qb][ch] = lvl2[(4 + 2 * qb + ch) * 256];
    }

    // ---- merge the two half-waves' partial softmaxes and write (M = Ms: p = 2^(score + M))
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = qw + 32 * qb;
        const float M = (l[qb] == 0.f) ? 3.0e38f : Ms[qb];
        const float M2 = __shfl_xor(M, 32);
        const float l2 = __shfl_xor(l[qb], 32);
        const float MM = fminf(M, M2);
        const float f1 = fast_exp2(MM - M), f2 = fast_exp2(MM - M2);
        const float lt = l[qb] * f1 + l2 * f2;
        if (a.partial) {
            float* pr = a.partial + (((long)slot * a.nbatch + b) * a.Lq + qi) * (2 + NV);
            if (half == 0 && qi < a.Lq) {
                pr[0] = MM;
                pr[1] = lt;
            }
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) {
                const float a2 = __shfl_xor(acc[qb][ch], 32);
                if (half == 0 && qi < a.Lq) pr[2 + ch] = acc[qb][ch] * f1 + a2 * f2;
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < NV; ++ch) {
                const float a2 = __shfl_xor(acc[qb][ch], 32);
                const float at = acc[qb][ch] * f1 + a2 * f2;
                if (half == 0 && qi < a.Lq) {
                    float r = a.alpha * (at / lt);
                    if (a.beta != 0.f) r += a.beta * vbase[ch * a.v_chan_stride + qi];
                    a.out[((long)b * NV + ch) * a.Lq + qi] = r;
                }
            }
        }
    }
  }   // segments
}

// merge a query's segments (gsv4_kernel): its tile's units [qtile * KT, (qtile + 1) * KT) lie in chunks w0 .. w1 = slots 0 .. w1 - w0
template <int NV>
__global__ void gsv4_combine_kernel(GsvArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.nbatch * a.Lq;
    if (i >= total) return;
    const int b = (int)(i / a.Lq), q = (int)(i - (long)b * a.Lq);
    const int KT = a.Lk / 64;
    const long qtile = (long)b * a.qtiles + q / 256;
    const int nseg = (int)(((qtile + 1) * KT - 1) / a.chunk - (qtile * KT) / a.chunk) + 1;
    float MM = 3.0e38f;
    for (int sp = 0; sp < nseg; ++sp) MM = fminf(MM, a.partial[((long)sp * total + i) * (2 + NV)]);
    float lt = 0.f, at[NV];
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) at[ch] = 0.f;
    for (int sp = 0; sp < nseg; ++sp) {
        const float* pr = a.partial + ((long)sp * total + i) * (2 + NV);
        const float f = fast_exp2(MM - pr[0]);
        lt += pr[1] * f;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) at[ch] += pr[2 + ch] * f;
    }
    const float* vbase = a.v + (long)b * a.v_batch_stride;
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) {
        float r = a.alpha * (at[ch] / lt);
        if (a.beta != 0.f) r += a.beta * vbase[ch * a.v_chan_stride + q];
        a.out[((long)b * NV + ch) * a.Lq + q] = r;
    }
}

// merge the per-split partial softmaxes: out = alpha * sum_s(acc_s 2^(M-M_s)) / sum_s(l_s 2^(M-M_s)) + beta * v[query]
template <int NV>
__global__ void gsv_combine_kernel(GsvArgs a, int nbatch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)nbatch * a.Lq;
    if (i >= total) return;
    const int b = (int)(i / a.Lq), q = (int)(i - (long)b * a.Lq);
    float MM = 3.0e38f;
    for (int sp = 0; sp < a.nsplit; ++sp) MM = fminf(MM, a.partial[((long)sp * total + i) * (2 + NV)]);
    float lt = 0.f, at[NV];
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) at[ch] = 0.f;
    for (int sp = 0; sp < a.nsplit; ++sp) {
        const float* pr = a.partial + ((long)sp * total + i) * (2 + NV);
        const float f = fast_exp2(MM - pr[0]);
        lt += pr[1] * f;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) at[ch] += pr[2 + ch] * f;
    }
    const float* vbase = a.v + (long)b * a.v_batch_stride;
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) {
        float r = a.alpha * (at[ch] / lt);
        if (a.beta != 0.f) r += a.beta * vbase[ch * a.v_chan_stride + q];
        a.out[((long)b * NV + ch) * a.Lq + q] = r;
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

#define GSV_MAX_SPLIT 8
// Key-range split so that the launch is a whole number of balanced rounds: 256 CUs x 2 resident workgroups.
static int gsv_choose_split(int qtiles, int nbatch, int ktiles) {
    const long wgs = (long)qtiles * nbatch;
    int best = 1;
    if (wgs >= 1536 || ktiles < 16) return 1;
    for (int sp = 1; sp <= GSV_MAX_SPLIT && ktiles / sp >= 8; ++sp) {
        best = sp;
        if (wgs * sp >= 1536) break;
    }
    return best;
}

// Kernel choice: gsv4_kernel (one wave per SIMD, 64 queries per wave) wherever the key count is whole 64-key tiles, the launch
// is not causal and fills the chip, else gsv3_kernel (two waves per SIMD) -- a pure function of the call's arguments.
// (Diagnostic builds only: UM_GSV_V3=1 forces gsv3 for same-box A/B runs.)  Operand planes carry sqrt(log2(e) / sqrt(C)) on both sides.
static int gsv_version() {
    static const int v = [] {
        const char* e3 = um_debug_env("UM_GSV_V3");
        return (e3 && *e3 == '1') ? 3 : 4;
    }();
    return v;
}
static float gsv_plane_scale(float scale_log2) { return sqrtf(scale_log2); }

// gsv4: one workgroup per CU, all of them resident at once ("stream-K"): the (batch, 256-query tile, 64-key tile) units are cut
// into equal chunks.  A chunk is at least 8 key tiles and a query tile is cut into at most GSV_MAX_SPLIT segments.
static int gsv_num_cus() { return um_num_cus(); }      // per device (common.h)

template <int NV, bool CAUSAL>
static hipError_t launch_gsv(GsvArgs a, int nbatch, int mode, float* partial, hipStream_t stream) {
    const int ver = gsv_version();
    // gsv4 wants every CU busy with a chunk of >= 8 key tiles of a 256-query tile; smaller launches (config 1: one sample of
    // 2240 tokens = 315 units) are better served by gsv3's 128-query workgroups with split keys (0.021 against 0.032 ms)
    const long units4 = (long)nbatch * ((a.Lq + 255) / 256) * (a.Lk / 64);
    if (!CAUSAL && ver == 4 && a.Lk % 64 == 0 && a.Lk >= 512 && units4 >= 8L * gsv_num_cus()) {
        const int KT = a.Lk / 64;
        a.nbatch = nbatch;
        a.qtiles = (a.Lq + 255) / 256;
        const long units = (long)nbatch * a.qtiles * KT;
        long chunk = (units + gsv_num_cus() - 1) / gsv_num_cus();
        const long min_chunk = (KT + GSV_MAX_SPLIT - 3) / (GSV_MAX_SPLIT - 2);     // at most GSV_MAX_SPLIT segments per query tile
        if (chunk < min_chunk) chunk = min_chunk;
        if (chunk < 8) chunk = 8;
        static const bool whole = [] { const char* e = um_debug_env("UM_GSV4_WHOLE_TILES"); return e && *e == '1'; }();   // A/B timing
        if (!partial || (whole && chunk > KT)) chunk = ((chunk + KT - 1) / KT) * KT;   // no partial buffer: whole query tiles per workgroup
        a.chunk = (int)chunk;
        const bool direct = (chunk % KT) == 0;
        a.partial = direct ? nullptr : partial;
        a.nsplit = 1;
        const unsigned wgs = (unsigned)((units + chunk - 1) / chunk);
        {
            ScopedKernelTimer timer(UM_K_GLOBAL_SOFTMAX, stream);
            um_census_hit(UM_V_GSV4);
            if (mode == 0)
                hipLaunchKernelGGL((gsv4_kernel<Fp16, 2, NV>), dim3(wgs), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL((gsv4_kernel<Bf16, 1, NV>), dim3(wgs), dim3(256), 0, stream, a);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && !direct) {
            const long total = (long)nbatch * a.Lq;
            hipLaunchKernelGGL((gsv4_combine_kernel<NV>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
            e = hipGetLastError();
        }
        return e;
    }
    const int qtiles = (a.Lq + 127) / 128;
    a.nsplit = (CAUSAL || !partial) ? 1 : gsv_choose_split(qtiles, nbatch, (a.Lk + 63) / 64);
    a.partial = partial;
    dim3 grid(qtiles, nbatch, a.nsplit), block(256);
    {
        ScopedKernelTimer timer(UM_K_GLOBAL_SOFTMAX, stream);
        um_census_hit(UM_V_GSV3);
        if (mode == 0) {
            hipLaunchKernelGGL((gsv3_kernel<Fp16, 2, NV, CAUSAL>), grid, block, 0, stream, a);
        } else {
            hipLaunchKernelGGL((gsv3_kernel<Bf16, 1, NV, CAUSAL>), grid, block, 0, stream, a);
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && a.nsplit > 1) {
        const long total = (long)nbatch * a.Lq;
        hipLaunchKernelGGL((gsv_combine_kernel<NV>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, nbatch);
        e = hipGetLastError();
    }
    return e;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t um_global_corr_workspace_bytes(int batch, int tokens, int channels, int mode) {
    if (batch <= 0 || tokens <= 0 || channels != UM_CHANNELS || (mode != 0 && mode != 1)) return 0;
    return 2 * align256(planes_bytes((long)batch * tokens, mode)) + align256((size_t)tokens * 2 * sizeof(float)) +
           align256((size_t)GSV_MAX_SPLIT * batch * tokens * 4 * sizeof(float));
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
    if ((long)batch * h * w * UM_CHANNELS * 2 >= (1L << 32)) {
        um_set_error("batch*tokens = %ld exceeds the 32-bit plane addressing of this kernel", (long)batch * h * w);
        return -4;
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
    float* partial = (float*)(ws + 2 * pb + align256((size_t)L * 2 * sizeof(float)));
    hipError_t e;
    const float pscale = gsv_plane_scale(UM_LOG2E / sqrtf((float)channels));
    if ((e = launch_split_planes(f0, p0, batch * L, pscale, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(f1, p1, batch * L, pscale, mode, stream)) != hipSuccess) return (int)e;
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
    if ((e = launch_gsv<2, false>(a, batch, mode, partial, stream)) != hipSuccess) return (int)e;
    if (bidir) {   // backward flow: softmax over the other axis of the same correlation (matching.py:23-27)
        a.qp = p1;
        a.kp = p0;
        a.out = flow + (long)batch * 2 * L;
        if ((e = launch_gsv<2, false>(a, batch, mode, partial, stream)) != hipSuccess) return (int)e;
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
    const float pscale = gsv_plane_scale(UM_LOG2E / sqrtf((float)channels));
    if ((e = launch_split_planes(f0, p0, batch * L, pscale, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(f1, p1, batch * L, pscale, mode, stream)) != hipSuccess) return (int)e;
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
    if ((e = launch_gsv<1, true>(a, batch * h, mode, nullptr, stream)) != hipSuccess) return (int)e;
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
    float* partial = (float*)(ws + 2 * pb + align256((size_t)L * 2 * sizeof(float)));
    hipError_t e;
    const float pscale = gsv_plane_scale(UM_LOG2E / sqrtf((float)channels));
    if ((e = launch_split_planes(q, pq, batch * L, pscale, mode, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(k, pk, batch * L, pscale, mode, stream)) != hipSuccess) return (int)e;
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
        e = launch_gsv<2, false>(a, batch, mode, partial, stream);
    else
        e = launch_gsv<1, false>(a, batch, mode, partial, stream);
    return (int)e;
}

// The factor the operand planes of this file's kernels carry on BOTH sides (q and k): sqrt(log2(e) / sqrt(C)), so that the MFMA
// result is the softmax logit in log2 units.
extern "C" float um_global_corr_plane_scale(int channels) {
    return channels > 0 ? gsv_plane_scale(UM_LOG2E / sqrtf((float)channels)) : 0.f;
}

// um_prop_global_attn on operands that are ALREADY planes carrying um_global_corr_plane_scale() -- the output of
// um_linear_bias_fwd(out_planes) -- so the propagation layer's q / k never exist in fp32 (attention.py:196-213).
extern "C" int um_prop_global_attn_planes(const void* q_planes, const void* k_planes, const float* value, float* out, int batch,
                                          int h, int w, int channels, int value_channels, int mode, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long L = (long)h * w;
    const size_t need = um_global_corr_workspace_bytes(batch, (int)L, channels, mode);
    if (int e = check_common(q_planes, k_planes, out, batch, h, w, channels, mode, workspace, workspace_bytes, need)) return e;
    if (!value || (value_channels != 1 && value_channels != 2)) {
        um_set_error("value_channels=%d: the reference propagates flow (2) or disparity/depth (1)", value_channels);
        return -1;
    }
    unsigned char* ws = (unsigned char*)workspace;
    const size_t pb = align256(planes_bytes(batch * L, mode));
    float* partial = (float*)(ws + 2 * pb + align256((size_t)L * 2 * sizeof(float)));
    GsvArgs a;
    a.qp = (const unsigned short*)q_planes;
    a.kp = (const unsigned short*)k_planes;
    a.q_plane_stride = a.k_plane_stride = batch * L * UM_CHANNELS;
    a.v = value;
    a.v_batch_stride = value_channels * L;
    a.v_chan_stride = L;
    a.out = out;
    a.Lq = a.Lk = (int)L;
    a.scale_log2 = UM_LOG2E / sqrtf((float)channels);
    a.alpha = 1.f;
    a.beta = 0.f;
    hipError_t e = (value_channels == 2) ? launch_gsv<2, false>(a, batch, mode, partial, stream)
                                         : launch_gsv<1, false>(a, batch, mode, partial, stream);
    return (int)e;
}
