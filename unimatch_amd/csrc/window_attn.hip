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
// Decomposition: workgroup = 4 waves = 128 query tokens of one window; wave = 32 queries; K/V tiles of 32
// window tokens stream into a 2-deep LDS ring by LDS-DMA (global_load_lds, no staging registers) while the
// previous tile is consumed; one barrier per tile.  Two workgroups share a CU (65 KB LDS each in exact mode).  Everything is computed transposed so that the MFMA
// column index n is the query: lane l owns query (l & 31) in BOTH products,
//       S^T = K . Q^T      (A = K rows from LDS via ds_read_b128,            B = Q^T held in registers)
//       O^T = V^T . P^T    (A = V^T via ds_read_b64_tr_b16 transpose reads,  B = P^T built in registers)
// which makes the softmax a per-lane affair (one exchange with lane^32 per tile for the row max) and lets
// the probabilities go from the S^T accumulators to the P^T operand with v_permlane32_swap only.
//
// Precision (mode): exact = fp16 hi+lo operands, 3 MFMA products per contraction (lo*hi, hi*lo, hi*hi),
// probabilities scaled by 2^14 before the fp16 split so that small p keep 22 bits; fast = bf16, 1 product.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "planes.h"
#include "timing.h"

// -DUM_TRACE: wave 0 / lane 0 of every 37th workgroup stamps s_memtime at section boundaries of its first 24 tiles
// into the buffer given to um_debug_set_trace() (diagnostics only; tools/trace_attn.py).
#ifdef UM_TRACE
__device__ unsigned long long* g_um_trace = nullptr;
#define UM_STAMP(slot) do { if (tracing && t < 24) trace_buf[(t) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UM_STAMP(slot) do { } while (0)
#endif

// -DUM_WATTN_P1=1 (diagnostic builds: profiles/r03_precision_budget.txt): the PV product with P in ONE fp16 plane -- 2 MFMA
// products (V_lo.P + V_hi.P) instead of 3, no lo split of P -- and the softmax denominator summed over the SAME rounded
// probabilities (v_dot2c_f32_f16), so that the result is an exact weighted mean with weights perturbed by <= 2^-11 relative.
#ifndef UM_WATTN_P1
#define UM_WATTN_P1 0
#endif
// Round-4 experiments on the tile loop (diagnostic builds, profiles/r04_attention_experiments.txt):
//   UM_WATTN_DMA_POS  0: the tile's four LDS-DMA statements ride on QK^T k-steps 0, 2, 4, 6 (rounds 1-3)
//                     1: they are issued in the softmax phase (VALU only) instead of inside an MFMA phase
//                     2: on QK^T k-steps 1, 3, 5, 7 (the address preparation no longer stands in front of the first MFMA)
//   UM_WATTN_OFF32    1: staging sources as 32-bit byte offsets from a scalar base (global_load_lds ... v, s[base]) instead of
//                        64-bit per-lane pointers: half the address registers, no 64-bit arithmetic per tile
#ifndef UM_WATTN_DMA_POS
#define UM_WATTN_DMA_POS 0
#endif
#ifndef UM_WATTN_OFF32
#define UM_WATTN_OFF32 1      // round 4: -2.7 % per launch in the model, +0.9 % end to end (profiles/r04_attention_experiments.txt)
#endif

struct WattnArgs {
    const unsigned short* qp;    // planes [NS][S][L][128]
    const unsigned short* kp;
    const unsigned short* vp;
    long plane_stride;           // q planes: rows * ldq
    long kv_plane_stride;        // k, v planes: rows * ldkv
    int ldq, ldkv;               // row strides in elements (128 for dedicated tensors; 256 / 384 for fused projections)
    float* out;                  // [S][L][128]
    int h, w, win_h, win_w, shift_h, shift_w;
    int nwx, nwin;               // windows per row, windows per stream
    int streams, kv_rotate;
    int n;                       // tokens per window
    int nqt;                     // 128-query tiles per window
    int total;                   // workgroups
    float scale_log2;            // log2(e) / sqrt(C)
    float mask_raw;              // -100 * sqrt(C): the shifted-window mask in raw q.k units
    // MERGE variant: out = LayerNorm(attention . Wm^T) (+ residual)   (transformer.py:137-138, 144)
    const unsigned short* wm;    // planes [NS][128][128] of the merge weight, pre-scaled by 2^wshift
    long wm_plane_stride;
    const float* gamma;
    const float* beta;
    const float* residual;       // optional [S][L][128]
    float wm_scale, eps;         // 2^-wshift
    float headroom;              // powers of two by which a moving softmax offset overshoots (exact mode: 8)
    // QPROJ variant: q = x . Wq^T computed in the prologue (transformer.py:58); qp is unused
    const float* x;              // [S][L][128] fp32 source tokens
    const unsigned short* wq;    // planes [NS][128][128] of the query weight, pre-scaled by 2^wshift (stride wm_plane_stride)
    // KSPLIT instantiation (small launches): the workgroups come in groups of `split` per query tile, each on 1 / split of the
    // window's key tiles
    int split;
    float* ks_part;              // [split tiles][split][17][256][4] fp32: O^T (16 vectors), (M, l, -, -) of every part
    unsigned* ks_flag;           // [split tiles] arrival counters, zero between launches
    unsigned* range_flag;        // um_range_flags: sticky operand-range word (device address; nullptr: none)
    // Round 6: masked-tile skip (see "class-major order" below)
    float skip_raw;              // a wholly masked key tile is dropped iff its hi.hi probe stays <= m_run + skip_raw  (raw q.k units;
                                 //   (100 - margin) * sqrt(C), margin = UM_WATTN_SKIP_MARGIN); < 0: never (every tile is computed)
    int spx;                     // > 0: streams per XCD for the heaviest-window-first grid order (0: natural order)
    unsigned long long* tile_census;   // nullptr, or 4 counters: tiles computed in full / probed / probed and then computed / workgroups
};

// window-local token -> global token index and its mask class.
// Class = 2*[rolled row in the wrapped band] + [rolled col in the wrapped band].  Inside ONE window this is
// equivalent to the reference's 3x3 region label (unimatch/utils.py:90-100): two tokens of a window carry
// different labels iff their classes differ (the first label band never shares a window with the others).
__device__ __forceinline__ int window_token(const WattnArgs& a, int wy, int wx, int t, int& cls) {
    const int ly = t / a.win_w, lx = t - ly * a.win_w;
    const int ry = wy * a.win_h + ly, rx = wx * a.win_w + lx;
    int oy = ry + a.shift_h, ox = rx + a.shift_w;
    oy = oy >= a.h ? oy - a.h : oy;
    ox = ox >= a.w ? ox - a.w : ox;
    const int rb = (a.shift_h > 0 && ry >= a.h - a.shift_h) ? 1 : 0;
    const int cb = (a.shift_w > 0 && rx >= a.w - a.shift_w) ? 1 : 0;
    cls = 2 * rb + cb;
    return oy * a.w + ox;
}

// ---- class-major order of a window's tokens (round 6) -----------------------------------------------------------------------------
// Softmax is invariant under a permutation of the keys and queries are independent, so the ORDER in which a workgroup walks its
// window is free.  In a window that carries the shift mask the tokens are laid out class by class -- class 0 (rows < R0, columns
// < C0 of the window), then 1 (columns >= C0), 2 (rows >= R0), 3 -- each class row-major inside its rectangle.  With that order a
// 32-key tile and a 128-query workgroup are uniformly of ONE class whenever the class sizes are multiples of 32 / 128 (config 2:
// 768 | 768 and 4 x 384), and a (query tile, key tile) pair of different classes is a block of nothing but -100 logits
// (unimatch/utils.py:84-108, unimatch/attention.py:88-89): 43.75 % of a shifted launch at K = 2.  Windows without the mask have
// R0 = win_h, C0 = win_w: one class, plain row-major order.
struct WinClasses {
    int R0, C0;              // first row / column of the wrapped band inside this window (win_h / win_w: none)
    int e0, e1, e2;          // class-major positions at which classes 1, 2, 3 begin
};
__device__ __forceinline__ WinClasses win_classes(const WattnArgs& a, int wy, int wx) {
    WinClasses k;
    k.R0 = (a.shift_h > 0 && (wy + 1) * a.win_h == a.h) ? a.win_h - a.shift_h : a.win_h;
    k.C0 = (a.shift_w > 0 && (wx + 1) * a.win_w == a.w) ? a.win_w - a.shift_w : a.win_w;
    k.e0 = k.R0 * k.C0;
    k.e1 = k.R0 * a.win_w;
    k.e2 = k.e1 + (a.win_h - k.R0) * k.C0;
    return k;
}
__device__ __forceinline__ int cm_class(const WinClasses& k, int p) { return (p >= k.e0 ? 1 : 0) + (p >= k.e1 ? 1 : 0) + (p >= k.e2 ? 1 : 0); }
// class-major position p (< n) of window (wy, wx) -> global token index and mask class
__device__ __forceinline__ int cm_token(const WattnArgs& a, const WinClasses& k, int wy, int wx, int p, int& cls) {
    cls = cm_class(k, p);
    const int start = cls == 0 ? 0 : (cls == 1 ? k.e0 : (cls == 2 ? k.e1 : k.e2));
    const int cw = (cls & 1) ? a.win_w - k.C0 : k.C0;
    const int idx = p - start;
    const int iy = idx / cw, ix = idx - iy * cw;
    const int ry = wy * a.win_h + ((cls & 2) ? k.R0 : 0) + iy, rx = wx * a.win_w + ((cls & 1) ? k.C0 : 0) + ix;
    int oy = ry + a.shift_h, ox = rx + a.shift_w;
    oy = oy >= a.h ? oy - a.h : oy;
    ox = ox >= a.w ? ox - a.w : ox;
    return oy * a.w + ox;
}

// 16 bytes per lane, global -> LDS without passing through VGPRs: LDS address = wave-uniform dst + 16 * lane.
// Issued through inline asm on purpose: hipcc cannot prove that the ring slot being filled and the slot being
// read do not alias and would drain the DMA (s_waitcnt vmcnt(0)) in front of the first LDS read; hidden from its
// scoreboard, the transfer stays in flight for the whole tile and is waited for explicitly before the barrier.
__device__ __forceinline__ void lds_dma16(const void* gsrc, const unsigned char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

template <class T, int NS, bool MERGE, bool QPROJ = false, bool KSPLIT = false>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void window_attn_kernel(WattnArgs a) {
    // One LDS buffer = one tile of TK window tokens: K planes, V planes (linear 256-byte rows, 16-byte chunks
    // XOR-swizzled by the SOURCE address because global_load_lds writes lane-linear), and the additive bias
    // table [4 query classes][TK].  Two buffers: tile t+1 streams in by LDS-DMA while tile t is consumed.
    constexpr int TK = 32;
    constexpr int PLANE = TK * 256;
    constexpr int BIAS_OFF = 2 * NS * PLANE;
    constexpr int BUF = BIAS_OFF + 4 * TK * 4;
    constexpr int PSHIFT = (NS == 2) ? 14 : 0;
    // + window-local token -> (global token << 2 | mask class), tabulated once per workgroup when the window has at most
    // TAB_BYTES / 4 tokens (rounded up to tiles): a tile's staging arithmetic is then two LDS reads instead of ~40 VALU
    constexpr int TAB_BYTES = 8192;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + TAB_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const float neg1 = um_opaque_neg1();
    // KSPLIT: a query tile whose key walk is shared by `split` workgroups, neighbours in the grid (same XCD); part p takes the
    // p-th share of the key tiles.  Used for launches with few query tiles (batch 1: 40 - 96 tiles on 256 CUs, each walking the
    // whole window, are latency-bound by that walk) and for the REMAINDER round of big launches (768 tiles on 512 resident slots:
    // the last 256 tiles as 512 half-walks keep every slot busy to the end).  Hand-off without waiting: every part publishes
    // (O^T, M, l) in its memory slot and takes a ticket from the tile's arrival counter; the LAST arriver merges all slots in part
    // order (integer offsets, power-of-two factors: exact, and the order is fixed: bitwise reproducible) and runs the epilogue,
    // the others exit.  Nobody spins, so nothing is assumed about residency or dispatch order.
    int wg, wl = 0, part = 0, nsplit = 1;
    if constexpr (KSPLIT) {
        // (a mixed grid -- whole tiles first, key-split remainder behind -- was measured in round 3 and lost; the plan only ever
        // produces all-split launches, so the kernel no longer carries that path)
        const int j = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        nsplit = a.split;
        wl = j / nsplit;                                           // split tile (indexes the slots and the counter)
        part = j - wl * nsplit;
        wg = wl;                                                   // query tile of the call
    } else {
        wg = xcd_remap(blockIdx.x, gridDim.x);
    }
    int qt, s, wy, wx;
    if (!KSPLIT && a.spx > 0) {
        // Heaviest windows first (round 6).  With the masked tiles dropped a workgroup of an interior window does 100 % of the
        // key walk, one of a last-column / last-row window ~58 %, one of the corner window ~37 %; the launch is 1.5 rounds of the
        // resident slots, so the order decides what the tail is made of.  Every XCD keeps its own `spx` streams (xcd_remap hands
        // XCD x the ids [x * per, (x + 1) * per)); inside them the order is window class, stream, query tile: interior windows,
        // last column, last row, corner.
        const int per = a.spx * a.nwin * a.nqt;
        const int xcd = wg / per, i = wg - xcd * per;
        const int wr = i / (a.spx * a.nqt), r2 = i - wr * (a.spx * a.nqt);
        const int sl = r2 / a.nqt;
        qt = r2 - sl * a.nqt;
        s = xcd * a.spx + sl;
        const int nwy = a.nwin / a.nwx, nA = (nwy - 1) * (a.nwx - 1);
        if (wr < nA) {
            wy = wr / (a.nwx - 1);
            wx = wr - wy * (a.nwx - 1);
        } else if (wr < nA + nwy - 1) {
            wy = wr - nA;
            wx = a.nwx - 1;
        } else if (wr < a.nwin - 1) {
            wy = nwy - 1;
            wx = wr - nA - (nwy - 1);
        } else {
            wy = nwy - 1;
            wx = a.nwx - 1;
        }
    } else {
        qt = wg % a.nqt;
        const int win = (wg / a.nqt) % a.nwin;
        s = wg / (a.nqt * a.nwin);
        wy = win / a.nwx;
        wx = win - wy * a.nwx;
    }
    const bool has_mask = (a.shift_h > 0 && (wy + 1) * a.win_h == a.h) || (a.shift_w > 0 && (wx + 1) * a.win_w == a.w);
    const float c = a.scale_log2;
    const long sbase = (long)s * a.h * a.w;
    // keys / values of stream (s + kv_rotate) mod streams: cross attention between the two halves of one stream tensor
    // ([f0; f1] attends [f1; f0], unimatch/transformer.py:271-291) without materialising the swapped copy
    const long kvbase = (long)((s + a.kv_rotate) % a.streams) * a.h * a.w;

    if constexpr (QPROJ) {       // Wq -> the idle K/V ring, in flight while the token table is being built
        const int row4 = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * (8 * wave + i) + row4;
            const int cw = pc ^ (row & 15);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                lds_dma16(a.wq + pl * a.wm_plane_stride + row * UM_CHANNELS + 8 * cw, lds + pl * 32768 + (4 * (8 * wave + i)) * 256);
        }
    }
    const int ntiles = (a.n + TK - 1) / TK;
    const int t0 = (part * ntiles) / nsplit;                              // this workgroup's key tiles [t0, t1)
    const int t1 = ((part + 1) * ntiles) / nsplit;
#ifdef UM_TRACE
    const bool tracing = g_um_trace != nullptr && (blockIdx.x % 37) == 0 && tid == 0;
    unsigned long long* trace_buf = g_um_trace + (size_t)(blockIdx.x / 37) * (24 * 8 + 8);
    if (tracing) trace_buf[24 * 8] = __builtin_amdgcn_s_memtime();
#endif

    const bool use_tab = ntiles * TK * 4 <= TAB_BYTES;                  // uniform
    const unsigned* tab = reinterpret_cast<const unsigned*>(lds + 2 * BUF);
    const WinClasses wc = win_classes(a, wy, wx);
    if (use_tab) {                                                      // class-major order (see cm_token)
        for (int tl = tid; tl < ntiles * TK; tl += 256) {
            int cls;
            const int tokn = cm_token(a, wc, wy, wx, min(tl, a.n - 1), cls);    // past the window's last token: any valid row (masked)
            reinterpret_cast<unsigned*>(lds + 2 * BUF)[tl] = ((unsigned)tokn << 2) | (unsigned)cls;
        }
        if (!QPROJ) __syncthreads();                                    // QPROJ: the prologue's barriers below cover it
    }
    // ---- which key tiles this workgroup computes, probes, or may drop (class-major windows only) -----------------------------
    //   own_pure: tiles uniformly of the workgroup's own query class       -> computed, and without the bias table;
    //   probe   : tiles uniformly of ANOTHER class (every logit carries -100) -> probed after the own tiles (see the probe phase);
    //   the rest (tiles that straddle a class boundary, or a workgroup whose 128 queries are of two classes): computed with the
    //   per-lane bias table as before.
    // All wave-uniform (ballots); tile index = bit index, at most 64 tiles because the token table holds 2048 tokens.
    unsigned long long own_pure = 0, probe = 0;
#ifdef UM_WATTN_KSPLIT_NOSKIP          // diagnostic builds: key-split launches walk every tile (the same-box A/B of profiles/r06_ksplit_skip.txt)
    if (has_mask && use_tab && !KSPLIT) {
#else
    if (has_mask && use_tab) {
#endif
        const int q0 = qt * 128, q1 = min(q0 + 127, a.n - 1);
        const int cq = cm_class(wc, q0);
        if (cq == cm_class(wc, q1)) {
            const int p0 = min(lane * TK, a.n - 1), p1 = min(lane * TK + TK - 1, a.n - 1);
            const int cf = cm_class(wc, p0), cl = cm_class(wc, p1);
            own_pure = __ballot(lane < ntiles && cf == cl && cf == cq);
            if (a.skip_raw >= 0.f) probe = __ballot(lane < ntiles && cf == cl && cf != cq);
        }
    }
    const bool bymask = use_tab;                                        // tile walk driven by a bit mask (else: the range [t0, t1))
    // KSPLIT (round 6: the key-split small launches -- batch-1 latency, the reference's own evaluation protocol -- skip as well): part p
    // of a query tile takes the p-th share BY RANK of the tiles to compute and the p-th share of the tiles to probe; its running
    // maximum covers its own share only, a weaker (still valid) lower bound of the row's final maximum -- a part without own-class
    // tiles probes against -1e30, i.e. computes its share of the masked tiles.
    // (only when a part's share is at least four probe tiles: below that -- config 1's 560-token windows, two probe tiles per part -- the
    // probe phase's barriers cost a latency-bound launch more than the tiles it saves: +2.7 % measured, profiles/r06_ksplit_skip.txt)
    if (KSPLIT && (int)__builtin_popcountll(probe) < 4 * nsplit) probe = 0;
    unsigned long long walk = (ntiles >= 64 ? ~0ull : ((1ull << ntiles) - 1ull)) & ~probe;
    if constexpr (KSPLIT) {
        if (bymask && nsplit > 1) {
            auto share = [&](unsigned long long mask) -> unsigned long long {
                const int cnt = (int)__builtin_popcountll(mask);
                const int lo = (part * cnt) / nsplit, hi = ((part + 1) * cnt) / nsplit;
                unsigned long long out = 0;
                int r = 0;
                while (mask != 0) {
                    const unsigned long long bit = mask & (0ull - mask);
                    if (r >= lo && r < hi) out |= bit;
                    mask ^= bit;
                    ++r;
                }
                return out;
            };
            if (probe != 0) {
                walk = share(walk);
                probe = share(probe);
            } else {                        // nothing to probe: the contiguous share [t0, t1) of all tiles, no rank loop
                walk = (t1 >= 64 ? ~0ull : ((1ull << t1) - 1ull)) & ~((1ull << t0) - 1ull);
            }
        }
    }

    // ---- this lane's query -----------------------------------------------------------------------
    const int tq = qt * 128 + wave * 32 + (lane & 31);
    int clsq;
    const int tokq = use_tab ? cm_token(a, wc, wy, wx, min(tq, a.n - 1), clsq) : window_token(a, wy, wx, min(tq, a.n - 1), clsq);
    // per-lane LDS read offsets of an A-operand row tile (K tile, Wm / Wq rows): loop invariant
    int koff[8];
    {
        const int r = lane & 31;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) koff[ks] = r * 256 + (((2 * ks + half) ^ (r & 15)) << 4);
    }
    i16x8 qf[NS][8];
    if constexpr (!QPROJ) {
        const unsigned short* qb = a.qp + (sbase + tokq) * a.ldq + 8 * half;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[pl][ks] = ld_global_16B(qb + pl * a.plane_stride + 16 * ks);
        // Make hipcc retire these loads HERE: inside the tile loop its scoreboard must be empty, otherwise its
        // counted vmcnt(N) waits for them would also wait for the (uncounted) LDS-DMA issued by inline asm.
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[pl][ks]));
    } else {
        // ---- Q^T = Wq . X^T for the workgroup's 128 queries (transformer.py:58): a query row is consumed by exactly one
        // workgroup, so projecting it here costs no recompute (96 of the ~2400 MFMAs of a 1536-token window) and the q
        // planes -- one write and one read of [S*L, 128] operands per layer -- never exist.  Wq rides the still idle K/V ring
        // with the K tile's swizzle (as Wm does in the epilogue); X^T is the B operand, one token per lane, split into
        // hi | lo in registers exactly as split_planes_kernel would; the accumulators turn into the Q^T operand
        // fragments the way P^T does.
        i16x8 xf[NS][8];
        float rmx = 0.f;                 // largest magnitude turned into an fp16 operand (um_range_flags)
        {
            const float* xb = a.x + (sbase + tokq) * UM_CHANNELS + 8 * half;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(xb + 16 * ks);
                const f32x4 v = *reinterpret_cast<const f32x4*>(xb + 16 * ks + 4);
                const float y[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                u32x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) rmx = fmaxf(rmx, __builtin_fabsf(y[j]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hi[j] = T::pack2(y[2 * j], y[2 * j + 1]);
                    if (NS == 2) lo[j] = T::lo2(y[2 * j], y[2 * j + 1], hi[j], neg1);
                }
                xf[0][ks] = __builtin_bit_cast(i16x8, hi);
                if (NS == 2) xf[NS - 1][ks] = __builtin_bit_cast(i16x8, lo);
            }
        }
        um_range_note<T>(a.range_flag, rmx, UM_RANGE_ATTN_TOKENS);
        rmx = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const i16x8 wh = *reinterpret_cast<const i16x8*>(lds + ot * (32 * 256) + koff[ks]);
                if (NS == 2) {
                    const i16x8 wl = *reinterpret_cast<const i16x8*>(lds + 32768 + ot * (32 * 256) + koff[ks]);
                    acc = T::mfma(wl, xf[0][ks], acc);
                    acc = T::mfma(wh, xf[NS - 1][ks], acc);
                }
                acc = T::mfma(wh, xf[0][ks], acc);
            }
            // lane holds, for its query, channels 32 ot + 8 g + 4 half + i (register 4 g + i): k-step 2 ot + kk of the Q^T
            // operand = registers 8 kk .. 8 kk + 7
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                unsigned wh[4], wl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = acc[8 * kk + 2 * j] * a.wm_scale, p1 = acc[8 * kk + 2 * j + 1] * a.wm_scale;
                    rmx = fmaxf(rmx, fmaxf(__builtin_fabsf(p0), __builtin_fabsf(p1)));
                    wh[j] = T::pack2(p0, p1);
                    if (NS == 2) wl[j] = T::lo2(p0, p1, wh[j], neg1);
                }
                {
                    const auto xx = __builtin_amdgcn_permlane32_swap(wh[0], wh[2], false, false);
                    const auto yy = __builtin_amdgcn_permlane32_swap(wh[1], wh[3], false, false);
                    const u32x4 f = {xx[0], yy[0], xx[1], yy[1]};
                    qf[0][2 * ot + kk] = __builtin_bit_cast(i16x8, f);
                }
                if (NS == 2) {
                    const auto xx = __builtin_amdgcn_permlane32_swap(wl[0], wl[2], false, false);
                    const auto yy = __builtin_amdgcn_permlane32_swap(wl[1], wl[3], false, false);
                    const u32x4 f = {xx[0], yy[0], xx[1], yy[1]};
                    qf[NS - 1][2 * ot + kk] = __builtin_bit_cast(i16x8, f);
                }
            }
        }
        um_range_note<T>(a.range_flag, rmx, UM_RANGE_ATTN_QUERY);
        __syncthreads();        // every wave is done with Wq: the ring may take tile 0
    }

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = UM_NEG_INIT, M = -ceilf(UM_NEG_INIT * c), l = 0.f;

    // ---- LDS-DMA staging.  One wave instruction moves 64 lanes x 16 B = 4 token rows; wave w owns rows
    // 8w..8w+7 of the tile (2 instructions per plane).  LDS chunk position cp of row r holds source chunk
    // cp ^ swz(r): swzK = r & 15 (A-fragment ds_read_b128 conflict free), swzV = (r & 3) << 2 (the 4 rows of a
    // ds_read_b64_tr_b16 group land in 4 different bank quarters).
    // The kernel is instruction-issue bound (rocprof + ablations, profiles/), so the per-tile addressing is kept
    // division free: every lane carries the window-local (ly, lx) of the rows it stages and advances them by
    // TK tokens per tile with one add / compare / select each.
    const int adv_y = TK / a.win_w, adv_x = TK - adv_y * a.win_w;      // TK tokens = adv_y rows + adv_x columns
    const int y0 = wy * a.win_h, x0 = wx * a.win_w;
    int sly[2], slx[2];                                                  // staged rows (j = 0, 1)
    long ssrc_k[2], ssrc_v[2];                                           // lane-constant part of the source offsets
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = t0 * TK + 8 * wave + 4 * j + ((lane >> 4) & 3);     // TK is a multiple of 16: row & 15 is tile-local
        sly[j] = row / a.win_w;
        slx[j] = row - sly[j] * a.win_w;
        const int cp = lane & 15;
        ssrc_k[j] = (cp ^ (row & 15)) << 3;
        ssrc_v[j] = (cp ^ ((row & 3) << 2)) << 3;
    }
    int bly = 0, blx = 0;                                                // bias-table key of this thread (tid < 4*TK)
    {
        const int key = t0 * TK + (tid & (TK - 1));
        bly = key / a.win_w;
        blx = key - bly * a.win_w;
    }
    // the additive bias table of tile t is needed for the ragged tail and for every tile that is not uniformly of the workgroup's
    // own class
    auto need_bias = [&](int t) -> bool { return (t + 1) * TK > a.n || (has_mask && !((own_pure >> t) & 1ull)); };
    auto token_at = [&](int ly, int lx, int& cls) -> int {             // (ly, lx) may run past the window: clamp
        ly = min(ly, a.win_h - 1);
        const int ry = y0 + ly, rx = x0 + lx;
        int oy = ry + a.shift_h, ox = rx + a.shift_w;
        oy = oy >= a.h ? oy - a.h : oy;
        ox = ox >= a.w ? ox - a.w : ox;
        cls = 2 * ((a.shift_h > 0 && ry >= a.h - a.shift_h) ? 1 : 0) + ((a.shift_w > 0 && rx >= a.w - a.shift_w) ? 1 : 0);
        return oy * a.w + ox;
    };
    auto advance = [&](int& ly, int& lx) {
        lx += adv_x;
        ly += adv_y;
        const bool wrap = lx >= a.win_w;
        lx = wrap ? lx - a.win_w : lx;
        ly = wrap ? ly + 1 : ly;
    };
    // Staging of tile t is split in two: prepare (source addresses of this lane's two rows, bias table) and
    // NPIECE single-instruction DMA pieces.  All waves leave the barrier together, and a burst of 8 DMA
    // instructions per wave queues up in the CU's single address unit (measured: ~1100 of ~5400 cycles per
    // tile); issued one per k-step of the QK^T loop instead, each piece hides under three MFMAs.
    constexpr int NPIECE = 4 * NS;
    const unsigned short* spk[2];
    const unsigned short* spv[2];
    unsigned sok[2], sov[2];                                             // UM_WATTN_OFF32: byte offsets from a.kp / a.vp
    auto stage_prepare = [&](int t, unsigned char* base) {
        // bias[class][key]: 0, the -100 mask (raw units), or "no such key"; only tiles that need it read it
        const bool need = need_bias(t);
        if (use_tab) {
            const unsigned* tp = tab + t * TK + 8 * wave + ((lane >> 4) & 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (UM_WATTN_OFF32) {
                    const unsigned goff = ((unsigned)kvbase + (tp[4 * j] >> 2)) * (unsigned)a.ldkv;
                    sok[j] = 2u * (goff + (unsigned)ssrc_k[j]);
                    sov[j] = 2u * (goff + (unsigned)ssrc_v[j]);
                } else {
                    const long goff = (kvbase + (long)(tp[4 * j] >> 2)) * a.ldkv;
                    spk[j] = a.kp + goff + ssrc_k[j];
                    spv[j] = a.vp + goff + ssrc_v[j];
                }
            }
            if (need && tid < 4 * TK) {
                const int cq = tid >> 5, key = tid & (TK - 1);
                const int cls = (int)(tab[t * TK + key] & 3u);
                const float bv = t * TK + key >= a.n ? UM_NEG_MASK : ((has_mask && cls != cq) ? a.mask_raw : 0.f);
                reinterpret_cast<float*>(base + BIAS_OFF)[cq * TK + key] = bv;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int cls;
            const int tok = token_at(sly[j], slx[j], cls);
            advance(sly[j], slx[j]);
            if (UM_WATTN_OFF32) {
                const unsigned goff = ((unsigned)kvbase + (unsigned)tok) * (unsigned)a.ldkv;
                sok[j] = 2u * (goff + (unsigned)ssrc_k[j]);
                sov[j] = 2u * (goff + (unsigned)ssrc_v[j]);
            } else {
                const long goff = (kvbase + tok) * a.ldkv;
                spk[j] = a.kp + goff + ssrc_k[j];
                spv[j] = a.vp + goff + ssrc_v[j];
            }
        }
        if (need && tid < 4 * TK) {
            const int cq = tid >> 5, key = tid & (TK - 1);
            int cls;
            (void)token_at(bly, blx, cls);
            const float bv = t * TK + key >= a.n ? UM_NEG_MASK : ((has_mask && cls != cq) ? a.mask_raw : 0.f);
            reinterpret_cast<float*>(base + BIAS_OFF)[cq * TK + key] = bv;
        }
        advance(bly, blx);
    };
    // one statement = the wave's two 4-row groups of one plane of K (or V): ONE M0 write, the second request carries the LDS
    // (and global) displacement of 1024 bytes in its instruction offset, so its source comes in 1024 bytes low.  M0 is saved and
    // restored inside the statement (hipcc manages M0 itself for the prologue's / epilogue's lds_dma16 and does not honour an
    // "m0" clobber): 6 instructions per pair instead of 10.
    constexpr int NPAIR = 2 * NS;
    auto stage_pair = [&](int ip, unsigned char* base) {        // ip = 0 .. NPAIR-1, compile-time after unrolling
        const int pl = ip >> 1, isv = ip & 1;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)(
            base + (8 * wave) * 256 + (isv * NS + pl) * PLANE));
        unsigned keep;
        if (UM_WATTN_OFF32) {
            // the second request's instruction offset (1024: its LDS displacement) also moves its SOURCE: compensated in the SCALAR
            // base, not in the offset register -- that one is zero-extended, so "offset - 1024" of a row in the first kilobyte of
            // the plane would address 4 GiB further on
            const unsigned short* base = (isv ? a.vp : a.kp) + pl * a.kv_plane_stride;       // wave-uniform: SGPR pair
            const unsigned short* base1 = base - 512;
            const unsigned o0 = isv ? sov[0] : sok[0];
            const unsigned o1 = isv ? sov[1] : sok[1];
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "global_load_lds_dwordx4 %2, %4 offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(o0), "v"(o1), "s"(base), "s"(base1), "s"(dst) : "memory");
            return;
        }
        const unsigned short* s0 = (isv ? spv[0] : spk[0]) + pl * a.kv_plane_stride;
        const unsigned short* s1 = (isv ? spv[1] : spk[1]) + pl * a.kv_plane_stride - 512;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(s0), "v"(s1), "s"(dst) : "memory");
    };

    const int li = lane & 15, lg = (lane >> 4) & 1;
    int voff[4];
    {
        const int r3 = (li >> 2) & 3;                       // (row & 3) of every row this lane addresses
        const int rowb = 8 * half + (li >> 2);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            voff[dt] = rowb * 256 + ((((dt ^ r3) << 2) + 2 * lg + ((li & 3) >> 1)) << 4) + 8 * (li & 1);
    }

    // One tile.  SLOT is a compile-time constant so that every LDS offset of the tile is an instruction immediate.  tn = the tile
    // that follows in this workgroup's walk (staged into the other slot meanwhile), < 0: none.
    auto tile = [&](auto slot_c, int t, int tn) {
        constexpr int SLOT = decltype(slot_c)::value;
        UM_STAMP(0);
        const bool staging = tn >= 0;
        unsigned char* nxt = lds + (SLOT ^ 1) * BUF;
        if (staging) stage_prepare(tn, nxt);
        const unsigned char* kb = lds + SLOT * BUF;
        const unsigned char* vb = kb + NS * PLANE;

        UM_STAMP(1);
        // ---- S^T = K . Q^T for the 32 keys of the tile ---------------------------------------------------
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        {   // fragment reads run two k-steps ahead of the MFMAs that consume them
            i16x8 fh[3], fl[3];
            __builtin_amdgcn_s_setprio(1);   // MFMA phases outrank the sibling wave's VALU phases (measured +3-4 %)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fh[ks] = *reinterpret_cast<const i16x8*>(kb + koff[ks]);
                if (NS == 2) fl[ks] = *reinterpret_cast<const i16x8*>(kb + PLANE + koff[ks]);
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 2 < 8) {
                    fh[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(kb + koff[ks + 2]);
                    if (NS == 2) fl[(ks + 2) % 3] = *reinterpret_cast<const i16x8*>(kb + PLANE + koff[ks + 2]);
                }
                if (UM_WATTN_DMA_POS == 0 && staging && (ks * NPAIR) % 8 == 0) stage_pair(ks * NPAIR / 8, nxt);
                if (UM_WATTN_DMA_POS == 2 && staging && NPAIR == 4 && (ks & 1)) stage_pair(ks >> 1, nxt);   // k-steps 1, 3, 5, 7
                if (UM_WATTN_DMA_POS == 2 && staging && NPAIR == 2 && (ks == 3 || ks == 7)) stage_pair(ks >> 2, nxt);
                if (NS == 2) {
                    sc = T::mfma(fl[ks % 3], qf[0][ks], sc);
                    sc = T::mfma(fh[ks % 3], qf[NS - 1][ks], sc);
                }
                sc = T::mfma(fh[ks % 3], qf[0][ks], sc);
            }
            // Scheduling contract for hipcc (it otherwise emits read -> full wait -> MFMAs per k-step):
            // fragment reads stay two k-steps ahead of the MFMAs that consume them.
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

        UM_STAMP(2);
        // ---- additive bias: shifted-window mask (-100, unimatch/utils.py:106) and the ragged window tail -----
        if (need_bias(t)) {
            const float* bt = reinterpret_cast<const float*>(kb + BIAS_OFF) + clsq * TK + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bt + 8 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[4 * g + i] += bv[i];
            }
        }

        // ---- online softmax (row max shared by the lane pair l, l^32) ------------------------------------
        if (UM_WATTN_DMA_POS == 1 && staging) stage_pair(0, nxt);
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
        {   // max with the partner lane (l ^ 32) without touching LDS
            float u, v2;
            half_wave_pair(mx, u, v2);
            mx = fmaxf(u, v2);
        }
        m = fmaxf(m, mx);
        if (UM_WATTN_DMA_POS == 1 && staging && NPAIR > 1) stage_pair(1, nxt);
        // Lazy, exact rescale.  The exponent offset M is an integer (every rescale factor is a power of two) and
        // is allowed to lag the true running max by up to LAG: then p <= 2^(LAG + PSHIFT) = 2^15 still fits fp16,
        // and the 64-accumulator rescale -- which otherwise fires on almost every tile because SOME of the wave's
        // 32 rows moves -- becomes rare.  The branch is wave uniform; rows that did not move multiply by 1.
        // When the offset does move it OVERSHOOTS by HEADROOM powers of two, so that the next move needs the row maximum to
        // grow by another 2^(HEADROOM + LAG): with row maxima that creep up over the tiles (a maximum of n samples grows like
        // sqrt(2 ln n)) SOME row moved by more than 2^LAG in most tiles.  Cost: small probabilities keep 22 bits down to
        // 2^(-17 + HEADROOM) of the row maximum instead of 2^-17; below that the fp16 lo plane goes subnormal: absolute error
        // 2^(-39 + HEADROOM) of the row maximum = 2^-31, far below fp32 resolution.  (UM_WATTN_HEADROOM=0: A/B timing.)
        constexpr float LAG = (NS == 2) ? 1.f : 8.f;
        const float Mn = -ceilf(m * c);
        const bool move = Mn + LAG < M;
        if (__any(move)) {
            const float Mh = Mn - a.headroom;
            const float resc = move ? fast_exp2(Mh - M) : 1.f;
            M = move ? Mh : M;
            l *= resc;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= resc;
        }
        if (UM_WATTN_DMA_POS == 1 && staging && NPAIR > 2) stage_pair(2, nxt);
        constexpr bool P1 = (NS == 2) && UM_WATTN_P1;           // one P plane (see UM_WATTN_P1)
        constexpr int NSP = P1 ? 1 : NS;
        const float mc = M + (float)PSHIFT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = fast_exp2(__builtin_fmaf(sc[r], c, mc));
            sc[r] = p;
            if (!P1) l += p;
        }

        if (UM_WATTN_DMA_POS == 1 && staging && NPAIR > 3) stage_pair(3, nxt);
        // ---- P^T operand fragments: cvt + v_permlane32_swap, no LDS ---------------------------------------
        // k-step ks (16 keys) uses regs 8*ks..8*ks+7; after the swaps a lane holds keys 8*half .. 8*half+7 of
        // the step for its own query (B operand layout).
        i16x8 pf[NS][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r0 = 8 * ks;
            unsigned wh[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = sc[r0 + 2 * j], p1 = sc[r0 + 2 * j + 1];
                wh[j] = T::pack2(p0, p1);
                if (NSP == 2) wl[j] = T::lo2(p0, p1, wh[j], neg1);
                if constexpr (P1) {
                    const f16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
                    l = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wh[j]), one, l, false);
                }
            }
            {
                const auto x = __builtin_amdgcn_permlane32_swap(wh[0], wh[2], false, false);
                const auto y = __builtin_amdgcn_permlane32_swap(wh[1], wh[3], false, false);
                const u32x4 f = {x[0], y[0], x[1], y[1]};
                pf[0][ks] = __builtin_bit_cast(i16x8, f);
            }
            if (NSP == 2) {
                const auto x = __builtin_amdgcn_permlane32_swap(wl[0], wl[2], false, false);
                const auto y = __builtin_amdgcn_permlane32_swap(wl[1], wl[3], false, false);
                const u32x4 f = {x[0], y[0], x[1], y[1]};
                pf[NS - 1][ks] = __builtin_bit_cast(i16x8, f);
            }
        }

        // ---- O^T += V^T . P^T ----------------------------------------------------------------------------------
        UM_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* va = vb + voff[dt] + ks * 16 * 256;
                i16x8 vh, vl;
                {
                    const i16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va));
                    const i16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va + 4 * 256));
                    vh = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (NS == 2) {
                    const i16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va + PLANE));
                    const i16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) i16x4*)(va + PLANE + 4 * 256));
                    vl = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = T::mfma(vl, pf[0][ks], o[dt]);
                    if (NSP == 2) o[dt] = T::mfma(vh, pf[NS - 1][ks], o[dt]);
                }
                o[dt] = T::mfma(vh, pf[0][ks], o[dt]);
            }
        }
        {   // transpose reads two (k-step, d-tile) groups ahead of the MFMAs
            constexpr int RD = 2 * NS, MF = NS + NSP - 1;
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 1);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, MF, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, RD, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 1);
        }
        __builtin_amdgcn_s_setprio(0);
        UM_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile t+1 has landed in LDS
        UM_STAMP(5);
        __syncthreads();  // ... and every wave's has; tile t is fully consumed, its slot may be refilled
        UM_STAMP(6);
    };
    // ---- the key walk.  Pass 0: the tiles that are computed unconditionally (all of them, unless the workgroup is class-uniform in
    // a masked window); then the PROBE of the wholly masked tiles; pass 1: the masked tiles the probe could not clear.
    unsigned long long fail = 0;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long rem = pass == 0 ? walk : fail;
        auto next_tile = [&](int t) -> int {
            if (bymask) {
                if (rem == 0) return -1;
                const int r = (int)__builtin_ctzll(rem);
                rem &= rem - 1;
                return r;
            }
            return t + 1 < t1 ? t + 1 : -1;
        };
        int t = bymask ? next_tile(0) : t0;
        if (t >= 0) {                      // (a key-split part may have no tile in a pass; its probe phase still runs)
            stage_prepare(t, lds);
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) stage_pair(i, lds);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (;;) {
                const int tn = next_tile(t);
                tile(std::integral_constant<int, 0>{}, t, tn);
                if (tn < 0) break;
                t = next_tile(tn);
                tile(std::integral_constant<int, 1>{}, tn, t);
                if (t < 0) break;
            }
        }
        if (pass == 1 || probe == 0) break;
#ifdef UM_WATTN_ASSUME_PASS          // diagnostic builds ONLY (wrong results for adversarial inputs): what the probe phase costs
        break;                       // (profiles/r06_probe_cost.txt)
#endif

        // ---- probe phase.  Every key of a `probe` tile is of another class than all 128 queries of the workgroup: its logit is
        // q.k / sqrt(C) - 100.  The own-class tiles are done, so m (the running row maximum, raw units) is a lower bound of the
        // row's final maximum; a key whose logit stays UM_WATTN_SKIP_MARGIN below it contributes e^-margin of the largest term
        // at most -- times <= 2048 keys still orders of magnitude below fp32 resolution of the sum (and of every output channel:
        // the same weights multiply bounded values).  A tile is dropped iff that holds for EVERY (query, key) pair, judged on the
        // hi.hi product alone: K_hi of up to 2 NS tiles lands in the plane areas of a ring slot, 8 MFMAs per tile (a sixth of the
        // tile's 48), in-lane maximum, one vote.  The probe's error against the exact product is <= 2^-10 |q| |k| / sqrt(C) --
        // part of the margin.  Tiles that fail the test in ANY wave are computed in pass 1 exactly as they were before round 6.
        {
            constexpr int G = 2 * NS;                                   // tiles per ring slot (one K_hi plane each)
            unsigned long long pm = probe, myfail = 0;
            unsigned* failw = reinterpret_cast<unsigned*>(lds + BUF + BIAS_OFF);      // slot 1's bias table is idle here
            if (tid == 0) {
                failw[0] = 0u;
                failw[1] = 0u;
            }
            const float thr = m + a.skip_raw;
            // (a group that runs out of tiles re-stages its last one: every group has G valid tiles, and the product below has no branch)
            int last_tile = 0;
            auto pstage = [&](int (&g)[G], unsigned char* base) {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    if (pm != 0) {
                        last_tile = (int)__builtin_ctzll(pm);
                        pm &= pm - 1;
                    }
                    g[i] = last_tile;
                    const unsigned* tp = tab + last_tile * TK + 8 * wave + ((lane >> 4) & 3);
                    const unsigned o0 = 2u * (((unsigned)kvbase + (tp[0] >> 2)) * (unsigned)a.ldkv + (unsigned)ssrc_k[0]);
                    const unsigned o1 = 2u * (((unsigned)kvbase + (tp[4] >> 2)) * (unsigned)a.ldkv + (unsigned)ssrc_k[1]);
                    const unsigned dst = __builtin_amdgcn_readfirstlane(
                        (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)(base + i * PLANE + (8 * wave) * 256));
                    const unsigned short* b0 = a.kp;
                    const unsigned short* b1 = a.kp - 512;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                                 "global_load_lds_dwordx4 %2, %4 offset:1024\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(o0), "v"(o1), "s"(b0), "s"(b1), "s"(dst) : "memory");
                }
            };
            // two tiles at a time on two accumulators: a lone chain of 8 dependent MFMAs behind 8 exposed LDS reads ran the matrix pipe at
            // a fraction of its rate (the probe cost 2.5 x its MFMA share, profiles/r06_probe_cost.txt); interleaved, each chain's
            // dependency and fragment latency hide under the other's MFMAs
            auto pcompute = [&](const int (&g)[G], const unsigned char* base) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < G; i += 2) {
                    f32x16 pa, pb;
#pragma unroll
                    for (int r = 0; r < 16; ++r) pa[r] = pb[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const i16x8 fa = *reinterpret_cast<const i16x8*>(base + i * PLANE + koff[ks]);
                        const i16x8 fb = *reinterpret_cast<const i16x8*>(base + (i + 1) * PLANE + koff[ks]);
                        pa = T::mfma(fa, qf[0][ks], pa);
                        pb = T::mfma(fb, qf[0][ks], pb);
                    }
                    float mxa = pa[0], mxb = pb[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) {
                        mxa = fmaxf(mxa, pa[r]);
                        mxb = fmaxf(mxb, pb[r]);
                    }
                    if (__any(!(mxa <= thr))) myfail |= 1ull << g[i];
                    if (__any(!(mxb <= thr))) myfail |= 1ull << g[i + 1];
                }
                __builtin_amdgcn_s_setprio(0);
            };
            int g0[G], g1[G];
            pstage(g0, lds);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (;;) {
                const bool more = pm != 0;
                if (more) pstage(g1, lds + BUF);
                pcompute(g0, lds);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (!more) break;
                const bool more2 = pm != 0;
                if (more2) pstage(g0, lds);
                pcompute(g1, lds + BUF);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (!more2) break;
            }
            if (lane == 0 && myfail != 0) {
                __hip_atomic_fetch_or(failw, (unsigned)myfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(failw + 1, (unsigned)(myfail >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            fail = (unsigned long long)__builtin_amdgcn_readfirstlane(failw[0]) |
                   ((unsigned long long)__builtin_amdgcn_readfirstlane(failw[1]) << 32);
        }
    }
    if (a.tile_census != nullptr && tid == 0) {
        const unsigned long long nfail = (unsigned long long)__builtin_popcountll(fail);
        const unsigned long long nfull = (unsigned long long)(bymask ? __builtin_popcountll(walk) : t1 - t0) + nfail;
        __hip_atomic_fetch_add(a.tile_census + 0, nfull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(a.tile_census + 1, (unsigned long long)__builtin_popcountll(probe), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(a.tile_census + 2, nfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(a.tile_census + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if constexpr (KSPLIT) {
        // The slots are written and read ONLY by 16-byte agent-scope accesses (sc1: write-through / L1-bypassing; the parts of a tile
        // share an XCD and therefore an L2); a release / acquire FENCE at agent scope would write back / invalidate the XCD's whole
        // L2 (measured).  What is needed is the completion of the stores before the ticket is taken: vmcnt(0) + the barrier.
        // Slot layout: 17 vectors of 16 bytes per thread, [vector][thread] -- O^T (16 vectors: tile dt, register group g), (M, l, -, -).
        constexpr int KS_SLOT = 17 * 256 * 4;                        // floats per slot
        if (nsplit > 1) {
            float* mine = a.ks_part + ((long)wl * nsplit + part) * KS_SLOT + 4 * tid;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    st_agent_16B(mine + (dt * 4 + g) * 1024, v);
                }
            {
                const f32x4 v = {M, l, 0.f, 0.f};
                st_agent_16B(mine + 16 * 1024, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                         // every thread's stores are complete
            unsigned* ticket = reinterpret_cast<unsigned*>(lds);     // (the K/V ring is idle now)
            if (tid == 0) *ticket = __hip_atomic_fetch_add(a.ks_flag + wl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*ticket != (unsigned)(nsplit - 1)) return;           // not the last arriver: done
            // Acquire side, last arriver only (one workgroup per tile): invalidate non-local lines at agent scope so that the sc1
            // loads below cannot hit a copy of a slot this XCD's L2 kept from an EARLIER launch if the parts ever land on
            // different XCDs (the dispatch-order assumption behind xcd_remap is "speed only").  Unlike an agent-scope RELEASE
            // this writes nothing back.
            asm volatile("buffer_inv sc1" ::: "memory");
            __syncthreads();                                         // (the ring is about to take Wm)
            if (tid == 0) __hip_atomic_store(a.ks_flag + wl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next launch
            // merge in part order, starting from part 0's slot (so that the result does not depend on who arrived last)
            for (int p = 0; p < nsplit; ++p) {
                const float* pr = a.ks_part + ((long)wl * nsplit + p) * KS_SLOT + 4 * tid;
                const f32x4 ml = ld_agent_16B(pr + 16 * 1024);
                const float Mo = ml[0], lo = ml[1];
                const float Ms = p == 0 ? Mo : fminf(M, Mo);          // offsets are integers: the factors are powers of two
                const float fa = p == 0 ? 0.f : fast_exp2(Ms - M), fb = fast_exp2(Ms - Mo);
                l = l * fa + lo * fb;
                M = Ms;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    f32x4 w[4];
                    const float* q = pr + dt * 4 * 1024;
                    ld_agent_16Bx4(q, q + 1024, q + 2048, q + 3072, w[0], w[1], w[2], w[3]);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[dt][4 * g + i] = o[dt][4 * g + i] * fa + w[g][i] * fb;
                }
            }
        }
    }

#ifdef UM_TRACE
    if (tracing) trace_buf[24 * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
    // ---- normalise and scatter back to the original token positions -----------------------------------------
    float lt;
    {
        float u, v2;
        half_wave_pair(l, u, v2);
        lt = u + v2;
    }
    const float inv = 1.0f / lt;
    if (!MERGE) {
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
        return;
    }
    // ---- MERGE: out = LayerNorm(message . Wm^T) (+ residual), message = O / l still in the accumulators -------------------
    // O^T goes accumulator -> operand exactly like P^T (k-step ks of 16 channels = registers 8 (ks & 1) .. +7 of tile ks >> 1);
    // Wm (planes [NS][128][128]) is pulled into the now idle K/V ring by LDS-DMA with the K tile's swizzle while the
    // fragments are being converted; Y^T = Wm . O^T keeps lane = query, so the LayerNorm statistics are in-lane sums plus
    // one exchange with lane ^ 32.
    {
        const int row4 = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * (8 * wave + i) + row4;             // wave w stages rows 32 w .. 32 w + 31
            const int c = pc ^ (row & 15);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                lds_dma16(a.wm + pl * a.wm_plane_stride + row * UM_CHANNELS + 8 * c, lds + pl * 32768 + (4 * (8 * wave + i)) * 256);
        }
    }
    i16x8 of[NS][8];
    float omx = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int dt = ks >> 1, r0 = 8 * (ks & 1);
        unsigned wh[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p0 = o[dt][r0 + 2 * j] * inv, p1 = o[dt][r0 + 2 * j + 1] * inv;
            omx = fmaxf(omx, fmaxf(__builtin_fabsf(p0), __builtin_fabsf(p1)));
            wh[j] = T::pack2(p0, p1);
            if (NS == 2) wl[j] = T::lo2(p0, p1, wh[j], neg1);
        }
        {
            const auto x = __builtin_amdgcn_permlane32_swap(wh[0], wh[2], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(wh[1], wh[3], false, false);
            const u32x4 f = {x[0], y[0], x[1], y[1]};
            of[0][ks] = __builtin_bit_cast(i16x8, f);
        }
        if (NS == 2) {
            const auto x = __builtin_amdgcn_permlane32_swap(wl[0], wl[2], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(wl[1], wl[3], false, false);
            const u32x4 f = {x[0], y[0], x[1], y[1]};
            of[NS - 1][ks] = __builtin_bit_cast(i16x8, f);
        }
    }
    um_range_note<T>(a.range_flag, omx, UM_RANGE_ATTN_OUTPUT);
    // the residual rows are requested HERE, the merge GEMM ahead of their use (the accumulators' 64 registers are free since the
    // conversion above): their latency was exposed at the very end of the workgroup
    f32x4 rres[16];
    if (a.residual != nullptr && tq < a.n) {
        const float* rb0 = a.residual + (sbase + tokq) * UM_CHANNELS + 4 * half;
#pragma unroll
        for (int q = 0; q < 16; ++q) rres[q] = *reinterpret_cast<const f32x4*>(rb0 + 8 * q);
    }
    f32x16 yv[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[ot][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const i16x8 wh = *reinterpret_cast<const i16x8*>(lds + ot * (32 * 256) + koff[ks]);
                if (NS == 2) {
                    const i16x8 wl = *reinterpret_cast<const i16x8*>(lds + 32768 + ot * (32 * 256) + koff[ks]);
                    yv[ot] = T::mfma(wl, of[0][ks], yv[ot]);
                    yv[ot] = T::mfma(wh, of[NS - 1][ks], yv[ot]);
                }
                yv[ot] = T::mfma(wh, of[0][ks], yv[ot]);
            }
    }
    float s1 = 0.f;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            yv[ot][r] *= a.wm_scale;
            s1 += yv[ot][r];
        }
    float u, v2;
    half_wave_pair(s1, u, v2);
    const float mean = (u + v2) * (1.0f / 128.0f);
    float s2 = 0.f;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = yv[ot][r] - mean;
            s2 = __builtin_fmaf(d, d, s2);
        }
    half_wave_pair(s2, u, v2);
    const float rstd = 1.0f / sqrtf((u + v2) * (1.0f / 128.0f) + a.eps);
    if (tq < a.n) {
        // lane holds, for its query, features 32 ot + 8 g + 4 half + i  (reg 4 g + i of tile ot)
        float* ob = a.out + (sbase + tokq) * UM_CHANNELS + 4 * half;
        const float* rb = a.residual ? a.residual + (sbase + tokq) * UM_CHANNELS + 4 * half : nullptr;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * ot + 8 * g;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + n + 4 * half);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + n + 4 * half);
                f32x4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (yv[ot][4 * g + i] - mean) * rstd * gm[i] + bt[i];
                if (rb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] += rres[4 * ot + g][i];
                }
                *reinterpret_cast<f32x4*>(ob + n) = y;
            }
    }
}

// (Round 5 built window_attn8_kernel here -- one 8-wave workgroup per CU and query tile, two wave groups in anti-phase, exactly three rounds
// at config 2 -- parity-green, measured +4.7 % slower than the kernel above, and removed it: profiles/r05_attention_experiments.txt has the
// design, the same-call timings and the section stamps; the code is in git at 06bbabb .. 4c4d2cd.)

// ------------------------------------------------------------------------------------ host side
extern void um_set_error(const char* fmt, ...);

static size_t align256w(size_t x) { return (x + 255) & ~(size_t)255; }

#ifdef UM_TRACE
extern "C" int um_debug_set_trace(void* ptr) {
    unsigned long long* p = (unsigned long long*)ptr;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_um_trace), &p, sizeof(p));
}
#endif

// ---- tile census (um_window_attn_tile_census): four 64-bit counters in device memory, per device, counted by one thread per
// workgroup while enabled: key tiles computed in full / wholly masked tiles probed / probed tiles that had to be computed after all /
// workgroups.  Off by default (the kernels get a null pointer).
static unsigned long long* g_tile_census[64] = {nullptr};
static bool g_tile_census_on = false;
static unsigned long long* wattn_census_ptr() {
    if (!g_tile_census_on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_tile_census[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 4 * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        (void)hipMemset(p, 0, 4 * sizeof(unsigned long long));
        g_tile_census[dev] = (unsigned long long*)p;
    }
    return g_tile_census[dev];
}

extern "C" int um_window_attn_tile_census(int enable, unsigned long long* counts4) {
    // counts4 != nullptr: the counters of the current device as of all work submitted so far (synchronises the device), then zeroed
    if (counts4) {
        counts4[0] = counts4[1] = counts4[2] = counts4[3] = 0;
        const bool was = g_tile_census_on;
        g_tile_census_on = true;
        unsigned long long* p = wattn_census_ptr();
        g_tile_census_on = was;
        if (p) {
            hipError_t e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(counts4, p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemset(p, 0, 4 * sizeof(unsigned long long));
            if (e != hipSuccess) return (int)e;
        }
    }
    g_tile_census_on = enable != 0;
    return 0;
}

#define UM_WATTN_SKIP_MARGIN 40.0f      // natural-log units below the running row maximum at which a masked key is dropped

static int launch_window_attn(const unsigned short* pq, const unsigned short* pk, const unsigned short* pv, float* out,
                              int streams, int h, int w, int ldq, int ldkv, long q_plane_stride, long kv_plane_stride,
                              int win_h, int win_w, int shift_h, int shift_w, int kv_rotate, int mode, hipStream_t stream,
                              const unsigned short* wm = nullptr, const float* gamma = nullptr, const float* beta = nullptr,
                              const float* residual = nullptr, float eps = 0.f, int wshift = 0, const float* x = nullptr,
                              const unsigned short* wq = nullptr, void* ks_ws = nullptr, size_t ks_ws_bytes = 0);

// ---- key split for small launches (see KSPLIT in the kernel): while tiles x parts fit the chip (two workgroups per CU)
static int wattn_num_cus() { return um_num_cus(); }      // per device (common.h)

static int wattn_key_split(int total, int ntiles) {
    static const bool off = um_debug_env("UM_WATTN_NO_KSPLIT") != nullptr;      // A/B switch (diagnostic builds)
    if (off) return 1;
    const int cus = wattn_num_cus();
    int split = 1;
    // all workgroups resident at once (two per CU), at least 4 key tiles per part
    while (split < 4 && total * (2 * split) <= 2 * cus && ntiles >= 4 * (2 * split)) split *= 2;
    return split;
}

// ---- launch plan: a pure function of the geometry (and the device's CU count).  The chip holds 2 * CUs workgroups at once.
//   * a small launch (total <= slots): every tile key-split `split` ways while the launch fits the chip (batch-1 latency);
//   * a big launch: one workgroup per tile.
// Round 3 measured five restructurings of the big launch on the GPU and dropped them all (profiles/r03_attention_experiments.txt;
// code in git at 85b86af): the remainder round of 768 tiles on 512 slots key-split -- as a second launch
// (0.2514 against 0.2403 ms) and, with the ticket hand-off, at the end of the same grid (0.2570 against 0.2473 ms): the half walks
// pay a second prologue and the hand-off, and the tail they replace is less idle than a round count suggests; 256-query / 8-wave
// workgroups with a 4-slot K/V ring and the DMA three tiles ahead (equal per-round time); a software-pipelined one-wave-per-SIMD
// instantiation with Q^T and O^T in AGPRs (0.293 against 0.260 ms); alternating accumulators in QK^T / PV (0.262 against 0.241).
struct WattnPlan {
    int full;       // tiles served one workgroup each (0: none)
    int rem;        // tiles served key-split (0: none)
    int split;      // parts per key-split tile (1 when rem == 0)
};

static WattnPlan wattn_plan(int total, int ntiles, bool can_split) {
    const int slots = 2 * wattn_num_cus();
    // diagnostic builds (-DUM_DEBUG_SWITCHES): UM_WATTN_FORCE_SPLIT=2|4 key-splits EVERY tile of a big launch (config 2: 768 tiles ->
    // 1536 half walks = exactly 3 rounds of the 512 resident slots; profiles/r04_attention_experiments.txt)
    static const int force = [] { const char* e = um_debug_env("UM_WATTN_FORCE_SPLIT"); return e ? atoi(e) : 0; }();
    if (can_split && force > 1 && ntiles >= 4 * force) return {0, total, force};
    if (can_split && total <= slots) {
        const int split = wattn_key_split(total, ntiles);
        if (split > 1) return {0, total, split};
    }
    return {total, 0, 1};
}

static size_t wattn_ks_bytes(int tiles, int split) {
    return align256w((size_t)tiles * sizeof(unsigned)) + (size_t)tiles * split * (17 * 256 * 4 * sizeof(float));
}

extern "C" size_t um_window_attn_ksplit_workspace_bytes(int streams, int h, int w, int win_h, int win_w) {
    if (streams <= 0 || h <= 0 || w <= 0 || win_h <= 0 || win_w <= 0 || h % win_h || w % win_w) return 0;
    const int n = win_h * win_w, total = ((n + 127) / 128) * (h / win_h) * (w / win_w) * streams;
    const WattnPlan p = wattn_plan(total, (n + 31) / 32, true);
    return p.rem > 0 ? wattn_ks_bytes(p.rem, p.split) : 0;
}

// launch plan of um_window_attn_qproj_merge_fwd for a geometry: tiles served whole / tiles served key-split / parts per split tile
extern "C" int um_window_attn_plan(int streams, int h, int w, int win_h, int win_w, int* full_tiles, int* split_tiles, int* parts) {
    if (streams <= 0 || h <= 0 || w <= 0 || win_h <= 0 || win_w <= 0 || h % win_h || w % win_w || !full_tiles || !split_tiles || !parts)
        return UM_ERR_BAD_ARG;
    const int n = win_h * win_w, total = ((n + 127) / 128) * (h / win_h) * (w / win_w) * streams;
    const WattnPlan p = wattn_plan(total, (n + 31) / 32, true);
    *full_tiles = p.full;
    *split_tiles = p.rem;
    *parts = p.split;
    return 0;
}

static int check_attn_geometry(int streams, int h, int w, int channels, int win_h, int win_w, int shift_h, int shift_w,
                               int mode) {
    if (streams <= 0 || h <= 0 || w <= 0) {
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
    return 0;
}

extern "C" int um_window_attn_planes_fwd(const void* qp, const void* kp, const void* vp, float* out, int streams,
                                         int h, int w, int channels, int ldq, int ldkv, long q_plane_stride,
                                         long kv_plane_stride, int win_h, int win_w, int shift_h, int shift_w,
                                         int kv_rotate, int mode, void* stream) {
    if (!qp || !kp || !vp || !out) {
        um_set_error("null pointer");
        return -1;
    }
    if (int e = check_attn_geometry(streams, h, w, channels, win_h, win_w, shift_h, shift_w, mode)) return e;
    if (ldq < UM_CHANNELS || ldkv < UM_CHANNELS || ldq % 8 || ldkv % 8) {
        um_set_error("row strides ldq=%d ldkv=%d must be multiples of 8 and >= %d", ldq, ldkv, UM_CHANNELS);
        return -1;
    }
    return launch_window_attn((const unsigned short*)qp, (const unsigned short*)kp, (const unsigned short*)vp, out,
                              streams, h, w, ldq, ldkv, q_plane_stride, kv_plane_stride, win_h, win_w, shift_h, shift_w,
                              kv_rotate, mode, (hipStream_t)stream);
}

extern "C" int um_window_attn_merge_fwd(const void* qp, const void* kp, const void* vp, const void* wm_planes, const float* gamma,
                                        const float* beta, const float* residual, float eps, int wshift, float* out, int streams,
                                        int h, int w, int channels, int ldq, int ldkv, long q_plane_stride,
                                        long kv_plane_stride, int win_h, int win_w, int shift_h, int shift_w, int kv_rotate,
                                        int mode, void* stream) {
    if (!qp || !kp || !vp || !out || !wm_planes || !gamma || !beta || wshift < 0 || wshift > 14) {
        um_set_error("um_window_attn_merge_fwd: null pointer or bad wshift");
        return -1;
    }
    if (int e = check_attn_geometry(streams, h, w, channels, win_h, win_w, shift_h, shift_w, mode)) return e;
    if (ldq < UM_CHANNELS || ldkv < UM_CHANNELS || ldq % 8 || ldkv % 8) {
        um_set_error("row strides ldq=%d ldkv=%d must be multiples of 8 and >= %d", ldq, ldkv, UM_CHANNELS);
        return -1;
    }
    return launch_window_attn((const unsigned short*)qp, (const unsigned short*)kp, (const unsigned short*)vp, out, streams, h, w,
                              ldq, ldkv, q_plane_stride, kv_plane_stride, win_h, win_w, shift_h, shift_w, kv_rotate, mode,
                              (hipStream_t)stream, (const unsigned short*)wm_planes, gamma, beta, residual, eps, wshift);
}

extern "C" int um_window_attn_qproj_merge_fwd(const float* x, const void* wq_planes, const void* kp, const void* vp,
                                              const void* wm_planes, const float* gamma, const float* beta, const float* residual,
                                              float eps, int wshift, float* out, int streams, int h, int w, int channels, int ldkv,
                                              long kv_plane_stride, int win_h, int win_w, int shift_h, int shift_w, int kv_rotate,
                                              int mode, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !wq_planes || !kp || !vp || !out || !wm_planes || !gamma || !beta || wshift < 0 || wshift > 14) {
        um_set_error("um_window_attn_qproj_merge_fwd: null pointer or bad wshift");
        return -1;
    }
    if (int e = check_attn_geometry(streams, h, w, channels, win_h, win_w, shift_h, shift_w, mode)) return e;
    if (ldkv < UM_CHANNELS || ldkv % 8) {
        um_set_error("row stride ldkv=%d must be a multiple of 8 and >= %d", ldkv, UM_CHANNELS);
        return -1;
    }
    return launch_window_attn(nullptr, (const unsigned short*)kp, (const unsigned short*)vp, out, streams, h, w, UM_CHANNELS, ldkv,
                              0, kv_plane_stride, win_h, win_w, shift_h, shift_w, kv_rotate, mode, (hipStream_t)stream,
                              (const unsigned short*)wm_planes, gamma, beta, residual, eps, wshift, x,
                              (const unsigned short*)wq_planes, workspace, workspace_bytes);
}

extern "C" size_t um_window_attn_workspace_bytes(int streams, int tokens, int channels, int mode) {
    if (streams <= 0 || tokens <= 0 || channels != UM_CHANNELS || (mode != 0 && mode != 1)) return 0;
    return 3 * align256w(planes_bytes((long)streams * tokens, mode));
}

extern "C" int um_window_attn_fwd(const float* q, const float* k, const float* v, float* out, int streams, int h,
                                  int w, int channels, int win_h, int win_w, int shift_h, int shift_w, int mode,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !k || !v || !out) {
        um_set_error("null pointer");
        return -1;
    }
    if (int e = check_attn_geometry(streams, h, w, channels, win_h, win_w, shift_h, shift_w, mode)) return e;
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

    return launch_window_attn(pq, pk, pv, out, streams, h, w, UM_CHANNELS, UM_CHANNELS, streams * L * UM_CHANNELS,
                              streams * L * UM_CHANNELS, win_h, win_w, shift_h, shift_w, 0, mode, stream);
}

static int launch_window_attn(const unsigned short* pq, const unsigned short* pk, const unsigned short* pv, float* out,
                              int streams, int h, int w, int ldq, int ldkv, long q_plane_stride, long kv_plane_stride,
                              int win_h, int win_w, int shift_h, int shift_w, int kv_rotate, int mode, hipStream_t stream,
                              const unsigned short* wm, const float* gamma, const float* beta, const float* residual,
                              float eps, int wshift, const float* x, const unsigned short* wq, void* ks_ws, size_t ks_ws_bytes) {
    // the staging sources are 32-bit byte offsets from the k / v plane bases (UM_WATTN_OFF32): one plane must stay below 4 GiB
    if (UM_WATTN_OFF32 && (long)streams * h * w * ldkv * 2 >= (1L << 32)) {
        um_set_error("window attention: a k / v operand plane of %ld bytes is beyond the 4 GiB this kernel addresses",
                     (long)streams * h * w * ldkv * 2);
        return UM_ERR_UNSUPPORTED;
    }
    WattnArgs a;
    a.x = x;
    a.wq = wq;
    a.split = 1;
    a.ks_part = nullptr;
    a.ks_flag = nullptr;
    a.range_flag = (mode == 0) ? um_range_flag_dev() : nullptr;
    a.wm = wm;
    a.wm_plane_stride = (long)UM_CHANNELS * UM_CHANNELS;
    a.gamma = gamma;
    a.beta = beta;
    a.residual = residual;
    a.eps = eps;
    a.wm_scale = ldexpf(1.f, -wshift);
    a.streams = streams;
    a.kv_rotate = ((kv_rotate % streams) + streams) % streams;
    a.qp = pq;
    a.kp = pk;
    a.vp = pv;
    a.plane_stride = q_plane_stride;
    a.kv_plane_stride = kv_plane_stride;
    a.ldq = ldq;
    a.ldkv = ldkv;
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
    a.scale_log2 = UM_LOG2E / sqrtf((float)UM_CHANNELS);
    a.mask_raw = -100.0f * sqrtf((float)UM_CHANNELS);
    static const float headroom = [] { const char* e = um_debug_env("UM_WATTN_HEADROOM"); return e ? (float)atof(e) : 8.f; }();
    a.headroom = (mode == 0) ? headroom : 0.f;
    // masked-tile skip and heaviest-first order: a pure function of the call (diagnostic builds: UM_WATTN_NO_SKIP / UM_WATTN_NO_LPT)
    static const bool no_skip = um_debug_env("UM_WATTN_NO_SKIP") != nullptr, no_lpt = um_debug_env("UM_WATTN_NO_LPT") != nullptr;
    const bool shifted = shift_h > 0 || shift_w > 0;
    a.skip_raw = (shifted && !no_skip) ? (100.0f - UM_WATTN_SKIP_MARGIN) * sqrtf((float)UM_CHANNELS) : -1.0f;
    a.spx = (shifted && !no_skip && !no_lpt && streams % 8 == 0) ? streams / 8 : 0;
    a.tile_census = wattn_census_ptr();
    ScopedKernelTimer timer(UM_K_WINDOW_ATTN, stream);
    if (wm && wq) {
        // the layer kernel (query projection + attention + merge + LayerNorm): wattn_plan; without workspace one workgroup per tile
        WattnPlan p = wattn_plan(a.total, (a.n + 31) / 32, ks_ws != nullptr);
        if (p.rem > 0 && ks_ws_bytes < wattn_ks_bytes(p.rem, p.split)) p = WattnPlan{a.total, 0, 1};
        if (p.rem == 0) {
            um_census_hit(UM_V_WATTN_TILE);
            if (mode == 0)
                hipLaunchKernelGGL((window_attn_kernel<Fp16, 2, true, true>), dim3(p.full), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL((window_attn_kernel<Bf16, 1, true, true>), dim3(p.full), dim3(256), 0, stream, a);
            return (int)hipGetLastError();
        }
        um_census_hit(UM_V_WATTN_KSPLIT);                          // (wattn_plan: a launch is all whole tiles or all key-split)
        a.split = p.split;
        a.ks_flag = (unsigned*)ks_ws;
        a.ks_part = (float*)((unsigned char*)ks_ws + align256w((size_t)p.rem * sizeof(unsigned)));
        const unsigned grid = (unsigned)(p.rem * p.split);
        if (mode == 0)
            hipLaunchKernelGGL((window_attn_kernel<Fp16, 2, true, true, true>), dim3(grid), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((window_attn_kernel<Bf16, 1, true, true, true>), dim3(grid), dim3(256), 0, stream, a);
        return (int)hipGetLastError();
    }
    um_census_hit(UM_V_WATTN_TILE);
    if (wm) {
        if (mode == 0)
            hipLaunchKernelGGL((window_attn_kernel<Fp16, 2, true>), dim3(a.total), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((window_attn_kernel<Bf16, 1, true>), dim3(a.total), dim3(256), 0, stream, a);
    } else if (mode == 0)
        hipLaunchKernelGGL((window_attn_kernel<Fp16, 2, false>), dim3(a.total), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((window_attn_kernel<Bf16, 1, false>), dim3(a.total), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
