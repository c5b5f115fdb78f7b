// K4 on the matrix cores: local cost volume at flow-displaced positions for locally coherent flow.   gfx950 / wave64 / MFMA
//
//   cost[b, (dy, dx), p] = < f0(p), bilinear f1(p + (dx, dy) + flow(p)) > / sqrt(C),   |dx|, |dy| <= 4, zeros outside the image
//   (unimatch/matching.py:86-123, called from the refinement loop unimatch/unimatch.py:315-331)
//
// local_corr_with_flow_kernel (local_ops.hip) turns the 81 x 4 bilinear gathers of a pixel into the 100 dot products of its
// 10 x 10 integer neighbourhood plus 81 four-tap blends, and is bound by the L2 -> CU traffic of those neighbour rows (51 KB
// per pixel, 0.59 ms per call at 4 x 128 x 192).  A pixel alone is a GEMV -- nothing for the matrix cores -- but the 10 x 10
// neighbourhoods of ADJACENT pixels overlap almost completely when the flow is locally coherent, which is what a trained model
// produces (with the conditioned weights of unimatch_amd/synth.py 96-98 % of the 32-pixel groups of config 4 spread their integer
// flow over <= 6 pixels; at random init the flow is incoherent everywhere, profiles/r02_k4_flow_coherence.txt).  So:
//   * a wave owns an 8 x 4 pixel tile; the union of its neighbourhoods is a (8 + 9 + rx) x (4 + 9 + ry) window of f1;
//   * per window ROW one 32 x 32 x 128 product S^T = F1row . F0^T on the matrix cores (A = up to 32 window positions of the
//     row, read straight from the fp16 hi | lo operand planes of f1 in L2; B = the tile's 32 pixels; exact mode: 3 products),
//     24 MFMAs per row, <= 24 rows: every dot product a pixel needs (and ~3x as many it does not) at ~10 KB per pixel;
//   * the accumulators are scattered to the per-pixel [10 x 10] dot tables in LDS (lane = pixel: the slot of a register is a
//     per-lane base plus a constant), then the same four-tap blends as the VALU kernel, and the same two output forms
//     ([B, 81, h, w] fp32, or the channels-last operand planes of the motion encoder's 1x1 convolution);
//   * tiles whose window does not fit 32 x 24 (motion boundaries; everything at random init) take the pixel-at-a-time VALU
//     path inside the same kernel -- the dot products of that path are the VALU kernel's.
// The feature planes are split once per scale (um_local_corr_feat_planes) and serve every refinement iteration.
//
// Target-ordered tiles (round 5).  Where the flow is INCOHERENT (random-init weights; motion boundaries) a natural 8 x 4 tile
// has no common window: round 4 measured the pixel path at 20 x its compulsory bytes (2.1 GB per launch at 4 x 128 x 192) with
// the waves parked 69 % of the time.  But pixels whose TARGETS lie in one 16 x 8 cell of f1 always share a 26 x 18 window,
// wherever they sit in f0.  So a launch first takes a census of its natural tiles (k4s_census_kernel); if more than a quarter
// of them are incoherent the pixels are counting-sorted by (image, target cell) -- a few large workgroups each histogram their
// share of the pixels in LDS and publish the counts, the last one to finish turns the [workgroup][cell] counts into start slots
// (two-level prefix sum), and k4s_scatter_kernel's workgroups hand their pixels out from those slots with LDS atomics: no
// global atomic per pixel or per cell anywhere (a first version had them: 98 k device-scope atomics on 32 cache lines made
// census + scatter 153 us of the launch's 288) -- into per-cell groups padded to whole 32-pixel tiles, and
// k4m_kernel walks those groups: lane = any pixel of the group, everything downstream (B operand, scatter addresses, blend
// weights, output row) was per lane already.  Pixels whose whole neighbourhood misses the image go to one group that writes
// zeros.  Properties kept: the decision is a pure function of the call's flow (no state, no host read-back); a pixel's 100 dot
// products do not depend on which other pixels share its tile (one MFMA column per pixel), and every sorted tile is coherent by
// construction, so the result does not depend on the order the scatter's atomics happen to produce: bitwise reproducible.
#include "common.h"
#include "planes.h"

extern void um_set_error(const char* fmt, ...);

#define K4M_RADIUS 4
#define K4M_KW 9
#define K4M_N1 10
#define K4M_TAPS 81
#define K4M_TW 8
#define K4M_TH 4
#define K4M_MAXROWS 24
#ifndef K4S_CW
#define K4S_CW 16                // target cell: 16 x 8 positions of f1  ->  window (16 + 10) x (8 + 10) <= 32 x 24
#define K4S_CH 8
#endif

struct K4mArgs {
    const float* f0;             // [B][L][128] fp32 tokens (VALU path)
    const float* f1;
    const unsigned short* fp0;   // operand planes [2][B*L][128] fp16 hi | lo of f0, f1 (MFMA path)
    const unsigned short* fp1;
    long plane_stride;
    const float* flow;           // [B][2][h][w]
    float* cost;                 // [B][81][h][w]   (or null)
    unsigned short* planes;      // [2][rows + 1][ld]   (or null)
    long out_plane_stride;
    int ld;
    int batch, h, w;
    int force_valu;              // diagnostics: every tile on the VALU path
    // local_correlation_softmax (matching.py:39-83) on the same machinery: flow == nullptr (all windows are the pixel's own
    // 9 x 9 neighbourhood: every tile is coherent), softmax over the 81 taps, expected offset -> flow_out [B][2][h][w]
    float* flow_out;
    unsigned* stats;             // optional: [0] += tiles on the product path, [1] += tiles on the pixel path (adaptive dispatch)
    // target-ordered mode (see the header): scratch of the launch, filled by the k4s_* kernels that run in front of it
    const int* perm;             // [ncell groups padded to 32] pixel ids (b * L + p), -1 = empty slot
    const unsigned* ctl;         // [0] = 1: walk the sorted groups, [1] = number of sorted tiles
};

// ---- scratch layout of the target-ordered mode (inside the feature-plane workspace, behind the planes)
#define K4S_WG 1024              // threads of a sorting workgroup
#define K4S_MAXG 32              // sorting workgroups per launch (the last one reads all their counts in one batch of loads)
#define K4S_MAXGROUPS 15360      // (image, cell) groups whose counters fit a workgroup's LDS (60 KB); beyond: natural tiles only
struct K4sLayout {
    int cx, cy, ncell;           // cells per image: (cx x cy) inside + 1 "nothing to sample" cell
    long rows;
    int ng;                      // sorting workgroups
    long span;                   // (natural tile, slot) units per sorting workgroup: whole waves
    bool ok;                     // the geometry can be target-ordered at all
    size_t off_ctl, off_part, off_cell, off_perm, bytes;
};
static inline K4sLayout k4s_layout(int batch, int h, int w) {
    K4sLayout l;
    l.cx = (w + K4S_CW - 1) / K4S_CW + 2;        // targets up to a window outside the image still touch it
    l.cy = (h + K4S_CH - 1) / K4S_CH + 2;
    l.ncell = l.cx * l.cy + 1;
    l.rows = (long)batch * h * w;
    const size_t nb = (size_t)batch * l.ncell;
    l.ok = nb <= K4S_MAXGROUPS;
    l.ng = (int)((l.rows + K4S_WG - 1) / K4S_WG);
    if (l.ng > K4S_MAXG) l.ng = K4S_MAXG;
    l.span = ((l.rows + l.ng - 1) / l.ng + 63) & ~63L;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 255) & ~(size_t)255; return at; };
    l.off_ctl = take(64);                        // [0] mode, [1] sorted tiles, [2] incoherent natural tiles, [3] arrival ticket
    l.off_part = take((size_t)K4S_MAXG * nb * 4);  // [workgroup][group]: counts, then start slots
    l.off_cell = take((size_t)l.rows * 4);       // group of every pixel, in (natural tile, slot) order
    l.off_perm = take(((size_t)l.rows + 32 * nb) * 4);
    l.bytes = o;
    return l;
}

// cell of a pixel's target (wave-uniform geometry, per-lane flow): cells are laid over f1 shifted by one cell so that targets up
// to a cell outside the image keep their neighbours; a pixel none of whose 10 x 10 positions can lie inside goes to the last cell
__device__ __forceinline__ int k4s_cell(int bx, int by, int h, int w, int cx, int cy) {
    const bool nothing = bx + K4M_N1 - 1 - K4M_RADIUS < 0 || bx - K4M_RADIUS >= w || by + K4M_N1 - 1 - K4M_RADIUS < 0 || by - K4M_RADIUS >= h;
    if (nothing) return cx * cy;
    const int ix = min(max((bx + K4S_CW) / K4S_CW, 0), cx - 1), iy = min(max((by + K4S_CH) / K4S_CH, 0), cy - 1);
    return iy * cx + ix;
}

// exclusive prefix sum over the 1024 threads of a sorting workgroup (wsum: 16 words of LDS; *total = the sum)
__device__ __forceinline__ unsigned k4s_block_exscan(unsigned v, unsigned* wsum, unsigned* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned u = __shfl_up(inc, s);
        if (lane >= s) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const unsigned w = lane < K4S_WG / 64 ? wsum[lane] : 0u;
        unsigned winc = w;
#pragma unroll
        for (int s = 1; s < K4S_WG / 64; s <<= 1) {
            const unsigned u = __shfl_up(winc, s);
            if (lane >= s) winc += u;
        }
        if (lane < K4S_WG / 64) wsum[lane] = winc - w;
        if (lane == K4S_WG / 64 - 1) *total = winc;
    }
    __syncthreads();
    return wsum[wave] + inc - v;
}

// (natural tile, slot) unit t -> image b, pixel p of the image, its coordinates
__device__ __forceinline__ void k4s_pixel(long t, int L, int tiles_x, int w, int& b, int& p, int& x, int& y) {
    const long tile = t >> 5;
    const int n = (int)(t & 31);
    b = (int)(tile / (L / 32));
    const int tt = (int)(tile - (long)b * (L / 32));
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    x = tx * K4M_TW + (n & 7);
    y = ty * K4M_TH + (n >> 3);
    p = y * w + x;
}

// Start slots from the published counts + the mode decision: run by ONE workgroup (the last census workgroup to finish).
// part [ng][ngroups]: counts in, start slots (in pixels) out; tot_s [ngroups] LDS.
__device__ __forceinline__ void k4s_scan(unsigned* ctl, unsigned* part, int* perm, int ngroups, int ng, unsigned natural_tiles,
                                         unsigned* tot_s, unsigned* wsum, unsigned* total_s) {
    const int tid = threadIdx.x;
    // (the counts were published by workgroups on every XCD with agent-scope stores: read them the same way, not through this XCD's L2)
    auto ld = [](const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const bool sorted = ld(ctl + 2) * 4u > natural_tiles;               // more than a quarter of the natural tiles incoherent
    __syncthreads();                                                     // every thread has read the census word before it is reset
    if (!sorted) {
        if (tid == 0) {
            ctl[0] = 0u;
            ctl[1] = 0u;
            ctl[2] = 0u;                                                 // census word and ticket back to zero for the next launch
            ctl[3] = 0u;
        }
        return;
    }
    // a thread owns `per` consecutive groups; all of a group's ng counts are requested before the first is used
    const int per = (ngroups + K4S_WG - 1) / K4S_WG;
    const int c0 = tid * per, c1 = min(ngroups, c0 + per);
    unsigned mine = 0;
    for (int c = c0; c < c1; ++c) {
        unsigned v[K4S_MAXG];
#pragma unroll
        for (int g = 0; g < K4S_MAXG; ++g) v[g] = g < ng ? ld(part + (long)g * ngroups + c) : 0u;
        unsigned tot = 0;
#pragma unroll
        for (int g = 0; g < K4S_MAXG; ++g) tot += v[g];
        tot_s[c] = tot;
        mine += (tot + 31) / 32;
    }
    unsigned run = k4s_block_exscan(mine, wsum, total_s);               // first tile of this thread's first group
    for (int c = c0; c < c1; ++c) {
        unsigned v[K4S_MAXG];
#pragma unroll
        for (int g = 0; g < K4S_MAXG; ++g) v[g] = g < ng ? ld(part + (long)g * ngroups + c) : 0u;
        const unsigned tot = tot_s[c];
        unsigned at = run * 32u;
#pragma unroll
        for (int g = 0; g < K4S_MAXG; ++g)
            if (g < ng) {
                part[(long)g * ngroups + c] = at;                        // read by the scatter kernel (next launch on the stream)
                at += v[g];
            }
        run += (tot + 31) / 32;
        for (unsigned sl = at; sl < run * 32u; ++sl) perm[sl] = -1;      // the padding of the group's last tile
    }
    if (tid == 0) {
        ctl[0] = 1u;
        ctl[1] = *total_s;
        ctl[2] = 0u;
        ctl[3] = 0u;
    }
}

// census of the natural tiles + per-workgroup histogram of the pixels' groups (grid = ng workgroups of 1024 threads, each owning a
// contiguous span of (natural tile, slot) units: one natural 8 x 4 tile per 32 consecutive threads); the last workgroup scans
__global__ __launch_bounds__(K4S_WG) void k4s_census_kernel(const float* flow, int batch, int h, int w, int cx, int cy, unsigned* ctl,
                                                            unsigned* part, int* cell_t, int* perm, int ngroups, long span,
                                                            unsigned natural_tiles) {
    extern __shared__ unsigned hist_s[];                                 // [ngroups]
    __shared__ unsigned wsum[K4S_WG / 64];
    __shared__ unsigned total_s, ticket_s, incoh_s;
    const int tid = threadIdx.x;
    const int L = h * w, tiles_x = w / K4M_TW;
    const long rows = (long)batch * L;
    for (int i = tid; i < ngroups; i += K4S_WG) hist_s[i] = 0u;
    if (tid == 0) incoh_s = 0u;
    __syncthreads();
    const long t0 = (long)blockIdx.x * span, t1 = min(t0 + span, rows);
    for (long base = t0; base < t1; base += K4S_WG) {                    // (spans and rows are whole half-waves: a natural tile is never split)
        const long t = base + tid;
        const bool active = t < t1;
        int b, p, x, y;
        k4s_pixel(active ? t : t1 - 1, L, tiles_x, w, b, p, x, y);
        const float fbx = floorf((float)x + flow[((long)b * 2 + 0) * L + p]), fby = floorf((float)y + flow[((long)b * 2 + 1) * L + p]);
        const int bx = (int)fminf(fmaxf(fbx, -32768.f), 32768.f), by = (int)fminf(fmaxf(fby, -32768.f), 32768.f);
        // the natural tile's window, as k4m_kernel computes it (the 32 lanes of one half-wave hold one tile here)
        int x0 = bx, x1 = bx, y0 = by, y1 = by;
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) {
            x0 = min(x0, __shfl_xor(x0, s));
            x1 = max(x1, __shfl_xor(x1, s));
            y0 = min(y0, __shfl_xor(y0, s));
            y1 = max(y1, __shfl_xor(y1, s));
        }
        if (active) {
            const bool coherent = x1 - x0 + K4M_N1 <= 32 && y1 - y0 + K4M_N1 <= K4M_MAXROWS;
            if ((t & 31) == 0 && !coherent) atomicAdd(&incoh_s, 1u);
            const int cell = b * (cx * cy + 1) + k4s_cell(bx, by, h, w, cx, cy);
            cell_t[t] = cell;
            atomicAdd(hist_s + cell, 1u);
        }
    }
    __syncthreads();
    // publish without an agent-scope fence (that would write back the XCD's L2): agent-scope stores, waited for; the census word's atomic
    // RETURNS, so it has been performed when its result is here; the ticket is taken behind the barrier; the last arriver reads everything
    // with agent-scope loads
    for (int i = tid; i < ngroups; i += K4S_WG)
        __hip_atomic_store(part + (long)blockIdx.x * ngroups + i, hist_s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned seen = 0;
    if (tid == 0 && incoh_s) seen = atomicAdd(ctl + 2, incoh_s);
    asm volatile("s_waitcnt vmcnt(0)" : : "v"(seen) : "memory");
    __syncthreads();
    if (tid == 0) ticket_s = atomicAdd(ctl + 3, 1u);
    __syncthreads();
    if (ticket_s != gridDim.x - 1) return;
    k4s_scan(ctl, part, perm, ngroups, (int)gridDim.x, natural_tiles, hist_s, wsum, &total_s);
}

// pixel ids into their groups (sorted mode only): the workgroups of the census, over the same spans, hand their pixels out from their
// start slots with LDS atomics.  (The order inside a group depends on the order of those atomics; the result of k4m_kernel does not.)
__global__ __launch_bounds__(K4S_WG) void k4s_scatter_kernel(const unsigned* ctl, const int* cell_t, const unsigned* part, int* perm,
                                                             int batch, int h, int w, int ngroups, long span) {
    extern __shared__ unsigned cursor_s[];                               // [ngroups]
    if (ctl[0] == 0u) return;
    const int tid = threadIdx.x;
    const int L = h * w, tiles_x = w / K4M_TW;
    const long rows = (long)batch * L;
    for (int i = tid; i < ngroups; i += K4S_WG) cursor_s[i] = part[(long)blockIdx.x * ngroups + i];
    __syncthreads();
    const long t0 = (long)blockIdx.x * span, t1 = min(t0 + span, rows);
    for (long t = t0 + tid; t < t1; t += K4S_WG) {
        int b, p, x, y;
        k4s_pixel(t, L, tiles_x, w, b, p, x, y);
        const unsigned pos = atomicAdd(cursor_s + cell_t[t], 1u);
        perm[pos] = b * L + p;
    }
}


__device__ __forceinline__ int k4m_wave_min(int v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v = min(v, __shfl_xor(v, s));
    return v;
}
__device__ __forceinline__ int k4m_wave_max(int v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v = max(v, __shfl_xor(v, s));
    return v;
}

__global__ __launch_bounds__(128) void k4m_kernel(K4mArgs a) {
    // per wave: the [slot][pixel] dot tables (one spare slot: the blend reads one past a row); the [tap][pixel] output tile
    // takes the same storage once the tables are consumed (the blends are held in registers across a wave barrier).
    // Occupancy is deliberately low (one wave per SIMD: 300+ registers for the double-buffered window rows): measured, both
    // paths are bound by L2 -> CU traffic and lose with more waves in flight (2-3 waves per SIMD: pixel path +6...+20 %).
    __shared__ float dots_s[2][(K4M_N1 * K4M_N1 + 1) * 32];
    __shared__ float pix_s[2][K4M_N1 * K4M_N1 + 4];              // VALU path: the dot table of one pixel
    __shared__ int pid_s[2][32];                                  // the tile's pixel ids (b * L + p; -1: empty slot of a sorted group)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, n = lane & 31;
    const int L = a.h * a.w;
    const int tiles_x = a.w / K4M_TW, tiles_y = a.h / K4M_TH;
    const bool sorted = a.ctl != nullptr && a.ctl[0] != 0u;      // target-ordered groups instead of the natural tiles (uniform)
    const int ntile = sorted ? (int)a.ctl[1] : a.batch * tiles_x * tiles_y;
    const float scale = 1.0f / sqrtf((float)UM_CHANNELS);
    float* dots = dots_s[wave];
    float* outt = dots_s[wave];

    for (int tile = blockIdx.x * 2 + wave; tile < ntile; tile += gridDim.x * 2) {
        // this lane's pixel: slot n of the natural 8 x 4 tile, or slot n of a sorted group (an empty slot borrows slot 0's pixel
        // -- every group's first slot is taken -- and writes nothing)
        int pid;
        if (sorted) {
            pid = a.perm[(long)tile * 32 + n];
        } else {
            const int bb = tile / (tiles_x * tiles_y);
            const int tt = tile - bb * (tiles_x * tiles_y);
            const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
            pid = bb * L + (ty * K4M_TH + (n >> 3)) * a.w + tx * K4M_TW + (n & 7);
        }
        const bool live = pid >= 0;
        if (half == 0) pid_s[wave][n] = pid;
        const int pid0 = __shfl(pid, 0);
        if (!live) pid = pid0;
        const int b = pid0 / L;                                               // one image per tile: groups are per image
        const int p = pid - b * L;
        const int y = p / a.w, x = p - y * a.w;
        const bool softmax_mode = a.flow_out != nullptr;                      // uniform
        const int n1 = softmax_mode ? K4M_KW : K4M_N1;                        // integer neighbourhood: 9 x 9 (no blend) or 10 x 10
        const float px = (float)x + (softmax_mode ? 0.f : a.flow[((long)b * 2 + 0) * L + p]);
        const float py = (float)y + (softmax_mode ? 0.f : a.flow[((long)b * 2 + 1) * L + p]);
        const float fbx = floorf(px), fby = floorf(py);
        const float wx = px - fbx, wy = py - fby;
        // clamp far-away bases so the int conversion is defined; such samples are all-zero anyway
        const int bx = (int)fminf(fmaxf(fbx, -32768.f), 32768.f);
        const int by = (int)fminf(fmaxf(fby, -32768.f), 32768.f);
        const int ux0 = k4m_wave_min(bx) - K4M_RADIUS, ux1 = k4m_wave_max(bx) - K4M_RADIUS + n1 - 1;
        const int uy0 = k4m_wave_min(by) - K4M_RADIUS, uy1 = k4m_wave_max(by) - K4M_RADIUS + n1 - 1;
        const int bw = ux1 - ux0 + 1, bh = uy1 - uy0 + 1;
        const bool coherent = softmax_mode || (!a.force_valu && bw <= 32 && bh <= K4M_MAXROWS);     // wave uniform

        // a tile none of whose pixels can sample inside the image (flows that leave the map: a quarter of all pixels at random init;
        // the sorted walk collects them in groups of their own) is all zeros (matching.py:113, zeros padding): no loads, no products
        const bool off_image = !softmax_mode && (bx + K4M_N1 - 1 - K4M_RADIUS < 0 || bx - K4M_RADIUS >= a.w ||
                                                 by + K4M_N1 - 1 - K4M_RADIUS < 0 || by - K4M_RADIUS >= a.h);
        const bool all_off = __all(off_image);                                                        // wave uniform
        if (a.stats && lane == 0) atomicAdd(a.stats + ((coherent || all_off) ? 0 : 1), 1u);
        if (all_off) {
            for (int idx = lane; idx < K4M_TAPS * 32; idx += 64) outt[idx] = 0.f;
        } else if (coherent) {
            // ---- B operand: the tile's 32 pixels (f0), both planes
            i16x8 qf[2][8];
            {
                const unsigned short* q0 = a.fp0 + ((long)b * L + p) * UM_CHANNELS + 8 * half;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) qf[pl][ks] = ld_global_16B(q0 + pl * a.plane_stride + 16 * ks);
            }
            // window position of this lane in a row, its validity, and the per-lane part of the scatter address
            const int xx = ux0 + n;
            const bool colok = n < bw && xx >= 0 && xx < a.w;
            const int dx0 = ux0 - (bx - K4M_RADIUS) + 4 * half;            // ix of accumulator register r = dx0 + c_r
            auto load_row = [&](int yy, i16x8 (&ka)[2][8]) __attribute__((always_inline)) {
                const bool ok = colok && yy >= 0 && yy < a.h;
                const unsigned short* k0 = a.fp1 + ((long)b * L + (ok ? yy * a.w + xx : 0)) * UM_CHANNELS + 8 * half;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        i16x8 v = ld_global_16B(k0 + pl * a.plane_stride + 16 * ks);
                        if (!ok) v = i16x8{0, 0, 0, 0, 0, 0, 0, 0};                  // zeros padding (matching.py:113)
                        ka[pl][ks] = v;
                    }
            };
            auto do_row = [&](int yy, const i16x8 (&ka)[2][8]) __attribute__((always_inline)) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    acc = Fp16::mfma(ka[1][ks], qf[0][ks], acc);               // lo_k . hi_q
                    acc = Fp16::mfma(ka[0][ks], qf[1][ks], acc);               // hi_k . lo_q
                    acc = Fp16::mfma(ka[0][ks], qf[0][ks], acc);               // hi_k . hi_q
                }
                const int iy = yy - (by - K4M_RADIUS);
                if ((unsigned)iy < (unsigned)n1) {
                    float* d = dots + (iy * K4M_N1 + dx0) * 32 + n;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cr = (r & 3) + 8 * (r >> 2);
                        if ((unsigned)(dx0 + cr) < (unsigned)n1) d[cr * 32] = acc[r];
                    }
                }
            };
            i16x8 kaA[2][8], kaB[2][8];                              // the next row's loads fly under this row's MFMAs
            load_row(uy0, kaA);
            for (int r0 = 0; r0 < bh; r0 += 2) {
                if (r0 + 1 < bh) load_row(uy0 + r0 + 1, kaB);
                do_row(uy0 + r0, kaA);
                if (r0 + 1 < bh) {
                    if (r0 + 2 < bh) load_row(uy0 + r0 + 2, kaA);
                    do_row(uy0 + r0 + 1, kaB);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (softmax_mode) {
                // ---- softmax over the 81 taps and expected offset; lanes 0-31 take tap rows 0..4, lanes 32-63 rows 5..8.
                // Out-of-image taps take part with logit -1e9 (matching.py:73)
                const int r0 = half ? 5 : 0, r1 = half ? K4M_KW : 5;
                float mxl = -3.0e38f;
                for (int trow = r0; trow < r1; ++trow)
#pragma unroll
                    for (int tcol = 0; tcol < K4M_KW; ++tcol) {
                        const int yy = y + trow - K4M_RADIUS, xq = x + tcol - K4M_RADIUS;
                        const bool ok = yy >= 0 && yy < a.h && xq >= 0 && xq < a.w;
                        const float lg = ok ? dots[(trow * K4M_N1 + tcol) * 32 + n] * scale : -1.0e9f;
                        mxl = fmaxf(mxl, lg);
                    }
                mxl = fmaxf(mxl, __shfl_xor(mxl, 32));
                float sp = 0.f, sx = 0.f, sy = 0.f;
                for (int trow = r0; trow < r1; ++trow)
#pragma unroll
                    for (int tcol = 0; tcol < K4M_KW; ++tcol) {
                        const int yy = y + trow - K4M_RADIUS, xq = x + tcol - K4M_RADIUS;
                        const bool ok = yy >= 0 && yy < a.h && xq >= 0 && xq < a.w;
                        const float lg = ok ? dots[(trow * K4M_N1 + tcol) * 32 + n] * scale : -1.0e9f;
                        const float e = __expf(lg - mxl);
                        sp += e;
                        sx += e * (float)(tcol - K4M_RADIUS);
                        sy += e * (float)(trow - K4M_RADIUS);
                    }
                sp += __shfl_xor(sp, 32);
                sx += __shfl_xor(sx, 32);
                sy += __shfl_xor(sy, 32);
                if (half == 0) {
                    a.flow_out[((long)b * 2 + 0) * L + p] = sx / sp;
                    a.flow_out[((long)b * 2 + 1) * L + p] = sy / sp;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            // ---- blends: lanes 0-31 take tap columns 0..4, lanes 32-63 columns 5..8 of every tap row
            const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy);
            const float w10 = (1.f - wx) * wy, w11 = wx * wy;
            const int col0 = 5 * half;
            float d0[6], d1[6], res[K4M_KW][5];
#pragma unroll
            for (int c = 0; c < 6; ++c) d0[c] = dots[(col0 + c) * 32 + n];
#pragma unroll
            for (int trow = 0; trow < K4M_KW; ++trow) {
#pragma unroll
                for (int c = 0; c < 6; ++c) d1[c] = dots[((trow + 1) * K4M_N1 + col0 + c) * 32 + n];
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    res[trow][j] = (w00 * d0[j] + w01 * d0[j + 1] + w10 * d1[j] + w11 * d1[j + 1]) * scale;
#pragma unroll
                for (int c = 0; c < 6; ++c) d0[c] = d1[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                        // every lane has read its dots: the storage becomes the tile
#pragma unroll
            for (int trow = 0; trow < K4M_KW; ++trow)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (col0 + j < K4M_KW) outt[(trow * K4M_KW + col0 + j) * 32 + n] = res[trow][j];
        } else {
            // ---- VALU path, one pixel at a time: lanes = 16 neighbour slots x 4 channel quarters, channels interleaved in groups
            // of 4 so that a load covers 64 contiguous bytes per slot (as local_ops.hip)
            const int slot = lane >> 2, quarter = lane & 3;
            for (int j = 0; j < 32; ++j) {
                if (__shfl(live ? 1 : 0, j) == 0) continue;                   // empty slot of a sorted group (uniform)
                const int jbx = __shfl(bx, j), jby = __shfl(by, j);
                const float jwx = __shfl(wx, j), jwy = __shfl(wy, j);
                const int jp = __shfl(p, j);
                f32x4 av[8];
                {
                    const float* ap = a.f0 + ((long)b * L + jp) * UM_CHANNELS + 4 * quarter;
#pragma unroll
                    for (int i = 0; i < 8; ++i) av[i] = reinterpret_cast<const f32x4*>(ap)[4 * i];
                }
                for (int r = 0; r * 16 < K4M_N1 * K4M_N1; ++r) {
                    const int t = r * 16 + slot;
                    const int iy = t / K4M_N1, ix = t - iy * K4M_N1;
                    const int yy = jby + iy - K4M_RADIUS, xx = jbx + ix - K4M_RADIUS;
                    const bool ok = t < K4M_N1 * K4M_N1 && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
                    float d = 0.f;
                    if (ok) {
                        const float* bp = a.f1 + ((long)b * L + yy * a.w + xx) * UM_CHANNELS + 4 * quarter;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const f32x4 bv = reinterpret_cast<const f32x4*>(bp)[4 * i];
                            d = __builtin_fmaf(av[i][0], bv[0], d);
                            d = __builtin_fmaf(av[i][1], bv[1], d);
                            d = __builtin_fmaf(av[i][2], bv[2], d);
                            d = __builtin_fmaf(av[i][3], bv[3], d);
                        }
                    }
                    d += __shfl_xor(d, 1);
                    d += __shfl_xor(d, 2);
                    if (quarter == 0 && t < K4M_N1 * K4M_N1) pix_s[wave][t] = d;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float w00 = (1.f - jwx) * (1.f - jwy), w01 = jwx * (1.f - jwy);
                const float w10 = (1.f - jwx) * jwy, w11 = jwx * jwy;
                for (int k = lane; k < K4M_TAPS; k += 64) {
                    const int trow = k / K4M_KW, tcol = k - trow * K4M_KW;
                    const float* dd = &pix_s[wave][trow * K4M_N1 + tcol];
                    const float v = w00 * dd[0] + w01 * dd[1] + w10 * dd[K4M_N1] + w11 * dd[K4M_N1 + 1];
                    outt[k * 32 + j] = v * scale;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- output: the [81][32] tile of this wave, every slot to its own pixel
        if (a.planes) {
            // channels-last operand planes for um_conv2d_ex: pixel row = ld channels, taps first, zeros up to ld
            const int pairs = a.ld >> 1;
            for (int idx = lane; idx < pairs * 32; idx += 64) {
                const int j = idx / pairs, c = (idx - j * pairs) * 2;
                const long pidj = pid_s[wave][j];
                if (pidj < 0) continue;
                const float v0 = c < K4M_TAPS ? outt[c * 32 + j] : 0.f, v1 = c + 1 < K4M_TAPS ? outt[(c + 1) * 32 + j] : 0.f;
                const unsigned hh = Fp16::pack2(v0, v1);
                const f32x2 u = Fp16::unpack2(hh);
                *reinterpret_cast<unsigned*>(a.planes + pidj * a.ld + c) = hh;
                *reinterpret_cast<unsigned*>(a.planes + a.out_plane_stride + pidj * a.ld + c) = Fp16::pack2(v0 - u[0], v1 - u[1]);
            }
        } else {
            for (int idx = lane; idx < K4M_TAPS * 32; idx += 64) {
                const int k = idx >> 5, j = idx & 31;
                const int pidj = pid_s[wave][j];
                if (pidj < 0) continue;
                a.cost[((long)b * K4M_TAPS + k) * L + (pidj - b * L)] = outt[k * 32 + j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------ host side
static size_t k4m_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// bytes of the feature planes of one scale: [f0 | f1], each [2][B * L][128] fp16
extern "C" size_t um_local_corr_feat_planes_bytes(int batch, int h, int w, int channels) {
    if (batch <= 0 || h <= 0 || w <= 0 || channels != UM_CHANNELS) return 0;
    return 2 * k4m_align256(planes_bytes((long)batch * h * w, 0)) + k4s_layout(batch, h, w).bytes;     // + target-ordering scratch
}

// split f0, f1 ([B, h*w, 128] fp32 tokens) into the operand planes the matrix-core cost volume reads; once per scale
extern "C" int um_local_corr_feat_planes(const float* f0, const float* f1, void* feat_planes, int batch, int h, int w, int channels,
                                         void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!f0 || !f1 || !feat_planes || batch <= 0 || h <= 0 || w <= 0 || channels != UM_CHANNELS) {
        um_set_error("um_local_corr_feat_planes: bad argument");
        return -1;
    }
    const long rows = (long)batch * h * w;
    unsigned char* ws = (unsigned char*)feat_planes;
    hipError_t e;
    if ((e = launch_split_planes(f0, (unsigned short*)ws, rows, 1.f, 0, stream)) != hipSuccess) return (int)e;
    if ((e = launch_split_planes(f1, (unsigned short*)(ws + k4m_align256(planes_bytes(rows, 0))), rows, 1.f, 0, stream)) != hipSuccess)
        return (int)e;
    // the target-ordering scratch behind the planes: the census word and the ticket start at zero (every launch leaves them zero)
    const K4sLayout l = k4s_layout(batch, h, w);
    if ((e = hipMemsetAsync(ws + 2 * k4m_align256(planes_bytes(rows, 0)), 0, l.off_part, stream)) != hipSuccess) return (int)e;
    return 0;
}

// 1 when um_local_corr_with_flow_feat serves this geometry (radius 4, maps of whole 8 x 4 pixel tiles)
extern "C" int um_local_corr_with_flow_feat_supported(int h, int w, int channels, int radius) {
    return (channels == UM_CHANNELS && radius == K4M_RADIUS && h > 0 && w > 0 && h % K4M_TH == 0 && w % K4M_TW == 0) ? 1 : 0;
}

// Local cost volume (matching.py:86-123) with the feature planes of um_local_corr_feat_planes.  Exactly one of `cost`
// ([B, 81, h, w] fp32) and `planes_out` (channels-last operand planes [2][plane_rows][ld], as um_local_corr_with_flow_planes)
// is non-null.  flags bit 0: force the VALU path for every tile (diagnostics / A-B timing).  stats (optional, device, 2 x
// unsigned, caller zeroes it): tiles that took the product path / the pixel path, for a caller that wants to route launches
// with mostly incoherent flow to um_local_corr_with_flow[_planes] (which block four pixels at a time and are ~3 % faster there).
extern "C" int um_local_corr_with_flow_feat(const float* f0, const float* f1, const void* feat_planes, const float* flow, float* cost,
                                            void* planes_out, int ld, long plane_rows, int batch, int h, int w, int channels,
                                            int radius, int flags, unsigned* stats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!f0 || !f1 || !feat_planes || !flow || (!cost == !planes_out) || batch <= 0) {
        um_set_error("um_local_corr_with_flow_feat: null pointer, or not exactly one output");
        return -1;
    }
    if (!um_local_corr_with_flow_feat_supported(h, w, channels, radius)) {
        um_set_error("um_local_corr_with_flow_feat: radius %d on a %dx%d map is not served (radius 4, h %% 4 == 0, w %% 8 == 0)", radius, h, w);
        return -2;
    }
    if (planes_out && (ld < K4M_TAPS + 1 || (ld & 1) || plane_rows < (long)batch * h * w)) {
        um_set_error("um_local_corr_with_flow_feat: bad planes geometry (ld=%d rows=%ld)", ld, plane_rows);
        return -1;
    }
    const long rows = (long)batch * h * w;
    K4mArgs a;
    a.f0 = f0;
    a.f1 = f1;
    a.fp0 = (const unsigned short*)feat_planes;
    a.fp1 = (const unsigned short*)((const unsigned char*)feat_planes + k4m_align256(planes_bytes(rows, 0)));
    a.plane_stride = rows * UM_CHANNELS;
    a.flow = flow;
    a.cost = cost;
    a.planes = (unsigned short*)planes_out;
    a.out_plane_stride = plane_rows * ld;
    a.ld = ld;
    a.batch = batch;
    a.h = h;
    a.w = w;
    a.force_valu = flags & 1;
    a.flow_out = nullptr;
    a.stats = stats;
    a.perm = nullptr;
    a.ctl = nullptr;
    const long ntile = rows / 32;
    ScopedKernelTimer timer(UM_K_COST_VOLUME, stream);
    const K4sLayout l = k4s_layout(batch, h, w);
    if (!(flags & (1 | 2)) && l.ok) {
        // target-ordered mode (flags bit 1 switches it off: A/B timing): census + per-workgroup histograms, start slots, scatter -- all on
        // device, the choice between natural tiles and sorted groups is made there from this call's flow alone
        unsigned char* sc = (unsigned char*)feat_planes + 2 * k4m_align256(planes_bytes(rows, 0));
        unsigned* ctl = (unsigned*)(sc + l.off_ctl);
        unsigned* part = (unsigned*)(sc + l.off_part);
        const int nb = batch * l.ncell;
        hipLaunchKernelGGL(k4s_census_kernel, dim3((unsigned)l.ng), dim3(K4S_WG), (size_t)nb * 4, stream, flow, batch, h, w, l.cx, l.cy, ctl, part,
                           (int*)(sc + l.off_cell), (int*)(sc + l.off_perm), nb, l.span, (unsigned)ntile);
        hipLaunchKernelGGL(k4s_scatter_kernel, dim3((unsigned)l.ng), dim3(K4S_WG), (size_t)nb * 4, stream, ctl, (const int*)(sc + l.off_cell),
                           (const unsigned*)part, (int*)(sc + l.off_perm), batch, h, w, nb, l.span);
        a.perm = (const int*)(sc + l.off_perm);
        a.ctl = ctl;
    }
    // (the sorted walk has at most rows / 32 + groups tiles; the grid is the natural one either way, tiles are grid-strided)
    long blocks = (ntile + 1) / 2;
    if (blocks > 4096) blocks = 4096;
    um_census_hit(UM_V_K4_MFMA);
    hipLaunchKernelGGL(k4m_kernel, dim3((unsigned)blocks), dim3(128), 0, stream, a);
    return (int)hipGetLastError();
}

// local_correlation_softmax (matching.py:39-83), radius 4, 2-D, on the matrix cores: every pixel's taps are its own 9 x 9
// neighbourhood, so every 8 x 4 tile's window is 16 x 12 and the whole map runs on the product path (the VALU kernel gathers
// 81 x 512 B per pixel: 0.59 ms at 4 x 128 x 192).  workspace: um_local_corr_feat_planes_bytes().  out: flow [B, 2, h, w].
extern "C" int um_local_corr_softmax_mfma(const float* f0, const float* f1, float* out, int batch, int h, int w, int channels, int radius,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!f0 || !f1 || !out || batch <= 0) {
        um_set_error("um_local_corr_softmax_mfma: null pointer");
        return -1;
    }
    if (!um_local_corr_with_flow_feat_supported(h, w, channels, radius)) {
        um_set_error("um_local_corr_softmax_mfma: radius %d on a %dx%d map is not served (radius 4, h %% 4 == 0, w %% 8 == 0)", radius, h, w);
        return -2;
    }
    if (!workspace || workspace_bytes < um_local_corr_feat_planes_bytes(batch, h, w, channels)) {
        um_set_error("um_local_corr_softmax_mfma: workspace too small");
        return -3;
    }
    if (int e = um_local_corr_feat_planes(f0, f1, workspace, batch, h, w, channels, stream_)) return e;
    const long rows = (long)batch * h * w;
    K4mArgs a;
    a.f0 = f0;
    a.f1 = f1;
    a.fp0 = (const unsigned short*)workspace;
    a.fp1 = (const unsigned short*)((const unsigned char*)workspace + k4m_align256(planes_bytes(rows, 0)));
    a.plane_stride = rows * UM_CHANNELS;
    a.flow = nullptr;
    a.cost = nullptr;
    a.planes = nullptr;
    a.out_plane_stride = 0;
    a.ld = 0;
    a.batch = batch;
    a.h = h;
    a.w = w;
    a.force_valu = 0;
    a.flow_out = out;
    a.stats = nullptr;
    a.perm = nullptr;
    a.ctl = nullptr;
    const long ntile = rows / 32;
    long blocks = (ntile + 1) / 2;
    if (blocks > 4096) blocks = 4096;
    {
        ScopedKernelTimer timer(UM_K_LOCAL_CORR, stream);
        um_census_hit(UM_V_K3_MFMA);
        hipLaunchKernelGGL(k4m_kernel, dim3((unsigned)blocks), dim3(128), 0, stream, a);
    }
    return (int)hipGetLastError();
}
