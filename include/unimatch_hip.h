/*
 * unimatch_hip.h — C ABI of the MI355X (gfx950) global-matching hot path.
 *
 * The reference (autonomousvision/unimatch) has no FFI: its hot path is a set of Python functions
 * made of ATen ops.  Each entry point below replaces ONE of those functions with one fused HIP
 * launch sequence; the citation on each is the reference function it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - Plain C: pointers and sizes only, no torch types.  All pointers are DEVICE pointers.
 *   - The caller owns every buffer (inputs, outputs, workspace).  The library never allocates or
 *     frees device memory and never keeps a pointer after the call returns.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and performs no host synchronisation.  Stateless and re-entrant.
 *   - Return value: 0 = enqueued; negative = argument error (UM_ERR_*); positive = hipError_t.
 *     um_last_error_string() describes the last failure on the calling thread.
 *   - Feature tensors are "token major": [N, L, C] fp32, L = h*w tokens in row-major (y, x) order,
 *     C = 128 channels contiguous per token (the transpose of the reference's [N, C, h, w]).
 *     Flow-like outputs are [N, V, h, w] fp32 exactly as the reference returns them.
 *   - `mode` selects the operand precision of the MFMA contractions:
 *       UM_MODE_EXACT  fp16 hi+lo split operands, 3 MFMA products per tile, fp32 accumulate
 *                      (~2^-22 relative operand error: meets the fp32 reference to its own noise floor)
 *       UM_MODE_FAST   bf16 operands, 1 MFMA product per tile, fp32 accumulate
 *     Softmax, accumulators and all non-MFMA kernels are fp32 in both modes.
 *   - Operand range (UM_MODE_EXACT): x = hi + lo with hi = rn_fp16(x), lo = rn_fp16(x - hi) gives 22 significant bits
 *     while lo stays in fp16's normal range, i.e. for element magnitudes in about [2^-3, 2^15]; below, the absolute
 *     error floor is 2^-25 (fp16 subnormal granularity), above 65504 the hi plane saturates to inf.  Weights are
 *     pre-scaled by 2^wshift (2^10) so that |w| in [2^-13, 2^5) is covered.  The model's activations on this path are
 *     O(1)..O(10) (LayerNorm / InstanceNorm outputs, features of std 1..4); tests/test_hip_parity_gpu.py sweeps
 *     0.1 .. 100.  There is no per-tensor rescaling: callers with other magnitudes scale by a power of two around the call.
 *     An element of magnitude >= 65504 on its way into an fp16 operand is NOT silent (round 5): the kernel that converts it raises
 *     a bit of a sticky, process-wide flag word (um_range_flags below) that says which kind of operand it was.
 */
#ifndef UNIMATCH_HIP_H
#define UNIMATCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UM_VERSION 220

#define UM_MODE_EXACT 0
#define UM_MODE_FAST 1

#define UM_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, unsupported channel count ... */
#define UM_ERR_BAD_GEOMETRY (-2) /* window does not tile the map, shift >= window ...             */
#define UM_ERR_WORKSPACE (-3)    /* workspace missing or too small                                  */
#define UM_ERR_UNSUPPORTED (-4)  /* valid in the reference but not implemented by this library      */
#define UM_ERR_COLLECTIVE (-5)   /* RCCL reported an error / the communicator bootstrap failed       */

int um_version(void);
const char* um_last_error_string(void);

/* Operand-range flags (UM_MODE_EXACT only; bf16 operands have fp32's exponent range).  Kernels on the Transformer / matching path
 * that turn fp32 values into fp16 hi | lo operands track the largest magnitude they convert and OR a bit into one sticky word in
 * pinned host memory when it reaches 65504 (the hi plane would be inf and NaN follows downstream; the fp32 reference has no such
 * limit -- unimatch/transformer.py:58-60).  Reading the word costs no device synchronisation; it reflects every launch that has
 * FINISHED, so read it after synchronising the stream(s) when the answer must cover the last call.  `reset` != 0 clears it.
 * The convolution kernels of the CNN encoder / refinement block are not instrumented (instance-normalised / bounded inputs). */
#define UM_RANGE_PLANES 1u        /* um_*_fwd entry points that split whole fp32 tensors (q, k, v, features): split_planes_kernel */
#define UM_RANGE_ATTN_TOKENS 2u   /* um_window_attn_qproj_merge_fwd: the source tokens x                                       */
#define UM_RANGE_ATTN_QUERY 4u    /* ... its projected queries Wq.x                                                           */
#define UM_RANGE_ATTN_OUTPUT 8u   /* um_window_attn_*merge*_fwd: the attention output fed to the merge Linear                 */
#define UM_RANGE_FFN_TOKENS 16u   /* um_ffn_*fwd: the concatenated input [x | y]                                              */
#define UM_RANGE_FFN_HIDDEN 32u   /* ... the hidden activations gelu(W1.[x | y])                                              */
#define UM_RANGE_KV_TOKENS 64u    /* um_kv4_fwd / the k | v epilogue of um_ffn_kv_fwd: tokens, and the projected keys / values */
#define UM_RANGE_LINEAR 128u      /* um_linear_fwd: fp32 inputs and plane outputs                                             */
int um_range_flags(unsigned* flags_out, int reset);

/* =============================================================================================
 * MEASUREMENT ABI (um_timing_*, um_census_*): not part of the reference's operator interface.  It exists so that bench.py and
 * the parity tests can (a) time individual kernels with hipEvents on the stream they are launched on and (b) assert which
 * kernel instantiation served a call.  Both are OFF by default; while off the library keeps no state at all and every
 * entry point below is a pure function of its arguments.  Micro-benchmarks of the hardware itself (um_debug_*) are NOT in
 * the shipped library: they live in diagnostic builds only (end of this header).
 * =============================================================================================
 * Kernel timing: when enabled, every launch of the kernels below is bracketed by a pair of
 * hipEvents recorded on the launch stream itself; um_timing_collect() waits for them, returns the summed
 * kernel time and launch count since the last collect, and recycles the events.  Off by default (then the
 * library keeps no state at all).  Two event records per launch are not free (~6 % of a 14 ms forward with every
 * kernel timed): select only the kernels of interest inside a timed region. */
#define UM_K_WINDOW_ATTN 0   /* window_attn_kernel (um_window_attn_fwd)                                  */
#define UM_K_GLOBAL_SOFTMAX 1 /* gsv_kernel (um_global_corr_softmax_flow/_stereo, um_prop_global_attn)    */
#define UM_K_SPLIT_PLANES 2  /* split_planes_kernel (operand conversion pre-pass of the MFMA kernels)     */
#define UM_K_LOCAL_CORR 3    /* local_corr_softmax_kernel                                                */
#define UM_K_COST_VOLUME 4   /* local_corr_with_flow_kernel                                              */
#define UM_K_PROP_LOCAL 5    /* prop_local_attn_kernel                                                   */
#define UM_K_DEPTH_CORR 6    /* depth_corr_softmax_kernel                                                */
#define UM_K_LINEAR 7        /* linear_kernel (um_linear_fwd)                                            */
#define UM_K_INSTANCE_NORM 8 /* instance_norm_kernel (um_instance_norm_fwd)                              */
#define UM_K_CONVEX_UPSAMPLE 9 /* convex_upsample_kernel (um_convex_upsample)                            */
#define UM_K_FFN 10          /* ffn_kernel (um_ffn_fwd)                                                  */
#define UM_K_CONV 11         /* conv_kernel (um_conv2d_fwd)                                              */
#define UM_K_COUNT 12
/* Launch census (diagnostic, off by default): with um_census_enable(1) every launch below counts the kernel INSTANTIATION that
 * served it, so a test can assert that the configuration it checks ran the same kernels the bench times (and not, say, the
 * small-launch split variants).  um_census_enable(1) also zeroes the counters; um_census_count(v) reads one. */
#define UM_V_WATTN_TILE 0     /* window_attn_kernel, one workgroup per 128-query tile                                   */
#define UM_V_WATTN_KSPLIT 1   /* window_attn_kernel<..., KSPLIT>: 2 / 4 workgroups share a query tile (small launches)  */
#define UM_V_FFN_TILE 2       /* ffn_kernel, one workgroup per 128-token tile                                           */
#define UM_V_FFN_HSPLIT 3     /* ffn_kernel<..., HSPLIT>: hidden slices split over 2 / 4 workgroups (small launches)    */
#define UM_V_GSV4 4           /* gsv4_kernel (stream-K global correlation / propagation)                                */
#define UM_V_GSV3 5           /* gsv3_kernel (causal / ragged / small launches)                                         */
#define UM_V_K4_MFMA 6        /* k4m_kernel cost volume (matrix cores, per-tile gather path inside)                      */
#define UM_V_K4_VALU 7        /* local_corr_with_flow_kernel cost volume (VALU)                                         */
#define UM_V_K3_MFMA 8        /* local correlation softmax on k4m_kernel                                                */
#define UM_V_K3_VALU 9        /* local_corr_softmax_kernel (VALU)                                                       */
#define UM_V_CONV_PATCH 10    /* conv_patch_kernel (3x3 stride 1, 2-D halo patch)                                       */
#define UM_V_CONV_ROWS 11     /* conv_rows_kernel (row window shared by the horizontal taps)                            */
#define UM_V_CONV_GENERIC 12  /* conv_kernel (tap-by-tap implicit GEMM)                                                 */
#define UM_V_WATTN_W8 13      /* reserved: an 8-wave attention variant measured and dropped in round 5 (never counted) */
#define UM_V_COUNT 14
int um_census_enable(int on);
long um_census_count(int variant);
/* Key-tile census of the window attention (round 6; diagnostic, off by default).  In a window that carries the shifted-window mask
 * (unimatch/utils.py:84-108) a (128-query workgroup, 32-key tile) pair whose classes differ holds nothing but -100 logits
 * (unimatch/attention.py:88-89); the kernel walks such windows class by class, PROBES those tiles (hi.hi product only) after the
 * workgroup's own-class tiles and drops a tile iff every probed logit stays UM margin (40 natural-log units) below the running
 * row maximum -- otherwise the tile is computed exactly as an unmasked one.  While enabled, every workgroup adds to four counters
 * of the current device: [0] key tiles computed in full, [1] tiles probed, [2] probed tiles that then had to be computed (they are
 * part of [0] as well), [3] workgroups.  um_window_attn_tile_census(enable, counts4): counts4 != NULL synchronises the device,
 * copies the counters out and zeroes them; then the census is switched on / off as `enable` says. */
int um_window_attn_tile_census(int enable, unsigned long long* counts4);
/* Box probes (round 6, csrc/probe.hip): three kernels bench.py times inside its own run so that a bench line describes the box it was
 * measured on (MI355X boxes of one pool differ by several per cent in what they sustain).  Asynchronous on `stream`; the caller times them.
 *   um_probe_mfma(sink, iters): a memory-free loop of independent 32x32x16 fp16 MFMAs with pseudo-random operands on every SIMD;
 *                               executes um_probe_mfma_flops(iters) FLOPs.  sink: one float of device memory (never written).
 *   um_probe_copy(src, dst, bytes): float4 copy, bytes a multiple of 16 (reads `bytes`, writes `bytes`).
 *   um_probe_chase(ring, out, hops): one lane follows ring[i] -> next index for `hops` dependent loads (the caller builds the ring). */
double um_probe_mfma_flops(int iters);
int um_probe_mfma(float* sink, int iters, void* stream);
int um_probe_copy(const void* src, void* dst, size_t bytes, void* stream);
int um_probe_chase(const unsigned* ring, unsigned* out, int hops, void* stream);
int um_timing_enable(int kernel_mask);   /* bit k set: time kernel id UM_K_* = k; -1: all; 0: off */
int um_timing_collect(int kernel_id, double* total_ms, int* launches);
/* ===================================== end of the measurement ABI ============================= */

/* ---------------------------------------------------------------------------------------------
 * Windowed single-head attention  softmax(q k^T / sqrt(C) + shift_mask) v   inside windows.
 * Replaces (one geometry each):
 *   unimatch/attention.py:45-104  single_head_split_window_attention   win=(h/K,w/K), shift=win/2|0
 *   unimatch/attention.py:8-16    single_head_full_attention           win=(h,w)
 *   unimatch/attention.py:19-42   single_head_full_attention_1d        win=(1,w)
 *   unimatch/attention.py:107-163 single_head_split_window_attention_1d win=(1,w/K), shift=(0,win_w/2)|0
 * together with torch.roll, split_feature/merge_splits (unimatch/utils.py:34-81,155-196) and the
 * additive -100 masks (unimatch/utils.py:84-108,199-216), which become index arithmetic.
 * q, k, v, out: [streams, h*w, C] fp32.  C must be 128.
 * ------------------------------------------------------------------------------------------- */
size_t um_window_attn_workspace_bytes(int streams, int tokens, int channels, int mode);
int um_window_attn_fwd(const float* q, const float* k, const float* v, float* out,
                       int streams, int h, int w, int channels,
                       int win_h, int win_w, int shift_h, int shift_w,
                       int mode, void* workspace, size_t workspace_bytes, void* stream);

/* Same attention core on operands that are ALREADY in the MFMA plane format ([NS][rows][ld] 16-bit, NS = 2 fp16
 * planes hi|lo in UM_MODE_EXACT, 1 bf16 plane in UM_MODE_FAST; plane stride = rows * ld elements), e.g. the output
 * of um_linear_fwd(epilogue = UM_EPI_PLANES).  q/k/v may be column slices of wider projections: ldq / ldkv are
 * the row strides in elements (multiples of 8), rows = streams * h * w.  k and v share ldkv.  No workspace.
 * kv_rotate: stream s reads the keys / values of stream (s + kv_rotate) mod streams -- with streams = 2B and kv_rotate = B this
 * is the reference's cross attention of [f0; f1] against [f1; f0] (unimatch/transformer.py:271-291) without the swapped copy. */
int um_window_attn_planes_fwd(const void* q_planes, const void* k_planes, const void* v_planes, float* out,
                              int streams, int h, int w, int channels, int ldq, int ldkv,
                              long q_plane_stride, long kv_plane_stride,
                              int win_h, int win_w, int shift_h, int shift_w, int kv_rotate, int mode, void* stream);

/* um_window_attn_planes_fwd with the layer's tail folded into the epilogue (unimatch/transformer.py:137-138, 144):
 *     out = LayerNorm(attention . Wm^T; gamma, beta, eps) (+ residual)
 * wm_planes: um_weight_planes() of the merge weight [128,128] with `wshift`; residual: optional fp32 [streams*h*w][128]
 * (the self-attention layers' `source + message`).  The attention output never reaches HBM. */
int um_window_attn_merge_fwd(const void* q_planes, const void* k_planes, const void* v_planes, const void* wm_planes,
                             const float* gamma, const float* beta, const float* residual, float eps, int wshift, float* out,
                             int streams, int h, int w, int channels, int ldq, int ldkv, long q_plane_stride,
                             long kv_plane_stride, int win_h, int win_w, int shift_h, int shift_w, int kv_rotate, int mode,
                             void* stream);

/* Same, with the query projection of unimatch/transformer.py:58 folded into the kernel's prologue:
 *     q = x . Wq^T   for the workgroup's 128 query tokens, straight into the MFMA operand registers
 * x: fp32 [streams*h*w][128] source tokens; wq_planes: um_weight_planes() of the query weight [128,128] with the same
 * `wshift` as wm_planes.  A query row is consumed by exactly one workgroup, so nothing is recomputed and the q operand
 * planes (one write + one read of [streams*h*w][128] per layer) never exist; k / v planes as above.
 * workspace (optional, NULL = none): um_window_attn_ksplit_workspace_bytes() bytes that are ZERO before the first launch
 * (the kernel leaves them zero; one workspace serves one launch at a time, launches on one stream are fine).  With it, a
 * launch of few query tiles (batch 1: 40 - 96 workgroups on 256 CUs, each walking a whole window) gives every query tile
 * to 2 or 4 neighbouring workgroups, each on its share of the window's keys; the first merges the others' partial softmaxes
 * (exact).  The byte count is 0 for launches that are not split. */
/* Split launches.  um_window_attn_qproj_merge_fwd and um_ffn_ws_fwd with `workspace` let several workgroups share the key walk of a
 * query tile / the hidden slices of a token tile when the launch is small (launch plans: um_window_attn_plan,
 * um_ffn_split_workspace_bytes).  Every part publishes its partial result in the workspace and takes a ticket from the tile's arrival
 * counter; the last arriver combines all parts in part order (exact / fixed order: bitwise reproducible) and finishes the layer.
 * Nobody waits for anybody: no assumption about residency or dispatch order.  The workspace must be zero before the first launch
 * that uses it, is left with zero counters by every launch, and must not be shared by launches in flight on different streams. */
size_t um_window_attn_ksplit_workspace_bytes(int streams, int h, int w, int win_h, int win_w);
/* The launch plan of um_window_attn_qproj_merge_fwd for a geometry, a pure function of the arguments and the current device's CU
 * count: `full_tiles` 128-query tiles are served one workgroup each, `split_tiles` tiles by `parts` workgroups each on a share of
 * the window's key tiles (launches whose tiles all fit the chip at once: batch-1 latency; big launches are never split -- measured). */
int um_window_attn_plan(int streams, int h, int w, int win_h, int win_w, int* full_tiles, int* split_tiles, int* parts);
int um_window_attn_qproj_merge_fwd(const float* x, const void* wq_planes, const void* k_planes, const void* v_planes,
                                   const void* wm_planes, const float* gamma, const float* beta, const float* residual, float eps,
                                   int wshift, float* out, int streams, int h, int w, int channels, int ldkv,
                                   long kv_plane_stride, int win_h, int win_w, int shift_h, int shift_w, int kv_rotate, int mode,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Transformer-layer linears  C[M,N] = A[M,K] . W[N,K]^T  (nn.Linear without bias) on MFMA with fused
 * prologue / epilogue.  Replaces, per layer of unimatch/transformer.py: q/k/v projections (:58-60), merge +
 * LayerNorm (+ residual) (:137-138,:144), and the FFN  cat[source, message] -> 8C -> GELU -> C -> LayerNorm ->
 * + source (:141-144).  Same operand arithmetic as the attention kernels (`mode`).
 *   weights : um_weight_planes() turns W [N,K] fp32 into planes pre-scaled by 2^wshift (exact; keeps the fp16
 *             lo plane out of the subnormal range); pass the same wshift to um_linear_fwd.
 *   input   : exactly one of  a0 (fp32 [M,K])  |  a0 + a1 (fp32 [M,K/2] each: K-concatenation without the
 *             concatenated tensor)  |  a_planes ([NS][M][K]).
 *   epilogue: UM_EPI_PLANES       out = planes [NS][M][N]
 *             UM_EPI_LN           out = fp32 [M,N], N == 128: LayerNorm(gamma, beta, eps) (+ residual [M,N] if not NULL)
 *             UM_EPI_GELU_PLANES  out = planes of erf-GELU(C)
 * N must be a multiple of 128, K of 32 (64 with a1).
 * ------------------------------------------------------------------------------------------- */
#define UM_EPI_PLANES 0
#define UM_EPI_LN 1
#define UM_EPI_GELU_PLANES 2
size_t um_planes_bytes(long rows, int cols, int mode);
int um_weight_planes(const float* w, void* planes, int n, int k, int wshift, int mode, void* stream);
int um_linear_fwd(const float* a0, const float* a1, const void* a_planes, const void* w_planes,
                  int m, int n, int k, int wshift, int epilogue, void* out,
                  const float* gamma, const float* beta, const float* residual, float eps,
                  int mode, void* stream);

/* nn.Linear WITH bias -- the propagation layer's q / k projections (unimatch/attention.py:196-205 global, :229-232 local):
 *     C = (A . W^T) * out_mul + bias * bias_mul
 * input: exactly one of a0 (fp32 [M,K]) | a_planes ([NS][M][K]);  output: exactly one of out_planes ([NS][M][N], feeds
 * um_prop_global_attn_planes) | out_f32 ([M,N], feeds um_prop_local_attn).  w_planes / wshift as for um_linear_fwd; bias: fp32
 * [N], 16-byte aligned.  With A already scaled by s (planes carrying um_global_corr_plane_scale()), out_mul = 1 and
 * bias_mul = s give s * (W a + b). */
int um_linear_bias_fwd(const float* a0, const void* a_planes, const void* w_planes, const float* bias, int m, int n, int k,
                       int wshift, float out_mul, float bias_mul, void* out_planes, float* out_f32, int mode, void* stream);

/* Whole FFN of a Transformer layer in ONE kernel:
 *     out = x + LayerNorm( W2 . gelu( W1 . [x | y] ) )          (unimatch/transformer.py:141-144, mlp :44-50)
 * x (source, also the residual) and y (message): fp32 [M,128]; W1 [hidden, 256] and W2 [128, hidden] as
 * um_weight_planes() planes with the same wshift; gamma/beta/eps: the layer's norm2.  The [M, hidden]
 * activations stay on chip (the two-launch form um_linear_fwd(GELU_PLANES) + um_linear_fwd(LN) writes and reads them
 * through HBM).  hidden: multiple of 32, >= 64.  out: fp32 [M,128], may not alias x or y. */
int um_ffn_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
               int wshift, const float* gamma, const float* beta, float eps, float* out, int mode, void* stream);
/* The same with an optional workspace of um_ffn_split_workspace_bytes(m, hidden) bytes that are ZERO before the first launch
 * (the kernel leaves them zero; one workspace serves one launch at a time).  With it a launch of few token tiles (batch 1:
 * 35 - 96 workgroups on 256 CUs, each walking all hidden slices) gives every tile to 2 or 4 neighbouring workgroups, each on
 * its share of the hidden units; the last one to finish sums the partial outputs (in part order) before LayerNorm.  The byte
 * count is 0 for launches that are not split. */
size_t um_ffn_split_workspace_bytes(int m, int hidden);
int um_ffn_ws_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden,
                  int wshift, const float* gamma, const float* beta, float eps, float* out, int mode, void* workspace,
                  size_t workspace_bytes, void* stream);

/* The key / value projections of BOTH layers of a Transformer block (unimatch/transformer.py:58-60: k_proj / v_proj of the block's
 * self_attn and of its cross_attn_ffn, four bias-free 128 x 128 Linears of the token stream as it enters the block) in ONE launch
 * (round 4; two um_linear_fwd launches before):  x fp32 [m][128]  ->  out_planes [NS][4][m][128], the attention kernels' operand
 * planes in blocked form: projection j (0 k_self, 1 v_self, 2 k_cross, 3 v_cross) is the contiguous [m][128] tensor at element
 * offset j * m * 128 of every plane (plane stride 4 * m * 128, row stride 128: what um_window_attn_qproj_merge_fwd takes as
 * k_planes / v_planes + ldkv + kv_plane_stride).  wc_planes = um_weight_planes of the packed [256][256] fp32 matrix
 *     Wc[32 c + r][0:128] = W4[32 c + r][:],   Wc[32 c + r][128:256] = W4[256 + 32 c + (r ^ 16)][:],   c = 0..7, r = 0..31,
 * W4 = the four weights stacked [512][128] (unimatch_amd/ops.py::kv4_weight_planes builds it). */
int um_kv4_fwd(const float* x, const void* wc_planes, int m, int wshift, void* out_planes, int mode, void* stream);
/* um_ffn_ws_fwd of block i AND um_kv4_fwd of block i + 1 on its result, from ONE launch (closed in round 4):
 * every workgroup hands the 128-token tile it has just normalised to the k | v projection in its epilogue, so the next block's
 * keys / values (kv_planes, blocked [NS][4][m][128]) leave the FFN kernel directly and `out` is not read back.  Launches small
 * enough for the hidden split (um_ffn_split_workspace_bytes > 0) run the two kernels back to back inside the call. */
int um_ffn_kv_fwd(const float* x, const float* y, const void* w1_planes, const void* w2_planes, int m, int hidden, int wshift,
                  const float* gamma, const float* beta, float eps, float* out, const void* wc_planes, void* kv_planes, int mode,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolutions either side of the matching path, NHWC, on the matrix cores with the same
 * operand arithmetic as everything else (`mode`): nn.Conv2d of unimatch/reg_refine.py:6-119 (3x3, 1x1, 1x5, 5x1, 7x7)
 * and unimatch/backbone.py:7-133 (3x3 at stride 1/2, 1x1 projections), dilation 1, groups 1.
 *   a_planes : activations as operand planes [NS][batch*hi*wi + 1][cin], NHWC, whose LAST row is all zeros (taps outside
 *              the image read it).  Written by um_nhwc_instance_norm / um_nchw_to_nhwc.  cin: multiple of 32.
 *   w_planes : um_weight_planes() of the weight permuted to [cout][kh][kw][cin] (n = cout, k = kh*kw*cin), same wshift.
 *   out      : fp32 [batch*ho*wo][cout] (NHWC), ho = (hi + 2 pad_h - kh) / stride + 1; + bias[cout] if not NULL (16-byte
 *              aligned); ReLU if relu.
 * ------------------------------------------------------------------------------------------- */
int um_conv2d_fwd(const void* a_planes, const void* w_planes, const float* bias, float* out, float* stats_out, int batch,
                  int hi, int wi, int cin, int cout, int kh, int kw, int stride, int pad_h, int pad_w, int relu, int wshift,
                  int mode, void* stream);
/* General form.  The input planes are a column slice [a_coff, a_coff + cin) of a buffer [NS][a_rows][a_ld] whose LAST row
 * (a_rows - 1 >= batch*hi*wi) is all zeros; the result goes to fp32 `out` [.][out_ld] at column out_coff and / or, as operand
 * planes, to `out_planes` [NS][outp_rows][outp_ld] at column outp_coff -- so a chain of convolutions needs no fp32 round trip
 * and channel concatenations (torch.cat of reg_refine.py:33-35, 50-53, 62-66) are column offsets.  act: 0 none, 1 ReLU,
 * 2 sigmoid, 3 tanh (the GRU gates of reg_refine.py:55-76).  Leading dimensions / offsets: multiples of 8 (input) / 4 (output). */
int um_conv2d_ex(const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes, const float* bias, float* out,
                 int out_ld, int out_coff, void* out_planes, int outp_ld, int outp_coff, long outp_rows, float* stats_out,
                 int batch, int hi, int wi, int cin, int cout, int kh, int kw, int stride, int pad_h, int pad_w, int act,
                 int wshift, int mode, void* stream);
/* The two convolutions of a SepConvGRU half (unimatch/reg_refine.py:66-74) with the gate arithmetic in the epilogue.
 *   gate 1: weights = [z ; r] stacked (2*channels outputs), sigmoid; z -> z_out [.][z_out_ld] (fp32), and r * hidden is written
 *           as operand planes at (out_planes, outp_ld, outp_coff) -- the q convolution's input;
 *   gate 2: weights = q (channels outputs), tanh; hidden <- (1 - z) * hidden + z * q in place (fp32 [.][channels], z = z[.][z_ld])
 *           and as operand planes (the next convolution's input).
 * Input slice / stride 1 / padding as um_conv2d_ex. */
int um_conv2d_gru_fwd(int gate, const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes, const float* bias,
                      float* hidden, const float* z, int z_ld, float* z_out, int z_out_ld, void* out_planes, int outp_ld,
                      int outp_coff, long outp_rows, int batch, int hi, int wi, int cin, int channels, int kh, int kw, int pad_h,
                      int pad_w, int wshift, int mode, void* stream);
/* The same with a per-pixel addend (fp32 [rows][addend_ld]; column = output channel of the convolution: 2 * channels for gate 1,
 * channels for gate 2), added to the scaled accumulator (+ bias) BEFORE the gate's activation.  A convolution is linear in its input
 * channels: the refinement loop (unimatch/unimatch.py:315-331) restarts its hidden state from the same net0 and feeds the same
 * context features `inp` in every iteration, so their share of every gate convolution is computed ONCE per scale (um_conv2d_ex,
 * no activation) and the per-iteration convolutions only read the channels that change (round 4; unimatch_amd/refine_nhwc.py).
 * gate 2 (v211): `z_out`, when non-null, receives the NEW hidden state (fp32 [rows][z_out_ld >= channels]) and `hidden` is only read --
 * the loop's first q convolution reads net0 and writes the working state, so net0 is never copied (null: `hidden` updated in place). */
int um_conv2d_gru_add_fwd(int gate, const void* a_planes, int a_ld, int a_coff, long a_rows, const void* w_planes, const float* bias,
                          const float* addend, int addend_ld, float* hidden, const float* z, int z_ld, float* z_out, int z_out_ld,
                          void* out_planes, int outp_ld, int outp_coff, long outp_rows, int batch, int hi, int wi, int cin,
                          int channels, int kh, int kw, int pad_h, int pad_w, int wshift, int mode, void* stream);
/* stats_out (optional): per output tile part (<= 128 pixels, in the order of the serving kernel's tiles) the (mean, pixel
 * count, sum of squared deviations) of every output channel, computed in the epilogue from the tile that is in LDS anyway.
 * um_conv_stats_parts() = parts per image for that geometry (the stem, um_conv7_fwd / um_stem_conv_fwd: kh = kw = 7,
 * stride 2, pad 3); the buffer holds um_conv_stats_bytes(batch, parts, cout) bytes; pass both to
 * um_nhwc_instance_norm(conv_stats, conv_stats_parts) and the normalisation skips its own statistics pass. */
int um_conv_stats_parts(int hi, int wi, int cout, int kh, int kw, int stride, int pad_h, int pad_w);
size_t um_conv_stats_bytes(int batch, int parts, int channels);

/* The encoder's 7x7 / stride-2 / pad-3 stem (unimatch/backbone.py:49; 3 input channels, no bias) on the same kernel, always
 * in the exact arithmetic.  image: fp32 NCHW [batch,3,h,w]; with `normalize` the reference's input normalisation
 * ((x / 255 - mean) / std, unimatch/unimatch.py:122-124) is applied while packing.  image_planes: scratch of
 * um_stem_planes_bytes() bytes (the zero-bordered NHWC-4 operand planes).  w_planes: um_weight_planes() of the weight
 * rearranged to [cout][7 ky][8 kx][4 ci] with zeros at kx = 7 and ci = 3 (n = cout, k = 224).  out: fp32 NHWC
 * [batch * ho * wo][cout], ho = (h - 1) / 2 + 1.  stats_out: as um_conv2d_fwd. */
size_t um_stem_planes_bytes(int batch, int h, int w);
/* General form: 7x7 / pad 3 convolution of an fp32 NCHW image with few channels (stride 2: <= 3 channels, packed 4 per pixel;
 * stride 1: <= 8 channels, packed 8 per pixel -- the motion encoder's flow branch, unimatch/reg_refine.py:13), outputs as
 * um_conv2d_ex.  w_planes: the weight rearranged to [cout][7 ky][8 kx][cpp] with zeros at kx = 7 and the missing channels
 * (k = 56 cpp).  mean3 / std3 are HOST arrays. */
size_t um_conv7_planes_bytes(int batch, int h, int w, int stride);
int um_conv7_fwd(const float* image, int channels, int normalize, const float* mean3, const float* std3, void* image_planes,
                 const void* w_planes, const float* bias, float* out, int out_ld, int out_coff, void* out_planes, int outp_ld,
                 int outp_coff, long outp_rows, float* stats_out, int batch, int h, int w, int cout, int stride, int act,
                 int wshift, void* stream);
int um_stem_conv_fwd(const float* image, int normalize, const float* mean3, const float* std3, void* image_planes,
                     const void* w_planes, float* out, float* stats_out, int batch, int h, int w, int cout, int wshift,
                     void* stream);

/* nn.InstanceNorm2d (affine=False, biased variance) + ReLU (+ shortcut add + ReLU) of unimatch/backbone.py:7-36 in NHWC:
 *   y = x (normalize == 0) | (x - mean_{b,c}) * rsqrt(var_{b,c} + eps);  y = relu(y) if relu;  y = relu(shortcut + y) if
 * shortcut.  x, shortcut: fp32 [batch*pixels][channels]; alternatively (shortcut == NULL) shortcut_planes: the shortcut as
 * operand planes [NS][batch*pixels + 1][channels] (hi + lo is added) -- a residual block's identity shortcut is the planes
 * its first convolution read, so no fp32 copy of the block input has to exist.  Outputs (either may be NULL): operand planes
 * [NS][batch*pixels + 1][channels] including the zero row um_conv2d_fwd expects, and fp32 [batch*pixels][channels].
 * Statistics are deterministic (fixed reduction order, chunk-shifted sums merged in fp64).  channels: multiple of 8, <= 256. */
size_t um_nhwc_norm_workspace_bytes(int batch, int pixels, int channels);
int um_nhwc_instance_norm(const float* x, const float* shortcut, const void* shortcut_planes, void* planes_out, float* f32_out,
                          int batch, int pixels, int channels, float eps, int normalize, int relu, const float* conv_stats,
                          int conv_stats_parts, void* workspace, size_t workspace_bytes, int mode, void* stream);

/* Channels-last element-wise helpers of the refinement block (SepConvGRU, unimatch/reg_refine.py:55-76); every result is
 * written as operand planes into columns [coff, coff + channels) of a buffer [NS = 2][plane_rows][ld]:
 *   mode 0  planes = src[rows][src_ld] (first `channels` columns)                      -- column scatter (flow, hidden state)
 *   mode 1  planes = r * h,   r = zr[rows][2C] columns C..2C,  h = hbuf[rows][C]
 *   mode 2  h <- (1 - z) * h + z * q  (z = zr columns 0..C, q = src), written back to hbuf and (if planes_out) as planes. */
int um_nhwc_gate(int mode, const float* src, int src_ld, const float* zr, float* hbuf, void* planes_out, int ld, int coff,
                 long plane_rows, long rows, int channels, void* stream);

/* fp32 NCHW [batch][channels][pixels] (the output of a MIOpen convolution) -> NHWC operand planes (with the zero row)
 * and / or fp32 NHWC.  channels: multiple of 8, <= 256. */
int um_nchw_to_nhwc(const float* x, void* planes_out, float* f32_out, int batch, int channels, int pixels, int mode,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * All-pairs correlation + softmax + expected coordinate (flow).
 * Replaces unimatch/matching.py:7-36 global_correlation_softmax (the [B,L,L] correlation and
 * probability tensors are never formed).  f0, f1: [B, h*w, C].  flow: [B or 2B, 2, h, w]
 * (bidir != 0 appends the backward flow, matching.py:23-27).
 * ------------------------------------------------------------------------------------------- */
size_t um_global_corr_workspace_bytes(int batch, int tokens, int channels, int mode);
int um_global_corr_softmax_flow(const float* f0, const float* f1, float* flow,
                                int batch, int h, int w, int channels, int bidir,
                                int mode, void* workspace, size_t workspace_bytes, void* stream);

/* Per-scanline W x W correlation, targets right of the query masked, disparity = x - E[x'].
 * Replaces unimatch/matching.py:126-151 global_correlation_softmax_stereo.  disp: [B, 1, h, w]. */
int um_global_corr_softmax_stereo(const float* f0, const float* f1, float* disp,
                                  int batch, int h, int w, int channels,
                                  int mode, void* workspace, size_t workspace_bytes, void* stream);

/* Global self-attention propagation  softmax(q k^T / sqrt(C)) value.
 * Replaces the attention core of unimatch/attention.py:196-213 SelfAttnPropagation.forward
 * (q/k projections stay with the caller).  q, k: [B, h*w, C]; value, out: [B, V, h, w], V in {1,2}. */
int um_prop_global_attn(const float* q, const float* k, const float* value, float* out,
                        int batch, int h, int w, int channels, int value_channels,
                        int mode, void* workspace, size_t workspace_bytes, void* stream);
/* The same on operands that are already MFMA planes ([NS][batch*h*w][128]) carrying um_global_corr_plane_scale(channels) on
 * both q and k (= sqrt(log2(e) / sqrt(C)): the kernels take the softmax logit in log2 units straight from the MFMA) -- the
 * output of um_linear_bias_fwd(out_planes).  Workspace as um_global_corr_workspace_bytes(). */
float um_global_corr_plane_scale(int channels);
int um_prop_global_attn_planes(const void* q_planes, const void* k_planes, const float* value, float* out, int batch, int h,
                               int w, int channels, int value_channels, int mode, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Local-window kernels (fp32 VALU, wavefront-shuffle reductions; `mode` does not apply).
 * ------------------------------------------------------------------------------------------- */

/* (2r+1)^2 (or 2r+1 when one_d) integer-offset correlation, out-of-image taps -> -1e9, softmax,
 * expected offset.  Replaces unimatch/matching.py:39-83 local_correlation_softmax (flow [B,2,h,w])
 * and :154-200 local_correlation_softmax_stereo (one_d != 0: returns -flow_x as [B,1,h,w]). */
int um_local_corr_softmax(const float* f0, const float* f1, float* out,
                          int batch, int h, int w, int channels, int radius, int one_d, void* stream);

/* Local cost volume at flow-displaced positions: out[k,p] = f0(p) . bilinear(f1, p + d_k + flow(p)) / sqrt(C),
 * zeros outside.  Replaces unimatch/matching.py:86-123 local_correlation_with_flow (dilation 1).
 * flow: [B, 2, h, w]; cost: [B, (2r+1)^2, h, w]. */
int um_local_corr_with_flow(const float* f0, const float* f1, const float* flow, float* cost,
                            int batch, int h, int w, int channels, int radius, void* stream);
/* The same with the reference's `dilation` argument (matching.py:86-91): taps at p + dilation * d_k + flow(p).  dilation = 1 is
 * um_local_corr_with_flow; larger dilations take a plain four-corner gather per tap (no caller of the reference uses them). */
int um_local_corr_with_flow_dilated(const float* f0, const float* f1, const float* flow, float* cost,
                                    int batch, int h, int w, int channels, int radius, int dilation, void* stream);
/* Same cost volume written channels-last as the operand planes of the motion encoder's 1x1 convolution
 * (unimatch/reg_refine.py:11,20): planes_out [2][plane_rows][ld], pixel row = (2r+1)^2 taps then zeros up to ld; the fp32
 * [B, taps, h, w] volume is never formed ("K4 fused into convc1"). */
int um_local_corr_with_flow_planes(const float* f0, const float* f1, const float* flow, void* planes_out, int ld, long plane_rows,
                                   int batch, int h, int w, int channels, int radius, void* stream);

/* The same cost volume on the matrix cores for locally coherent flow (csrc/local_corr_mfma.hip): an 8 x 4 pixel tile whose
 * 10 x 10 integer neighbourhoods fit a 32 x 24 window of f1 gets every dot product from one 32 x 32 x 128 MFMA product per
 * window row; other tiles (motion boundaries) take the pixel-at-a-time path inside the same kernel.  The refinement loop
 * (unimatch/unimatch.py:315-331) calls matching.py:86-123 with the SAME feature0 / feature1 in every iteration, so their fp16
 * hi | lo operand planes are built once per scale:
 *   um_local_corr_feat_planes_bytes(...)                bytes of `feat_planes`
 *   um_local_corr_feat_planes(f0, f1, feat_planes, ...) fill it ([B, h*w, 128] fp32 tokens in)
 *   um_local_corr_with_flow_feat_supported(...)         1 for radius 4 on maps of whole 8 x 4 tiles
 *   um_local_corr_with_flow_feat(...)                   exactly one of `cost` ([B, 81, h, w] fp32) and `planes_out` (as
 *                                                       um_local_corr_with_flow_planes) non-null; flags bit 0 = every tile on
 *                                                       the pixel-at-a-time path, bit 1 = natural 8 x 4 tiles only (A/B timing).
 *   Where more than a quarter of a launch's natural tiles have incoherent flow (no common 32 x 24 window) the pixels are grouped by
 *   the 16 x 8 cell of f1 they sample (counting sort on the device, scratch inside `feat_planes`) and the groups run on the matrix
 *   cores: the choice is a function of the call's flow alone and the result is bitwise reproducible (profiles/r05_k4_target_order.txt). */
size_t um_local_corr_feat_planes_bytes(int batch, int h, int w, int channels);
int um_local_corr_feat_planes(const float* f0, const float* f1, void* feat_planes, int batch, int h, int w, int channels,
                              void* stream);
int um_local_corr_with_flow_feat_supported(int h, int w, int channels, int radius);
/* um_local_corr_softmax (2-D, radius 4, maps of whole 8 x 4 tiles) on the same kernel: every pixel's 81 taps are its own 9 x 9
 * neighbourhood, so the whole map takes the matrix-core path.  workspace: um_local_corr_feat_planes_bytes(). */
int um_local_corr_softmax_mfma(const float* f0, const float* f1, float* out, int batch, int h, int w, int channels, int radius,
                               void* workspace, size_t workspace_bytes, void* stream);
int um_local_corr_with_flow_feat(const float* f0, const float* f1, const void* feat_planes, const float* flow, float* cost,
                                 void* planes_out, int ld, long plane_rows, int batch, int h, int w, int channels, int radius,
                                 int flags, unsigned* stats /* optional device [2]: += tiles on the product / pixel path */,
                                 void* stream);

/* (2r+1)^2 local self-attention propagation with zero-padded keys/values (out-of-image neighbours
 * have logit 0, value 0 and take part in the softmax).
 * Replaces the core of unimatch/attention.py:217-253 forward_local_window_attn. */
int um_prop_local_attn(const float* q, const float* k, const float* value, float* out,
                       int batch, int h, int w, int channels, int value_channels, int radius,
                       void* stream);

/* Plane-sweep depth correlation + softmax over D inverse-depth candidates + soft-argmin / argmax.
 * Replaces unimatch/matching.py:203-282 correlation_softmax_depth + warp_with_pose_depth_candidates.
 * cam: [B, 30] fp32 per-sample camera constants, row major:  K^-1 (9) | R (9) | t (3) | K (9), with K the
 * intrinsics already divided by the feature stride (unimatch/unimatch.py:147-150) and [R|t] the relative
 * pose; the kernel applies them in the reference's order (back-project, rotate, scale by depth, translate,
 * project, clamp z >= 1e-3).  candidates: [D] inverse depths.  out: [B, 1, h, w] inverse depth. */
int um_depth_corr_softmax(const float* f0, const float* f1, const float* cam, const float* candidates,
                          float* out, int batch, int h, int w, int channels, int num_candidates,
                          int from_argmax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RAFT convex upsampling (SURVEY.md 8(f) "next" row): softmax over the 9 neighbours of each of factor^2 sub-pixels,
 * weighted sum of the 3x3 (zero padded) neighbourhood of mult * flow, pixel shuffle.  Replaces
 * upsample_flow_with_mask (unimatch/utils.py:134-152).  flow: [B, V, h, w] (V = 1 | 2), mask: [B, 9*factor^2, h, w]
 * (channel = k*factor^2 + fy*factor + fx), up: [B, V, factor*h, factor*w]; mult = 1 if is_depth else factor;
 * factor 4 or 8.
 * ------------------------------------------------------------------------------------------- */
int um_convex_upsample(const float* flow, const float* mask, float* up, int batch, int channels, int h, int w,
                       int factor, int is_depth, int mask_nhwc, void* stream);
/* mask_nhwc: the mask is [batch][h*w][9*factor^2] (the output of um_conv2d_fwd) instead of NCHW. */

/* flow_warp (unimatch/geometry.py:41-72, call sites unimatch/unimatch.py:166-168): bilinear warp of a token-major feature
 * [batch][h*w][channels] by flow [batch][2][h][w] (x, y), zeros outside, align_corners -> out_tokens, same layout.
 * SURVEY.md 8(f) rank 2. */
int um_flow_warp(const float* feature_tokens, const float* flow, float* out_tokens, int batch, int h, int w, int channels,
                 void* stream);

/* The per-scale loop's small glue ops (round 3: they were torch calls):
 *   um_flow_upsample2x  out[B,V,2h,2w] = mult * bilinear_up2(flow[B,V,h,w]), align_corners = True -- unimatch/unimatch.py:162-163
 *                       (F.interpolate(..., scale_factor=2, mode='bilinear', align_corners=True) * 2: pass mult = 2)
 *   um_depth_cam_pack   cam[B or 2B][30] = Kinv | R | t | K (row major) from intrinsics [B,3,3] with rows 0-1 divided by stride_div
 *                       (unimatch.py:147-150) and pose [B,4,4]; with bidir, entries B..2B-1 carry the inverse pose
 *                       (matching.py:226-233).  Closed-form inverses (adjugate / determinant, evaluated in fp64, rounded once):
 *                       no torch.inverse (which synchronises the device).  A SINGULAR K or rotation cannot raise as
 *                       torch.inverse does: its inverse is all NaN, so that sample's predictions are NaN (other samples untouched).
 *                       Intrinsics / poses held in double by the caller are rounded to fp32 before the call.
 *   um_rigid_flow       flow[B,2,h,w] induced by inv_depth [B,1,h,w] and cam [B][30]: unimatch/geometry.py:99-195 as called
 *                       from the depth refinement (unimatch.py:295-305) */
int um_flow_upsample2x(const float* flow, float* out, int batch, int channels, int h, int w, float mult, void* stream);
int um_depth_cam_pack(const float* intrinsics, const float* pose, float* cam, int batch, float stride_div, int bidir, void* stream);
int um_rigid_flow(const float* inv_depth, const float* cam, float* flow, int batch, int h, int w, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder helper (outside the hot path of SURVEY.md section 8; added because the element-wise tail of the CNN encoder
 * had become the largest non-convolution cost):  fused InstanceNorm2d(affine=False) + ReLU (+ shortcut + ReLU),
 *   t = (x - mean) * rsqrt(var + eps) per (image, channel) plane; relu != 0: t = max(t, 0);
 *   shortcut != NULL: t = max(t + shortcut, 0).
 * Replaces nn.InstanceNorm2d + nn.ReLU (+ residual add + ReLU) of unimatch/backbone.py:7-36.
 * x, shortcut, y: [planes, hw] fp32 contiguous (NCHW with planes = N*C, hw = H*W, hw % 4 == 0).
 * ------------------------------------------------------------------------------------------- */
int um_instance_norm_fwd(const float* x, const float* shortcut, float* y, long planes, int hw, float eps,
                         int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: collecting the predictions (SURVEY.md 8b "allgather_preds", 8e).
 *
 * The path shards by sample (one process per GPU, contiguous batch split, weights replicated) and needs no collective
 * to COMPUTE; the one exchange is an all-gather of the final prediction over RCCL / xGMI.  The reference has no
 * inference-time collective; its process-group bring-up (utils/dist_utils.py:12-30, launcher
 * scripts/gmflow_scale1_train.sh:12) is what um_comm_* replaces, torch-free:
 *   um_comm_unique_id   rank 0 obtains the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *                       (torch.distributed store, MPI, a file);
 *   um_comm_init_rank   every rank: ncclCommInitRank on the calling thread's current HIP device;
 *   um_comm_init_file   both steps through a file on a filesystem all ranks see: rank 0 publishes `path` atomically, the
 *                       others poll for it up to timeout_seconds (< 0: forever).  The caller chooses a job-unique path; rank 0
 *                       removes a leftover at `path` first, the record carries the world size and a wall-clock stamp (records
 *                       older than 10 minutes are ignored) and rank 0 deletes it once every rank has joined.
 *   um_comm_init_file_nonce  the same with a caller-chosen per-job nonce stored in the record: readers skip records of another
 *                       nonce, so a job relaunched under the same path right after a crash (leftover record still fresh) never
 *                       joins the dead id.  um_comm_init_file is nonce 0 on both sides.
 *   um_allgather_preds  ncclAllGather(send, recv, count_per_rank floats) enqueued on `stream` (the caller's compute or side
 *                       stream; no host synchronisation).  recv: [world][count_per_rank], rank major.
 *   um_comm_world       number of ranks of the communicator (ncclCommCount);  um_comm_destroy releases it.
 * RCCL is bound at first use by soname (librccl.so.1); without it these return UM_ERR_UNSUPPORTED.  The communicator is
 * the ONLY state the library ever holds on behalf of a caller, and it is an explicit handle.
 * ------------------------------------------------------------------------------------------- */
#define UM_COMM_ID_BYTES 128
int um_comm_unique_id(void* id_out /* UM_COMM_ID_BYTES, host */);
int um_comm_init_rank(void** comm_out, const void* id /* UM_COMM_ID_BYTES, host */, int rank, int world);
int um_comm_init_file(void** comm_out, const char* path, int rank, int world, int timeout_seconds);
int um_comm_init_file_nonce(void** comm_out, const char* path, int rank, int world, int timeout_seconds, int nonce);
int um_comm_world(void* comm);
int um_comm_destroy(void* comm);
int um_allgather_preds(void* comm, const float* send, float* recv, size_t count_per_rank, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SURVEY.md 8(b) names.  The survey's contract lists um_<op> / um_workspace_bytes_<op>; where this header spells an
 * entry point differently the literal name is exported too (csrc/aliases.hip, thin forwards):
 *   um_swin_attn_fwd          = um_window_attn_fwd                             (attention.py:8-16, 45-104)
 *   um_attn1d_fwd             = um_window_attn_fwd with win_h = 1, shift_h = 0 (attention.py:19-42, 107-163)
 *   um_local_corr_softmax_1d  = um_local_corr_softmax with one_d = 1           (matching.py:154-200)
 *   um_workspace_bytes_<op>(batch_or_streams, h, w, channels, mode)            (0 where um_<op> takes no workspace)
 * ------------------------------------------------------------------------------------------- */
int um_swin_attn_fwd(const float* q, const float* k, const float* v, float* out, int streams, int h, int w, int channels,
                     int win_h, int win_w, int shift_h, int shift_w, int mode, void* workspace, size_t workspace_bytes,
                     void* stream);
int um_attn1d_fwd(const float* q, const float* k, const float* v, float* out, int streams, int h, int w, int channels,
                  int win_w, int shift_w, int mode, void* workspace, size_t workspace_bytes, void* stream);
int um_local_corr_softmax_1d(const float* f0, const float* f1, float* out, int batch, int h, int w, int channels, int radius,
                             void* stream);
size_t um_workspace_bytes_swin_attn_fwd(int streams, int h, int w, int channels, int mode);
size_t um_workspace_bytes_attn1d_fwd(int streams, int h, int w, int channels, int mode);
size_t um_workspace_bytes_global_corr_softmax_flow(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_global_corr_softmax_stereo(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_prop_global_attn(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_local_corr_softmax(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_local_corr_softmax_1d(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_local_corr_with_flow(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_prop_local_attn(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_depth_corr_softmax(int batch, int h, int w, int channels, int mode);
size_t um_workspace_bytes_allgather_preds(int batch, int h, int w, int channels, int mode);

/* =============================================================================================
 * DIAGNOSTIC BUILDS ONLY (`python -m unimatch_amd.build --variant diag` -> unimatch_amd/_variants/libdiag.so, loaded through
 * UM_LIB): hardware micro-benchmarks behind -DUM_DIAGNOSTIC_BUILD (csrc/microbench.hip).  The shipped libunimatch_hip.so
 * exports NO um_debug_* symbol (tests/test_host_logic_cpu.py::test_library_exports_every_declared_symbol).
 * ============================================================================================= */
#ifdef UM_DIAGNOSTIC_BUILD
/* Diagnostic: a memory-free loop of independent 32x32x16 fp16 MFMAs on every CU (8 * iters MFMAs per wave, 1024 workgroups of
 * 8 waves): the sustained matrix-pipe rate of this part under its power limit, with one constant operand value or with
 * pseudo-random operands (data toggling costs clock).  sink: any device float. */
int um_debug_mfma_peak(float* sink, int iters, int random_operands, void* stream);
/* Diagnostic: the same loop with s_memtime stamps -- 256 workgroups of `waves` (4 | 8) waves, 8 * iters MFMAs per wave, on one accumulator
 * (`chain`) or four; ticks[256 * waves] receives every wave's elapsed s_memtime ticks.  With the host's wall time of the launch this
 * calibrates the tick rate and the ticks per MFMA the attention kernel's section stamps are read against (tools/mfma_ticks.py). */
int um_debug_mfma_ticks(unsigned long long* ticks, float* sink, int iters, int random_operands, int waves, int chain, void* stream);
/* Diagnostic: the attention kernel's QK^T pattern alone -- 256 workgroups of 4 waves, per "tile" 16 ds_read_b128 two k-steps ahead of the 24
 * MFMAs they feed (mode 0), or the same stream with the MFMAs on loop-invariant A registers (mode 1); ticks[1024]. */
int um_debug_mfma_lds(unsigned long long* ticks, float* sink, int tiles, int mode, void* stream);
#endif /* UM_DIAGNOSTIC_BUILD */

#ifdef __cplusplus
}
#endif
#endif /* UNIMATCH_HIP_H */
