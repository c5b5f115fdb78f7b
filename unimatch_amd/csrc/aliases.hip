// The kernel-library contract of SURVEY.md 8(b), by its literal names.
//
// The survey lists the C ABI a replacement kernel library must export as um_<op> for <op> in {swin_attn_fwd, attn1d_fwd,
// global_corr_softmax_flow, global_corr_softmax_stereo, local_corr_softmax, local_corr_softmax_1d, local_corr_with_flow,
// prop_global_attn, prop_local_attn, depth_corr_softmax, allgather_preds} plus um_workspace_bytes_<op>(dims...).  This library
// grew one windowed-attention entry point that serves the 2-D, 1-D, windowed and full variants by geometry, a `one_d` flag
// on the local correlation, and <op>_workspace_bytes spellings; the entry points below are thin forwards so that a caller
// written against the survey's list links unchanged.  (INTEGRATION.md has the name map.)
#include <hip/hip_runtime.h>
#include "../../include/unimatch_hip.h"

extern void um_set_error(const char* fmt, ...);

// unimatch/attention.py:45-104 single_head_split_window_attention (and :8-16 with win = map)
extern "C" int um_swin_attn_fwd(const float* q, const float* k, const float* v, float* out, int streams, int h, int w, int channels,
                                int win_h, int win_w, int shift_h, int shift_w, int mode, void* workspace, size_t workspace_bytes,
                                void* stream) {
    return um_window_attn_fwd(q, k, v, out, streams, h, w, channels, win_h, win_w, shift_h, shift_w, mode, workspace,
                              workspace_bytes, stream);
}

// unimatch/attention.py:19-42 single_head_full_attention_1d (win_w = w, shift_w = 0) and :107-163
// single_head_split_window_attention_1d (win_w = w / K, shift_w = win_w / 2 on shifted layers): attention along x inside every row
extern "C" int um_attn1d_fwd(const float* q, const float* k, const float* v, float* out, int streams, int h, int w, int channels,
                             int win_w, int shift_w, int mode, void* workspace, size_t workspace_bytes, void* stream) {
    return um_window_attn_fwd(q, k, v, out, streams, h, w, channels, 1, win_w, 0, shift_w, mode, workspace, workspace_bytes, stream);
}

// unimatch/matching.py:154-200 local_correlation_softmax_stereo: 2r+1 taps along x, returns -flow_x as [B, 1, h, w]
extern "C" int um_local_corr_softmax_1d(const float* f0, const float* f1, float* out, int batch, int h, int w, int channels, int radius,
                                        void* stream) {
    return um_local_corr_softmax(f0, f1, out, batch, h, w, channels, radius, 1, stream);
}

// ---- um_workspace_bytes_<op>: scratch a caller must provide for um_<op> (0 = the op takes no workspace)
extern "C" size_t um_workspace_bytes_swin_attn_fwd(int streams, int h, int w, int channels, int mode) {
    return (h > 0 && w > 0) ? um_window_attn_workspace_bytes(streams, h * w, channels, mode) : 0;
}
extern "C" size_t um_workspace_bytes_attn1d_fwd(int streams, int h, int w, int channels, int mode) {
    return (h > 0 && w > 0) ? um_window_attn_workspace_bytes(streams, h * w, channels, mode) : 0;
}
extern "C" size_t um_workspace_bytes_global_corr_softmax_flow(int batch, int h, int w, int channels, int mode) {
    return (h > 0 && w > 0) ? um_global_corr_workspace_bytes(batch, h * w, channels, mode) : 0;
}
extern "C" size_t um_workspace_bytes_global_corr_softmax_stereo(int batch, int h, int w, int channels, int mode) {
    return (h > 0 && w > 0) ? um_global_corr_workspace_bytes(batch, h * w, channels, mode) : 0;
}
extern "C" size_t um_workspace_bytes_prop_global_attn(int batch, int h, int w, int channels, int mode) {
    return (h > 0 && w > 0) ? um_global_corr_workspace_bytes(batch, h * w, channels, mode) : 0;
}
extern "C" size_t um_workspace_bytes_local_corr_softmax(int, int, int, int, int) { return 0; }
extern "C" size_t um_workspace_bytes_local_corr_softmax_1d(int, int, int, int, int) { return 0; }
extern "C" size_t um_workspace_bytes_local_corr_with_flow(int, int, int, int, int) { return 0; }
extern "C" size_t um_workspace_bytes_prop_local_attn(int, int, int, int, int) { return 0; }
extern "C" size_t um_workspace_bytes_depth_corr_softmax(int, int, int, int, int) { return 0; }
extern "C" size_t um_workspace_bytes_allgather_preds(int, int, int, int, int) { return 0; }
