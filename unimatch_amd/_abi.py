"""ctypes binding of ``libunimatch_hip.so`` (the C ABI declared in ``include/unimatch_hip.h``).

There is deliberately NO fallback: if the library is missing or a symbol cannot be resolved the import
of the hot path fails loudly.  The product never computes the hot path any other way.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UM_LIB') or os.path.join(_HERE, 'libunimatch_hip.so')   # UM_LIB: diagnostics builds only

MODE_EXACT = 0
MODE_FAST = 1

# launch census ids (UM_V_* of the header): which kernel instantiation served a call
CENSUS = {'wattn_tile': 0, 'wattn_ksplit': 1, 'ffn_tile': 2, 'ffn_hsplit': 3, 'gsv4': 4, 'gsv3': 5, 'k4_mfma': 6, 'k4_valu': 7,
          'k3_mfma': 8, 'k3_valu': 9, 'conv_patch': 10, 'conv_rows': 11, 'conv_generic': 12, 'wattn_w8': 13}

_c_int, _c_size_t, _c_void_p = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/unimatch_hip.h one to one
SIGNATURES = {
    'um_version': (_c_int, []),
    'um_last_error_string': (ctypes.c_char_p, []),
    'um_range_flags': (_c_int, [ctypes.POINTER(ctypes.c_uint), _c_int]),
    'um_timing_enable': (_c_int, [_c_int]),
    'um_timing_collect': (_c_int, [_c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_int)]),
    'um_window_attn_workspace_bytes': (_c_size_t, [_c_int] * 4),
    'um_window_attn_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 9 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_window_attn_planes_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [ctypes.c_long] * 2 + [_c_int] * 6 + [_c_void_p]),
    'um_window_attn_merge_fwd': (_c_int, [_c_void_p] * 7 + [ctypes.c_float, _c_int, _c_void_p] + [_c_int] * 6 + [ctypes.c_long] * 2 +
                                 [_c_int] * 6 + [_c_void_p]),
    'um_window_attn_qproj_merge_fwd': (_c_int, [_c_void_p] * 8 + [ctypes.c_float, _c_int, _c_void_p] + [_c_int] * 5 + [ctypes.c_long] +
                                       [_c_int] * 6 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_window_attn_ksplit_workspace_bytes': (_c_size_t, [_c_int] * 5),
    'um_window_attn_plan': (_c_int, [_c_int] * 5 + [ctypes.POINTER(_c_int)] * 3),
    'um_planes_bytes': (_c_size_t, [ctypes.c_long, _c_int, _c_int]),
    'um_weight_planes': (_c_int, [_c_void_p] * 2 + [_c_int] * 4 + [_c_void_p]),
    'um_linear_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 5 + [_c_void_p] * 4 + [ctypes.c_float, _c_int, _c_void_p]),
    'um_linear_bias_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 4 + [ctypes.c_float] * 2 + [_c_void_p] * 2 + [_c_int, _c_void_p]),
    'um_ffn_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p] * 2 + [ctypes.c_float, _c_void_p, _c_int, _c_void_p]),
    'um_ffn_ws_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p] * 2 + [ctypes.c_float, _c_void_p, _c_int, _c_void_p, _c_size_t,
                                _c_void_p]),
    'um_ffn_split_workspace_bytes': (_c_size_t, [_c_int] * 2),
    'um_ffn_kv_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p] * 2 + [ctypes.c_float, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                                _c_size_t, _c_void_p]),
    'um_kv4_fwd': (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_void_p]),
    'um_conv2d_fwd': (_c_int, [_c_void_p] * 5 + [_c_int] * 13 + [_c_void_p]),
    'um_conv2d_ex': (_c_int, [_c_void_p, _c_int, _c_int, ctypes.c_long, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p,
                              _c_int, _c_int, ctypes.c_long, _c_void_p] + [_c_int] * 13 + [_c_void_p]),
    'um_nhwc_gate': (_c_int, [_c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, ctypes.c_long, ctypes.c_long,
                              _c_int, _c_void_p]),
    'um_local_corr_with_flow_planes': (_c_int, [_c_void_p] * 4 + [_c_int, ctypes.c_long] + [_c_int] * 5 + [_c_void_p]),
    'um_conv2d_gru_fwd': (_c_int, [_c_int, _c_void_p, _c_int, _c_int, ctypes.c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                   _c_void_p, _c_int, _c_void_p, _c_int, _c_int, ctypes.c_long] + [_c_int] * 11 + [_c_void_p]),
    'um_conv2d_gru_add_fwd': (_c_int, [_c_int, _c_void_p, _c_int, _c_int, ctypes.c_long, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                        _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, ctypes.c_long] + [_c_int] * 11 + [_c_void_p]),
    'um_conv_stats_bytes': (_c_size_t, [_c_int] * 3),
    'um_conv_stats_parts': (_c_int, [_c_int] * 8),
    'um_stem_planes_bytes': (_c_size_t, [_c_int] * 3),
    'um_conv7_planes_bytes': (_c_size_t, [_c_int] * 4),
    'um_conv7_fwd': (_c_int, [_c_void_p, _c_int, _c_int] + [_c_void_p] * 6 + [_c_int, _c_int, _c_void_p, _c_int, _c_int, ctypes.c_long,
                              _c_void_p] + [_c_int] * 7 + [_c_void_p]),
    'um_stem_conv_fwd': (_c_int, [_c_void_p, _c_int] + [_c_void_p] * 6 + [_c_int] * 5 + [_c_void_p]),
    'um_nhwc_norm_workspace_bytes': (_c_size_t, [_c_int] * 3),
    'um_nhwc_instance_norm': (_c_int, [_c_void_p] * 5 + [_c_int] * 3 + [ctypes.c_float, _c_int, _c_int, _c_void_p, _c_int, _c_void_p,
                                       _c_size_t, _c_int, _c_void_p]),
    'um_nchw_to_nhwc': (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_void_p]),
    'um_flow_warp': (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_void_p]),
    'um_convex_upsample': (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_void_p]),
    'um_flow_upsample2x': (_c_int, [_c_void_p] * 2 + [_c_int] * 4 + [ctypes.c_float, _c_void_p]),
    'um_depth_cam_pack': (_c_int, [_c_void_p] * 3 + [_c_int, ctypes.c_float, _c_int, _c_void_p]),
    'um_rigid_flow': (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p]),
    'um_instance_norm_fwd': (_c_int, [_c_void_p] * 3 + [ctypes.c_long, _c_int, ctypes.c_float, _c_int, _c_void_p]),
    'um_global_corr_workspace_bytes': (_c_size_t, [_c_int] * 4),
    'um_global_corr_softmax_flow': (_c_int, [_c_void_p] * 3 + [_c_int] * 6 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_global_corr_softmax_stereo': (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_prop_global_attn': (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_global_corr_plane_scale': (ctypes.c_float, [_c_int]),
    'um_prop_global_attn_planes': (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_local_corr_softmax': (_c_int, [_c_void_p] * 3 + [_c_int] * 6 + [_c_void_p]),
    'um_local_corr_with_flow': (_c_int, [_c_void_p] * 4 + [_c_int] * 5 + [_c_void_p]),
    'um_local_corr_with_flow_dilated': (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p]),
    'um_local_corr_feat_planes_bytes': (ctypes.c_size_t, [_c_int] * 4),
    'um_local_corr_feat_planes': (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_void_p]),
    'um_local_corr_with_flow_feat_supported': (_c_int, [_c_int] * 4),
    'um_local_corr_softmax_mfma': (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p, ctypes.c_size_t, _c_void_p]),
    'um_local_corr_with_flow_feat': (_c_int, [_c_void_p] * 6 + [_c_int, ctypes.c_long] + [_c_int] * 6 + [_c_void_p, _c_void_p]),
    'um_prop_local_attn': (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p]),
    'um_depth_corr_softmax': (_c_int, [_c_void_p] * 5 + [_c_int] * 6 + [_c_void_p]),
    'um_census_enable': (_c_int, [_c_int]),
    'um_census_count': (ctypes.c_long, [_c_int]),
    'um_window_attn_tile_census': (_c_int, [_c_int, ctypes.POINTER(ctypes.c_ulonglong)]),
    'um_probe_mfma_flops': (ctypes.c_double, [_c_int]),
    'um_probe_mfma': (_c_int, [_c_void_p, _c_int, _c_void_p]),
    'um_probe_copy': (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    'um_probe_chase': (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p]),
    'um_swin_attn_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 9 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_attn1d_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 7 + [_c_void_p, _c_size_t, _c_void_p]),
    'um_local_corr_softmax_1d': (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p]),
    **{f'um_workspace_bytes_{op}': (_c_size_t, [_c_int] * 5) for op in (
        'swin_attn_fwd', 'attn1d_fwd', 'global_corr_softmax_flow', 'global_corr_softmax_stereo', 'prop_global_attn',
        'local_corr_softmax', 'local_corr_softmax_1d', 'local_corr_with_flow', 'prop_local_attn', 'depth_corr_softmax',
        'allgather_preds')},
    'um_comm_unique_id': (_c_int, [_c_void_p]),
    'um_comm_init_rank': (_c_int, [ctypes.POINTER(_c_void_p), _c_void_p, _c_int, _c_int]),
    'um_comm_init_file': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_char_p, _c_int, _c_int, _c_int]),
    'um_comm_init_file_nonce': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_char_p, _c_int, _c_int, _c_int, _c_int]),
    'um_comm_world': (_c_int, [_c_void_p]),
    'um_comm_destroy': (_c_int, [_c_void_p]),
    'um_allgather_preds': (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
}
COMM_ID_BYTES = 128
# hardware micro-benchmarks: exported by diagnostic builds only (`python -m unimatch_amd.build --variant diag`, UM_LIB=...)
DIAG_SIGNATURES = {
    'um_debug_mfma_peak': (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p]),
    'um_debug_mfma_lds': (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p]),
    'um_debug_mfma_ticks': (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
}

# um_range_flags bits (UM_RANGE_* of the header) -> what overflowed fp16's range on its way into an exact-mode operand
RANGE_ROLES = {1: 'a tensor converted to operand planes (q / k / v / features)', 2: 'attention source tokens', 4: 'projected queries',
               8: 'attention output (input of the merge Linear)', 16: 'FFN input [x | y]', 32: 'FFN hidden activations',
               64: 'tokens / projected keys and values of the k | v projection', 128: 'inputs / plane outputs of a Linear'}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def load():
    """Load the shared library once and attach prototypes.  Raises HipExtensionError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f'{LIB_PATH} is missing: build it with `python -m unimatch_amd.build` '
            '(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the hot path.')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise HipExtensionError(f'cannot load {LIB_PATH}: {exc}') from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise HipExtensionError(f'{LIB_PATH} does not export {name}') from exc
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SIGNATURES.items():        # present in diagnostic builds only
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def load_diagnostic():
    """The library with the um_debug_* micro-benchmarks: a diagnostic build loaded through UM_LIB."""
    lib = load()
    if not hasattr(lib, 'um_debug_mfma_peak'):
        raise HipExtensionError('the shipped library carries no um_debug_* symbol: build a diagnostic variant '
                                '(`python -m unimatch_amd.build --variant diag`) and set UM_LIB=unimatch_amd/_variants/libdiag.so')
    return lib


def census(lib=None):
    """Launch counts since the last ``um_census_enable(1)`` as ``{name: count}``."""
    lib = lib or load()
    return {k: int(lib.um_census_count(v)) for k, v in CENSUS.items()}


def attn_tile_census(enable):
    """``um_window_attn_tile_census``: read-and-zero the key-tile counters of the current device, then switch the census on / off.
    Returns ``{'full', 'probed', 'probed_then_computed', 'workgroups'}``."""
    out = (ctypes.c_ulonglong * 4)()
    check(load().um_window_attn_tile_census(1 if enable else 0, out), 'um_window_attn_tile_census')
    return dict(zip(('full', 'probed', 'probed_then_computed', 'workgroups'), (int(v) for v in out)))


def check(code, what):
    """Translate the C ABI's return convention into Python exceptions."""
    if code == 0:
        return
    msg = load().um_last_error_string().decode('utf-8', 'replace')
    if code < 0:
        raise ValueError(f'{what}: {msg} (code {code})')
    raise RuntimeError(f'{what}: HIP error {code}')


class OperandRangeError(FloatingPointError):
    """An activation reached fp16's largest finite value (65504) on its way into an exact-mode MFMA operand."""


def range_flags(reset=False):
    """The sticky operand-range word (``um_range_flags``): a bit per kind of operand that overflowed in a FINISHED launch."""
    out = ctypes.c_uint(0)
    check(load().um_range_flags(ctypes.byref(out), 1 if reset else 0), 'um_range_flags')
    return out.value


def check_operand_range(where=''):
    """Raise :class:`OperandRangeError` naming the operands that overflowed since the last check (and clear the flags)."""
    flags = range_flags(reset=True)
    if flags:
        roles = '; '.join(v for k, v in RANGE_ROLES.items() if flags & k)
        raise OperandRangeError(
            f'{where}exact mode splits fp32 values into fp16 hi | lo operands, and an element of magnitude >= 65504 reached: {roles} '
            '(the hi plane is inf there and NaN follows downstream; the fp32 reference has no such limit).  Scale the inputs / '
            "weights by a power of two, or run precision='fast' (bf16 operands: fp32's exponent range).")
