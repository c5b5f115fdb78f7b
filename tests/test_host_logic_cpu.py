"""CPU tests of the product's HOST side: the drop-in boundary (constructor, forward signature, output dict,
state_dict names), the per-scale orchestration with the oracle injected as hot-path backend, the C-ABI
library (loads, exports every declared symbol) and the loud failure when no GPU / extension is present."""
import ctypes
import inspect
import os
import re
import subprocess
import sys

import pytest
import torch

from unimatch_amd import UniMatch, _abi
from unimatch_amd.model import attention_windows, sine_position_tokens
from unimatch_amd.synth import CONFIGS, synth_camera, synth_images, synth_state_dict
from oracle import hotpath as hp
from oracle import model as om
from tests.oracle_ops import OracleOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = {'gmflow_s1': (64, 96), 'gmstereo_s1': (64, 96), 'gmdepth_s1': (96, 128), 'gmdepth_s1_rr1': (96, 128),
         'gmflow_s2_rr6': (128, 192), 'gmstereo_s2_rr3': (128, 192)}


def build(name, batch=1):
    ck, fk = CONFIGS[name]
    model = UniMatch(**ck).eval()
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=0.02)
    model.load_state_dict(sd)
    hh, ww = SIZES[name]
    i0, i1 = synth_images(batch, hh, ww, seed=1000, kind='shift', normalized=(fk['task'] != 'flow'))
    kw = dict(fk)
    if fk['task'] == 'depth':
        k, pose = synth_camera(batch, hh, ww)
        kw.update(intrinsics=k, pose=pose)
    return model, sd, i0, i1, kw, ck


def oracle_out(sd, i0, i1, kw, ck):
    okw = dict(kw)
    okw.update(num_scales=ck['num_scales'], upsample_factor=ck['upsample_factor'], reg_refine=ck['reg_refine'])
    return om.unimatch_forward(sd, i0, i1, **okw)


@pytest.mark.parametrize('name', sorted(SIZES))
def test_forward_with_injected_oracle_matches_oracle_model(name):
    """Same maths on both sides (oracle kernels), so any difference is a host-logic bug in the product:
    layouts, sign flips, clamps, bidirectional concatenation, refinement loop.  fp32 re-association only."""
    model, sd, i0, i1, kw, ck = build(name)
    ops = OracleOps()
    out = model.bind_ops(ops)(i0, i1, **kw)
    assert list(out.keys()) == ['flow_preds'] and len(out['flow_preds']) == 1
    pred = out['flow_preds'][0]
    ref = oracle_out(sd, i0, i1, kw, ck)
    assert pred.shape == ref.shape
    tol = 2e-2 if 'rr' in name and 's2' in name else 1e-3       # two-scale refine: ill-conditioned (see synth)
    assert (pred - ref).abs().mean().item() < tol
    used = {c[0] for c in ops.calls}
    assert 'window_attention' in used and len([c for c in ops.calls if c[0] == 'window_attention']) == 12 * ck['num_scales']


def test_bidirectional_outputs():
    model, sd, i0, i1, kw, ck = build('gmflow_s1', batch=2)
    pred = model.bind_ops(OracleOps())(i0, i1, pred_bidir_flow=True, **kw)['flow_preds'][0]
    assert pred.shape == (4, 2, 64, 96)
    ref = oracle_out(sd, i0, i1, dict(kw, pred_bidir_flow=True), ck)
    assert (pred - ref).abs().mean().item() < 1e-3
    model, sd, i0, i1, kw, ck = build('gmdepth_s1_rr1', batch=1)
    pred = model.bind_ops(OracleOps())(i0, i1, pred_bidir_depth=True, **kw)['flow_preds'][0]
    ref = oracle_out(sd, i0, i1, dict(kw, pred_bidir_depth=True), ck)
    assert pred.shape == ref.shape == (2, 96, 128)
    assert (pred - ref).abs().mean().item() < 1e-3


def test_signature_matches_reference_boundary():
    """unimatch/unimatch.py:17-26 and :95-111."""
    ctor = list(inspect.signature(UniMatch.__init__).parameters)[1:]
    assert ctor == ['num_scales', 'feature_channels', 'upsample_factor', 'num_head', 'ffn_dim_expansion',
                    'num_transformer_layers', 'reg_refine', 'task']
    fwd = inspect.signature(UniMatch.forward).parameters
    assert list(fwd)[1:] == ['img0', 'img1', 'attn_type', 'attn_splits_list', 'corr_radius_list',
                             'prop_radius_list', 'num_reg_refine', 'pred_bidir_flow', 'task', 'intrinsics', 'pose',
                             'min_depth', 'max_depth', 'num_depth_candidates', 'depth_from_argmax',
                             'pred_bidir_depth', 'kwargs']
    assert fwd['min_depth'].default == 2.0 and fwd['max_depth'].default == 0.1 and fwd['num_reg_refine'].default == 1


def test_state_dict_names_and_param_counts():
    expected = {'gmflow_s1': 4680288, 'gmflow_s2_rr6': 7360688, 'gmstereo_s2_rr3': 7354416, 'gmdepth_s1_rr1': 7322592}
    for name, count in expected.items():
        sd = UniMatch(**CONFIGS[name][0]).state_dict()
        assert sum(v.numel() for v in sd.values()) == count
    sd = UniMatch(**CONFIGS['gmflow_s2_rr6'][0]).state_dict()
    for key in ('backbone.conv1.weight', 'backbone.layer2.0.downsample.0.bias', 'backbone.trident_conv.weight',
                'transformer.layers.5.cross_attn_ffn.mlp.2.weight', 'transformer.layers.0.self_attn.norm1.bias',
                'feature_flow_attn.k_proj.bias', 'refine_proj.weight', 'refine.encoder.convc1.weight',
                'refine.gru.convq2.bias', 'refine.flow_head.conv2.weight', 'refine.mask.2.weight'):
        assert key in sd, key
    assert 'upsampler.0.weight' in UniMatch().state_dict()
    assert 'upsampler.0.weight' not in sd


@pytest.mark.skipif(not os.path.isdir('/root/reference/unimatch'), reason='needs the reference checkout (build container only)')
@pytest.mark.parametrize('name', sorted(CONFIGS))
def test_constructor_weights_match_reference_seed_326(name):
    """SURVEY.md 8(d) prescribes the weights ``torch.manual_seed(326)`` (main_flow.py:56) + the reference constructor.  This
    package's constructor registers and initialises its parameters in the same order with the same initialisers, so the same seed
    gives BIT-IDENTICAL weights -- which is what lets the GPU box (no reference there) rebuild the reference-constructor
    weight set (tools/parity_fullsize.py, weight set ``ctor326``)."""
    sys.path.insert(0, '/root/reference')
    try:
        from unimatch.unimatch import UniMatch as RefUniMatch
    finally:
        sys.path.remove('/root/reference')
    ck = CONFIGS[name][0]
    torch.manual_seed(326)
    ref = RefUniMatch(**ck).state_dict()
    torch.manual_seed(326)
    own = UniMatch(**ck).state_dict()
    assert list(ref) == list(own)
    for k in ref:
        assert torch.equal(ref[k], own[k]), k


def test_reference_error_behaviour():
    model, sd, i0, i1, kw, ck = build('gmflow_s1')
    model.bind_ops(OracleOps())
    with pytest.raises(AssertionError):                         # list lengths must equal num_scales
        model(i0, i1, **dict(kw, attn_splits_list=[2, 8]))
    with pytest.raises(AssertionError):                         # bidirectional flow is a flow-only feature
        model(i0, i1, **dict(kw, task='stereo', pred_bidir_flow=True))
    with pytest.raises(NotImplementedError):
        UniMatch(num_head=2)
    with pytest.raises(RuntimeError):                           # inference only
        UniMatch().train()(i0, i1, **kw)
    with pytest.raises(AssertionError):                         # map not divisible by the splits
        model(i0, i1, **dict(kw, attn_splits_list=[5]))


def test_attention_windows_agree_with_oracle_dispatch():
    for attn_type in ('swin', 'self_swin2d_cross_1d', 'self_swin2d_cross_swin1d', 'other'):
        for splits in (1, 2, 4):
            for is_self in (True, False):
                for shift in (True, False):
                    assert attention_windows(attn_type, is_self, splits, 8, 16, shift) == \
                        hp.attention_geometry(attn_type, is_self, splits, 8, 16, shift)


def test_position_tokens_bit_exact(golden):
    g = golden('position')
    tab = sine_position_tokens(4, 6, 2, 2, 128)                 # [8*12, 128] tokens
    f0 = g['f0']
    want = g['add_k2_0'] - f0                                    # not exact (a+b-a), compare the sum instead
    got = f0 + tab.t().reshape(1, 128, 8, 12)
    assert torch.equal(got, g['add_k2_0'])
    assert want.shape == (2, 128, 8, 12)


# ------------------------------------------------------------------ the C-ABI library
def declared_symbols(diagnostic=False):
    """Entry points the header declares for the shipped library (or, with ``diagnostic``, inside #ifdef UM_DIAGNOSTIC_BUILD)."""
    text = open(os.path.join(ROOT, 'include', 'unimatch_hip.h')).read()
    head, _, rest = text.partition('#ifdef UM_DIAGNOSTIC_BUILD')
    diag, _, tail = rest.partition('#endif /* UM_DIAGNOSTIC_BUILD */')
    return sorted(set(re.findall(r'\b(um_[a-z0-9_]+)\s*\(', diag if diagnostic else head + tail)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_abi.LIB_PATH), 'build it with python -m unimatch_amd.build'
    lib = ctypes.CDLL(_abi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_abi.SIGNATURES) == names                     # the ctypes table mirrors the header
    assert _abi.load().um_version() == 220
    # harness / product separation: hardware micro-benchmarks exist in diagnostic builds only
    assert not any(n.startswith('um_debug_') for n in names)
    assert sorted(_abi.DIAG_SIGNATURES) == declared_symbols(diagnostic=True)
    exported = subprocess.run(['nm', '-D', '--defined-only', _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert 'um_debug_' not in exported, [ln for ln in exported.splitlines() if 'um_debug_' in ln]


def test_operand_range_flag_word_without_gpu():
    """um_range_flags: argument check, and a readable (zero) word on a host without a GPU."""
    lib = _abi.load()
    assert lib.um_range_flags(None, 0) == -1
    assert _abi.range_flags() == 0 and _abi.range_flags(reset=True) == 0
    _abi.check_operand_range()                                          # nothing raised: no exception


def test_abi_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it can be exercised on a GPU-less host."""
    lib = _abi.load()
    assert lib.um_window_attn_workspace_bytes(2, 96, 64, 0) == 0            # unsupported channel count
    assert lib.um_window_attn_workspace_bytes(2, 96, 128, 0) == 3 * 2 * 96 * 128 * 2 * 2
    assert lib.um_window_attn_workspace_bytes(2, 96, 128, 1) == 3 * 2 * 96 * 128 * 2
    rc = lib.um_window_attn_fwd(None, None, None, None, 1, 8, 12, 128, 4, 6, 0, 0, 0, None, 0, None)
    assert rc == -1 and b'null' in lib.um_last_error_string()
    fake = ctypes.c_void_p(4096)
    rc = lib.um_window_attn_fwd(fake, fake, fake, fake, 1, 8, 12, 128, 3, 6, 0, 0, 0, None, 0, None)
    assert rc == -2 and b'does not tile' in lib.um_last_error_string()
    rc = lib.um_window_attn_fwd(fake, fake, fake, fake, 1, 8, 12, 128, 4, 6, 4, 0, 0, None, 0, None)
    assert rc == -2
    rc = lib.um_window_attn_fwd(fake, fake, fake, fake, 1, 8, 12, 128, 4, 6, 2, 3, 0, None, 0, None)
    assert rc == -3 and b'workspace' in lib.um_last_error_string()
    rc = lib.um_local_corr_softmax(fake, fake, fake, 1, 8, 12, 128, 40, 0, None)
    assert rc == -4


def test_collective_entry_points_argument_errors_without_gpu():
    """um_comm_* / um_allgather_preds: exported, and their argument checks run before RCCL is touched."""
    lib = _abi.load()
    comm = ctypes.c_void_p()
    uid = (ctypes.c_ubyte * _abi.COMM_ID_BYTES)()
    assert lib.um_comm_unique_id(None) == -1
    assert lib.um_comm_init_rank(ctypes.byref(comm), uid, 2, 2) == -1 and b'outside' in lib.um_last_error_string()
    assert lib.um_comm_init_rank(None, uid, 0, 1) == -1
    assert lib.um_comm_init_file(ctypes.byref(comm), b'', 0, 1, 1) == -1
    assert lib.um_comm_init_file(ctypes.byref(comm), b'/tmp/x', 3, 2, 1) == -1
    fake = ctypes.c_void_p(4096)
    assert lib.um_allgather_preds(None, fake, fake, 16, None) == -1 and b'null communicator' in lib.um_last_error_string()
    assert lib.um_allgather_preds(fake, fake, fake, 0, None) == -1
    assert lib.um_comm_destroy(None) == 0


def test_conv_statistics_bookkeeping_without_gpu():
    """um_conv_stats_parts() (host-side dispatch, no launch): the number of statistics parts per image a convolution's
    epilogue writes -- 128-pixel parts for the generic / row-window kernels, two parts per 8 x 32 tile where the 2-D patch
    kernel serves (3x3, stride 1, pad 1, at least 3/4 of the tiling inside the image); and the argument checks of the
    entry points that consume them."""
    lib = _abi.load()
    parts = lib.um_conv_stats_parts
    assert parts(256, 384, 64, 3, 3, 1, 1, 1) == 2 * 32 * 12                 # patch kernel: whole tiles
    assert parts(22, 60, 64, 3, 3, 1, 1, 1) == 2 * 3 * 2                     # patch kernel: ragged tiles
    assert parts(40, 7, 64, 3, 3, 1, 1, 1) == (40 * 7 + 127) // 128          # 7 of 32 columns: row-window kernel instead
    assert parts(8, 12, 128, 3, 3, 1, 1, 1) == 1                             # < 256 pixels: generic kernel
    assert parts(21, 30, 96, 3, 3, 2, 1, 1) == (11 * 15 + 127) // 128        # stride 2
    assert parts(512, 768, 64, 7, 7, 2, 3, 3) == 256 * 384 // 128            # the stem
    assert parts(16, 24, 128, 1, 5, 1, 0, 2) == 3                            # GRU 1x5: row-window kernel
    assert parts(0, 24, 128, 3, 3, 1, 1, 1) == -1 and parts(4, 4, 64, 7, 7, 1, 0, 0) == -1
    assert lib.um_conv_stats_bytes(3, 24, 64) == 3 * 24 * 3 * 64 * 4 and lib.um_conv_stats_bytes(3, 0, 64) == 0
    fake = ctypes.c_void_p(4096)
    rc = lib.um_conv2d_fwd(fake, fake, ctypes.c_void_p(4100), fake, None, 1, 16, 32, 64, 64, 3, 3, 1, 1, 1, 0, 10, 0, None)
    assert rc == -1 and b'aligned' in lib.um_last_error_string()             # bias must be 16-byte aligned
    rc = lib.um_conv2d_fwd(fake, fake, None, fake, None, 1, 16, 32, 48, 64, 3, 3, 1, 1, 1, 0, 10, 0, None)
    assert rc == -1                                                          # cin not a multiple of 32


def test_small_launch_split_plans_without_gpu():
    """The host-side plans of the split variants for small launches (no launch; the CU count defaults to 256 when there is no
    device): attention splits the keys of a query tile over 2 or 4 workgroups only while all of them are resident (two per
    CU) and each part keeps at least 4 key tiles; the FFN splits its hidden slices while tiles x parts fit the CUs (one
    workgroup per CU).  A full-batch launch is never split: the byte count is 0."""
    lib = _abi.load()
    ks, hs = lib.um_window_attn_ksplit_workspace_bytes, lib.um_ffn_split_workspace_bytes
    slot_a, slot_f = 17 * 256 * 16, 64 * 256 * 4

    def ks_bytes(tiles, split):                                  # arrival counters + one slot per part
        return ((tiles * 4 + 255) // 256) * 256 + tiles * split * slot_a

    def hs_bytes(tiles, split):
        return ((tiles * 4 + 255) // 256) * 256 + tiles * split * slot_f

    assert ks(16, 64, 96, 32, 48) == 0                           # config 2, batch 8: 768 query tiles, one workgroup each

    def plan(*geo):
        f, r, k = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.um_window_attn_plan(*geo, ctypes.byref(f), ctypes.byref(r), ctypes.byref(k)) == 0
        return f.value, r.value, k.value
    assert plan(16, 64, 96, 32, 48) == (768, 0, 1)               # config 2, batch 8: a big launch is never key-split (measured: loses)
    assert plan(8, 128, 192, 16, 24) == (1536, 0, 1)             # config 4 scale 1 (4 pairs): 384-token windows, 128-query tiles
    assert plan(2, 64, 96, 32, 48) == (0, 96, 4)                 # batch 1: small launch, every tile in 4 parts
    assert ks(2, 64, 96, 32, 48) == ks_bytes(96, 4)              # batch 1 at 512x768: 96 tiles x 48 key tiles -> 4 parts
    assert ks(4, 64, 96, 32, 48) == ks_bytes(192, 2)             # batch 2: 192 tiles -> 2 parts (4 would not be resident)
    assert ks(2, 40, 56, 20, 28) == ks_bytes(40, 4)              # config 1: 560-token windows, 18 key tiles -> 4 parts of >= 4
    assert ks(2, 16, 24, 8, 12) == 0                             # 96-token windows: 3 key tiles, nothing to split
    assert ks(2, 64, 96, 30, 48) == 0 and ks(0, 64, 96, 32, 48) == 0      # windows must tile the map; bad arguments
    assert hs(16 * 6144, 1024) == 0                              # config 2, batch 8: 768 token tiles
    assert hs(2 * 2240, 1024) == hs_bytes(35, 4)                 # config 1: 35 tiles x 32 slices -> 4 parts
    assert hs(2 * 6144, 1024) == hs_bytes(96, 2)                 # batch 1 at 512x768: 96 tiles -> 2 parts
    assert hs(128, 64) == 0 and hs(128, 96) == 0                 # 2 / 3 slices: too few (or not divisible) to split
    assert hs(0, 1024) == 0 and hs(128, 1000) == 0               # bad arguments


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the GPU-less failure mode')
def test_hot_path_fails_loudly_without_gpu():
    model, sd, i0, i1, kw, ck = build('gmflow_s1')
    with pytest.raises(_abi.HipExtensionError):
        model(i0, i1, **kw)


def test_kv4_weight_packing_matches_the_header():
    """include/unimatch_hip.h (um_kv4_fwd): Wc[32 c + r][0:128] = W4[32 c + r], Wc[32 c + r][128:256] = W4[256 + 32 c + (r ^ 16)]."""
    from unimatch_amd.ops import pack_kv4_weights
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(128, 128, generator=g) for _ in range(4)]
    wc = pack_kv4_weights(ws)
    w4 = torch.cat(ws, 0)
    assert wc.shape == (256, 256) and wc.is_contiguous()
    for c in range(8):
        for r in range(32):
            assert torch.equal(wc[32 * c + r, :128], w4[32 * c + r])
            assert torch.equal(wc[32 * c + r, 128:], w4[256 + 32 * c + (r ^ 16)])
    # every output row of the four projections appears exactly once
    rows = torch.cat([wc[:, :128], wc[:, 128:]], 0)
    assert torch.equal(rows.sort(0).values, w4.sort(0).values)


def test_concurrent_wrapper_splits_and_reassembles_a_batch():
    """``ConcurrentUniMatch`` (one batch as several forwards; concurrent on HIP streams, one after the other on the CPU): uneven parts,
    the [forward; backward] layout of bidirectional outputs and per-sample camera arguments come back in the order of the one-forward
    result (the samples of a batch are independent: same result to fp32 re-association)."""
    from unimatch_amd.streams import ConcurrentUniMatch
    model, sd, i0, i1, kw, ck = build('gmflow_s1', batch=3)
    model.bind_ops(OracleOps())
    whole = model(i0, i1, pred_bidir_flow=True, **kw)['flow_preds'][0]
    for parts in (2, 3, 5):
        wrapped = ConcurrentUniMatch(model, parts=parts)
        for _ in range(2):                                           # second call: the geometry is known (same path on the CPU)
            got = wrapped(i0, i1, pred_bidir_flow=True, **kw)['flow_preds'][0]
            assert got.shape == whole.shape == (6, 2, 64, 96)
            assert (got - whole).abs().max().item() < 1e-4
    model, sd, i0, i1, kw, ck = build('gmdepth_s1_rr1', batch=2)
    model.bind_ops(OracleOps())
    whole = model(i0, i1, **kw)['flow_preds'][0]
    got = ConcurrentUniMatch(model, parts=2)(i0, i1, **kw)['flow_preds'][0]
    assert got.shape == whole.shape and (got - whole).abs().max().item() < 1e-4 * whole.abs().max().item()
    assert ConcurrentUniMatch(model, parts=1)(i0, i1, **kw)['flow_preds'][0].shape == whole.shape


def test_forward_parts_plan_is_a_pure_function_of_the_call():
    """``streams.forward_parts``: the five BASELINE configs at their own batch sizes -- two concurrent forwards for
    configs 2 - 5 (and config 4 as written), one for the single pair of config 1 and for batches whose halves no longer fill the chip
    (the thresholds of profiles/r06_forward_parts.txt)."""
    from unimatch_amd.streams import forward_parts
    from unimatch_amd.synth import CONFIGS
    plan = lambda name, b, h, w: forward_parts(CONFIGS[name][1]['task'], CONFIGS[name][1]['attn_type'], CONFIGS[name][0]['num_scales'],
                                               CONFIGS[name][0]['reg_refine'], b, h, w)
    assert plan('gmflow_s1', 1, 320, 448) == 1
    assert plan('gmflow_s1', 8, 512, 768) == 2
    assert plan('gmstereo_s2_rr3', 4, 512, 960) == 2 and plan('gmstereo_s2_rr3', 2, 512, 960) == 1
    assert plan('gmflow_s2_rr6', 4, 512, 768) == 2 and plan('gmflow_s2_rr6', 32, 512, 768) == 2 and plan('gmflow_s2_rr6', 2, 512, 768) == 2
    assert plan('gmdepth_s1', 16, 480, 640) == 2 and plan('gmdepth_s1', 2, 480, 640) == 1 and plan('gmdepth_s1', 4, 480, 640) == 2
    assert plan('gmflow_s1', 2, 512, 768) == 1 and plan('gmflow_s1', 3, 512, 768) == 2 and plan('gmflow_s1', 16, 512, 768) == 2
    assert plan('gmflow_s1', 4, 320, 448) == 1 and plan('gmflow_s1', 8, 320, 448) == 2 and plan('gmflow_s1', 8, 128, 192) == 1
    assert plan('gmstereo_s1', 4, 512, 960) == 1 and plan('gmstereo_s1', 8, 512, 960) == 2
    assert plan('gmflow_s2_rr6', 1, 512, 768) == 1
