"""Build recipe for the HIP extension: ``hipcc --offload-arch=gfx950`` -> ``unimatch_amd/libunimatch_hip.so``.

In-tree on purpose: the built ``.so`` travels with a snapshot of the repository (it is git-ignored, so the
history stays source-only).  hipcc cross-compiles without a GPU.  Usage: ``python -m unimatch_amd.build``.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libunimatch_hip.so')
SOURCES = ['capi.hip', 'global_match.hip', 'window_attn.hip', 'local_ops.hip', 'linear.hip', 'ffn.hip', 'conv.hip', 'nhwc_ops.hip', 'norm_ops.hip', 'upsample.hip',
           'rccl_gather.hip', 'local_corr_mfma.hip', 'aliases.hip', 'probe.hip']
# hardware micro-benchmarks (um_debug_*): diagnostic builds only, never in the shipped library
DIAG_SOURCES = ['microbench.hip']
HEADERS = ['common.h', 'planes.h', 'timing.h', os.path.join('..', '..', 'include', 'unimatch_hip.h')]
# per-file extras: the FFN kernel's hand-placed scalar VALU stream must not be re-packed into v_pk_* by the SLP vectorizer
EXTRA_FLAGS = {'ffn.hip': ['-fno-slp-vectorize'], 'global_match.hip': ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form', '-Wno-inline-asm'], 'window_attn.hip': ['-fno-slp-vectorize'],
               'linear.hip': ['-fno-slp-vectorize']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result']


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP extension cannot be built')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library.  Returns the library path."""
    hipcc = find_hipcc()
    objdir = os.path.join(HERE, '_obj')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True, cwd=CSRC)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_variant(name, defines, verbose=False, only=None):
    """A diagnostic build ``unimatch_amd/_variants/lib<name>.so`` of the whole library with extra ``-D`` flags (A/B switches
    behind -DUM_DEBUG_SWITCHES, precision-budget experiments ...); load it with ``UM_LIB=<path>`` (tools/ab_bench.py).
    ``only``: source files the flags concern (``--only window_attn.hip,ffn.hip``) -- the other objects are the shipped build's
    (brought up to date first), which turns a 90 s variant build into a 10 s one."""
    hipcc = find_hipcc()
    objdir = os.path.join(HERE, '_variants', '_obj_' + name)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    defines = list(defines) + ['-DUM_DIAGNOSTIC_BUILD']
    if only:
        build(verbose=verbose)
    for src in SOURCES + DIAG_SOURCES:
        if only and src not in only and src not in DIAG_SOURCES:
            objs.append(os.path.join(HERE, '_obj', src.replace('.hip', '.o')))
            continue
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + list(defines) + ['-c', os.path.join(CSRC, src), '-o', o]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
        objs.append(o)
    lib = os.path.join(HERE, '_variants', f'lib{name}.so')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-ldl'], check=True)
    shutil.rmtree(objdir, ignore_errors=True)
    return lib


if __name__ == '__main__':
    if '--variant' in sys.argv:            # python -m unimatch_amd.build --variant NAME -DFOO=1 -DBAR
        i = sys.argv.index('--variant')
        only = sys.argv[sys.argv.index('--only') + 1].split(',') if '--only' in sys.argv else None
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith('-D')], verbose=True, only=only))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
