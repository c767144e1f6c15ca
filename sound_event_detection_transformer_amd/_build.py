"""Build libsedt_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsedt_hip.so')
SOURCES = ['igemm.hip', 'igemm2.hip', 'igemm3.hip', 'wgrad2.hip', 'wgrad3.hip', 'wgrad4.hip', 'misc.hip', 'stem.hip', 'conv3x3_c64.hip', 'norm_attn.hip', 'attn_mfma.hip', 'criterion.hip', 'postproc.hip', 'input.hip', 'skinny.hip', 'host.cpp']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'sedt_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libsedt_hip.so')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-pass-failed',
           *[os.path.join(CSRC, s) for s in SOURCES], '-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
