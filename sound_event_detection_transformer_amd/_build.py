"""Build libsedt_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

One object per source file (cached under build/obj, recompiled when the source or any header is newer), compiled in
parallel, then one link: editing a kernel costs the compile of its own file, not of the library."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsedt_hip.so')
OBJ = os.path.join(HERE, '..', 'build', 'obj')
SOURCES = ['igemm.hip', 'igemm3.hip', 'wgrad3.hip', 'wgrad4.hip', 'misc.hip', 'stem.hip', 'conv3x3_c64.hip',
           'norm_attn.hip', 'attn_mfma.hip', 'attn_f32_mfma.hip', 'enc_slab.hip', 'heads_slab.hip', 'bneck.hip', 'bneck3.hip', 'criterion.hip', 'postproc.hip', 'input.hip', 'skinny.hip', 'split3.hip', 'pool_at.hip', 'host.cpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed']
if os.environ.get('SEDT_DEV_BUILD') == '1':
    # developer build: the tuning switches read the environment (csrc/common.h: dev_getenv) and the ablation hooks are compiled in.
    # Written beside the product library, never loaded unless SEDT_DEV=1 SEDT_LIB_AB=<path> asks for it (lib.py)
    FLAGS = FLAGS + ['-DSEDT_DEV']
    LIB = os.path.join(HERE, '..', 'build', 'dev', 'libsedt_hip_dev.so')
    OBJ = os.path.join(HERE, '..', 'build', 'dev', 'obj')


def source_stamp():
    """sha256 (16 hex digits) over everything that decides which kernels a step launches and what they do: csrc/*, the C-ABI header and
    the package's Python sources.  bench.py prints it (`build_stamp`); tools/pmc_step_summary.py writes it into profiles/rNN_pmc_*.json,
    and bench.py reports the profile's HBM traffic only while the stamps agree (a stale profile gives `traffic: null` + the reason)"""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h', '.cpp'))]
    files.append(os.path.join(HERE, '..', 'include', 'sedt_hip.h'))
    for d, _, fs in sorted(os.walk(HERE)):
        if '__pycache__' in d or os.sep + 'csrc' in d:
            continue
        files += [os.path.join(d, f) for f in sorted(fs) if f.endswith('.py')]
    for f in files:
        h.update(os.path.relpath(f, HERE).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return hs + [os.path.join(HERE, '..', 'include', 'sedt_hip.h'), os.path.abspath(__file__)]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, jobs=None):
    """concurrent importers (torchrun ranks, mp.spawn test workers) serialise on build/.lock: one of them compiles, the others find
    everything up to date; objects and the library are written to a temporary name and renamed"""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, '..', '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objs = [os.path.join(OBJ, s + '.o') for s in SOURCES]
    heads = _headers()
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _newer(o, [s] + heads)]
    if not todo and not _newer(LIB, objs):
        return LIB
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libsedt_hip.so')
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(so):
        tmp = f'{so[1]}.{os.getpid()}.tmp'
        cmd = [hipcc, *FLAGS, '-c', so[0], '-o', tmp]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {so[0]}:\n' + r.stdout + r.stderr)
        os.replace(tmp, so[1])

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    tmp = f'{LIB}.{os.getpid()}.tmp'
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
