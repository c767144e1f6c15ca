"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/sedt_hip.h declares."""
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from sound_event_detection_transformer_amd import _build, lib
    _build.build()
    l = lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'sedt_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(sedt_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    for name in declared:
        assert hasattr(l, name), name
    assert l.sedt_version() >= 1
    assert l.sedt_igemm_splitk(64, 576, 128000, 1) > 1          # host-only helper: no GPU needed


def test_struct_mirrors_have_the_sizes_the_library_was_compiled_with():
    """every hand-written mirror of an argument struct of include/sedt_hip.h - the ctypes Structures of lib.py, the numpy record dtypes
    of packing.py / optim.py that fill DEVICE job tables - has the sizeof the loaded library reports (sedt_sizeof): a field added to the
    header but not to a mirror (or the reverse) fails here, on the CPU, instead of as a kernel reading garbage"""
    import ctypes
    from sound_event_detection_transformer_amd import _build, lib, optim, packing
    _build.build()
    l = lib.load()
    want = {0: ctypes.sizeof(lib.SedtIgemm), 1: ctypes.sizeof(lib.SedtReduceJob), 2: ctypes.sizeof(lib.SedtSplitJob),
            3: ctypes.sizeof(lib.SedtPrefetch), 4: ctypes.sizeof(lib.SedtCriterion), 5: ctypes.sizeof(lib.SedtMatch),
            6: optim._DT.itemsize, 7: packing._BN.itemsize, 8: packing._PK.itemsize, 9: packing._FJ.itemsize,
            10: ctypes.sizeof(lib.SedtPoolAt), 11: ctypes.sizeof(lib.SedtCopyJob)}
    got = {k: l.sedt_sizeof(k) for k in want}
    assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    assert l.sedt_sizeof(99) == -1
    # ... and the one struct whose fields a kernel-argument copy depends on most: spot-check offsets against the header's order
    f = dict((n, getattr(lib.SedtIgemm, n).offset) for n, _ in lib.SedtIgemm._fields_)
    assert f['M'] == 0 and f['A'] == 16 and f['split_out'] % 8 == 0 and f['awrap'] == f['split_out'] + 8
    assert f['btap'] == f['btap_on'] + 4 and f['omap'] == f['f32ep'] + 4


def test_launch_log_counts_by_entry_point():
    from sound_event_detection_transformer_amd import lib
    assert lib.LAUNCH_LOG is None
    lib.check(0, 'outside')                       # no scope: nothing recorded, nothing raised
    with lib.launch_log() as log:
        lib.check(0, 'sedt_igemm')
        lib.check(0, 'sedt_igemm')
        with lib.launch_log() as inner:
            lib.check(0, 'split3')
        assert inner['split3'] == 1 and log['split3'] == 0
        lib.check(0, 'encoder_qkv_fwd')
    assert log['sedt_igemm'] == 2 and log['encoder_qkv_fwd'] == 1 and log['outside'] == 0 and lib.LAUNCH_LOG is None


def test_source_stamp_and_traffic_gate(tmp_path, monkeypatch):
    """bench.py reports the PMC profile's HBM traffic only for the build it was taken on (stamp + kernel count), else null + the reason"""
    import json
    import sys
    from sound_event_detection_transformer_amd import _build
    st = _build.source_stamp()
    assert len(st) == 16 and st == _build.source_stamp()
    sys.path.insert(0, ROOT)
    import bench
    (tmp_path / 'profiles').mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    prof, why = bench.pmc_traffic('c2', st, 200)
    assert prof is None and 'no PMC profile' in why
    (tmp_path / 'profiles' / 'r05_pmc_c2.json').write_text(json.dumps({'build_stamp': st, 'kernels_per_step': 198, 'hbm_bytes_per_step': 7}))
    prof, why = bench.pmc_traffic('c2', st, 200)
    assert why is None and prof['hbm_bytes_per_step'] == 7
    prof, why = bench.pmc_traffic('c2', st, 230)
    assert prof is None and '198 kernels per step' in why
    prof, why = bench.pmc_traffic('c2', 'deadbeefdeadbeef', 200)
    assert prof is None and 'was taken on build' in why
