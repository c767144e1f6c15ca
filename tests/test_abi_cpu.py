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
