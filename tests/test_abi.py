"""The C-ABI shared library loads and exports every symbol include/p2s_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(REPO, 'include', 'p2s_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(p2s_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    from points2surf_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    from points2surf_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), 'symbol %s declared in include/p2s_hip.h is not exported' % s
    # the ctypes prototype table covers exactly the header
    assert sorted(_lib.PROTOTYPES) == syms


def test_abi_version_and_error_string(lib):
    assert lib.p2s_abi_version() == 5
    assert isinstance(lib.p2s_last_error(), bytes)
    assert lib.p2s_device_count() >= 0


def test_struct_sizes_match_header():
    from points2surf_amd import weights, _lib
    assert ctypes.sizeof(weights.ModelCfg) == 16 * 4 and weights.ModelCfg.fixed_subsample.offset == 8 * 4
    assert ctypes.sizeof(weights.EncoderOffsets) == 22 * 8
    assert ctypes.sizeof(weights.QstnOffsets) == 12 * 8
    assert ctypes.sizeof(weights.WeightOffsets) == (2 * 22 + 12 + 10) * 8
    assert ctypes.sizeof(_lib.Counters) == 7 * 8 + 2 * 8 + 8 * 8


def test_no_device_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    with pytest.raises(RuntimeError):
        engine.Model(w, cfg)
    # straight through the C ABI as well: an error code, not a crash / silent CPU path
    from points2surf_amd import weights
    blob, offs, mc = weights.build_blob(w, cfg)
    h = ctypes.c_void_p()
    rc = lib.p2s_model_create(ctypes.byref(mc), blob.ctypes.data_as(ctypes.c_void_p), blob.size, ctypes.byref(offs), 0,
                              ctypes.byref(h))
    assert rc == -5 and b'device' in lib.p2s_last_error()


def test_integration_stub_uses_only_declared_entry_points():
    """the reference-side ctypes stub in INTEGRATION.md binds symbols that include/p2s_hip.h declares"""
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    used = set(re.findall(r'\b(p2s_[a-z0-9_]+)\b', text))
    declared = set(_declared_symbols()) | {'p2s_hip', 'p2s_model_cfg', 'p2s_counters', 'p2s_mi355x', 'p2s_max', 'p2s_vanilla', 'p2s_uniform',
                                          'p2s_no_qstn', 'p2s_small_knn', 'p2s_large_knn', 'p2s_regression', 'p2s_shared_encoder',
                                          'p2s_shared_transformer', 'p2s_vanilla_ablation', 'p2s_vanilla_lower_lr',
                                          'p2s_vanilla_uniform_subsample', 'p2s_'}   # + non-symbols (model names)
    unknown = {u for u in used if u not in declared and not u.endswith('_t')}
    assert not unknown, unknown


def test_every_environment_switch_is_documented():
    """each P2S_* variable the library, the Python package or bench.py reads appears in INTEGRATION.md"""
    import glob
    names = set()
    files = glob.glob(os.path.join(REPO, 'points2surf_amd', 'csrc', '*')) + glob.glob(os.path.join(REPO, 'points2surf_amd', '*.py')) + \
        glob.glob(os.path.join(REPO, 'points2surf_amd', 'dropin', '**', '*.py'), recursive=True) + [os.path.join(REPO, 'bench.py')]
    for f in files:
        t = open(f).read()
        names |= set(re.findall(r'getenv\("(P2S_[A-Z0-9_]+)"\)', t))
        names |= set(re.findall(r"environ(?:\.get)?\(?\[?'(P2S_[A-Z0-9_]+)'", t))
    doc = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    missing = sorted(n for n in names if n not in doc)
    assert 10 <= len(names) <= 20 and not missing, (len(names), missing)        # at most 20 switches


def test_kernels_named_in_the_docs_exist():
    """every `..._kernel` name in DESIGN.md / INTEGRATION.md is a kernel of points2surf_amd/csrc (HISTORY.md and
    profiles/README.md are records of earlier rounds and may name kernels that are gone)"""
    import glob
    src = ''.join(open(f).read() for f in glob.glob(os.path.join(REPO, 'points2surf_amd', 'csrc', '*')))
    for doc in ('DESIGN.md', 'INTEGRATION.md'):
        t = open(os.path.join(REPO, doc)).read()
        names = set(re.findall(r'`((?:p2s|wc|vol|mc|mt|rs)_[a-z0-9_]*_kernel)(?:<[^`]*>)?`', t))
        missing = sorted(n for n in names if ('void ' + n) not in src)
        assert (len(names) >= 10 or doc != 'DESIGN.md') and not missing, (doc, missing)
