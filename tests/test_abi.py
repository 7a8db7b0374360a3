"""CPU-side checks of the C-ABI library: it builds, loads and exports every symbol that
include/evreal_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def built():
    from evreal_amd import build
    return build.build(verbose=False)


def test_library_builds_and_loads(built):
    from evreal_amd import lib
    l = lib.load()
    assert l.evr_version() >= 1001      # (1001: percentile workspace contract, effective evr_model_arith)
    assert l.evr_last_error() is not None


def test_every_header_symbol_is_exported_and_typed(built):
    from evreal_amd import lib
    hdr = open(os.path.join(ROOT, 'include', 'evreal_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(evr_[a-z0-9_]+)\s*\(', hdr))
    assert declared, "no prototypes parsed"
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    l = lib.load()
    for name in declared:
        assert hasattr(l, name), name


def test_argument_validation_without_gpu(built):
    from evreal_amd import lib
    l = lib.load()
    # shape validation happens before any HIP call
    assert l.evr_voxelize_workspace_bytes(15000, 1, 5, 260, 346) > 15000 * 16
    rc = l.evr_voxelize(None, None, None, None, None, 1, 10, 0, 260, 346, None, None, None, 0, None)
    assert rc == -1 and b'evr_voxelize' in l.evr_last_error()
    assert l.evr_metrics_workspace_bytes(2, 260, 346) > 0


def test_no_fallback_without_gpu(built):
    import torch
    from evreal_amd import lib, model
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.EvrError):
        model.FireNet()
