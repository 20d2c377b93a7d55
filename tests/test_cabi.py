"""The C-ABI shared library loads without a GPU and exports exactly what include/dpk.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "dpk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dpk_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from dp_timeautodiff_amd import _native
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dpk.h but not exported"
    assert sorted(_native.EXPORTS) == names, "ctypes table and header disagree"


def test_host_only_entry_points():
    from dp_timeautodiff_amd import _native
    lib = _native.load()
    assert lib.dpk_version() == 100
    assert lib.dpk_error_string(0) == b"ok"
    assert b"aligned" in lib.dpk_error_string(-3)
    assert lib.dpk_linear_slots(400, 200) > 0 and lib.dpk_linear_slots(0, 5) < 0
    assert lib.dpk_normsq_slots(649400) == 80
    assert lib.dpk_clip_reduce_workspace(64, 649408) % 16 == 0
    assert lib.dpk_embedding_slots(48, 128) == 1 and lib.dpk_layernorm_slots(500) == 1


def test_product_refuses_cpu_tensors():
    import pytest
    import torch
    from dp_timeautodiff_amd import _native
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    with pytest.raises(_native.DpkError, match="no CPU"):
        GradSampleModule(torch.nn.Linear(3, 2))
    base = torch.zeros(2, 64)
    with pytest.raises(_native.DpkError):
        _native.clip_factors(torch.zeros(2, 4), 4, torch.zeros(2), None, 2, 1.0)
