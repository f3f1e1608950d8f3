"""The C-ABI library loads and exports every symbol include/nnmpc.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from industrial_nnmpc_2021_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nnmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nnmpc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product path refuses loudly instead of computing on the CPU."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    with pytest.raises(_lib.NnmpcError, match="no HIP device|HIP"):
        BatchedBoxQP(np.eye(4), np.eye(4), 2, Kunc=None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "industrial_nnmpc_2021_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
