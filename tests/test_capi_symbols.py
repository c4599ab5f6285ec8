"""CPU: the C-ABI library loads and exports every symbol include/aznet_hip.h declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "aznet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from aznet_hip import ffi
    if not os.path.exists(ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ffi.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libaznet_hip.so does not export %s" % n
    assert sorted(ffi.SYMBOLS) == names
    assert b"gfx950" in L.az_version()


def test_constants_the_binding_restates():
    from aznet_hip import ffi
    src = open(os.path.join(REPO, "include", "aznet_hip.h")).read()
    assert int(re.search(r"#define\s+AZ_BATCH_MAX\s+(\d+)", src).group(1)) == ffi.AZ_BATCH_MAX
    assert sorted(ffi.SEARCH_FORMS) == [0, 1, 2, 3, 4, 5]


def test_no_gpu_means_loud_failure():
    """Without a gfx950 device az_create must fail (AZ_ERR_NO_DEVICE) -- never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aznet_hip import ffi
    with pytest.raises(ffi.AzError) as e:
        ffi.AzContext(0)
    assert e.value.code == ffi.AZ_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under az-net_amd/ may reference it."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "az-net_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) or "liboracle" in s:
                    bad.append(os.path.join(root, f))
    assert not bad, bad
