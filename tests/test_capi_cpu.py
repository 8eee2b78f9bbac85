"""CPU-side checks of the C-ABI library: it builds, loads, and exports every symbol that
include/loraine_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import loraine_jl_amd
    if not os.path.exists(loraine_jl_amd.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return loraine_jl_amd.load_library()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "loraine_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lrn_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported(lib):
    from loraine_jl_amd import _capi
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/loraine_hip.h but not exported"
        assert n in _capi.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_capi.SIGNATURES) == names


def test_version_and_device_count(lib):
    assert lib.lrn_version() >= 100
    assert lib.lrn_device_count() >= 0


def test_no_cpu_fallback(lib):
    """The product path must fail loudly without a GPU."""
    import loraine_jl_amd
    if lib.lrn_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(loraine_jl_amd.LoraineHipError):
        loraine_jl_amd.Device(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "loraine.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
