"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/simhand_hip.h (the product surface) and include/simhand_hip_test.h (the
instruments: route counters, event profiler, test hooks) declare; the product header declares
no instrument; the product package has no CPU compute path."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="simhand_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(simhand_[a-z0-9_]+)\s*\(", text)))


_INSTRUMENT = ("simhand_test_", "simhand_prof_", "simhand_route_")


def test_library_exports_every_declared_symbol():
    from simhand_amd import _lib

    lib = _lib.load()
    product, instruments = _declared(), _declared("simhand_hip_test.h")
    assert len(product) >= 40
    for n in product + instruments:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == sorted(product + instruments)
    assert lib.simhand_abi_version() == _lib.ABI_VERSION == 5


def test_product_header_declares_no_instrument():
    """VERDICT r5 "Next" 6: the drop-in surface and the instruments live in separate headers."""
    product, instruments = _declared(), _declared("simhand_hip_test.h")
    assert not [n for n in product if n.startswith(_INSTRUMENT)], "instrument declared in the product header"
    assert instruments and all(n.startswith(_INSTRUMENT) for n in instruments), "product entry point declared in the test header"
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "simhand_hip.h")).read(), flags=re.S)  # (comments may name them)
    assert "sh_test_switch" not in text and "enum sh_route" not in text and "enum sh_prof_class" not in text


def test_compute_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from simhand_amd import _lib, ops

    with pytest.raises(_lib.SimhandHipError):
        ops.proj_stats(torch.zeros(4, 128))
    with pytest.raises(_lib.SimhandHipError):
        _lib.require_device()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "simhand_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src, f"{f} touches the reference tree"
