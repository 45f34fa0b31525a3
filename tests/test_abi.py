"""CPU: librfhip.so loads, exports every symbol include/rfhip.h declares, and fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re
import subprocess

import pytest

import reforge_amd as rf
from reforge_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 60
    L = ctypes.CDLL(rf.SO_PATH)
    for s in syms:
        assert hasattr(L, s), "librfhip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "reforge_amd/_lib.py and include/rfhip.h disagree"


def test_only_rf_symbols_are_public_api():
    out = subprocess.check_output(["nm", "-D", "--defined-only", rf.SO_PATH]).decode()
    exported_c = {l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("_Z")}
    assert {s for s in exported_c if s.startswith("rf_")} == set(declared_symbols())


def test_library_does_not_depend_on_the_oracle_or_torch():
    out = subprocess.check_output(["ldd", rf.SO_PATH]).decode()
    assert "oracle" not in out and "torch" not in out
    for path in ("reforge_amd/__init__.py", "reforge_amd/host.py", "reforge_amd/_lib.py"):
        src = open(os.path.join(ROOT, path)).read()
        assert "import oracle" not in src and "from oracle" not in src and "import torch" not in src


def test_abi_version_and_error_channel():
    assert rf.lib().rf_abi_version() == 4
    with pytest.raises(rf.RfError) as e:
        rf.Config("input -> aa")
    assert e.value.status == 2 and "'output' is never used" in str(e.value)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_fails_loudly():
    with pytest.raises(rf.RfError) as e:
        rf.Context(0)
    assert e.value.status == 4          # RF_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)
