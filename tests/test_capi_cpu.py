"""No-GPU checks of the boundary: the library builds, loads and exports every symbol
include/ebcsim.h declares; struct sizes agree with the header; no CPU fallback exists."""
import ctypes as C
import os
import re
import subprocess

import pytest

from ebcsim import _abi, _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ebcsim.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ebc_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_capi.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "eb-cadrl_amd", "csrc")])
    return _capi.lib()


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 14
    assert set(names) == set(_capi.SYMBOLS), "bindings and header disagree"
    for n in names:
        assert hasattr(lib, n), n


def test_struct_sizes_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(EbcParams),sizeof(EbcScene),sizeof(EbcStepArgs),sizeof(EbcLookaheadArgs),'
                   'sizeof(EbcStateView),sizeof(EbcStepKArgs),sizeof(EbcSceneGen),sizeof(EbcSceneOut));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    sizes = list(map(int, subprocess.check_output([str(exe)]).split()))
    assert sizes == [C.sizeof(_abi.EbcParams), C.sizeof(_abi.EbcScene), C.sizeof(_abi.EbcStepArgs),
                     C.sizeof(_abi.EbcLookaheadArgs), C.sizeof(_abi.EbcStateView), C.sizeof(_abi.EbcStepKArgs),
                     C.sizeof(_abi.EbcSceneGen), C.sizeof(_abi.EbcSceneOut)]


def test_params_default_matches_bindings(lib):
    p = _abi.EbcParams()
    assert lib.ebc_params_default(C.addressof(p)) == 0
    q = _abi.default_params()
    assert p.struct_size == q.struct_size == C.sizeof(_abi.EbcParams)
    assert p.orca_neighbor_dist == q.orca_neighbor_dist == 10.0
    assert p.orca_max_neighbors == q.orca_max_neighbors == 10
    assert p.time_good == q.time_good == 10.0


def test_argument_validation_without_device(lib):
    h = C.c_void_p()
    p = _abi.default_params()
    p.struct_size = 7
    assert lib.ebc_create(0, 4, 5, 0, C.addressof(p), C.byref(h)) == _abi.ERR_INVALID
    assert b"struct_size" in lib.ebc_last_error()
    p = _abi.default_params()
    assert lib.ebc_create(0, 4, 64, 0, C.addressof(p), C.byref(h)) == _abi.ERR_UNSUPPORTED
    assert lib.ebc_step(None, None) == _abi.ERR_INVALID


def test_no_cpu_fallback(lib):
    """Without a HIP device creation must fail loudly (the product never computes on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ebcsim.batched import BatchedEnv
    with pytest.raises(_capi.EbcError) as ei:
        BatchedEnv(_abi.default_params(), 4, 5, 0)
    assert ei.value.code == _abi.ERR_DEVICE


def test_product_does_not_import_oracle():
    """Nothing under eb-cadrl_amd/ may import, include, link or load the checker (comments that
    cite it are fine)."""
    pkg = os.path.join(ROOT, "eb-cadrl_amd")
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b)|(#\s*include\s*[\"<][^\">]*oracle)|"
                     r"(libebc_oracle)|(-lebc_oracle)|(oracle\.(py|so)\b)", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dp, f)).read()
                assert not bad.search(text), os.path.join(dp, f)
