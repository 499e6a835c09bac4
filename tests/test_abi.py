"""CPU: the C-ABI library loads, exports every symbol include/sxg_poa.h declares, and fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sxg_poa.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sxg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import smoothxg_amd as S
    lib = S.load_library()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    from smoothxg_amd import poa
    assert sorted(poa.EXPORTS) == names
    assert lib.sxg_poa_abi_version() == 4


def test_struct_layouts_match_header():
    from smoothxg_amd import poa
    assert C.sizeof(poa.Params) == 8
    assert C.sizeof(poa.BatchIn) == 8 + 5 * 8 + 5 * 4 + 4 + 8  # n_blocks(+pad), 5 pointers, 5 ints (+pad), bg_trim
    assert C.sizeof(poa.BatchOut) == 8 + 8 + 17 * 8 + 12 * 8 + 8   # n_blocks(+pad), n_seqs, 17 + 12 (block graph) pointers, _owner
    assert C.sizeof(poa.Stats) == 8 + 8 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 8 + 4 + 4 + 4 + 4 + 8


def test_xxh64_product_matches_python_xxhash():
    import xxhash
    import smoothxg_amd as S
    rng = np.random.default_rng(5)
    for n in (0, 1, 5, 31, 32, 33, 64, 1000):
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert S.xxh64(d) == xxhash.xxh64(d).intdigest()


def test_no_gpu_means_error_not_fallback():
    import torch
    import smoothxg_amd as S
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert S.load_library().sxg_poa_device_count() == 0
    with pytest.raises(S.PoaError):
        S.PoaEngine(0)
    h = C.c_void_p()
    assert S.load_library().sxg_poa_create(0, C.byref(h)) == -2  # SXG_E_NODEVICE
    assert b"HIP device" in S.load_library().sxg_poa_last_error()


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (judge rule)."""
    for root, _, files in os.walk(os.path.join(ROOT, "smoothxg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"#\s*include[^\n]*oracle", txt), f
                assert not re.search(r"^\s*(from|import)\s+[^\n]*oracle", txt, flags=re.M), f
                assert "libpoa_oracle" not in txt, f


def test_missing_rccl_is_an_error_code_not_a_crash():
    """sxg_poa_comm_* with no loadable librccl: SXG_E_NODEVICE and a message (the dlerror text used to be read twice,
    the second read returning NULL into a std::string)."""
    import subprocess
    import sys
    code = ("import ctypes as C, smoothxg_amd as S\n"
            "L = S.load_library()\n"
            "buf = (C.c_uint8 * 128)()\n"
            "rc = L.sxg_poa_comm_unique_id(buf)\n"
            "msg = L.sxg_poa_last_error().decode()\n"
            "assert rc == -2, rc\n"
            "assert 'librccl' in msg and 'no-such-rccl' in msg, msg\n"
            "print('ok')\n")
    env = dict(os.environ, SXG_POA_RCCL_LIB="/no-such-rccl.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
