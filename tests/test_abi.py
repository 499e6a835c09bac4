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
    assert lib.sxg_poa_abi_version() == 5


def test_struct_layouts_match_header():
    from smoothxg_amd import poa
    assert C.sizeof(poa.Params) == 8
    assert C.sizeof(poa.BatchIn) == 8 + 5 * 8 + 5 * 4 + 4 + 8  # n_blocks(+pad), 5 pointers, 5 ints (+pad), bg_trim
    assert C.sizeof(poa.BatchOut) == 8 + 8 + 17 * 8 + 12 * 8 + 8 + 8   # n_blocks(+pad), n_seqs, 17 + 12 (block graph) pointers, block_cycles, _owner
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


def test_python_mirror_unpacks_every_result_shape_without_a_gpu():
    """The ctypes mirror's result parser on hand-made sxg_poa_batch_out structs: a full result with MSA (the MSA loop once
    shadowed the "raw arrays present" flag: every block after the first one with an MSA lost its graph arrays) and a
    block-graphs-only result (want_block_graph = 3: per-node / per-edge arrays and cons_nodes NULL)."""
    from smoothxg_amd import poa as P
    eng = object.__new__(P.PoaEngine)
    blk_off = np.array([0, 2, 3], np.int32)
    seq_off = np.array([0, 3, 6, 8], np.int64)
    eng._shape = (blk_off, seq_off)
    keep = []

    def ptr(a, t):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(t))

    def base():
        o = P.BatchOut()
        o.n_blocks, o.n_seqs = 2, 3
        o.status = ptr(np.zeros(2, np.int32), C.c_int32)
        o.node_off = ptr(np.array([0, 3, 5], np.int64), C.c_int64)
        o.edge_off = ptr(np.array([0, 2, 3], np.int64), C.c_int64)
        o.score = ptr(np.array([3, 3, 2], np.int32), C.c_int32)
        o.cells = ptr(np.array([9, 9, 4], np.uint64), C.c_uint64)
        o.cons_off = ptr(np.array([0, 3, 5], np.int64), C.c_int64)
        return o

    full = base()
    full.node_code = ptr(np.array([0, 1, 2, 3, 0], np.uint8), C.c_uint8)
    full.node_rank = ptr(np.arange(5, dtype=np.int32), C.c_int32)
    full.node_group = ptr(np.arange(5, dtype=np.int32), C.c_int32)
    full.edge_tail = ptr(np.array([0, 1, 0], np.int32), C.c_int32)
    full.edge_head = ptr(np.array([1, 2, 1], np.int32), C.c_int32)
    full.edge_weight = ptr(np.array([2, 2, 1], np.uint32), C.c_uint32)
    full.seq_path_nodes = ptr(np.array([0, 1, 2, 0, 1, 2, 0, 1], np.int32), C.c_int32)
    full.cons_nodes = ptr(np.array([0, 1, 2, 0, 1], np.int32), C.c_int32)
    msa = b"ACGACG" + b"TA"
    full.msa_off = ptr(np.array([0, 6, 8], np.int64), C.c_int64)
    full.msa_cols = ptr(np.array([3, 2], np.int32), C.c_int32)
    mbuf = np.frombuffer(msa + b"\0", np.uint8).copy()
    keep.append(mbuf)
    full.msa = mbuf.ctypes.data
    res = eng._unpack(full)
    assert [r.msa for r in res] == [["ACG", "ACG"], ["TA"]]
    assert res[0].node_code.tolist() == [0, 1, 2] and res[1].node_code.tolist() == [3, 0]      # (block 1 keeps its arrays)
    assert res[1].edge_tail.tolist() == [0] and res[1].consensus.tolist() == [0, 1] and len(res[0].paths) == 2

    only = base()
    only.bg_node_off = ptr(np.array([0, 1, 2], np.int64), C.c_int64)
    only.bg_node_len = ptr(np.array([3, 2], np.int32), C.c_int32)
    only.bg_node_outdeg = ptr(np.array([0, 0], np.int32), C.c_int32)
    only.bg_node_indeg = ptr(np.array([0, 0], np.uint8), C.c_uint8)
    only.bg_seq_off = ptr(np.array([0, 3, 5], np.int64), C.c_int64)
    sq = np.frombuffer(b"ACGTA\0", np.uint8).copy()
    keep.append(sq)
    only.bg_seq = sq.ctypes.data
    only.bg_edge_off = ptr(np.array([0, 0, 0], np.int64), C.c_int64)
    only.bg_edge_to = ptr(np.zeros(1, np.int32), C.c_int32)
    only.bg_step_off = ptr(np.array([0, 1, 2, 3], np.int64), C.c_int64)
    only.bg_steps = ptr(np.array([0, 0, 0], np.int32), C.c_int32)
    res = eng._unpack(only)
    assert all(r.node_code is None and r.edge_tail is None and r.paths is None and r.consensus is None for r in res)
    assert res[0].bg.node_seq == ["ACG"] and res[1].bg.node_seq == ["TA"] and [p.tolist() for p in res[0].bg.paths] == [[0], [0]]
    assert res[1].scores.tolist() == [2]
