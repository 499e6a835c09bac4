"""CPU: the product's data-parallel graph code (poa_graph_dev.h, compiled with a one-thread
context by tests/csrc) must reproduce the oracle's graphs exactly (S6-S8)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import PARAM_SETS, oparams, random_block

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-s"])
    return C.CDLL(os.path.join(HERE, "csrc", "libgraph_emul.so"))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def run_emul(L, seqs, weights, params, pool=4096):
    bases = np.concatenate(seqs).astype(np.uint8)
    off = np.zeros(len(seqs) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in seqs])
    cap = int(off[-1]) + 8
    cnt = np.zeros(3, np.int32)
    code = np.zeros(cap, np.uint8)
    i32 = lambda: np.zeros(cap, np.int32)
    rank, ld, et, eh, paths, cons = i32(), i32(), i32(), i32(), i32(), i32()
    ew = np.zeros(cap, np.uint32)
    sc = np.zeros(len(seqs), np.int32)
    hints = i32()
    remain = i32()
    w = np.asarray(weights, np.uint32)
    st = L.emul_block_run(_p(bases, C.c_uint8), _p(off, C.c_int32), len(seqs), _p(w, C.c_uint32), C.byref(params),
                          pool, _p(cnt, C.c_int32), _p(code, C.c_uint8), _p(rank, C.c_int32), _p(ld, C.c_int32),
                          _p(et, C.c_int32), _p(eh, C.c_int32), _p(ew, C.c_uint32), _p(paths, C.c_int32),
                          _p(sc, C.c_int32), _p(cons, C.c_int32), _p(hints, C.c_int32), _p(remain, C.c_int32))
    n, e, nc = cnt
    return st, (code[:n], rank[:n], ld[:n], et[:e], eh[:e], ew[:e], paths[:off[-1]], sc, cons[:nc], hints[:n], remain[:n])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_graph_code_matches_oracle(emul, oracle, mode, pname):
    rng = np.random.default_rng(40 + mode)
    p = oparams(pname, mode)
    for trial in range(25):
        S = int(rng.integers(1, 16))
        if trial % 4 == 0:
            seqs = [rng.integers(0, 5, int(rng.integers(1, 30)), dtype=np.uint8) for _ in range(S)]
        else:
            seqs = random_block(rng, S, int(rng.integers(2, 300)), div=0.07)
        w = rng.integers(1, 4, len(seqs))
        g, sc, _ = oracle.block_run(seqs, w, p)
        st, r = run_emul(emul, seqs, w, p)
        assert st == 0
        code, rank, grp = g.nodes()
        t, h, ww = g.edges()
        assert (r[0] == code).all() and (r[1] == rank).all() and (r[2] == grp).all()
        assert (r[3] == t).all() and (r[4] == h).all() and (r[5] == ww).all()
        assert (r[6] == np.concatenate([g.seq_path(s) for s in range(g.n_seqs)])).all()
        assert (r[7] == sc).all()
        assert (r[8] == g.consensus()).all()
        assert (r[9] == g.row_hints()).all()      # backbone coordinates (band hints, decree B2)
        assert (r[10] == g.row_remain()).all()    # heaviest-path distance to a sink (adaptive band, decree B4)


def test_row_pool_overflow_is_reported(emul, oracle):
    """A ring too small for the live predecessor rows must surface ST_POOL_OVERFLOW (2)."""
    rng = np.random.default_rng(9)
    anc = rng.integers(0, 4, 300, dtype=np.uint8)
    dele = np.concatenate([anc[:50], anc[250:]])  # a 200-row deletion edge keeps a row alive
    seqs = [anc, anc.copy(), dele]
    seqs[1][100] = (seqs[1][100] + 1) % 4
    seqs[1][180] = (seqs[1][180] + 1) % 4
    st, _ = run_emul(emul, seqs + [anc.copy()], [1, 1, 1, 1], oparams("convex_default", 1), pool=1)
    assert st == 2
    st, _ = run_emul(emul, seqs + [anc.copy()], [1, 1, 1, 1], oparams("convex_default", 1), pool=64)
    assert st == 0
