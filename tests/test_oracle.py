"""CPU tests of the oracle itself (no GPU): the parts that CAN be pinned are pinned.

The reference holds no golden vector for this path (parity unpinned), so
  - XXH64 is pinned against the python `xxhash` wheel and the published empty-input vector;
  - the DP is pinned against an independent textbook pairwise implementation (numpy) on
    single-sequence graphs, and against an independent brute-force DAG DP written here;
  - every alignment is re-scored independently of the DP (poa_rescore);
  - graph invariants the reference itself relies on (src/main.cpp:770-810: every path spells
    its sequence) are asserted;
  - self-generated golden fixtures (tests/golden, labelled self-oracle) guard regressions.
"""
import json
import os

import numpy as np
import pytest

from helpers import PARAM_SETS, oparams, random_block
from smoothxg_amd import synth

NEG = -(1 << 29)


def test_xxh64_pinned(oracle):
    import xxhash
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999  # published vector
    rng = np.random.default_rng(0)
    for n in list(range(0, 70)) + [255, 256, 1000, 4097]:
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert oracle.xxh64(d) == xxhash.xxh64(d).intdigest()
        assert oracle.xxh64(d, 12345) == xxhash.xxh64(d, seed=12345).intdigest()


def _norm(p):
    m, n, g, e, q, c = p
    if g >= e:
        e = q = c = g
    elif g <= q or e >= c:
        q, c = g, e
    return m, n, g, e, q, c


def _dag_dp_score(codes, off, pred, sink, seq, p, sw):
    """Independent plain-python two-piece-affine DP over a DAG in rank order (score only)."""
    m, n, g, e, q, c = _norm(p)
    N, L = len(codes), len(seq)
    H = np.full((N + 1, L + 1), NEG, np.int64)
    F = np.full((N + 1, L + 1), NEG, np.int64)
    Oo = np.full((N + 1, L + 1), NEG, np.int64)
    H[0, 0] = 0
    for j in range(1, L + 1):
        H[0, j] = 0 if sw else max(g + (j - 1) * e, q + (j - 1) * c)
    best = 0 if sw else None
    for i in range(1, N + 1):
        ps = list(pred[off[i - 1]:off[i]]) or [0]
        E = Q = NEG
        for j in range(0, L + 1):
            f = max(max(H[x, j] + g, F[x, j] + e) for x in ps)
            o = max(max(H[x, j] + q, Oo[x, j] + c) for x in ps)
            h = max(f, o)
            if j > 0:
                d = max(H[x, j - 1] for x in ps) + (m if codes[i - 1] == seq[j - 1] else n)
                E = max(H[i, j - 1] + g, E + e)
                Q = max(H[i, j - 1] + q, Q + c)
                h = max(h, d, E, Q)
            if sw:
                h = max(h, 0)
                best = max(best, h)
            H[i, j], F[i, j], Oo[i, j] = h, f, o
        if not sw and sink[i - 1]:
            best = H[i, L] if best is None else max(best, H[i, L])
    return int(best)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_dp_score_matches_independent_dag_dp(oracle, mode, pname):
    rng = np.random.default_rng(3)
    for trial in range(6):
        seqs = random_block(rng, int(rng.integers(2, 7)), int(rng.integers(5, 40)), div=0.15)
        p = oparams(pname, mode)
        g, _, _ = oracle.block_run(seqs[:-1], None, p)
        codes, off, pred, sink, _ = g.rows()
        an, ap, sc = oracle.align_csr(codes, off, pred, sink, seqs[-1], p)
        assert sc == _dag_dp_score(codes, off, pred, sink, seqs[-1], PARAM_SETS[pname], mode == 0)


def _pairwise_gotoh(a, b, p, sw):
    """Textbook pairwise two-piece affine alignment score, vectorised per row (numpy)."""
    m, n, g, e, q, c = _norm(p)
    L = len(b)
    Hp = np.zeros(L + 1, np.int64)
    if not sw:
        j = np.arange(1, L + 1)
        Hp[1:] = np.maximum(g + (j - 1) * e, q + (j - 1) * c)
    Fp = np.full(L + 1, NEG, np.int64)
    Op = np.full(L + 1, NEG, np.int64)
    best = 0
    for i in range(1, len(a) + 1):
        F = np.maximum(Hp + g, Fp + e)
        Oo = np.maximum(Hp + q, Op + c)
        H = np.maximum(F, Oo)
        sub = np.where(np.asarray(b) == a[i - 1], m, n)
        D = Hp[:-1] + sub
        E = Q = NEG
        for j in range(1, L + 1):
            E = max(H[j - 1] + g, E + e)
            Q = max(H[j - 1] + q, Q + c)
            H[j] = max(H[j], D[j - 1], E, Q)
            if sw:
                H[j] = max(H[j], 0)
        if sw:
            H[0] = max(H[0], 0)
            best = max(best, int(H.max()))
        Hp, Fp, Op = H, F, Oo
    return best if sw else int(Hp[L])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", ["convex_default", "affine_4param", "linear"])
def test_single_sequence_graph_equals_pairwise(oracle, mode, pname):
    rng = np.random.default_rng(11)
    for trial in range(8):
        a = rng.integers(0, 4, int(rng.integers(1, 60)), dtype=np.uint8)
        b = a.copy()
        for _ in range(int(rng.integers(0, 6))):
            k = int(rng.integers(0, len(b)))
            b = np.delete(b, k) if rng.random() < 0.5 and len(b) > 1 else np.insert(b, k, rng.integers(0, 4))
        p = oparams(pname, mode)
        g = oracle.Graph()
        g.add_alignment([], [], a)
        an, ap, sc, cells = g.align(b, p)
        assert cells == len(a) * len(b)
        assert sc == _pairwise_gotoh(a, b, PARAM_SETS[pname], mode == 0)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_alignment_rescores_to_dp_score_and_graph_invariants(oracle, mode, pname):
    rng = np.random.default_rng(21)
    p = oparams(pname, mode)
    for trial in range(10):
        seqs = random_block(rng, int(rng.integers(2, 12)), int(rng.integers(3, 200)), div=0.08, alphabet=5)
        g = oracle.Graph()
        for k, s in enumerate(seqs):
            an, ap, sc, cells = g.align(s, p)
            if len(an):
                assert g.rescore(s, p, an, ap) == sc, f"trial {trial} seq {k}"
            else:
                assert sc == 0
            g.add_alignment(an, ap, s, weight=1 + k % 3)
        code, rank, grp = g.nodes()
        t, h, w = g.edges()
        # the reference's own self-check (src/main.cpp:770-803): paths spell their sequences
        for k, s in enumerate(seqs):
            assert (code[g.seq_path(k)] == s).all()
        assert (rank[t] < rank[h]).all(), "topological order violated"
        assert sorted(rank) == list(range(len(rank)))
        # aligned groups are contiguous in rank and hold distinct letters
        order = np.argsort(rank)
        gs = grp[order]
        changes = 1 + int((gs[1:] != gs[:-1]).sum()) if len(gs) else 0
        assert changes == len(set(grp.tolist()))
        for leader in set(grp.tolist()):
            letters = code[grp == leader]
            assert len(set(letters.tolist())) == len(letters)
        # every base contributes 2*w to each path edge (S6)
        tot = sum(2 * (1 + k % 3) * (len(s) - 1) for k, s in enumerate(seqs))
        assert int(w.sum()) == tot
        # MSA rows spell the sequences once gaps are removed; consensus is a path
        msa = g.msa(True)
        for k, s in enumerate(seqs):
            assert msa[k].replace("-", "") == synth.decode(s)
        cons = g.consensus()
        edges = set(zip(t.tolist(), h.tolist()))
        assert all((a, b) in edges for a, b in zip(cons[:-1], cons[1:]))


def test_gap_model_selection(oracle):
    """S1: linear == affine with e=g; 4-parameter form (q=g,c=e) == plain affine."""
    rng = np.random.default_rng(2)
    seqs = random_block(rng, 5, 60, div=0.1)
    for mode in (0, 1):
        a, sa, _ = oracle.block_run(seqs, None, oracle.mkparams(1, -4, -6, -2, -6, -2, mode))
        b, sb, _ = oracle.block_run(seqs, None, oracle.mkparams(1, -4, -6, -2, -8, -2, mode))  # g<=q? no: q<g and e>=c -> affine
        assert (sa == sb).all() and a.n_nodes == b.n_nodes
        c, sc_, _ = oracle.block_run(seqs, None, oracle.mkparams(2, -3, -5, -5, -9, -1, mode))  # g>=e -> linear
        d, sd, _ = oracle.block_run(seqs, None, oracle.mkparams(2, -3, -5, -5, -5, -5, mode))
        assert (sc_ == sd).all() and c.n_nodes == d.n_nodes


GOLD = os.path.join(os.path.dirname(__file__), "golden", "self_oracle_blocks.json")


def test_golden_self_oracle_fixtures(oracle):
    """Regression fixtures generated by tests/golden/make_golden.py FROM THIS ORACLE
    (self-oracle, not reference-derived: the reference has no vectors for this path)."""
    with open(GOLD) as f:
        gold = json.load(f)
    assert gold["provenance"].startswith("self-oracle")
    for case in gold["cases"]:
        seqs = [synth.encode(s) for s in case["seqs"]]
        p = oracle.mkparams(*case["params"], mode=case["mode"])
        g, sc, cells = oracle.block_run(seqs, case["weights"], p)
        assert sc.tolist() == case["scores"]
        assert g.n_nodes == case["n_nodes"] and g.n_edges == case["n_edges"]
        assert synth.decode(g.nodes()[0][g.consensus()]) == case["consensus"]
        assert g.msa(True) == case["msa"]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_derived_traceback_equals_recorded_traceback(oracle, mode, pname):
    """poa_vtb.c derives the alignment from the stored values (H, oF, oO) by re-applying the tie
    rules; poa_oracle.c replays recorded choices.  Two independent statements of S3/S5 must give the
    same pairs on graphs with many ties (2-letter alphabet), deep bubbles and N letters."""
    rng = np.random.default_rng(11 + mode)
    p = oparams(pname, mode)
    for trial in range(60):
        S = int(rng.integers(2, 10))
        L = int(rng.integers(3, 120))
        alpha = (2, 4, 5)[trial % 3]
        seqs = random_block(rng, S, L, div=(0.05, 0.15, 0.3)[trial % 3 if trial % 7 else 2], alphabet=alpha)
        g, _, _ = oracle.block_run(seqs[:-1], None, p)
        codes, off, pred, sink, _ = g.rows()
        q = seqs[-1] if trial % 5 else rng.integers(0, alpha, int(rng.integers(1, 2 * L)), dtype=np.uint8)
        a = oracle.align_csr(codes, off, pred, sink, q, p)
        b = oracle.align_csr(codes, off, pred, sink, q, p, vtb=True)
        assert a[2] == b[2], f"trial {trial}: score {a[2]} != {b[2]}"
        assert len(a[0]) == len(b[0]) and (a[0] == b[0]).all() and (a[1] == b[1]).all(), f"trial {trial}: pairs differ"


def test_derived_traceback_long_gaps(oracle):
    """Structural variants: gaps long enough for the second convex piece (Q / O states) in both
    directions, where E-versus-Q is decided by the bounded scan of poa_vtb.c."""
    rng = np.random.default_rng(5)
    for pname in ("convex_default", "convex_heavy", "adaptive_tier"):
        for mode in (0, 1):
            p = oparams(pname, mode)
            anc = rng.integers(0, 4, 400, dtype=np.uint8)
            ins = rng.integers(0, 4, 90, dtype=np.uint8)
            a = anc
            b = np.concatenate([anc[:150], ins, anc[150:]])      # insertion
            c = np.concatenate([anc[:220], anc[300:]])           # deletion
            d = np.concatenate([anc[:100], ins[:35], anc[100:260], anc[290:]])
            for order in ([a, b, c, d], [b, a, d, c], [c, d, b, a]):
                g, _, _ = oracle.block_run(order[:-1], None, p)
                codes, off, pred, sink, _ = g.rows()
                x = oracle.align_csr(codes, off, pred, sink, order[-1], p)
                y = oracle.align_csr(codes, off, pred, sink, order[-1], p, vtb=True)
                assert x[2] == y[2] and len(x[0]) == len(y[0])
                assert (x[0] == y[0]).all() and (x[1] == y[1]).all()


def test_fullshape_fixture_is_reproducible_on_its_small_cases(oracle):
    """tests/golden/fullshape_oracle.json: the cheap cases (config 2, config 4's smallest block) are
    recomputed here, which pins generator + oracle + digest code to the committed file; the 64 x 5 kbp
    and 128 x 10 kbp cases cost minutes and are only checked on the GPU side."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_fullshape as MF
    with open(os.path.join(here, "golden", "fullshape_oracle.json")) as f:
        cases = json.load(f)["cases"]
    names = {c["name"] for c in cases}
    assert {"ns_sw", "ns_nw", "c3", "c2", "c4_max", "c4_min", "c3a", "c3b"} <= names
    assert {c["name"] for c in cases if c.get("order") == "spoa"} >= {"ns_sw", "ns_nw", "ns_nw_affine", "c3", "c2", "c4_min"}
    for c in cases:
        assert len(c["scores"]) == c["n_seqs"] and len(c["seq_lens"]) == c["n_seqs"]
        if c["name"] not in ("c2", "c4_min"):
            continue
        # (round 6: a case also names its node order -- "spoa" = decree S7', "s7" = the incrementally kept one -- and its band mode)
        again = MF.run_case((c["name"], c["block_id"], c["n_seqs"], c["length"], tuple(c["params"]), c["mode"], c.get("order", "s7"), c.get("banded", 0)))
        for k in ("scores", "cells", "n_nodes", "n_edges", "digests", "seq_lens"):
            assert again[k] == c[k], (c["name"], k)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_avx2_row_sweep_equals_scalar_oracle(oracle, mode, pname):
    """oracle/poa_simd.c (the vectorised CPU baseline: AVX2 int16 row sweep, prefix-maximum in-row gaps,
    value-derived traceback) must reproduce the scalar recorded-choice oracle exactly: scores, graphs,
    paths, consensus -- including blocks whose lengths are not multiples of the vector width."""
    if not oracle.simd_available():
        pytest.skip("host without AVX2")
    rng = np.random.default_rng(21 + mode)
    p = oparams(pname, mode)
    for trial in range(25):
        S = int(rng.integers(2, 10))
        L = int(rng.integers(1, 260))
        seqs = random_block(rng, S, L, div=(0.04, 0.12, 0.3)[trial % 3], alphabet=(2, 4, 5)[trial % 3])
        g1, s1, c1 = oracle.block_run(seqs, None, p)
        g2, s2, c2 = oracle.block_run(seqs, None, p, impl=oracle.IMPL_AVX2)
        assert (s1 == s2).all() and (c1 == c2).all(), f"trial {trial}"
        assert all((a == b).all() for a, b in zip(g1.nodes(), g2.nodes()))
        assert all((a == b).all() for a, b in zip(g1.edges(), g2.edges()))
        for s in range(g1.n_seqs):
            assert (g1.seq_path(s) == g2.seq_path(s)).all()
        assert (g1.consensus() == g2.consensus()).all()


def test_avx2_row_sweep_on_the_committed_fullshape_fixture(oracle):
    """The vectorised variant against the committed full-shape oracle output (config 2 blocks: 16 x 1 kbp)."""
    if not oracle.simd_available():
        pytest.skip("host without AVX2")
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "fullshape_oracle.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["name"] in ("c2", "c4_min")]
    for c in cases:
        seqs = synth.make_block(c["block_id"], c["n_seqs"], c["length"])
        g, sc, cells = oracle.block_run(seqs, None, oracle.mkparams(*c["params"], mode=c["mode"]), impl=oracle.IMPL_AVX2)
        assert sc.tolist() == c["scores"] and g.n_nodes == c["n_nodes"] and g.n_edges == c["n_edges"]


def test_threaded_block_driver_with_workspaces(oracle):
    """poa_blocks_run_omp2: per-thread workspaces, both implementations, same aggregate results."""
    bases, seq_off, blk_off = synth.make_batch(6, 5, 300)
    p = oracle.mkparams()
    a = oracle.blocks_run_omp(bases, seq_off, blk_off, None, p, 3)
    b = oracle.blocks_run_omp(bases, seq_off, blk_off, None, p, 3, impl=oracle.IMPL_AVX2)
    assert (a[0] == b[0]).all() and a[1] == b[1] and (a[2] == b[2]).all() and (a[3] == b[3]).all()
    for blk in range(6):
        seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[blk], blk_off[blk + 1])]
        g, sc, _ = oracle.block_run(seqs, None, p)
        assert (a[0][blk_off[blk]:blk_off[blk + 1]] == sc).all() and a[2][blk] == g.n_nodes


def test_consensus_against_an_independent_heaviest_bundle(oracle):
    """The oracle's consensus (and the device's, which shares its text) against tests/helpers.heaviest_bundle_independent,
    written from Lee 2003 in a different form: deep blocks, weights, two-letter alphabets (many weight ties)."""
    from helpers import heaviest_bundle_independent
    rng = np.random.default_rng(77)
    for trial in range(60):
        S = int(rng.integers(2, 20))
        seqs = random_block(rng, S, int(rng.integers(5, 200)), div=(0.05, 0.15, 0.3)[trial % 3], alphabet=(2, 4)[trial % 2])
        w = rng.integers(1, 4, len(seqs))
        g, _, _ = oracle.block_run(seqs, w, oparams("convex_default", trial % 2))
        code, rank, _ = g.nodes()
        t, h, ww = g.edges()
        assert (heaviest_bundle_independent(code, rank, t, h, ww) == g.consensus()).all(), trial


def test_banded_decrees_static_and_adaptive(oracle):
    """Decrees B1-B4 of oracle/poa_oracle.c on the CPU.  (a) On a calm block both bands return the full matrix's scores
    and graph at a fraction of its cells.  (b) A 700-base insertion shared by two of four sequences (beyond
    w = 311 + 0.03 L) carries the alignment off the backbone: the STATIC band (B2) loses it, the ADAPTIVE band (B4, abPOA's
    rule: the band follows the best cells of the predecessor rows) keeps the full matrix's scores.  (c) remain() -- the
    walk along heaviest out-edges -- of a chain graph counts down to 0."""
    from helpers import random_block
    rng = np.random.default_rng(81)
    calm = random_block(rng, 5, 1500, div=0.02)
    ref = oracle.block_run(calm, None, oracle.mkparams(mode=0, banded=0))
    for banded in (1, 2):
        g, sc, cells = oracle.block_run(calm, None, oracle.mkparams(mode=0, banded=banded))
        assert (sc == ref[1]).all() and g.n_nodes == ref[0].n_nodes and (g.nodes()[1] == ref[0].nodes()[1]).all()
        assert int(cells.sum()) < 0.6 * int(ref[2].sum())
    anc = rng.integers(0, 4, 2500, dtype=np.uint8)
    with_ins = np.concatenate([anc[:1000], rng.integers(0, 4, 700, dtype=np.uint8), anc[1000:]])
    wild = [with_ins, anc.copy(), with_ins.copy(), anc.copy()]
    full = oracle.block_run(wild, None, oracle.mkparams(mode=0, banded=0))[1]
    static = oracle.block_run(wild, None, oracle.mkparams(mode=0, banded=1))[1]
    adaptive = oracle.block_run(wild, None, oracle.mkparams(mode=0, banded=2))
    assert (adaptive[1] == full).all() and not (static == full).all()
    assert int(adaptive[2].sum()) < 0.5 * sum(len(s) for s in wild[1:]) * adaptive[0].n_nodes
    chain = oracle.block_run([anc[:50]], None, oracle.mkparams())[0]
    assert (chain.row_remain() == np.arange(49, -1, -1)).all()
    # (d) global mode: the static band is not applied (its band need not hold the end column); the adaptive band is (round 4,
    # smooth_abpoa sets its band for both modes): full-matrix scores at a fraction of the cells on the calm block AND on the
    # block with the 700-base insertion
    glc = [oracle.block_run(calm[:3], None, oracle.mkparams(mode=1, banded=b)) for b in (0, 1, 2)]
    assert (glc[0][1] == glc[1][1]).all() and int(glc[1][2].sum()) == int(glc[0][2].sum())
    assert (glc[0][1] == glc[2][1]).all() and int(glc[2][2].sum()) < 0.6 * int(glc[0][2].sum())
    glw = [oracle.block_run(wild, None, oracle.mkparams(mode=1, banded=b)) for b in (0, 2)]
    assert (glw[0][1] == glw[1][1]).all() and int(glw[1][2].sum()) < 0.6 * int(glw[0][2].sum())
