"""CPU: host-side rows (A2-A4 collection, A9/A10 block graphs, lacing + GFA I/O) of
smoothxg_amd/csrc/sxg_smooth.cpp against the pure-Python oracle (oracle/smooth_oracle.py), with the
POA supplied by the C oracle through a callback (no GPU involved)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from oracle import smooth_oracle as SO
from smoothxg_amd import poa as P
from smoothxg_amd import smooth as S

HERE = os.path.dirname(os.path.abspath(__file__))
DRB1 = os.path.join(HERE, "golden", "DRB1-3123.seqwish.gfa")  # the reference's own test input (test/data/)


class OracleProvider:
    """sxg_poa_batch_run-shaped callback backed by oracle/poa_oracle.c (test infrastructure)."""

    def __init__(self):
        self.keep = []
        self.run = S.RUN_FN(self._run)
        self.free = S.FREE_FN(self._free)

    def provider(self):
        return C.cast(self.run, C.c_void_p), C.cast(self.free, C.c_void_p), None

    def _run(self, ctx, pin, pout):
        i, o = pin.contents, pout.contents
        nb = i.n_blocks
        blk = np.ctypeslib.as_array(i.blk_off, (nb + 1,)).copy()
        ns = int(blk[-1])
        so = np.ctypeslib.as_array(i.seq_off, (ns + 1,)).copy() if ns else np.zeros(1, np.int64)
        bases = np.ctypeslib.as_array(i.bases, (max(int(so[-1]), 1),)).copy()
        w = np.ctypeslib.as_array(i.weights, (max(ns, 1),)).copy() if i.weights else np.ones(max(ns, 1), np.uint32)
        node_off, cons_off, codes, paths, cons = [0], [0], [], [], []
        msa_off, msa_cols, msa_txt = [0], [], b""
        for b in range(nb):
            pr = i.params[b if i.per_block_params else 0]
            par = O.mkparams(pr.m, pr.n, pr.g, pr.e, pr.q, pr.c, pr.mode, pr.banded)
            seqs = [bases[so[s]:so[s + 1]] for s in range(blk[b], blk[b + 1])]
            if seqs:
                g, _, _ = O.block_run(seqs, w[blk[b]:blk[b + 1]], par)
                codes.append(g.nodes()[0])
                paths += [g.seq_path(k) for k in range(len(seqs))]
                cons.append(g.consensus())
                if i.want_msa:
                    rows = g.msa(bool(i.want_consensus))
                    msa_cols.append(len(rows[0]))
                    msa_txt += "".join(rows).encode()
            if not seqs or not i.want_msa:
                msa_cols.append(0)
            msa_off.append(len(msa_txt))
            node_off.append(node_off[-1] + (len(codes[-1]) if seqs else 0))
            cons_off.append(cons_off[-1] + (len(cons[-1]) if seqs else 0))
        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(1, dt), dt)
        arrs = dict(node_off=np.asarray(node_off, np.int64), cons_off=np.asarray(cons_off, np.int64),
                    node_code=cat(codes, np.uint8), paths=cat(paths, np.int32), cons=cat(cons, np.int32),
                    status=np.zeros(max(nb, 1), np.int32), msa_off=np.asarray(msa_off, np.int64),
                    msa_cols=np.asarray(msa_cols[:max(nb, 1)] or [0], np.int32), msa=np.frombuffer(msa_txt + b"\0", np.uint8).copy())
        self.keep.append(arrs)
        o.n_blocks, o.n_seqs = nb, ns
        o.status = arrs["status"].ctypes.data_as(C.POINTER(C.c_int32))
        o.node_off = arrs["node_off"].ctypes.data_as(C.POINTER(C.c_int64))
        o.node_code = arrs["node_code"].ctypes.data_as(C.POINTER(C.c_uint8))
        o.seq_path_nodes = arrs["paths"].ctypes.data_as(C.POINTER(C.c_int32))
        o.cons_off = arrs["cons_off"].ctypes.data_as(C.POINTER(C.c_int64))
        o.cons_nodes = arrs["cons"].ctypes.data_as(C.POINTER(C.c_int32))
        if i.want_msa:
            o.msa_off = arrs["msa_off"].ctypes.data_as(C.POINTER(C.c_int64))
            o.msa_cols = arrs["msa_cols"].ctypes.data_as(C.POINTER(C.c_int32))
            o.msa = arrs["msa"].ctypes.data
        return 0

    def _free(self, pout):
        pass


def synthetic_gfa(seed, n_paths=5, n_nodes=60, with_reverse=True):
    """A small variation graph: a backbone of nodes, paths that skip / substitute / reverse."""
    rng = np.random.default_rng(seed)
    lines, seqs = ["H\tVN:Z:1.0"], []
    for i in range(n_nodes):
        s = "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(1, 30))))
        if i % 17 == 5:
            s = s[:1] + "N" + s[1:]
        seqs.append(s)
        lines.append("S\t%d\t%s" % (i + 1, s))
    for p in range(n_paths):
        steps = []
        for i in range(n_nodes):
            r = rng.random()
            if r < 0.1:
                continue
            steps.append("%d%s" % (i + 1, "-" if (with_reverse and r > 0.93) else "+"))
        if with_reverse and p == n_paths - 1:   # one path walks the graph mostly in reverse
            steps = [s[:-1] + ("-" if s[-1] == "+" else "+") for s in reversed(steps)]
        lines.append("P\tpath%d\t%s\t*" % (p, ",".join(steps)))
    return "\n".join(lines) + "\n"


def haplotype_gfa(seed, n_paths=5, length=900, sub=0.01, node_bp=60):
    """Near-identical haplotypes, every path on its own chain of nodes: blocks of high identity
    (the adaptive score tiers of A14 need >= 0.90)."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, length)
    lines, nid = ["H\tVN:Z:1.0"], 1
    plines = []
    for p in range(n_paths):
        hap = anc.copy()
        mut = rng.random(length) < sub
        hap[mut] = (hap[mut] + rng.integers(1, 4, int(mut.sum()))) % 4
        text = "".join("ACGT"[c] for c in hap)
        steps = []
        for a in range(0, length, node_bp):
            lines.append("S\t%d\t%s" % (nid, text[a:a + node_bp]))
            steps.append("%d+" % nid)
            nid += 1
        plines.append("P\thap%d\t%s\t*" % (p, ",".join(steps)))
    return "\n".join(lines + plines) + "\n"


def inversion_gfa(seed, n_paths=5, length=900, node_bp=60, lo=300, hi=600):
    """Collinear haplotypes whose middle window [lo, hi) is stored reverse-complemented and walked with '-' steps:
    that block is collected in reverse (src/smooth.cpp:707-709) and can only join its neighbours' MAF group flipped."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, length)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    lines, plines, nid = ["H\tVN:Z:1.0"], [], 1
    for p in range(n_paths):
        hap = anc.copy()
        mut = rng.random(length) < 0.02
        hap[mut] = (hap[mut] + rng.integers(1, 4, int(mut.sum()))) % 4
        sq = "".join("ACGT"[x] for x in hap)
        steps = []
        for a in range(0, length, node_bp):
            piece = sq[a:a + node_bp]
            if lo <= a < hi:
                lines.append("S\t%d\t%s" % (nid, "".join(comp[c] for c in reversed(piece))))
                steps.append("%d-" % nid)
            else:
                lines.append("S\t%d\t%s" % (nid, piece))
                steps.append("%d+" % nid)
            nid += 1
        plines.append("P\thap%d\t%s\t*" % (p, ",".join(steps)))
    return "\n".join(lines + plines) + "\n"


@pytest.fixture(scope="module")
def prov():
    return OracleProvider()


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("target", [40, 150, 100000])
def test_collect_matches_reference_restatement(seed, target):
    text = synthetic_gfa(seed)
    sm = S.Smoother(text, target)
    g = SO.Graph(text)
    blocks = SO.blockset_by_path_windows(g, target)
    assert sm.n_blocks == len(blocks)
    for frac in (0.001, 0.0, 0.5):
        p = S.default_params(poa_padding_fraction=frac)
        for k in range(len(blocks)):
            assert sm.collect_text(k, p) == SO.collect_text(SO.collect(g, blocks[k], frac, 1000)), (seed, target, frac, k)


def test_padding_quirks_of_append_to_sequence():
    """src/smooth.cpp:75-126: a range starting at step 0 gets all-N on the left; the left walk takes
    the LAST characters of the range's own first node; N-fill at path ends."""
    text = "S\t1\tAAAA\nS\t2\tCCCCCC\nS\t3\tGG\nP\tp\t1+,2+,3+\t*\n"
    g = SO.Graph(text)
    left, f, r = SO.append_to_sequence(g, 0, 0, 5, True)
    assert left == "NNNNN"
    left, f, r = SO.append_to_sequence(g, 0, 1, 5, True)   # walks step 1 (CCCCCC) itself, then stops before step 0
    assert left == "CCCCC" and f == 5
    right, f, r = SO.append_to_sequence(g, 0, 2, 5, False)
    assert right == "GGNNN"
    sm = S.Smoother(text, 1000)
    assert "seq\t0\t1\t" in sm.collect_text(0, S.default_params())


@pytest.mark.parametrize("seed", [4, 5])
@pytest.mark.parametrize("cons", [0, 1])
def test_block_graph_and_full_iteration_match_oracle(prov, seed, cons):
    text = synthetic_gfa(seed, n_paths=6, n_nodes=80)
    g = SO.Graph(text)
    for target in (120, 500):
        sm = S.Smoother(text, target)
        blocks = SO.blockset_by_path_windows(g, target)
        p = S.default_params(add_consensus=cons, poa_padding_fraction=0.001)
        for k in range(min(len(blocks), 4)):
            c = SO.collect(g, blocks[k])
            if not c.seqs:
                continue
            code, paths, cn = SO.poa(c)
            want = SO.to_gfa(SO.build_block_graph(c, code, paths, cn, ("Consensus_%d" % k) if cons else ""))
            assert sm.block_graph_gfa(k, p, prov.provider()) == want
        want = SO.smooth(g, blocks, add_consensus=bool(cons))
        got = sm.smooth_gfa(p, prov.provider())
        assert got == want
        # the smoothed graph still spells every input path (src/main.cpp:770-810)
        out = SO.Graph(got)
        for q, nm in enumerate(g.pname):
            assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


@pytest.mark.parametrize("cons", [0, 1])
@pytest.mark.parametrize("scores", [None, (1, 4, 6, 2, 0, 0), (2, 3, 0, 3, 0, 0)])
def test_abpoa_path_iteration_matches_restatement(prov, cons, scores):
    """-A (smooth_abpoa, src/smooth.cpp:133-627): the scores in abPOA's convention (convex default, the 4-parameter form
    with q = c = 0 -> affine, g = 0 -> linear), the adaptive band (params.banded = 2 reaches the provider), the
    consensus path restricted to visited nodes (build_odgi_abPOA) -- C++ host rows == Python restatement, byte for byte."""
    text = synthetic_gfa(11, n_paths=6, n_nodes=90)
    g = SO.Graph(text)
    sm = S.Smoother(text, 300)
    blocks = SO.blockset_by_path_windows(g, 300)
    kw = {} if scores is None else dict(zip(("poa_m", "poa_n", "poa_g", "poa_e", "poa_q", "poa_c"), scores))
    okw = {} if scores is None else dict(zip(("m", "n", "g", "e", "q", "cc"), scores))
    p = S.default_params(add_consensus=cons, use_abpoa=1, **kw)
    seen = []
    inner = prov.provider()
    got = sm.smooth_gfa(p, inner)
    want = SO.smooth(g, blocks, add_consensus=bool(cons), abpoa=True, **okw)
    assert got == want
    assert got != SO.smooth(g, blocks, add_consensus=bool(cons), abpoa=False, **okw) or scores is not None
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
    # the engine parameters the provider is handed
    e = SO.engine_params(*(scores or (1, 4, 6, 2, 26, 1)), True, True)
    assert e.banded == 2 and e.mode == 0
    if scores is None:
        assert (e.m, e.n, e.g, e.e, e.q, e.c) == (1, -4, -8, -2, -27, -1)
    elif scores[2] == 0:
        assert (e.g, e.e, e.q, e.c) == (-3, -3, -3, -3)
    else:
        assert (e.g, e.e, e.q, e.c) == (-8, -2, -8, -2)


def test_abpoa_band_in_global_mode_and_the_local_mode_switch(prov):
    """smooth_abpoa sets its band for both alignment modes (src/smooth.cpp:259-271): -A -Z hands the provider banded = 2 with
    mode = 1, and the oracle honours the adaptive band in global mode.  abpoa_band_local = 0 (upstream abPOA may run local
    mode unbanded): abPOA's scores, banded = 0.  C++ == Python restatement in both."""
    text = haplotype_gfa(21, n_paths=5, length=1000)
    g = SO.Graph(text)
    sm = S.Smoother(text, 350)
    blocks = SO.blockset_by_path_windows(g, 350)
    got = sm.smooth_gfa(S.default_params(use_abpoa=1, local_alignment=0, add_consensus=1), prov.provider())
    assert got == SO.smooth(g, blocks, add_consensus=True, abpoa=True, local=False)
    e = SO.engine_params(1, 4, 6, 2, 26, 1, False, True)
    assert e.banded == 2 and e.mode == 1
    got = sm.smooth_gfa(S.default_params(use_abpoa=1, abpoa_band_local=0), prov.provider())
    assert got == SO.smooth(g, blocks, abpoa=True, band_local=False)
    e = SO.engine_params(1, 4, 6, 2, 26, 1, True, True, band_local=False)
    assert e.banded == 0 and (e.g, e.q) == (-8, -27)
    e = SO.engine_params(1, 4, 6, 2, 26, 1, False, True, band_local=False)
    assert e.banded == 2                              # global alignment is banded either way
    sm.close()


def test_spoa_order_switch_reaches_the_engine(prov):
    """poa_spoa_order = 1: the provider is handed mode | SXG_ORDER_SPOA (decree S7'); the iteration equals the restatement run
    with the same switch, still preserves every path, and differs from the default order's output on a divergent graph."""
    text = synthetic_gfa(5, n_paths=6, n_nodes=80)
    g = SO.Graph(text)
    sm = S.Smoother(text, 250)
    blocks = SO.blockset_by_path_windows(g, 250)
    got = sm.smooth_gfa(S.default_params(poa_spoa_order=1, add_consensus=1), prov.provider())
    assert got == SO.smooth(g, blocks, add_consensus=True, spoa_order=True)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
    assert SO.engine_params(1, 4, 6, 2, 26, 1, True, False, spoa_order=True).mode == 0x10
    sm.close()


def test_drb1_fixture_round_trip(prov):
    """The reference's own test input (CMakeLists.txt:562-567 runs the CLI on it and checks the exit
    code): one smoothing iteration must preserve all 12 paths and agree with the oracle."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    assert len(g.pname) == 12 and len(g.seq) == 3585
    sm = S.Smoother(text, 700)
    blocks = SO.blockset_by_path_windows(g, 700)
    got = sm.smooth_gfa(S.default_params(), prov.provider())
    assert got == SO.smooth(g, blocks)
    out = SO.Graph(got)
    assert sorted(out.pname) == sorted(g.pname)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


def test_three_chained_iterations_on_a_synthetic_graph(prov):
    """The reference runs its target lengths as a CHAIN (-l 700,900,1100, CMakeLists.txt:565): every iteration reads the
    graph the one before wrote (src/main.cpp:374-1065), consensus paths only in the last (:404).  Here on a small
    synthetic graph (the DRB1 chain is pinned by tests/golden/drb1_chain.json and runs in the GPU tier): real block
    discovery per iteration, C++ host rows == Python restatement after EVERY iteration, paths and path count kept."""
    text = synthetic_gfa(21, n_paths=6, n_nodes=120)
    g0 = SO.Graph(text)
    want_text = text
    for it, tl in enumerate((150, 220, 300)):
        last = it == 2
        gw = SO.Graph(want_text)
        blocks = SO.break_blocks(gw, SO.smoothable_blocks(gw, tl * 6, tl, 5000, 5000), 2 * tl)
        want_text = SO.smooth(gw, blocks, add_consensus=last)
        sm = S.Smoother(text, discover=dict(target_poa_length=tl, n_haps=6, max_path_jump=5000, max_edge_jump=5000))
        assert [sm.block_ranges(k) for k in range(sm.n_blocks)] == [[tuple(r) for r in blk] for blk in blocks]
        text = sm.smooth_gfa(S.default_params(add_consensus=1 if last else 0), prov.provider())
        sm.close()
        assert text == want_text, "iteration %d (-l %d)" % (it, tl)
        out = SO.Graph(text)
        names = [nm for nm in out.pname if not nm.startswith("Consensus_")]
        assert sorted(names) == sorted(g0.pname)
        for q, nm in enumerate(g0.pname):
            assert out.path_sequence(out.pname.index(nm)) == g0.path_sequence(q)
    assert sum(nm.startswith("Consensus_") for nm in SO.Graph(text).pname) >= 1


@pytest.mark.parametrize("chunk", ["1", "3", "7"])
def test_chunked_pipeline_gives_the_same_bytes(prov, monkeypatch, chunk):
    """Phases 1-3 run as a pipeline over chunks of blocks (collect / POA provider in its own thread / block graphs): the
    GFA, the MAF text and the flip set are the same for every chunk size -- here 1, 3 and 7 blocks per chunk against one
    chunk, with consensus paths, adaptive scores, and the in-order MAF consumer with merging."""
    text = synthetic_gfa(31, n_paths=6, n_nodes=140)
    g = SO.Graph(text)
    sm = S.Smoother(text, 120)
    blocks = SO.blockset_by_path_windows(g, 120)
    assert len(blocks) >= 12
    p = S.default_params(add_consensus=1, adaptive_poa_params=1, kmer_size=5)
    monkeypatch.setenv("SXG_SMOOTH_CHUNK_BLOCKS", "1000000")
    one = sm.smooth_gfa(p, prov.provider())
    one_maf = sm.smooth_maf_gfa(p, prov.provider(), merge_blocks=True, jaccard=0.5)
    assert one == SO.smooth(g, blocks, add_consensus=True, adaptive=True, kmer_size=5)
    monkeypatch.setenv("SXG_SMOOTH_CHUNK_BLOCKS", chunk)
    assert sm.smooth_gfa(p, prov.provider()) == one
    got = sm.smooth_maf_gfa(p, prov.provider(), merge_blocks=True, jaccard=0.5)
    assert got[0] == one_maf[0] and got[1] == one_maf[1] and got[2] == one_maf[2]


def test_drb1_chain_fixture_is_well_formed():
    import json
    j = json.load(open(os.path.join(os.path.dirname(DRB1), "drb1_chain.json")))
    assert [x["target_poa_length"] for x in j["iterations"]] == [700, 900, 1100]
    assert all(len(x["sha256"]) == 64 and x["gfa_bytes"] > 0 for x in j["iterations"])
    assert j["iterations"][0]["paths"] == j["iterations"][1]["paths"] == 12 and j["iterations"][2]["paths"] > 12


def test_adaptive_score_tiers_are_the_reference_table():
    """src/smooth.cpp:2032-2069, every tier and both sides of every cut."""
    table = {0.99: (1, 19, 39, 3, 81, 1), 0.98: (1, 13, 31, 3, 51, 1), 0.97: (1, 9, 16, 2, 41, 1),
             0.95: (1, 7, 11, 2, 33, 1), 0.90: (1, 4, 6, 2, 26, 1)}
    custom = (2, 5, 7, 3, 30, 2)
    for cut, tier in table.items():
        for thr in (np.float32(cut + 0.004), np.float32(cut)):
            want = tier if float(thr) >= cut else None
            if want is not None:
                assert S.adaptive_poa_scores(float(thr), custom) == want == SO.adaptive_scores(thr, custom)
    for thr in (0.7, 0.85, 0.8999):
        assert S.adaptive_poa_scores(thr, custom) == custom == SO.adaptive_scores(np.float32(thr), custom)
    for thr in np.linspace(0.69, 1.0, 63, dtype=np.float32):
        assert S.adaptive_poa_scores(float(thr), custom) == SO.adaptive_scores(thr, custom)


def test_identity_threshold_matches_oracle():
    """A14 estimator (exact canonical-k-mer Jaccard -> mash distance, 30 % percentile, floor 0.7)."""
    for seed, k in ((4, 5), (5, 7), (6, 11)):
        text = synthetic_gfa(seed, n_paths=6, n_nodes=80)
        g = SO.Graph(text)
        sm = S.Smoother(text, 150)
        for b, ranges in enumerate(SO.blockset_by_path_windows(g, 150)):
            thr, used = SO.identity_threshold(g, ranges, k)
            got_thr, got_used = sm.identity_threshold(b, k)
            assert got_used == used
            if used > 1:
                assert np.float32(got_thr) == thr, (seed, b, got_thr, thr)
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 700)
    blocks = SO.blockset_by_path_windows(g, 700)
    tiers = set()
    for b in (0, 3, 7, 11):
        thr, used = SO.identity_threshold(g, blocks[b], 17)
        got_thr, got_used = sm.identity_threshold(b, 17)
        assert (got_used, np.float32(got_thr)) == (used, thr)
        tiers.add(SO.adaptive_scores(thr))
    assert len(tiers) >= 1


@pytest.mark.parametrize("seed", [4, 6])
def test_adaptive_iteration_matches_oracle(prov, seed):
    """-a end to end: per-block scores reach the POA provider (per_block_params) and the smoothed
    graph equals the oracle stack's, paths preserved."""
    text = synthetic_gfa(seed, n_paths=6, n_nodes=80)
    g = SO.Graph(text)
    sm = S.Smoother(text, 200)
    blocks = SO.blockset_by_path_windows(g, 200)
    p = S.default_params(adaptive_poa_params=1, kmer_size=5)
    picked = {SO.block_scores(g, r, True, 5, 1000) for r in blocks}
    got = sm.smooth_gfa(p, prov.provider())
    assert got == SO.smooth(g, blocks, adaptive=True, kmer_size=5)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
    assert picked  # at least the default tier


@pytest.mark.parametrize("sub,tier", [(0.0005, (1, 19, 39, 3, 81, 1)), (0.004, (1, 13, 31, 3, 51, 1)), (0.012, (1, 9, 16, 2, 41, 1)),
                                      (0.02, (1, 7, 11, 2, 33, 1))])
def test_adaptive_iteration_high_identity_tiers(prov, sub, tier):
    """Haplotypes of graded divergence land in the upper score tiers; the iteration still equals
    the oracle stack's and preserves every path."""
    text = haplotype_gfa(int(sub * 1e5), sub=sub)
    g = SO.Graph(text)
    sm = S.Smoother(text, 450)
    blocks = SO.blockset_by_path_windows(g, 450)
    picked = [SO.block_scores(g, r, True, 15, 1000) for r in blocks]
    assert tier in picked, picked
    p = S.default_params(adaptive_poa_params=1, kmer_size=15)
    got = sm.smooth_gfa(p, prov.provider())
    assert got == SO.smooth(g, blocks, adaptive=True, kmer_size=15)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


@pytest.mark.parametrize("cons", [0, 1])
def test_maf_rows_match_oracle(prov, cons):
    """MSA -> MAF rows (src/smooth.cpp:782-905) and the MAF block text (src/maf.hpp:35-66): synthetic
    graphs with reverse steps (strand '-', start counted from the path's end) and DRB1 blocks."""
    cases = [(synthetic_gfa(4, n_paths=6, n_nodes=80), 150, None), (synthetic_gfa(5, n_paths=6, n_nodes=80), 400, None),
             (open(DRB1).read(), 700, (0, 9))]
    seen_rev = False
    for text, target, pick in cases:
        g = SO.Graph(text)
        sm = S.Smoother(text, target)
        blocks = SO.blockset_by_path_windows(g, target)
        p = S.default_params(add_consensus=cons)
        for k in (pick if pick else range(min(len(blocks), 4))):
            c = SO.collect(g, blocks[k])
            if not c.seqs:
                assert sm.block_maf_rows(k, p, prov.provider()) == ""
                continue
            msa, clen = SO.poa_msa(c, bool(cons))
            rows = SO.maf_rows(g, blocks[k], c, msa, ("Consensus_%d" % k) if cons else "", clen)
            seen_rev |= any(r[3] for r in rows)
            assert sm.block_maf_rows(k, p, prov.provider()) == SO.maf_rows_text(rows)
            maf = sm.block_maf(k, p, prov.provider())
            assert maf == SO.maf_block_text(rows)
            # every record spells the unpadded sequence of its range, in the record's strand
            for (src, start, size, rev, psize, txt), line in zip(rows, [l for l in maf.split("\n") if l.startswith("s ")]):
                letters = txt.replace("-", "")
                assert len(letters) == size
                if not src.startswith("Consensus_"):
                    full = g.path_sequence(g.pname.index(src))
                    want = SO.revcomp(full)[start:start + size] if rev else full[start:start + size]
                    assert letters == want
    assert seen_rev


def test_unchop_and_order_decrees():
    G = SO.OGraph()
    G.seq = ["A", "C", "G", "T", "A"]
    for a, b in [(0, 1), (1, 2), (1, 3), (2, 4), (3, 4)]:
        G.add_edge(a << 1, b << 1)
    G.paths = [("x", [0, 2, 4, 8]), ("y", [0, 2, 6, 8])]
    SO.unchop(G)
    assert G.seq == ["AC", "G", "T", "A"]
    SO.topo_renumber(G)
    assert SO.to_gfa(G).count("\nL\t") == 4


def test_host_library_exports_every_declared_symbol():
    import re
    src = open(os.path.join(os.path.dirname(HERE), "include", "sxg_smooth.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(sxg_[a-z0-9_]+)\s*\(", src)) - {"sxg_poa_run_fn", "sxg_poa_free_fn"})
    L = S.load_library()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(S.EXPORTS) == names


def test_blockset_from_callers_own_ranges(prov):
    """sxg_blockset_from_ranges: real block_t's (src/blocks.hpp:29-43) enter sxg_smooth_gfa.  The demo
    partition's ranges, handed back through the public constructor, give the same blocks and the same
    smoothed GFA; malformed ranges are rejected, not trusted."""
    text = synthetic_gfa(9, n_paths=6, n_nodes=80)
    a = S.Smoother(text, 150)
    blocks = [a.block_ranges(k) for k in range(a.n_blocks)]
    assert all(r[3] > 0 and r[2] > r[1] for blk in blocks for r in blk)
    b = S.Smoother(text, blocks=blocks)
    assert b.n_blocks == a.n_blocks
    assert [b.block_ranges(k) for k in range(b.n_blocks)] == blocks
    p = S.default_params(add_consensus=1)
    for k in range(a.n_blocks):
        assert a.collect_text(k, p) == b.collect_text(k, p)
    assert a.smooth_gfa(p, prov.provider()) == b.smooth_gfa(p, prov.provider())
    # length 0 = "compute it"; a wrong length, a range past its path or an unknown path are errors
    c = S.Smoother(text, blocks=[[(r[0], r[1], r[2]) for r in blk] for blk in blocks])
    assert [c.block_ranges(k) for k in range(c.n_blocks)] == blocks
    for bad in ([[(0, 0, 10 ** 6)]], [[(99, 0, 1)]], [[(0, 3, 2)]], [[(0, 0, 1, 123456)]]):
        with pytest.raises(S.SmoothError):
            S.Smoother(text, blocks=bad)
    # a blockset that does not cover a path is caught by the lacing check, as src/main.cpp:770-810 would
    with pytest.raises(S.SmoothError):
        S.Smoother(text, blocks=[blocks[0]]).smooth_gfa(p, prov.provider())


def test_integration_snippet_compiles_and_links_against_the_c_abi(tmp_path):
    """INTEGRATION.md's phase-2 binding (the code a smoothxg maintainer pastes into smooth_spoa) is kept as
    tests/csrc/integration_snippet.cpp: it must compile as C++17 against include/sxg_poa.h, link against
    libsxgpoa.so, and -- on this GPU-less box -- report the missing device through the error path."""
    import subprocess
    root = os.path.dirname(HERE)
    src = os.path.join(HERE, "csrc", "integration_snippet.cpp")
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = open(src).read()
    body = code[code.index("// ---- INTEGRATION.md snippet begin"):code.index("// ---- INTEGRATION.md snippet end")]
    for line in body.splitlines()[1:]:
        if line.strip():
            assert line in md, "INTEGRATION.md and tests/csrc/integration_snippet.cpp drifted apart: " + line
    P.load_library()
    exe = str(tmp_path / "snippet")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), src, "-o", exe,
                           "-L", os.path.join(root, "smoothxg_amd", "csrc"), "-lsxgpoa", "-L/opt/rocm/lib",
                           "-Wl,-rpath," + os.path.join(root, "smoothxg_amd", "csrc"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi 5" in r.stdout


def test_ready_made_and_multi_gpu_snippets_compile_and_link(tmp_path):
    """INTEGRATION.md's other two code blocks (block discovery -> sxg_smooth_gfa through the engine; the sharded entry
    point) are kept in tests/csrc/integration_snippet2.cpp: same text as the document, compiles with -Wall -Werror
    against both headers, links against both libraries, runs the host-only calls and stops at the missing device."""
    import subprocess
    root = os.path.dirname(HERE)
    src = os.path.join(HERE, "csrc", "integration_snippet2.cpp")
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = open(src).read()
    n = 0
    for part in code.split("// ---- INTEGRATION.md snippet begin")[1:]:
        for line in part[:part.index("// ---- INTEGRATION.md snippet end")].splitlines()[1:]:
            if line.strip():
                assert line in md, "INTEGRATION.md and tests/csrc/integration_snippet2.cpp drifted apart: " + line
                n += 1
    assert n >= 12
    P.load_library()
    S.load_library()
    exe = str(tmp_path / "snippet2")
    libdir = os.path.join(root, "smoothxg_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), src, "-o", exe,
                           "-L", libdir, "-lsxgsmooth", "-lsxgpoa", "-L/opt/rocm/lib", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "discovery:" in r.stdout


def gfa_with_links(text):
    """Adds the L lines the paths imply (a seqwish graph's edges are its path adjacencies)."""
    links = set()
    for line in text.split("\n"):
        f = line.split("\t")
        if f[0] == "P":
            st = f[2].split(",")
            for a, b in zip(st, st[1:]):
                links.add((a[:-1], a[-1], b[:-1], b[-1]))
    return text + "".join("L\t%s\t%s\t%s\t%s\t0M\n" % l for l in sorted(links))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_block_discovery_matches_restatement(seed):
    """smoothable_blocks (src/blocks.cpp:7-327) + the length cut of break_blocks (src/breaks.cpp:210-330): C++ against
    the Python restatement on graphs with shared and reversed nodes, several limit settings."""
    text = gfa_with_links(synthetic_gfa(seed, n_paths=6, n_nodes=90))
    g = SO.Graph(text)
    for tl, haps, jump, ejump in ((60, 6, 100, 0), (150, 3, 20, 0), (400, 6, 100, 50), (100000, 6, 100, 0)):
        want = SO.break_blocks(g, SO.smoothable_blocks(g, tl * haps, tl, jump, ejump), 2 * tl)
        sm = S.Smoother(text, discover=dict(target_poa_length=tl, n_haps=haps, max_path_jump=jump, max_edge_jump=ejump))
        got = [sm.block_ranges(k) for k in range(sm.n_blocks)]
        assert got == [[tuple(r) for r in blk] for blk in want], (seed, tl, haps, jump, ejump)
        # every step belongs to at most one range; ranges are inside their paths
        taken = set()
        for blk in got:
            for p, b, e, ln in blk:
                assert 0 <= b < e <= len(g.steps[p]) and ln == g.pos[p][e] - g.pos[p][b]
                for k in range(b, e):
                    assert (p, k) not in taken
                    taken.add((p, k))


def test_block_discovery_on_the_reference_fixture_and_round_trip(prov):
    """DRB1 with the flags of the reference's own test (CMakeLists.txt:565: -j 5k -e 5k -l 700 -r 12, first iteration):
    discovered blocks == restatement, cover every step, and the smoothing iteration on them preserves all paths."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    want = SO.break_blocks(g, SO.smoothable_blocks(g, 700 * 12, 700, 5000, 5000), 1400)
    sm = S.Smoother(text, discover=dict(target_poa_length=700, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
    got = [sm.block_ranges(k) for k in range(sm.n_blocks)]
    assert got == [[tuple(r) for r in blk] for blk in want]
    assert sum(e - b for blk in got for _, b, e, _ in blk) == sum(len(st) for st in g.steps)   # nothing left out here
    assert max(ln for blk in got if len(blk) > 1 for _, _, _, ln in blk) <= 1400 + max(len(sq) for sq in g.seq)
    out = SO.Graph(sm.smooth_gfa(S.default_params(), prov.provider()))
    assert sorted(out.pname) == sorted(g.pname)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


@pytest.mark.parametrize("cons", [0, 1])
@pytest.mark.parametrize("merge", [False, True])
def test_maf_merging_and_flips_match_restatement(prov, cons, merge):
    """A13 + 8f-4: the in-order MAF consumer (src/smooth.cpp:1600-1919) -- contiguous blocks merged by path Jaccard,
    blocks joined in the opposite orientation flipped and their graphs rebuilt (src/smooth.cpp:2352-2436), merged
    consensus paths (src/main.cpp:870-960) -- C++ against oracle/smooth_oracle.py: MAF text, flip set, laced GFA."""
    n_flips = n_merged = 0
    for seed, target, jac in ((4, 120, 1.0), (5, 200, 0.5), (6, 90, 0.0), (7, 150, 1.0), (-1, 300, 1.0), (-2, 300, 1.0)):
        text = synthetic_gfa(seed, n_paths=6, n_nodes=80) if seed >= 0 else (haplotype_gfa(3, n_paths=5, length=1500) if seed == -1 else inversion_gfa(8))
        g = SO.Graph(text)
        sm = S.Smoother(text, target)
        blocks = SO.blockset_by_path_windows(g, target)
        frac = 0.0 if seed == -2 else 0.001   # (with the default padding the inverted window is outvoted by its forward flanks)
        p = S.default_params(add_consensus=cons, poa_padding_fraction=frac)
        want = SO.smooth(g, blocks, add_consensus=bool(cons), fraction=frac,
                         merge=dict(merge_blocks=merge, jaccard=jac, header="##maf version=1"))
        got = sm.smooth_maf_gfa(p, prov.provider(), merge_blocks=merge, jaccard=jac, header="##maf version=1")
        assert got[1] == want[1], (seed, "MAF")
        assert got[2] == len(want[2]), (seed, "flips")
        assert got[0] == want[0], (seed, "GFA")
        n_flips += got[2]
        out = SO.Graph(got[0])
        for q, nm in enumerate(g.pname):   # flipped or not, every path still spells its sequence
            assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
        if merge:
            n_merged += got[1].count(" merged=true")
            if cons and " merged=true" in got[1]:
                assert any(nm.startswith("Consensus_") and "-" in nm for nm in out.pname)
        else:
            assert got[2] == 0 and " merged=true" not in got[1]
            assert got[0] == sm.smooth_gfa(p, prov.provider())     # without -M the GFA is the plain iteration's
    if merge:
        assert n_merged > 0, "the fixtures are meant to exercise merging"
        assert n_flips > 0, "the fixtures are meant to exercise the flip rebuild"


def test_flip_changes_the_gfa_exactly_as_the_restatement_says(prov):
    """A block whose sequences run reverse to its neighbours' joins their group flipped: the laced GFA differs from the
    unmerged one in that block's nodes (reverse-complemented) and equals the oracle's."""
    text = synthetic_gfa(21, n_paths=5, n_nodes=70, with_reverse=True)
    g = SO.Graph(text)
    for target in (80, 140, 260):
        sm = S.Smoother(text, target)
        blocks = SO.blockset_by_path_windows(g, target)
        p = S.default_params()
        plain = sm.smooth_gfa(p, prov.provider())
        got = sm.smooth_maf_gfa(p, prov.provider(), merge_blocks=True, jaccard=0.0)
        want = SO.smooth(g, blocks, merge=dict(merge_blocks=True, jaccard=0.0))
        assert got[0] == want[0] and got[1] == want[1] and got[2] == len(want[2])
        if got[2]:
            assert got[0] != plain


def test_parallel_forms_of_unchop_and_gfa_writer_equal_the_serial_ones():
    """unchop, the edge sort and the GFA writer switch to their OpenMP forms above 100 000 nodes (the laced graph of the
    headline workload); SXG_SMOOTH_PAR_MIN=0 forces them on a small graph in a fresh process: same bytes."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_smooth_host as H\nfrom smoothxg_amd import smooth as S\n"
            "text = H.haplotype_gfa(11, n_paths=6, length=1500)\n"
            "sm = S.Smoother(text, 300)\n"
            "sys.stdout.write(sm.smooth_gfa(S.default_params(add_consensus=1), H.OracleProvider().provider()))\n"
            % (HERE, os.path.dirname(HERE)))
    outs = []
    for env in ({}, {"SXG_SMOOTH_PAR_MIN": "0", "OMP_NUM_THREADS": "4"}):
        e = dict(os.environ)
        e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True, timeout=600).stdout)
    assert outs[0] == outs[1] and outs[0].startswith("H\tVN:Z:1.0")


@pytest.mark.parametrize("case", ["synthetic", "synthetic_cons", "haplotypes", "haplotypes_nopad", "inversion", "abpoa_cons", "drb1", "drb1_cons"])
def test_flat_lacing_equals_the_laced_graph_path(case, monkeypatch):
    """The iteration without the MAF consumer laces compact block graphs without building the laced graph (lace_fast:
    fragments, incremental global unchop, edges merged by position).  SXG_SMOOTH_LEGACY=1 takes the ograph_t path instead:
    same bytes on graphs with reverse steps, merges across block boundaries (no consensus paths to block them), padding on
    and off, the abPOA consensus filter and the reference's DRB1 input with real block discovery."""
    prov = OracleProvider()
    kw = {}
    if case.startswith("synthetic"):
        jobs = [(synthetic_gfa(seed), dict(target_bp=tb)) for seed in (0, 1, 2, 3) for tb in (90, 200)]
    elif case.startswith("haplotypes"):
        jobs = [(haplotype_gfa(seed, n_paths=6, length=1500), dict(target_bp=tb)) for seed in (0, 1) for tb in (300, 500)]
        if case.endswith("nopad"):
            kw["poa_padding_fraction"] = 0.0
    elif case == "inversion":
        jobs = [(inversion_gfa(3), dict(target_bp=300))]
    elif case == "abpoa_cons":
        jobs = [(haplotype_gfa(4, n_paths=5, length=1200), dict(target_bp=400))]
        kw.update(use_abpoa=1, add_consensus=1)
    else:
        jobs = [(open(DRB1).read(), dict(discover=dict(target_poa_length=700, n_haps=12, max_path_jump=5000, max_edge_jump=5000)))]
    if case.endswith("_cons"):
        kw["add_consensus"] = 1
    for gfa, how in jobs:
        sm = S.Smoother(gfa, **how)
        p = S.default_params(**kw)
        monkeypatch.delenv("SXG_SMOOTH_LEGACY", raising=False)
        fast = sm.smooth_gfa(p, prov.provider())
        monkeypatch.setenv("SXG_SMOOTH_LEGACY", "1")
        legacy = sm.smooth_gfa(p, prov.provider())
        monkeypatch.delenv("SXG_SMOOTH_LEGACY", raising=False)
        sm.close()
        assert fast == legacy


def test_parameter_struct_is_checked():
    """sxg_smooth_params carries its size (a caller built against another header is refused instead of having fields read from
    whatever follows its struct), and scores whose engine form does not fit int8 are refused instead of wrapping."""
    L = S.load_library()
    assert L.sxg_smooth_abi_version() == 2
    sm = S.Smoother(synthetic_gfa(0), target_bp=200)
    prov = OracleProvider()
    p = S.default_params()
    assert p.struct_size == C.sizeof(S.SmoothParams) and p.abpoa_band_local == 1
    p.struct_size -= 4
    with pytest.raises(S.SmoothError, match="struct_size"):
        sm.smooth_gfa(p, prov.provider())
    with pytest.raises(S.SmoothError, match="int8"):
        sm.smooth_gfa(S.default_params(poa_q=200), prov.provider())
    with pytest.raises(S.SmoothError, match="int8"):
        sm.smooth_gfa(S.default_params(use_abpoa=1, poa_q=100, poa_c=30), prov.provider())
    with pytest.raises(S.SmoothError):
        sm.collect_text(0, S.default_params(poa_n=-4))
    sm.close()


def tandem_repeat_gfa(seed, n_paths=4, unit=1300, copies=(3, 4, 3, 5), flank=400, node_bp=70, sub=0.01):
    """Haplotypes that carry a tandem repeat: `copies[p]` copies of a `unit`-base motif between unique flanks, every path
    on its own chain of nodes (so the block spans the whole locus and its ranges are several kbp long)."""
    rng = np.random.default_rng(seed)
    motif = rng.integers(0, 4, unit)
    left, right = rng.integers(0, 4, flank), rng.integers(0, 4, flank)
    lines, plines, nid = ["H\tVN:Z:1.0"], [], 1
    for p in range(n_paths):
        hap = np.concatenate([left] + [motif] * copies[p % len(copies)] + [right])
        mut = rng.random(len(hap)) < sub
        hap[mut] = (hap[mut] + rng.integers(1, 4, int(mut.sum()))) % 4
        text = "".join("ACGT"[c] for c in hap)
        steps = []
        for a in range(0, len(text), node_bp):
            lines.append("S\t%d\t%s" % (nid, text[a:a + node_bp]))
            steps.append("%d+" % nid)
            nid += 1
        plines.append("P\thap%d\t%s\t*" % (p, ",".join(steps)))
    return "\n".join(lines + plines) + "\n"


def test_repeat_aware_cut_length_of_break_blocks():
    """src/breaks.cpp:224-272 (break_repeats = true is what src/main.cpp:476 always passes): a block that has to be cut and
    whose ranges hold a tandem repeat is cut at half the mean repeat length, every range of it.  The detector is a decree
    (sautocorr is absent): C++ == Python restatement, the repeat is found at its true period, the cut differs from the blind
    one, and the iteration on the repeat-cut blocks still preserves every path."""
    text = tandem_repeat_gfa(11)
    g = SO.Graph(text)
    # one block per... the locus: every path one range of ~5-7 kbp
    blocks = [[(p, 0, len(g.steps[p]), len(g.path_sequence(p))) for p in range(len(g.pname))]]
    blocks[0].sort(key=lambda r: -r[3])
    seq = g.path_sequence(blocks[0][0][0])
    rl = SO.repeat_length(seq, 1000, 20000, 5.0, 50)
    assert rl == 1300.0                                    # the motif's period (first lag of greatest z: not 2600, 3900)
    assert SO.repeat_length(seq[:1900], 1000, 20000, 5.0, 50) == 0.0   # shorter than two minimal copies
    rng = np.random.default_rng(3)
    assert SO.repeat_length("".join("ACGT"[c] for c in rng.integers(0, 4, 6000)), 1000, 20000, 5.0, 50) == 0.0
    want = SO.break_blocks(g, blocks, 2000)
    blind = SO.break_blocks(g, blocks, 2000, repeats=None)
    assert want != blind
    assert max(r[3] for r in want[0]) <= 650 + 70          # pieces of just over 1300 / 2 bases (closed by a node of 70)
    sm = S.Smoother(text, blocks=blocks)
    L = sm.L
    for rep, ref in (((1000, 20000, 5.0, 50), want), (None, blind), ((500, 3000, 4.0, 25), SO.break_blocks(g, blocks, 2000, repeats=(500, 3000, 4, 25)))):
        out = C.c_void_p()
        if rep is None:
            assert L.sxg_blockset_break_ex(sm.g, sm.b, 2000, 0, 1000, 20000, 5.0, 50, 1, C.byref(out)) == 0
        else:
            assert L.sxg_blockset_break_ex(sm.g, sm.b, 2000, 1, rep[0], rep[1], rep[2], rep[3], 1, C.byref(out)) == 0
        keep, sm.b = sm.b, out
        got = [sm.block_ranges(k) for k in range(sm.n_blocks)]
        sm.b = keep
        L.sxg_blockset_free(out)
        assert got == [[tuple(r) for r in blk] for blk in ref]
    out = C.c_void_p()
    assert L.sxg_blockset_break(sm.g, sm.b, 2000, 1, C.byref(out)) == 0   # the reference's defaults
    keep, sm.b = sm.b, out
    assert [sm.block_ranges(k) for k in range(sm.n_blocks)] == [[tuple(r) for r in blk] for blk in want]
    sm.b = keep
    L.sxg_blockset_free(out)
    sm.close()
    # discovery with the defaults takes the repeat-aware cut, end to end
    sm = S.Smoother(text, discover=dict(target_poa_length=4000, n_haps=4, max_poa_length=2000))
    prov = OracleProvider()
    got = sm.smooth_gfa(S.default_params(), prov.provider())
    o = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert o.path_sequence(o.pname.index(nm)) == g.path_sequence(q)
    blocks2 = SO.break_blocks(g, SO.smoothable_blocks(g, 4000 * 4, 4000, 100, 0), 2000)
    assert got == SO.smooth(g, blocks2)
    sm.close()


@pytest.mark.parametrize("legacy", [False, True])
def test_validation_catches_a_corrupted_block(legacy, monkeypatch):
    """Every laced path must spell its original sequence (src/main.cpp:770-810).  A provider whose POA result has one
    letter wrong -- a node that paths visit, turned into another base -- must make the iteration fail with the path's
    name, on the flat path (run-wise comparison of consecutive node ids) and on the laced-graph path."""
    class Corrupting(OracleProvider):
        def _run(self, ctx, pin, pout):
            rc = OracleProvider._run(self, ctx, pin, pout)
            arrs = self.keep[-1]
            code, paths = arrs["node_code"], arrs["paths"]
            if len(paths) > 40:
                v = int(paths[len(paths) // 2])        # (a node of the first block that a path steps on)
                code[v] = (int(code[v]) + 1) % 4
            return rc
    if legacy:
        monkeypatch.setenv("SXG_SMOOTH_LEGACY", "1")
    else:
        monkeypatch.delenv("SXG_SMOOTH_LEGACY", raising=False)
    sm = S.Smoother(haplotype_gfa(2, n_paths=5, length=900), target_bp=300)
    p = S.default_params()
    good = sm.smooth_gfa(p, OracleProvider().provider())
    assert good.startswith("H\tVN:Z:1.0")
    with pytest.raises(S.SmoothError, match="corrupted"):
        sm.smooth_gfa(p, Corrupting().provider())
    if not legacy:   # several chunks: every block's ranges are checked when its chunk comes back, the failure surfaces the same way
        monkeypatch.setenv("SXG_SMOOTH_CHUNK_BLOCKS", "2")
        assert sm.smooth_gfa(p, OracleProvider().provider()) == good
        with pytest.raises(S.SmoothError, match="corrupted"):
            sm.smooth_gfa(p, Corrupting().provider())
    sm.close()
