"""GPU: one smoothing iteration end to end -- collection (A2-A4) -> ONE batched GPU POA call ->
block graphs (A9/A10) -> lacing -> GFA -- through the two C ABIs, against the oracle stack
(oracle/smooth_oracle.py + oracle/poa_oracle.c) on the reference's own DRB1 test input."""
import os

import pytest

from oracle import smooth_oracle as SO
from smoothxg_amd import smooth as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DRB1 = os.path.join(HERE, "golden", "DRB1-3123.seqwish.gfa")


@pytest.mark.parametrize("cons", [0, 1])
def test_drb1_smoothing_iteration_on_gpu(engine, cons):
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 700)           # the reference's ctest uses -l 700,900,1100 (CMakeLists.txt:565)
    got = sm.smooth_gfa(S.default_params(add_consensus=cons), S.gpu_provider(engine))
    want = SO.smooth(g, SO.blockset_by_path_windows(g, 700), add_consensus=bool(cons))
    assert got == want
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):     # src/main.cpp:770-803
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
    assert len(out.pname) == 12 + (20 if cons else 0)


def test_global_alignment_and_block_graph_on_gpu(engine):
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 1100)
    blocks = SO.blockset_by_path_windows(g, 1100)
    p = S.default_params(local_alignment=0)
    for k in (0, 5):
        c = SO.collect(g, blocks[k])
        code, paths, cn = SO.poa(c, local=False)
        assert sm.block_graph_gfa(k, p, S.gpu_provider(engine)) == SO.to_gfa(SO.build_block_graph(c, code, paths, cn, ""))


def _haplotype_gfa(seed, n_paths, length, sub, node_bp=80):
    import numpy as np
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, length)
    lines, plines, nid = ["H\tVN:Z:1.0"], [], 1
    for p in range(n_paths):
        hap = anc.copy()
        mut = rng.random(length) < sub * (1 + p % 3)          # blocks of different identity
        hap[mut] = (hap[mut] + rng.integers(1, 4, int(mut.sum()))) % 4
        text = "".join("ACGT"[c] for c in hap)
        steps = []
        for a in range(0, length, node_bp):
            lines.append("S\t%d\t%s" % (nid, text[a:a + node_bp]))
            steps.append("%d+" % nid)
            nid += 1
        plines.append("P\thap%d\t%s\t*" % (p, ",".join(steps)))
    return "\n".join(lines + plines) + "\n"


@pytest.mark.parametrize("sub", [0.001, 0.008])
def test_adaptive_scores_reach_the_gpu_per_block(engine, sub):
    """A14 (-a) end to end on the GPU: per-block score tiers (q up to 81) through per_block_params."""
    text = _haplotype_gfa(7, 8, 3000, sub)
    g = SO.Graph(text)
    sm = S.Smoother(text, 1000)
    blocks = SO.blockset_by_path_windows(g, 1000)
    picked = {SO.block_scores(g, r, True, 17, 1000) for r in blocks}
    assert picked - {(1, 4, 6, 2, 26, 1)}, picked            # some upper tier is in play
    got = sm.smooth_gfa(S.default_params(adaptive_poa_params=1), S.gpu_provider(engine))
    assert got == SO.smooth(g, blocks, adaptive=True, kmer_size=17)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


@pytest.mark.parametrize("cons", [0, 1])
def test_maf_rows_from_the_gpu_msa(engine, cons):
    """MSA -> MAF rows (src/smooth.cpp:782-905) with the MSA coming from the GPU engine."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 900)
    blocks = SO.blockset_by_path_windows(g, 900)
    p = S.default_params(add_consensus=cons)
    for k in (0, 4, 8):
        c = SO.collect(g, blocks[k])
        msa, clen = SO.poa_msa(c, bool(cons))
        rows = SO.maf_rows(g, blocks[k], c, msa, ("Consensus_%d" % k) if cons else "", clen)
        assert sm.block_maf(k, p, S.gpu_provider(engine)) == SO.maf_block_text(rows)


@pytest.mark.parametrize("tl", [700, 1100])
def test_drb1_real_block_discovery_round_trip_on_gpu(engine, tl):
    """Config 1/5 with REAL blocks: smoothable_blocks + break_blocks' length cut with the flags of the reference's
    ctest (CMakeLists.txt:565: -j 5k -e 5k -l 700,...,1100 -r 12), collection, ONE batched GPU POA call, lacing:
    GFA byte-identical to the oracle stack, all 12 paths preserved."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    blocks = SO.break_blocks(g, SO.smoothable_blocks(g, tl * 12, tl, 5000, 5000), 2 * tl)
    sm = S.Smoother(text, discover=dict(target_poa_length=tl, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
    assert [sm.block_ranges(k) for k in range(sm.n_blocks)] == [[tuple(r) for r in blk] for blk in blocks]
    got = sm.smooth_gfa(S.default_params(add_consensus=1), S.gpu_provider(engine))
    assert got == SO.smooth(g, blocks, add_consensus=True)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


def test_drb1_maf_merging_and_flips_on_gpu(engine):
    """A13 + 8f-4 with the GPU's MSAs: MAF text (merged groups), flip set and the laced GFA with merged consensus paths
    equal the oracle stack's on the reference's DRB1 input (-M -J 0.5, consensus on)."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 900)
    blocks = SO.blockset_by_path_windows(g, 900)
    p = S.default_params(add_consensus=1)
    got = sm.smooth_maf_gfa(p, S.gpu_provider(engine), merge_blocks=True, jaccard=0.5, header="##maf version=1")
    want = SO.smooth(g, blocks, add_consensus=True, merge=dict(merge_blocks=True, jaccard=0.5, header="##maf version=1"))
    assert got[1] == want[1]
    assert got[2] == len(want[2])
    assert got[0] == want[0]
    out = SO.Graph(got[0])
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


@pytest.mark.parametrize("cons", [0, 1])
def test_drb1_abpoa_path_on_gpu(engine, cons):
    """-A: the smooth_abpoa path (A11) end to end on the GPU -- abPOA's score convention, its ADAPTIVE band
    (params.banded = 2: the one-wave kernel finds every row's best cells while sweeping), consensus restricted to visited
    nodes (build_odgi_abPOA) -- equals the oracle stack byte for byte and preserves the 12 paths."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    blocks = SO.break_blocks(g, SO.smoothable_blocks(g, 900 * 12, 900, 5000, 5000), 1800)
    sm = S.Smoother(text, discover=dict(target_poa_length=900, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
    got = sm.smooth_gfa(S.default_params(add_consensus=cons, use_abpoa=1), S.gpu_provider(engine))
    assert engine.stats()["dom_row_mode"] == 3
    assert got == SO.smooth(g, blocks, add_consensus=bool(cons), abpoa=True)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)


def test_drb1_abpoa_path_global_and_unbanded_local_on_gpu(engine):
    """-A -Z: the smooth_abpoa path with GLOBAL alignment runs the adaptive band too (src/smooth.cpp:259-271 sets wb / wf for
    both modes).  -A with abpoa_band_local = 0: the reading in which upstream abPOA switches its band off in local mode --
    abPOA's scores on the full matrix.  Both equal the oracle stack byte for byte."""
    text = open(DRB1).read()
    g = SO.Graph(text)
    blocks = SO.break_blocks(g, SO.smoothable_blocks(g, 900 * 12, 900, 5000, 5000), 1800)
    sm = S.Smoother(text, discover=dict(target_poa_length=900, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
    got = sm.smooth_gfa(S.default_params(use_abpoa=1, local_alignment=0), S.gpu_provider(engine))
    assert engine.stats()["dom_row_mode"] == 3
    assert got == SO.smooth(g, blocks, abpoa=True, local=False)
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
    got = sm.smooth_gfa(S.default_params(use_abpoa=1, abpoa_band_local=0), S.gpu_provider(engine))
    assert engine.stats()["dom_row_mode"] != 3
    assert got == SO.smooth(g, blocks, abpoa=True, band_local=False)
    sm.close()


@pytest.mark.parametrize("spoa_order", [1, 0])
def test_drb1_three_chained_iterations_as_the_reference_ctest_runs_them(engine, spoa_order):
    """The reference's own test configuration (CMakeLists.txt:565: -l 700,900,1100 -j 5k -e 5k -r 12) is THREE smoothing
    iterations, each on the graph the one before wrote (src/main.cpp:374-1065); consensus paths only in the last
    (src/main.cpp:404).  Every iteration: real block discovery on the previous GFA, one batched GPU POA call, lacing.
    After each one the 12 paths still spell their sequences and the path count is kept (src/main.cpp:770-810); the GFA
    of every iteration is byte-identical to the oracle stack's, pinned by size and SHA-256 in
    tests/golden/drb1_chain.json (made by tests/golden/make_drb1_chain.py; the oracle chain takes minutes on a CPU) -- in spoa's node
    order (poa_spoa_order = 1, the default since round 6: "iterations") and in the incrementally kept one ("iterations_s7")."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(HERE, "golden", "drb1_chain.json")))["iterations" if spoa_order else "iterations_s7"]
    text = open(DRB1).read()
    g0 = SO.Graph(text)
    for it, tl in enumerate((700, 900, 1100)):
        last = it == 2
        sm = S.Smoother(text, discover=dict(target_poa_length=tl, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
        assert sm.n_blocks == gold[it]["blocks"]
        text = sm.smooth_gfa(S.default_params(add_consensus=1 if last else 0, poa_spoa_order=spoa_order), S.gpu_provider(engine))
        sm.close()
        assert len(text) == gold[it]["gfa_bytes"], f"iteration {it} (-l {tl})"
        assert hashlib.sha256(text.encode()).hexdigest() == gold[it]["sha256"], f"iteration {it} (-l {tl})"
        out = SO.Graph(text)
        names = [nm for nm in out.pname if not nm.startswith("Consensus_")]
        assert sorted(names) == sorted(g0.pname) and len(out.pname) == gold[it]["paths"]
        for q, nm in enumerate(g0.pname):
            assert out.path_sequence(out.pname.index(nm)) == g0.path_sequence(q)


@pytest.mark.parametrize("cons_mode", [0, 1, 2])
def test_device_block_graphs_equal_the_restatement(engine, oracle, cons_mode):
    """sxg_poa_batch_in::want_block_graph: A9 + A10 computed by the block-graph kernel right after the alignments (trim,
    path-supported edges, unchop, Kahn order, compact step lists) through the C ABI, against the Python restatement
    (oracle/smooth_oracle.py::build_block_graph) fed with the oracle's POA of the same block; with and without padding trim,
    spoa-style and abPOA-style (visited nodes only) consensus; mode 2 leaves the per-base paths out, mode 3 the raw POA graphs too."""
    import numpy as np
    import smoothxg_amd as SX
    from helpers import random_block
    rng = np.random.default_rng(170 + cons_mode)
    blocks, trims = [], []
    for trial in range(24):
        S_ = int(rng.integers(1, 14))
        if trial % 6 == 0:
            seqs = [rng.integers(0, 5, int(rng.integers(1, 40)), dtype=np.uint8) for _ in range(S_)]
        elif trial % 6 == 1:
            seqs = random_block(rng, S_, int(rng.integers(1500, 2600)), div=0.03)
        else:
            seqs = random_block(rng, S_, int(rng.integers(2, 500)), div=0.08)
        blocks.append(seqs)
        trims.append(0 if trial % 3 == 0 else int(rng.integers(1, 30)))
    prm = SX.Params(1, -4, -6, -2, -26, -1, 0, 0)
    flat = [s for blk in blocks for s in blk]
    bases = np.concatenate(flat).astype(np.uint8)
    seq_off = np.zeros(len(flat) + 1, np.int64)
    seq_off[1:] = np.cumsum([len(s) for s in flat])
    blk_off = np.zeros(len(blocks) + 1, np.int32)
    blk_off[1:] = np.cumsum([len(b) for b in blocks])
    for mode in (1, 2, 3):
        res = engine.run_flat(bases, seq_off, blk_off, None, prm, want_consensus=cons_mode > 0, block_graph=mode, bg_trim=trims,
                              bg_cons_visited_only=(cons_mode == 2))
        assert (res[0].paths is None) == (mode >= 2)
        assert (res[0].node_code is None) == (mode == 3) and (res[0].edge_tail is None) == (mode == 3)
        for b, seqs in enumerate(blocks):
            assert res[b].status == 0
            g, _, _ = oracle.block_run(seqs, None, oracle.mkparams(1, -4, -6, -2, -26, -1, 0))
            c = SO.Collected()
            c.poa_padding = trims[b]
            c.seqs = ["".join("ACGTN"[min(int(x), 4)] for x in s) for s in seqs]
            c.dup_seq_names = [["s%d" % i] for i in range(len(seqs))]
            c.dup_is_revs = [[False] for _ in seqs]
            c.all_names = ["s%d" % i for i in range(len(seqs))]
            G = SO.build_block_graph(c, g.nodes()[0], [g.seq_path(k) for k in range(len(seqs))], g.consensus(),
                                     "cons" if cons_mode else "", abpoa=(cons_mode == 2))
            assert res[b].bg.gfa(c.dup_seq_names, None, "cons" if cons_mode else None) == SO.to_gfa(G), (mode, b)


def test_device_block_graphs_feed_the_same_gfa_as_host_built_ones(engine, monkeypatch):
    """The iteration asks the engine for block graphs only (want_block_graph = 3) and laces them without the laced graph;
    SXG_SMOOTH_LEGACY=1 builds block graphs on the host from the per-base paths and laces through ograph_t: same bytes,
    with padding and consensus paths, on DRB1 with real block discovery."""
    text = open(DRB1).read()
    for cons in (0, 1):
        sm = S.Smoother(text, discover=dict(target_poa_length=700, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
        p = S.default_params(add_consensus=cons)
        monkeypatch.delenv("SXG_SMOOTH_LEGACY", raising=False)
        fast = sm.smooth_gfa(p, S.gpu_provider(engine))
        assert engine.stats()["bg_ms"] > 0
        monkeypatch.setenv("SXG_SMOOTH_LEGACY", "1")
        legacy = sm.smooth_gfa(p, S.gpu_provider(engine))
        monkeypatch.delenv("SXG_SMOOTH_LEGACY", raising=False)
        sm.close()
        assert fast == legacy


def test_sharded_runs_carry_block_graphs_built_by_the_owning_rank(request):
    """Multi-GPU: every rank builds the block graphs of ITS blocks on its device and sends compact graphs (not one node id
    per base) to the rank that laces.  Played with simulated ranks on the one GPU (partition, per-rank blobs with the bg
    sections, the root's assembly in batch order) and through a real one-rank communicator: the block graphs equal the
    single-GPU ones, mode 2 leaves the per-base paths out of the blobs, and sxg_smooth_gfa over the sharded entry gives the
    single-GPU bytes.  (In a process of its own: see helpers.rerun_in_own_process.)"""
    import numpy as np
    import smoothxg_amd as SX
    from helpers import random_block, rerun_in_own_process
    if rerun_in_own_process(request):
        return
    engine = request.getfixturevalue("engine")
    rng = np.random.default_rng(311)
    blocks = [random_block(rng, int(rng.integers(1, 10)), int(rng.choice([40, 300, 900])), div=0.06) if b != 4 else [] for b in range(13)]
    trims = [int(rng.integers(0, 12)) for _ in blocks]
    flat = [s for blk in blocks for s in blk]
    bases = np.concatenate(flat).astype(np.uint8)
    seq_off = np.zeros(len(flat) + 1, np.int64)
    seq_off[1:] = np.cumsum([len(s) for s in flat])
    blk_off = np.zeros(len(blocks) + 1, np.int32)
    blk_off[1:] = np.cumsum([len(b) for b in blocks])
    prm = SX.Params(1, -4, -6, -2, -26, -1, 0, 0)
    ref = engine.run_flat(bases, seq_off, blk_off, None, prm, want_consensus=True, block_graph=1, bg_trim=trims)
    names = lambda b: [["s%d" % i] for i in range(len(blocks[b]))]

    def same(res, mode):
        for b, (a, r) in enumerate(zip(res, ref)):
            assert a.status == r.status == 0
            assert (a.paths is None) == (mode == 2)
            if blocks[b]:
                assert a.bg.gfa(names(b), None, "c") == r.bg.gfa(names(b), None, "c")
                assert (a.bg.node_indeg == r.bg.node_indeg).all()
    for nranks in (1, 2, 3):
        for mode in (1, 2):
            same(engine.run_flat_sharded(bases, seq_off, blk_off, None, prm, want_consensus=True, simulate_ranks=nranks, block_graph=mode,
                                         bg_trim=trims), mode)
    engine.comm_init(engine.comm_unique_id(), 1, 0)
    try:
        same(engine.run_flat_sharded(bases, seq_off, blk_off, None, prm, want_consensus=True, block_graph=2, bg_trim=trims), 2)
        text = open(DRB1).read()
        sm = S.Smoother(text, 900)
        p = S.default_params(add_consensus=1)
        assert sm.smooth_gfa(p, S.gpu_provider(engine, sharded=True)) == sm.smooth_gfa(p, S.gpu_provider(engine))
        sm.close()
    finally:
        engine.lib.sxg_poa_comm_destroy(engine.h)


def test_step_lists_through_the_pinned_pool_equal_the_pageable_download(engine, monkeypatch):
    """From 16 M steps on, the step lists of the block graphs are downloaded into a pinned buffer that the handle lends to
    the result (PinPool; pinned in the background while the kernels run) and takes back when the result is freed.  Two
    runs in a row (the second borrows the same buffer) and a run with SXG_POA_NO_PINNED=1 must give the same block graphs."""
    import numpy as np
    import smoothxg_amd as SX
    from smoothxg_amd import synth
    bases, seq_off, blk_off = synth.make_batch(80, 64, 5000)     # 25.6 M bases, ~0.8 steps per base at this depth: past the pool's 16 M steps
    prm = SX.Params(1, -4, -6, -2, -26, -1, 0, 0)

    def digest(res):
        import hashlib
        h = hashlib.sha256()
        for r in res:
            assert r.status == 0 and r.paths is None
            for p in r.bg.paths:
                h.update(np.ascontiguousarray(p, np.int32).tobytes())
            h.update("".join(r.bg.node_seq).encode())
        return h.hexdigest()

    monkeypatch.delenv("SXG_POA_NO_PINNED", raising=False)
    first = engine.run_flat(bases, seq_off, blk_off, None, prm, block_graph=3)
    assert sum(len(p) for r in first for p in r.bg.paths) >= (16 << 20)
    a = digest(first)
    b = digest(engine.run_flat(bases, seq_off, blk_off, None, prm, block_graph=2))
    monkeypatch.setenv("SXG_POA_NO_PINNED", "1")
    c = digest(engine.run_flat(bases, seq_off, blk_off, None, prm, block_graph=2))
    assert a == b == c
