"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

from helpers import PARAM_SETS, assert_block_equal, gparams, oparams, random_block, rerun_in_own_process
from smoothxg_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_graph_after(oracle, seqs, k, p):
    g, _, _ = oracle.block_run(seqs[:k], None, p)
    return g


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_align_only_matches_oracle(engine, oracle, mode, pname):
    """sxg_poa_align_batch == oracle Align on graphs of growing depth (A6)."""
    rng = np.random.default_rng(7 + mode)
    problems, expect = [], []
    for trial in range(12):
        S = int(rng.integers(2, 9))
        L = int(rng.integers(8, 300))
        seqs = random_block(rng, S, L, div=0.08)
        p = oparams(pname, mode)
        g = _oracle_graph_after(oracle, seqs, S - 1, p)
        codes, off, pred, sink, row_node = g.rows()
        q = seqs[-1]
        an, ap, sc = oracle.align_csr(codes, off, pred, sink, q, p)
        problems.append((codes, off, pred, sink, q))
        expect.append((an, ap, sc))
    got = engine.align(problems, gparams(pname, mode))
    for k, ((pr, pp, sc, st), (an, ap, esc)) in enumerate(zip(got, expect)):
        assert st == 0
        assert sc == esc, f"problem {k}: score {sc} != {esc}"
        assert len(pr) == len(an), f"problem {k}: {len(pr)} pairs != {len(an)}"
        assert (pr == an).all() and (pp == ap).all(), f"problem {k}: alignment differs"


def test_align_only_global_alignment_with_a_strongly_negative_score(engine, oracle):
    """the align-only kernel admits packed global alignments under the strict range rule (all-gap corner above
    -15 800) and has no wider re-run; the packed walk's clamp threshold (-16 000 + m L + m, made for whole blocks that CAN be re-run)
    must not stop it at a legitimately low score.  Two homopolymers of different letters (3 900 / 3 000 letters), linear gaps of 2,
    mismatch 10: the optimum is all gaps (-15 526 / -11 926, below the thresholds -12 099 / -12 999 from the first cell on or soon
    after) -- pairs and score equal the oracle's."""
    import smoothxg_amd as S
    rng = np.random.default_rng(4711)
    problems, expect = [], []
    for L in (3900, 3000):
        a, b = np.zeros(L, np.uint8), np.ones(L - 37, np.uint8)
        b[::97] = rng.integers(0, 4, len(b[::97]), dtype=np.uint8)   # (a few chance matches, so that the walk has decisions to take)
        p = oracle.mkparams(1, -10, -2, -2, -2, -2, mode=1)
        g, _, _ = oracle.block_run([a], None, p)
        codes, off, pred, sink, _ = g.rows()
        an, ap, sc = oracle.align_csr(codes, off, pred, sink, b, p)
        assert sc < -11000, sc              # (a walk through cells at and below the threshold)
        problems.append((codes, off, pred, sink, b))
        expect.append((an, ap, sc))
    got = engine.align(problems, S.Params(1, -10, -2, -2, -2, -2, 1, 0))
    for k, ((pr, pp, sc, st), (an, ap, esc)) in enumerate(zip(got, expect)):
        assert st == 0 and sc == esc, (k, st, sc, esc)
        assert len(pr) == len(an) and (pr == an).all() and (pp == ap).all(), f"problem {k}: alignment differs"


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_blocks_match_oracle_small(engine, oracle, mode, pname):
    """sxg_poa_batch_run == oracle for whole blocks (A6+A7), many shapes in one batch."""
    rng = np.random.default_rng(100 + mode)
    blocks, weights = [], []
    for trial in range(24):
        S = int(rng.integers(1, 14))
        L = int(rng.integers(1, 260))
        if trial % 5 == 0:
            seqs = [rng.integers(0, 5, int(rng.integers(1, 40)), dtype=np.uint8) for _ in range(S)]
        else:
            seqs = random_block(rng, S, L, div=0.06, alphabet=5 if trial % 7 == 0 else 4)
        blocks.append(seqs)
        weights.append(rng.integers(1, 5, len(seqs)).astype(np.uint32))
    res = engine.run_blocks(blocks, gparams(pname, mode), weights=weights, want_consensus=True, want_msa=True)
    for b, (seqs, w) in enumerate(zip(blocks, weights)):
        g, sc, cells = oracle.block_run(seqs, w, oparams(pname, mode))
        assert_block_equal(res[b], g, sc, cells, label=f"{pname}/mode{mode}/block{b}")
        assert (res[b].consensus == g.consensus()).all(), f"block {b}: consensus differs"
        assert res[b].msa == g.msa(True), f"block {b}: MSA differs"


@pytest.mark.parametrize("mode", [0, 1])
def test_config2_shape_blocks(engine, oracle, mode):
    """BASELINE config 2 shape (16 seqs x 1 kbp), synthetic generator, default convex scores."""
    bases, seq_off, blk_off = synth.make_batch(6, 16, 1000)
    res = engine.run_flat(bases, seq_off, blk_off, None, gparams("convex_default", mode))
    for b in range(6):
        seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[b], blk_off[b + 1])]
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", mode))
        assert_block_equal(res[b], g, sc, cells, label=f"c2/mode{mode}/block{b}")


def test_mixed_lengths_pick_every_variant(engine, oracle):
    """Blocks of very different lengths in one batch exercise several (T,W) kernel variants."""
    rng = np.random.default_rng(5)
    blocks = []
    for L in (40, 500, 900, 1900, 2900, 4000, 5000, 6000):
        blocks.append(random_block(rng, 3, L, div=0.03))
    res = engine.run_blocks(blocks, gparams("convex_default", 0))
    for b, seqs in enumerate(blocks):
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
        assert_block_equal(res[b], g, sc, cells, label=f"len{len(seqs[0])}")


def test_edge_cases(engine, oracle):
    """Empty batch, empty block, single-sequence block, length-1 sequences, duplicates."""
    assert engine.run_blocks([], gparams("convex_default", 0)) == []
    one = np.array([2], np.uint8)
    blocks = [[], [one], [one, one.copy()], [np.array([0, 1, 2, 3, 4], np.uint8)] * 3,
              [np.zeros(50, np.uint8), np.ones(50, np.uint8)]]
    res = engine.run_blocks(blocks, gparams("convex_default", 0))
    assert res[0].status == 0 and len(res[0].node_code) == 0
    for b in range(1, len(blocks)):
        g, sc, cells = oracle.block_run(blocks[b], None, oparams("convex_default", 0))
        assert_block_equal(res[b], g, sc, cells, label=f"edge{b}")


def test_letters_above_four_read_as_n(engine, oracle):
    """Codes 5..255 in the input are N (4): clamped on the device after the upload (clamp_bases_kernel), at every
    alignment of the buffer and at its very end."""
    rng = np.random.default_rng(8)
    seqs = random_block(rng, 5, 333, div=0.05)
    dirty = [s.copy() for s in seqs]
    for s in dirty:
        idx = rng.integers(0, len(s), 9)
        s[idx] = rng.integers(5, 256, 9).astype(np.uint8)
        s[-1] = 200
    clean = [np.minimum(s, 4) for s in dirty]
    res = engine.run_blocks([dirty, [np.array([9], np.uint8)]], gparams("convex_default", 0))
    g, sc, cells = oracle.block_run(clean, None, oparams("convex_default", 0))
    assert_block_equal(res[0], g, sc, cells, label="dirty letters")
    assert res[1].status == 0 and res[1].node_code.tolist() == [4]


def test_too_long_is_reported_not_crashed(engine):
    import smoothxg_amd as S
    blk = [[np.zeros(30000, np.uint8), np.zeros(10, np.uint8)]]
    res = engine.run_blocks(blk, gparams("convex_default", 0), check=False)
    assert res[0].status == 5


def test_local_alignments_beyond_12_kbp_run_the_16_wave_packed_classes(engine, oracle):
    """smoothxg with -l 13k cuts ranges at 26 kbp (src/main.cpp:376): local alignments of 12-26 kbp run 1024-thread packed
    classes (8, 10, 12, 13 columns per strip); a global alignment beyond 14.5 kbp has no sweep and is reported."""
    rng = np.random.default_rng(1213)
    blocks = [random_block(rng, 3, L, div=0.02) for L in (14000, 18500, 23000, 26300)]
    res = engine.run_blocks(blocks, gparams("convex_default", 0))
    st = engine.stats()
    for b, seqs in enumerate(blocks):
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
        assert_block_equal(res[b], g, sc, cells, label=f"long-local{len(seqs[0])}")
    assert st["dom_row_mode"] == 2
    # global: 14 kbp still fits the clamped packed sweep (m L < 14 500, round 5), 18.5 kbp has no sweep and is reported
    g, sc, cells = oracle.block_run(blocks[0], None, oparams("convex_default", 1))
    near = engine.run_blocks([blocks[0]], gparams("convex_default", 1))
    assert_block_equal(near[0], g, sc, cells, label="long-global14000")
    far = engine.run_blocks([blocks[1]], gparams("convex_default", 1), check=False)
    assert far[0].status == 5


def test_band_miss_on_a_long_block_is_rerun_with_a_full_plane(engine, oracle, monkeypatch):
    """A local block of 14 kbp whose traceback cannot stay inside the plane's band (here: a 700 bp insertion and a plane
    narrowed to 96 columns by the test knob, so the in-kernel hint shifts cannot repair it) used to end on the 32-bit
    sweep -- which stops at 12 287 columns, i.e. ST_TOO_LONG for the block and SXG_E_BLOCK for the batch.  The engine now
    repeats the packed sweep with a plane that keeps every strip: same results as the oracle, one retry, still packed."""
    rng = np.random.default_rng(4711)
    base = random_block(rng, 1, 14000, div=0.0)[0]
    ins = rng.integers(0, 4, 700).astype(np.uint8)
    with_sv = np.concatenate([base[:6000], ins, base[6000:13300]])
    third = base.copy()
    third[rng.integers(0, len(third), 150)] = rng.integers(0, 4, 150).astype(np.uint8)
    seqs = [base, with_sv, third]
    g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
    monkeypatch.setenv("SXG_POA_BAND_COLS", "96")
    res = engine.run_blocks([seqs], gparams("convex_default", 0))
    st = engine.stats()
    assert st["retries"] >= 1 and st["dom_row_mode"] == 2, st
    assert_block_equal(res[0], g, sc, cells, label="band-miss-long")
    # with the default plane (1100 columns) the same block needs no retry: the hints follow the insertion
    monkeypatch.delenv("SXG_POA_BAND_COLS")
    res = engine.run_blocks([seqs], gparams("convex_default", 0))
    assert_block_equal(res[0], g, sc, cells, label="band-default-long")


def test_invalid_arguments(engine):
    import smoothxg_amd as S
    with pytest.raises(S.PoaError):
        engine.run_blocks([[np.zeros(4, np.uint8)]], S.Params(1, 4, -6, -2, -26, -1, 0, 0))  # n > 0


@pytest.mark.parametrize("mode", [0, 1])
def test_long_sequences_every_kernel_class(engine, oracle, mode):
    """Lengths that select the 256/512/1024-thread classes, all three strip widths and (in
    global mode beyond ~7 kbp) the 32-bit row words."""
    rng = np.random.default_rng(17 + mode)
    blocks = [random_block(rng, 3, L, div=0.02) for L in (1500, 2500, 3500, 7000, 9000, 11500)]
    res = engine.run_blocks(blocks, gparams("convex_default", mode))
    for b, seqs in enumerate(blocks):
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", mode))
        assert_block_equal(res[b], g, sc, cells, label=f"long{len(seqs[0])}/mode{mode}")


@pytest.mark.parametrize("pname", ["convex_default", "affine_4param"])
def test_deep_divergent_block_many_predecessors(engine, oracle, pname):
    """48 divergent sequences: rows with many predecessors (generic fold + parked register row)."""
    rng = np.random.default_rng(23)
    seqs = random_block(rng, 48, 120, div=0.1)
    for mode in (0, 1):
        g, sc, cells = oracle.block_run(seqs, None, oparams(pname, mode))
        codes, off, pred, sink, _ = g.rows()
        assert np.diff(off).max() >= 4
        res = engine.run_blocks([seqs], gparams(pname, mode), want_consensus=True)
        assert_block_equal(res[0], g, sc, cells, label=f"deep/{pname}/{mode}")
        assert (res[0].consensus == g.consensus()).all()


def test_lds_dma_prefetch_path_is_exact(engine, oracle, monkeypatch):
    """SXG_POA_PREFETCH=1 turns on the global_load_lds prefetch of predecessor rows."""
    monkeypatch.setenv("SXG_POA_PREFETCH", "1")
    rng = np.random.default_rng(31)
    blocks = [random_block(rng, 6, L, div=0.05) for L in (300, 900, 2000)]
    for mode in (0, 1):
        res = engine.run_blocks(blocks, gparams("convex_default", mode))
        for b, seqs in enumerate(blocks):
            g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", mode))
            assert_block_equal(res[b], g, sc, cells, label=f"prefetch/{b}/{mode}")


def test_north_star_shape_properties(engine, oracle):
    """One block of the headline shape (64 x 5 kbp): size-independent invariants on the full
    block, oracle equality on its first 6 sequences."""
    import smoothxg_amd as S
    seqs = synth.make_block(4242, 64, 5000)
    p = gparams("convex_default", 0)
    res = engine.run_blocks([seqs, seqs[:6]], p)
    full, head = res
    assert full.status == 0
    n = len(full.node_code)
    for s, q in enumerate(seqs):                       # the reference's own self-check (src/main.cpp:770-803)
        assert (full.node_code[full.paths[s]] == q).all()
    assert sorted(full.node_rank.tolist()) == list(range(n))
    assert (full.node_rank[full.edge_tail] < full.node_rank[full.edge_head]).all()
    assert int(full.edge_weight.sum()) == sum(2 * (len(q) - 1) for q in seqs)
    assert (full.scores[1:] > 0).all() and (full.scores <= np.array([len(q) for q in seqs])).all()
    order = np.argsort(full.node_rank)                  # aligned groups contiguous, distinct letters
    gs = full.node_group[order]
    assert 1 + int((gs[1:] != gs[:-1]).sum()) == len(set(full.node_group.tolist()))
    g, sc, cells = oracle.block_run(seqs[:6], None, oparams("convex_default", 0))
    assert_block_equal(head, g, sc, cells, label="ns-head")
    assert (full.scores[:6] == sc).all() and (full.cells[:6] == cells).all()


def test_mixed_depth_and_length_batch_properties(engine, oracle):
    """BASELINE config 4 in miniature: blocks of 8-128 sequences x 0.5-10 kbp in ONE batch (several
    launch geometries run concurrently, arenas are sized per geometry).  Invariants on every block,
    oracle equality on the cheapest one."""
    bases, seq_off, blk_off = synth.make_batch(10, 0, 0, first_block=900, mixed=True)
    res = engine.run_flat(bases, seq_off, blk_off, None, gparams("convex_default", 0))
    costs = []
    for b, r in enumerate(res):
        assert r.status == 0
        seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[b], blk_off[b + 1])]
        for s, q in enumerate(seqs):
            assert (r.node_code[r.paths[s]] == q).all()
        assert (r.node_rank[r.edge_tail] < r.node_rank[r.edge_head]).all()
        assert int(r.edge_weight.sum()) == sum(2 * (len(q) - 1) for q in seqs)
        costs.append(int(r.cells.sum()))
    b = int(np.argmin(costs))
    seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[b], blk_off[b + 1])]
    g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
    assert_block_equal(res[b], g, sc, cells, label="mixed-cheapest")


def _packed_geometry(length, spread=False):
    """The (threads, columns per lane) the engine's chooser gives a packed-sweep block whose longest
    sequence has `length` bases: fewest padded columns, wider strip on ties (sxg_poa.hip).  spread: what a batch too
    small to fill the device gets -- twice the waves at half the strip width while both stay legal."""
    best = None
    for W in (12, 11, 10, 9, 8, 7, 6, 5, 4):
        for NW in (1, 2, 3, 4, 8, 12, 16):
            cols = 128 * NW * W
            if cols < length + 1:
                continue
            if (W > 8 and NW > 8) or (W < 8 and NW > 4):
                continue
            if best is None or cols < best[0]:
                best = (cols, NW, W)
            break
    cols, NW, W = best
    while spread and NW <= 2 and W % 2 == 0 and W // 2 >= 4:
        NW, W = 2 * NW, W // 2
    return cols, 64 * NW, 2 * W


@pytest.mark.parametrize("spread", [False, True])
def test_every_packed_strip_width_and_wave_count(engine, oracle, monkeypatch, spread):
    """One block per packed geometry: strip widths 4..12 at 1, 2, 3, 4 and 8 waves (4..7: up to 4 waves); the engine's
    stats confirm the geometry that ran, the oracle confirms the result.  A one-block batch cannot fill the device, so
    the engine spreads it over more waves (spread=True); SXG_POA_NO_SPREAD pins the geometry a large batch would get."""
    if not spread:
        monkeypatch.setenv("SXG_POA_NO_SPREAD", "1")
    rng = np.random.default_rng(41)
    wanted = {}
    for W in (4, 5, 6, 7, 8, 9, 10, 11, 12):
        for NW in (1, 2, 3, 4, 8):
            L = 128 * NW * W - 3
            cols, T, cpl = _packed_geometry(L)
            if (T, cpl) == (64 * NW, 2 * W):
                wanted[(T, cpl)] = L
    assert {c for (_, c) in wanted} == {8, 10, 12, 14, 16, 18, 20, 22, 24}
    assert {t for (t, _) in wanted} >= {64, 128, 192, 256, 512}
    for (T, cpl), L in sorted(wanted.items()):
        seqs = random_block(rng, 3, L, div=0.02)
        top = max(len(s) for s in seqs)
        if _packed_geometry(top)[1:] != (T, cpl):      # indels moved the longest sequence out of the class
            seqs = [s[:L] for s in seqs]
        res = engine.run_blocks([seqs], gparams("convex_default", 0))
        st = engine.stats()
        assert st["dom_row_mode"] == 2
        assert (st["dom_threads"], st["dom_cols_per_lane"]) == _packed_geometry(max(len(s) for s in seqs), spread)[1:]
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
        assert_block_equal(res[0], g, sc, cells, label=f"packed T={T} cols/lane={cpl} L={L} spread={spread}")


def test_global_alignment_beyond_the_int16_corner_stays_packed_and_a_clamped_walk_is_rerun_wider(engine, oracle):
    """Affine global alignment (1,4,6,2): the all-gap corner of the matrix, -(g + 2(N-1)) - (g + 2(L-1)), leaves int16 from
    N + L ~ 8 000 on.  Round 5: the packed sweep clamps H at P16_NWFLOOR and the traceback checks that every cell it decides
    in lies above the level a clamped cell could have lifted -- a block of related sequences runs packed to the end, whatever
    its corner; two sequences with nothing in common walk through clamped cells, the kernel answers RANGE_OVERFLOW and the
    engine re-runs the block on the 32-bit sweep.  Same results as the oracle in both cases."""
    rng = np.random.default_rng(77)
    seqs = random_block(rng, 8, 3000, div=0.25)
    g, sc, cells = oracle.block_run(seqs, None, oparams("affine_4param", 1))
    assert len(g.nodes()[0]) > 5200                      # the corner is beyond -15 800 from N ~ 4 900 on
    res = engine.run_blocks([seqs], gparams("affine_4param", 1))
    st = engine.stats()
    assert st["dom_row_mode"] == 2       # (a retry here is the capacity ladder: at 25 % divergence the graph outgrows the first tier)
    assert_block_equal(res[0], g, sc, cells, label="clamped-packed")
    # nothing in common: the optimal walk runs thousands below zero, into what clamped cells can reach
    far = [rng.integers(0, 4, 9400, dtype=np.uint8), rng.integers(0, 4, 9300, dtype=np.uint8)]
    g3, sc3, cells3 = oracle.block_run(far, None, oparams("affine_4param", 1))
    assert sc3[1] < -16000 + 9400 + 1
    res3 = engine.run_blocks([far], gparams("affine_4param", 1))
    st = engine.stats()
    assert st["retries"] >= 1 and st["dom_row_mode"] != 2
    assert_block_equal(res3[0], g3, sc3, cells3, label="clamped-walk-rerun")
    # ... while a similar block that stays small never needed either
    calm = random_block(rng, 4, 3000, div=0.01)
    g2, sc2, cells2 = oracle.block_run(calm, None, oparams("affine_4param", 1))
    res2 = engine.run_blocks([calm], gparams("affine_4param", 1))
    assert engine.stats()["dom_row_mode"] == 2 and engine.stats()["retries"] == 0
    assert_block_equal(res2[0], g2, sc2, cells2, label="range-ok")


def test_per_block_scores_and_modes_in_one_batch(engine, oracle):
    """per_block_params: every block of one batch brings its own scores AND alignment type -- five
    score sets x local/global x two lengths, so packed and 32-bit sweeps, linear/affine/convex kernels
    and several geometries run side by side in one call."""
    rng = np.random.default_rng(91)
    blocks, gp, op = [], [], []
    for L in (300, 1400):
        for pname in PARAM_SETS:
            for mode in (0, 1):
                blocks.append(random_block(rng, 5, L, div=0.04))
                gp.append(gparams(pname, mode))
                op.append(oparams(pname, mode))
    res = engine.run_blocks(blocks, gp, want_consensus=True)
    assert len(res) == len(blocks) == 20
    for b, seqs in enumerate(blocks):
        g, sc, cells = oracle.block_run(seqs, None, op[b])
        assert_block_equal(res[b], g, sc, cells, label=f"per-block {b}")
        assert (res[b].consensus == g.consensus()).all()


def test_small_memory_budget_runs_blocks_through_few_slots(engine, oracle):
    """sxg_poa_set_memory_budget: with room for a handful of slot arenas, 40 blocks go through the
    work queue of a few persistent workgroups (first item fixed per slot, the rest pulled) -- same results."""
    rng = np.random.default_rng(123)
    blocks = [random_block(rng, 6, 500 + 40 * (b % 7), div=0.04) for b in range(40)]
    try:
        engine.set_memory_budget(64 << 20)
        res = engine.run_blocks(blocks, gparams("convex_default", 0))
        st = engine.stats()
        assert 1 <= st["n_slots"] < len(blocks)
    finally:
        engine.set_memory_budget(0)
    for b, seqs in enumerate(blocks):
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
        assert_block_equal(res[b], g, sc, cells, label=f"budget {b}")


def test_sharded_run_reassembles_in_block_order(request):
    """sxg_poa_batch_run_sharded (the multi-GPU entry of the C ABI): LPT partition by cost, per-rank blobs, assembly on
    the root in the batch's block order.  Played here with 1, 2, 3 and 5 SIMULATED ranks on the one GPU of the box (the
    test entry of the ABI: everything but ncclSend/ncclRecv) and with a real one-rank communicator; mixed block sizes,
    per-block scores, weights, consensus and MSA must all come back as from the single-GPU call.  (In a process of its
    own: see helpers.rerun_in_own_process.)"""
    if rerun_in_own_process(request):
        return
    engine = request.getfixturevalue("engine")
    rng = np.random.default_rng(202)
    blocks, gp = [], []
    names = list(PARAM_SETS)
    for b in range(17):
        L = int(rng.choice([30, 200, 700, 1600]))
        blocks.append(random_block(rng, int(rng.integers(1, 9)), L, div=0.05) if b != 6 else [])
        gp.append(gparams(names[b % len(names)], b % 2))
    seqs = [s for blk in blocks for s in blk]
    bases = np.concatenate(seqs)
    seq_off = np.zeros(len(seqs) + 1, np.int64)
    seq_off[1:] = np.cumsum([len(s) for s in seqs])
    blk_off = np.zeros(len(blocks) + 1, np.int32)
    blk_off[1:] = np.cumsum([len(b) for b in blocks])
    w = rng.integers(1, 4, len(seqs)).astype(np.uint32)
    ref = engine.run_flat(bases, seq_off, blk_off, w, gp, want_consensus=True, want_msa=True)

    def same(res):
        assert len(res) == len(ref)
        for a, b in zip(res, ref):
            assert a.status == b.status == 0
            for f in ("node_code", "node_rank", "node_group", "edge_tail", "edge_head", "edge_weight", "scores", "cells", "consensus"):
                assert (getattr(a, f) == getattr(b, f)).all(), f
            assert all((x == y).all() for x, y in zip(a.paths, b.paths)) and a.msa == b.msa

    for nranks in (1, 2, 3, 5):
        same(engine.run_flat_sharded(bases, seq_off, blk_off, w, gp, want_consensus=True, want_msa=True, simulate_ranks=nranks))
    engine.comm_init(engine.comm_unique_id(), 1, 0)         # a real RCCL communicator of one rank
    try:
        same(engine.run_flat_sharded(bases, seq_off, blk_off, w, gp, want_consensus=True, want_msa=True))
        # the staged form (what bench.py --gpus N times): inputs dealt and uploaded once, the collective stage repeated --
        # with a communicator the size all-gathers and the (empty) send/recv group really run through RCCL
        engine.upload_sharded(bases, seq_off, blk_off, w, gp, want_consensus=True, want_msa=True)
        for _ in range(2):
            engine.execute_sharded()
            info = engine.sharded_info()
            assert (info["ranks_seen"], info["bytes_received"]) == (1, 0) and info["pack_ms"] >= 0 and info["exchange_ms"] >= 0
        same(engine.download_sharded())
    finally:
        engine.lib.sxg_poa_comm_destroy(engine.h)
    # a share the caller cut himself (plain upload) can be executed and exchanged, but the root cannot reassemble a batch it never saw
    engine.upload(bases, seq_off, blk_off, w, gp, want_consensus=True)
    engine.execute_sharded()
    import smoothxg_amd as S
    with pytest.raises(S.PoaError):
        engine.download_sharded()


def test_device_view_tensors_are_the_downloaded_results():
    """What bench.py --gpus N hands to the lacing rank (shard.engine_result_tensors: zero-copy torch views of the engine's
    HBM results) equals what download() returns.  In a process of its own that imports torch FIRST, as bench.py does:
    torch brings its own HIP runtime, and a process that initialised the system one before finds no GPU through torch."""
    import subprocess
    import sys
    pytest.importorskip("torch")
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import smoothxg_amd as S
from smoothxg_amd import synth, shard
eng = S.PoaEngine(0)
bases, so, bo = synth.make_batch(12, 6, 400)
eng.upload(bases, so, bo, None, S.Params(1, -4, -6, -2, -26, -1, 0, 0))
eng.execute()
ts = shard.engine_result_tensors(eng)
res = eng.download()
summ = ts[0].cpu().numpy().reshape(3, -1)
assert (summ[0] == 0).all() and (summ[1] == [len(r.node_code) for r in res]).all()
assert (summ[2] == [len(r.edge_tail) for r in res]).all()
ref = np.concatenate([np.concatenate(r.paths) for r in res])
assert ts[1].is_cuda and (ts[1].cpu().numpy() == ref).all()
print("views ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "views ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n_seqs,length", [(4, 300), (4, 700), (4, 1500), (4, 3000), (3, 5000)])
def test_packed_sweep_handles_the_block_itself(engine, oracle, n_seqs, length):
    """One, two, three and four-wave workgroups of the packed sweep: the blocks are done in one round of launches (one per geometry), with no retry --
    a wrong sweep that the widening ladder silently repairs on the 32-bit kernels (band miss -> re-run) would still
    produce the oracle's results, only slower, and every other parity test would stay green."""
    bases, seq_off, blk_off = synth.make_batch(2, n_seqs, length)
    res = engine.run_flat(bases, seq_off, blk_off, None, gparams("convex_default", 0), want_consensus=False)
    st = engine.stats()
    assert st["retries"] == 0 and st["dp_launches"] <= 2 and st["dom_row_mode"] == 2, st   # (two blocks: at most one launch per geometry)
    for b in range(2):
        seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[b], blk_off[b + 1])]
        g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
        assert_block_equal(res[b], g, sc, cells, f"block {b}")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_spoa_order_option_on_the_device(engine, oracle, mode, monkeypatch):
    """sxg_poa_params::mode | SXG_ORDER_SPOA (decree S7': the depth-first re-sort after every AddAlignment that spoa is
    believed to do; restated from memory, unverified): on the device a walk per root, every thread its own roots, and only the pieces of the order
    the last alignment touched walked again (poa_graph_dev.h::spoa_resort_par; per-node words in LDS); scores,
    graphs, ranks, paths, consensus and MSA equal the oracle run with the same option, on blocks of several shapes
    (packed sweep, banded sweep, deep bubbles), and mixed with default-order blocks in one batch."""
    import smoothxg_amd as S
    if mode == 2:   # (the walk's states in the slot's scratch instead of LDS: what a graph beyond the workgroup's LDS gets)
        monkeypatch.setenv("SXG_POA_RESORT_NO_LDS", "1")
        mode = 0
    rng = np.random.default_rng(900 + mode)
    blocks = [random_block(rng, int(rng.integers(2, 12)), L, div=0.07) for L in (30, 200, 700, 1600)]
    blocks.append(random_block(rng, 24, 300, div=0.12))
    blocks.append(random_block(rng, 20, 2500, div=0.05))     # ~5 000 nodes, a four-wave class: groups, insertions, in-degrees beyond three
    base = rng.integers(0, 4, 80).astype(np.uint8)            # bushy: aligned groups of five, in-degrees of six and more (the record's overflow paths)
    blocks.append([np.where(rng.random(80) < 0.12, rng.integers(0, 5, 80), base).astype(np.uint8)[rng.random(80) > 0.03] for _ in range(40)])
    m, n, g, e, q, c = PARAM_SETS["convex_default"]
    prm = [S.Params(m, n, g, e, q, c, mode | (0x10 if b != 1 else 0), 2 if (b == 3 and mode == 0) else 0) for b in range(len(blocks))]
    res = engine.run_blocks(blocks, prm, want_consensus=True, want_msa=True)
    for b, seqs in enumerate(blocks):
        op = oracle.mkparams(m, n, g, e, q, c, mode=mode | (0x10 if b != 1 else 0), banded=2 if (b == 3 and mode == 0) else 0)
        gg, sc, cells = oracle.block_run(seqs, None, op)
        assert_block_equal(res[b], gg, sc, cells, f"spoa order, block {b}")
        assert (res[b].consensus == gg.consensus()).all() and res[b].msa == gg.msa(True)


def test_per_block_device_time_is_reported(engine):
    """SURVEY section 5 / the reference's POA_DEBUG table (src/smooth.cpp:2121-2265): every block comes back with the
    shader-clock cycles its slot spent on it.  A block of 12 x 1 kbp costs more than one of 3 x 300 bp, an empty block nothing
    worth mentioning, and the sum over a batch cannot exceed slots x kernel time."""
    from smoothxg_amd import synth
    big = synth.make_block(1, 12, 1000)
    small = synth.make_block(2, 3, 300)
    res = engine.run_blocks([big, small, [], big], gparams("convex_default", 0))
    st = engine.stats()
    cyc = [r.device_cycles for r in res]
    assert all(c is not None for c in cyc) and cyc[0] > 4 * cyc[1] > 0 and cyc[2] < cyc[1]
    assert abs(cyc[0] - cyc[3]) < 0.5 * cyc[0]
    ms = [c / (st["dom_clock_mhz"] * 1e3) for c in cyc]
    assert 0 < max(ms) <= st["kernel_ms"] * 1.05


def test_global_alignment_beyond_the_32_bit_sweeps_reach_runs_packed(engine, oracle):
    """13 kbp, global, affine 1,4,6,2: longer than the 32-bit sweeps reach (12 287 letters) and with an all-gap corner near
    -52 000 -- rounds 1-4 answered TOO_LONG.  The clamped packed sweep aligns it (m L < 14 500) and equals the oracle."""
    rng = np.random.default_rng(131)
    seqs = random_block(rng, 3, 13000, div=0.02)
    g, sc, cells = oracle.block_run(seqs, None, oparams("affine_4param", 1))
    res = engine.run_blocks([seqs], gparams("affine_4param", 1))
    assert engine.stats()["dom_row_mode"] == 2
    assert_block_equal(res[0], g, sc, cells, label="global-13k")


def test_default_score_classes_equal_the_generic_ones(engine, oracle, monkeypatch):
    """Blocks with smoothxg's default scores run kernel classes compiled FOR those scores (immediates instead of scalar
    registers); SXG_POA_NO_DEFAULT_CLASS sends the same blocks through the generic classes.  One-, two- and four-wave
    geometries, local and global: both equal the oracle."""
    rng = np.random.default_rng(606)
    blocks = [random_block(rng, 5, L, div=0.03) for L in (700, 1200, 3000)]
    for mode in (0, 1):
        want = [oracle.block_run(b, None, oparams("convex_default", mode)) for b in blocks]
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("SXG_POA_NO_DEFAULT_CLASS", env)
            else:
                monkeypatch.delenv("SXG_POA_NO_DEFAULT_CLASS", raising=False)
            res = engine.run_blocks(blocks, gparams("convex_default", mode))
            for r, (g, sc, cells) in zip(res, want):
                assert_block_equal(r, g, sc, cells, label=f"default-class env={env} mode={mode}")
    monkeypatch.delenv("SXG_POA_NO_DEFAULT_CLASS", raising=False)


def test_end_cell_keys_across_row_epochs_and_ties(engine, oracle):
    """The packed sweep finds the end cell of a local alignment with one key per strip -- (greatest H) << 16 | 0xffff - (row &
    0xffff) -- and folds the keys at every 65 536th row.  Four UNRELATED sequences of 18 kbp: the graph passes 65 536 rows for the
    fourth (the epoch fold runs), the best local score is small and reached in many cells (greatest score, then smallest row, then
    smallest column must pick the oracle's cell), and every alignment is short (most bases become new nodes)."""
    rng = np.random.default_rng(65536)
    seqs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (18000, 17900, 17800, 17700, 600)]
    g, sc, cells = oracle.block_run(seqs, None, oparams("convex_default", 0))
    assert len(g.nodes()[0]) > 66000
    res = engine.run_blocks([seqs], gparams("convex_default", 0))
    assert engine.stats()["dom_row_mode"] == 2
    assert_block_equal(res[0], g, sc, cells, label="epochs-and-ties")
