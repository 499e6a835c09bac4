"""GPU parity of the BANDED sweep (A11: the reference's abPOA path, src/smooth.cpp:133-627, band wb=311 / wf=0.03):
the one-wave sliding-window kernel of poa_band16.hip.h against the oracle's banded mode (decrees B1-B3 of
oracle/poa_oracle.c) -- scores, graphs, paths, consensus, and the number of band cells."""
import numpy as np
import pytest

from helpers import PARAM_SETS, assert_block_equal, random_block
from oracle import oracle_py as O
from smoothxg_amd import Params, synth

pytestmark = pytest.mark.gpu


def _p(pname, banded=1):
    m, n, g, e, q, c = PARAM_SETS[pname]
    return Params(m, n, g, e, q, c, 0, banded), O.mkparams(m, n, g, e, q, c, mode=0, banded=banded)


@pytest.mark.parametrize("pname", ["convex_default", "affine_4param", "linear"])
def test_banded_blocks_match_banded_oracle(engine, pname):
    """Blocks of many shapes in one batch: short ones (the band covers the whole matrix), long ones (the window
    slides), divergent ones (deep bubbles, >= 3 predecessors), structural variants that push the alignment towards
    the edge of the band."""
    rng = np.random.default_rng(61)
    blocks = []
    for L in (40, 300, 900, 1500, 2600, 4000):
        blocks.append(random_block(rng, int(rng.integers(3, 9)), L, div=0.04))
    blocks.append(random_block(rng, 24, 400, div=0.12))
    anc = rng.integers(0, 4, 3000, dtype=np.uint8)
    ins = rng.integers(0, 4, 280, dtype=np.uint8)
    blocks.append([np.concatenate([anc[:1200], ins, anc[1200:]]), anc.copy(), np.concatenate([anc[:700], anc[950:]]),
                   np.concatenate([anc[:2000], ins[:150], anc[2000:]])])
    gp, op = _p(pname)
    res = engine.run_blocks(blocks, gp, want_consensus=True)
    st = engine.stats()
    assert st["dom_row_mode"] == 3 and st["dom_threads"] == 64
    for b, seqs in enumerate(blocks):
        g, sc, cells = O.block_run(seqs, None, op)
        assert_block_equal(res[b], g, sc, cells, label=f"banded/{pname}/block{b}")
        assert (res[b].consensus == g.consensus()).all()
        full = sum(len(s) for s in seqs[1:])   # the band is narrower than the matrix for the long blocks
        if len(seqs[0]) > 2000:
            assert int(res[b].cells.sum()) < 0.8 * int(O.block_run(seqs, None, _p(pname, 0)[1])[2].sum()), full


def test_band_changes_results_only_where_it_must(engine):
    """A block whose alignments stay near the diagonal gives the same graph banded and unbanded; a 700-base insertion
    (beyond w = 311 + 0.03 L) does not -- and both agree with their oracles."""
    rng = np.random.default_rng(62)
    calm = random_block(rng, 5, 2500, div=0.02)
    anc = rng.integers(0, 4, 2500, dtype=np.uint8)
    wild = [np.concatenate([anc[:1000], rng.integers(0, 4, 700, dtype=np.uint8), anc[1000:]]), anc.copy(), anc.copy()]
    for seqs, same in ((calm, True), (wild, False)):
        rb = engine.run_blocks([seqs], _p("convex_default", 1)[0])[0]
        ru = engine.run_blocks([seqs], _p("convex_default", 0)[0])[0]
        gb, sb, cb = O.block_run(seqs, None, _p("convex_default", 1)[1])
        gu, su, cu = O.block_run(seqs, None, _p("convex_default", 0)[1])
        assert_block_equal(rb, gb, sb, cb, label="banded")
        assert_block_equal(ru, gu, su, cu, label="unbanded")
        assert (rb.scores == ru.scores).all() == same


def test_config3_shape_banded_full_block(engine):
    """BASELINE config 3 as the reference's -A path runs it: 64 x 5 kbp, affine 1,4,8,2 in abPOA's convention, BANDED.
    Whole block, every sequence, against the banded oracle (~15 s on the host)."""
    seqs = synth.make_block(0, 64, 5000)
    gp, op = Params(1, -4, -8, -2, -8, -2, 0, 1), O.mkparams(1, -4, -8, -2, -8, -2, mode=0, banded=1)
    res = engine.run_blocks([seqs], gp, want_consensus=True)[0]
    g, sc, cells = O.block_run(seqs, None, op)
    assert_block_equal(res, g, sc, cells, label="c3-banded")
    assert (res.consensus == g.consensus()).all()
    w = 311 + int(0.03 * 5000)
    assert int(cells.sum()) < 0.25 * sum(len(s) for s in seqs[1:]) * g.n_nodes   # ~(2w+11)/L of the matrix
    assert w == 461


def test_global_alignment_static_band_flag_runs_the_full_matrix(engine):
    """The STATIC band is for local mode only (its band need not hold the end column of a global alignment): global +
    banded = 1 runs the full matrix, in the oracle and on the device."""
    rng = np.random.default_rng(63)
    seqs = random_block(rng, 4, 800, div=0.03)
    m, n, g, e, q, c = PARAM_SETS["convex_default"]
    res = engine.run_blocks([seqs], Params(m, n, g, e, q, c, 1, 1))[0]
    gg, sc, cells = O.block_run(seqs, None, O.mkparams(m, n, g, e, q, c, mode=1, banded=1))
    assert_block_equal(res, gg, sc, cells, label="global+static band flag")
    assert engine.stats()["dom_row_mode"] != 3


@pytest.mark.parametrize("pname", ["convex_default", "affine_4param", "linear"])
def test_global_alignment_with_the_adaptive_band(engine, pname):
    """smooth_abpoa sets abPOA's band for BOTH alignment modes (src/smooth.cpp:259-271): global alignment (-Z) with the
    adaptive band (banded = 2, decree B4) on the one-wave kernel -- nothing clamped at 0, the virtual row's gap costs,
    end cell = column L of a sink row -- against the oracle: scores, graphs, paths, consensus, band cells.  Blocks of
    several shapes incl. structural variants and deep bubbles; all inside the packed range (a global alignment whose
    scores leave int16 runs the full matrix, see the last assertion)."""
    rng = np.random.default_rng(64)
    # (linear gaps cost 5 per base here: the int16 range ends near 1.4 kbp)
    lens, alen = ((40, 300, 900, 1500, 2400), 2200) if pname != "linear" else ((40, 300, 700, 1100), 1100)
    blocks = [random_block(rng, int(rng.integers(3, 9)), L, div=0.04) for L in lens]
    blocks.append(random_block(rng, 20, 400, div=0.12))
    anc = rng.integers(0, 4, alen, dtype=np.uint8)
    ins = rng.integers(0, 4, 280, dtype=np.uint8)
    a3 = alen // 3
    blocks.append([np.concatenate([anc[:a3], ins, anc[a3:]]), anc.copy(), np.concatenate([anc[:a3 - 200], anc[a3 + 50:]]),
                   np.concatenate([anc[:2 * a3], ins[:150], anc[2 * a3:]])])
    if pname != "linear":
        big_ins = rng.integers(0, 4, 600, dtype=np.uint8)   # an insertion beyond the band's half-width, shared by two sequences
        blocks.append([np.concatenate([anc[:900], big_ins, anc[900:]]), anc.copy(), np.concatenate([anc[:900], big_ins, anc[900:]]), anc.copy()])
    m, n, g, e, q, c = PARAM_SETS[pname]
    gp, op = Params(m, n, g, e, q, c, 1, 2), O.mkparams(m, n, g, e, q, c, mode=1, banded=2)
    res = engine.run_blocks(blocks, gp, want_consensus=True)
    st = engine.stats()
    assert st["dom_row_mode"] == 3 and st["dom_threads"] == 64
    for b, seqs in enumerate(blocks):
        gg, sc, cells = O.block_run(seqs, None, op)
        assert_block_equal(res[b], gg, sc, cells, label=f"global-adaptive/{pname}/block{b}")
        assert (res[b].consensus == gg.consensus()).all()
        if len(seqs[0]) > 2000:
            full = O.block_run(seqs, None, O.mkparams(m, n, g, e, q, c, mode=1, banded=0))
            assert int(res[b].cells.sum()) < 0.8 * int(full[2].sum())
    # beyond the packed range (affine global at 6 kbp): the block takes the widening ladder and runs the full matrix
    long_seqs = random_block(rng, 3, 6000, div=0.02)
    r = engine.run_blocks([long_seqs], Params(1, -4, -8, -2, -8, -2, 1, 2))[0]
    assert r.status == 0 and engine.stats()["dom_row_mode"] != 3
    gf, sf, cf = O.block_run(long_seqs, None, O.mkparams(1, -4, -8, -2, -8, -2, mode=1, banded=0))
    assert (r.scores == sf).all()


def test_banded_beyond_12_kbp(engine):
    """The window slides over any length the packed range allows: 15 and 24 kbp blocks in banded mode."""
    rng = np.random.default_rng(1500)
    blocks = [random_block(rng, 3, L, div=0.02) for L in (15000, 24000)]
    gp, op = _p("affine_4param")
    res = engine.run_blocks(blocks, gp)
    assert engine.stats()["dom_row_mode"] == 3
    for b, seqs in enumerate(blocks):
        g, sc, cells = O.block_run(seqs, None, op)
        assert_block_equal(res[b], g, sc, cells, label=f"banded-long{len(seqs[0])}")


def test_long_banded_block_with_an_indel_wider_than_the_window_margin(engine):
    """Beyond 12.7 kbp the half-width is capped at 693 (decree B1) so that the band fits the kernel's window.  A 900-bp
    insertion at 20 kbp pushes the best alignment ~900 columns off the backbone: outside the band in both the oracle
    and the kernel, which therefore still agree (before the cap the oracle's band was wider than the kernel's window)."""
    rng = np.random.default_rng(1501)
    anc = rng.integers(0, 4, 20000, dtype=np.uint8)
    ins = rng.integers(0, 4, 900, dtype=np.uint8)
    blocks = [[np.concatenate([anc[:6000], ins, anc[6000:]]), anc.copy(), np.concatenate([anc[:11000], anc[11800:]])]]
    gp, op = _p("affine_4param")
    res = engine.run_blocks(blocks, gp)
    assert engine.stats()["dom_row_mode"] == 3
    g, sc, cells = O.block_run(blocks[0], None, op)
    assert_block_equal(res[0], g, sc, cells, label="banded-long-indel")


# ---- ADAPTIVE band (params.banded = 2, decree B4: abPOA's rule) ------------------------------------------------

@pytest.mark.parametrize("pname", ["convex_default", "affine_4param", "linear"])
def test_adaptive_band_blocks_match_the_oracle(engine, pname):
    """The same batch of shapes as the static band, in the adaptive mode: the band of every row follows the best cells of
    its predecessor rows (found on the device while sweeping) and the node's distance to the end of the graph (pointer
    jumping over heaviest out-edges in the graph phase)."""
    rng = np.random.default_rng(71)
    blocks = []
    for L in (40, 300, 900, 1500, 2600, 4000):
        blocks.append(random_block(rng, int(rng.integers(3, 9)), L, div=0.04))
    blocks.append(random_block(rng, 24, 400, div=0.12))
    anc = rng.integers(0, 4, 3000, dtype=np.uint8)
    ins = rng.integers(0, 4, 280, dtype=np.uint8)
    blocks.append([np.concatenate([anc[:1200], ins, anc[1200:]]), anc.copy(), np.concatenate([anc[:700], anc[950:]]),
                   np.concatenate([anc[:2000], ins[:150], anc[2000:]])])
    blocks.append([rng.integers(0, 4, 5, dtype=np.uint8), rng.integers(0, 4, 3, dtype=np.uint8)])   # tiny: one strip
    gp, op = _p(pname, 2)
    res = engine.run_blocks(blocks, gp, want_consensus=True)
    st = engine.stats()
    assert st["dom_row_mode"] == 3 and st["dom_threads"] == 64
    for b, seqs in enumerate(blocks):
        g, sc, cells = O.block_run(seqs, None, op)
        assert_block_equal(res[b], g, sc, cells, label=f"adaptive/{pname}/block{b}")
        assert (res[b].consensus == g.consensus()).all()


def test_adaptive_band_follows_a_structural_variant_the_static_band_loses(engine):
    """Three sequences carry a 700-base insertion the first one lacks (beyond w = 311 + 0.03 L = 386).  The static band sits
    on the backbone coordinate and loses the alignment after the insertion; the adaptive band moves with the best cells
    of the rows above, keeps it, and returns the full matrix's scores -- abPOA's reason for the rule.  Each mode
    equals its own oracle."""
    rng = np.random.default_rng(72)
    anc = rng.integers(0, 4, 2500, dtype=np.uint8)
    ins = rng.integers(0, 4, 700, dtype=np.uint8)
    with_ins = np.concatenate([anc[:1000], ins, anc[1000:]])
    seqs = [with_ins, anc.copy(), with_ins.copy(), anc.copy()]
    seqs[2][[50, 1500, 3000]] ^= 1
    out = {}
    for banded in (0, 1, 2):
        gp, op = _p("convex_default", banded)
        r = engine.run_blocks([seqs], gp)[0]
        g, sc, cells = O.block_run(seqs, None, op)
        assert_block_equal(r, g, sc, cells, label=f"sv/banded={banded}")
        out[banded] = r
    assert (out[2].scores == out[0].scores).all()
    assert not (out[1].scores == out[0].scores).all()
    assert int(out[2].cells.sum()) < 0.6 * int(out[0].cells.sum())


def test_config3_shape_adaptive_band_full_block(engine):
    """BASELINE config 3 as the reference's -A path runs it, adaptive band: 64 x 5 kbp, affine, whole block."""
    seqs = synth.make_block(1, 64, 5000)
    gp, op = Params(1, -4, -8, -2, -8, -2, 0, 2), O.mkparams(1, -4, -8, -2, -8, -2, mode=0, banded=2)
    res = engine.run_blocks([seqs], gp, want_consensus=True)[0]
    g, sc, cells = O.block_run(seqs, None, op)
    assert_block_equal(res, g, sc, cells, label="c3-adaptive")
    assert (res.consensus == g.consensus()).all()
    assert int(cells.sum()) < 0.3 * sum(len(s) for s in seqs[1:]) * g.n_nodes


def test_adaptive_band_long_blocks_and_mixed_modes_in_one_batch(engine):
    """15 kbp (the window slides, strip width 11) next to short blocks; static, adaptive and unbanded blocks share a batch
    (per-block parameters), the two banded modes even a launch."""
    rng = np.random.default_rng(73)
    blocks = [random_block(rng, 3, 15000, div=0.02), random_block(rng, 6, 700, div=0.05), random_block(rng, 5, 1800, div=0.03),
              random_block(rng, 4, 2400, div=0.03)]
    m, n, g, e, q, c = PARAM_SETS["convex_default"]
    modes = [2, 1, 2, 0]
    gp = [Params(m, n, g, e, q, c, 0, b) for b in modes]
    res = engine.run_blocks(blocks, gp)
    for b, seqs in enumerate(blocks):
        gg, sc, cells = O.block_run(seqs, None, O.mkparams(m, n, g, e, q, c, mode=0, banded=modes[b]))
        assert_block_equal(res[b], gg, sc, cells, label=f"mixed-modes/block{b}/banded={modes[b]}")
