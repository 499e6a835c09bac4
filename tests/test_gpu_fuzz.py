"""MI355X: randomized parity -- every block of a batch brings its own random score set, alignment mode, node order and band flag
(the engine's per_block_params), lengths from 1 to 3 kbp, 1-20 sequences, N letters, weights.  HIP == oracle, bit for bit."""
import os

import numpy as np
import pytest

from helpers import assert_block_equal, random_block
from oracle import oracle_py as O
from smoothxg_amd import Params

pytestmark = pytest.mark.gpu


def random_scores(rng):
    """A valid score set in spoa's sign convention, covering the three gap models (S1) and the byte limits of the packed
    rows (|g|, |q| <= 120)."""
    m = int(rng.integers(1, 6))
    n = -int(rng.integers(1, 12))
    kind = int(rng.integers(0, 3))
    if kind == 0:       # linear
        g = -int(rng.integers(1, 12)); e = g; q = g; c = g
    elif kind == 1:     # affine
        e = -int(rng.integers(1, 6)); g = e - int(rng.integers(1, 30)); q = g; c = e
    else:               # convex: g > q (cheaper to open), e < c (dearer to extend)
        c = -int(rng.integers(1, 4)); e = c - int(rng.integers(1, 5))
        g = e - int(rng.integers(1, 12)); q = g - int(rng.integers(1, 60))
    return m, n, g, e, q, c


# SXG_FUZZ_SEEDS=<n>: a longer campaign (the default twelve seeds cost ~40 s on the GPU box)
@pytest.mark.parametrize("seed", [9001 + k for k in range(int(os.environ.get("SXG_FUZZ_SEEDS", "12")))])
def test_random_blocks_random_scores_per_block(engine, seed):
    rng = np.random.default_rng(seed)
    blocks, weights, gp, op = [], [], [], []
    for trial in range(36):
        L = int(rng.choice([1, 3, 17, 60, 200, 500, 900, 1400, 2100, 3000]))
        L = max(1, L + int(rng.integers(-L // 4, L // 4 + 1)))
        S = int(rng.integers(1, 21 if L < 1000 else 7))
        seqs = random_block(rng, S, L, div=float(rng.choice([0.01, 0.05, 0.15])), alphabet=5 if trial % 4 == 0 else 4)
        if trial % 6 == 0 and L > 40:   # a structural variant: one sequence loses a piece, one gains one
            seqs.append(np.concatenate([seqs[0][:L // 3], seqs[0][L // 3 + L // 8:]]))
            seqs.append(np.concatenate([seqs[0][:L // 2], rng.integers(0, 4, L // 10 + 1, dtype=np.uint8), seqs[0][L // 2:]]))
        m, n, g, e, q, c = random_scores(rng)
        mode = int(rng.integers(0, 2))
        banded = int(rng.integers(0, 3)) if mode == 0 else 0
        if mode == 1 and L <= 300 and trial % 2 == 1:
            banded = 2   # abPOA's band in global mode (round 4); short enough for every score set to stay in the packed range
        # round 6: the node order is drawn per block as well -- spoa's depth-first re-sort after every sequence (0x10, decree S7': the
        # parallel, incremental re-sort of the device against the oracle's one sequential walk) for two blocks out of three
        mode |= 0x10 if int(rng.integers(0, 3)) else 0
        blocks.append(seqs)
        weights.append(rng.integers(1, 6, len(seqs)).astype(np.uint32))
        gp.append(Params(m, n, g, e, q, c, mode, banded))
        op.append(O.mkparams(m, n, g, e, q, c, mode=mode, banded=banded))
    res = engine.run_blocks(blocks, gp, weights=weights, want_consensus=True)
    for b, (seqs, w) in enumerate(zip(blocks, weights)):
        g_, sc, cells = O.block_run(seqs, w, op[b])
        p = gp[b]
        label = f"fuzz{seed}/block{b} L={len(seqs[0])} S={len(seqs)} scores=({p.m},{p.n},{p.g},{p.e},{p.q},{p.c}) mode={p.mode} banded={p.banded}"
        assert_block_equal(res[b], g_, sc, cells, label=label)
        assert (res[b].consensus == g_.consensus()).all(), label


# SXG_FUZZ_LONG=<n>: more seeds of the long-block campaign (default three)
@pytest.mark.parametrize("seed", [9501 + k for k in range(int(os.environ.get("SXG_FUZZ_LONG", "3")))])
def test_random_long_banded_blocks(engine, seed):
    """Blocks of 3-9 kbp (strip widths 8 and 11, a window that slides and re-centres) in the two banded modes, with
    structural variants of 100-900 bases that push the alignment to and beyond the edge of the band: HIP == oracle."""
    rng = np.random.default_rng(seed)
    blocks, gp, op = [], [], []
    for trial in range(6):
        L = int(rng.integers(3000, 9000))
        seqs = random_block(rng, int(rng.integers(3, 7)), L, div=float(rng.choice([0.01, 0.04])))
        sv = int(rng.integers(100, 900))
        at = int(rng.integers(L // 5, 4 * L // 5))
        seqs.append(np.concatenate([seqs[0][:at], rng.integers(0, 4, sv, dtype=np.uint8), seqs[0][at:]]))
        seqs.append(np.concatenate([seqs[0][:at], seqs[0][min(L, at + sv):]]))
        if trial % 2:
            seqs.append(seqs[-2].copy())
        m, n, g, e, q, c = random_scores(rng)
        # (the banded kernel is the packed one: a local alignment with m * L >= 30 000 leaves int16 and runs the full matrix
        #  on the 32-bit sweep, band flag ignored -- documented in INTEGRATION.md; keep the campaign inside the band's domain)
        m = min(m, 29999 // max(len(x) for x in seqs))
        banded = 1 + trial % 2
        blocks.append(seqs)
        omode = 0x10 if trial % 3 == 0 else 0   # (round 6: spoa's order on the banded kernel as well)
        gp.append(Params(m, n, g, e, q, c, omode, banded))
        op.append(O.mkparams(m, n, g, e, q, c, mode=omode, banded=banded))
    res = engine.run_blocks(blocks, gp, want_consensus=True)
    for b, seqs in enumerate(blocks):
        g_, sc, cells = O.block_run(seqs, None, op[b])
        p = gp[b]
        label = f"long{seed}/block{b} L={len(seqs[0])} S={len(seqs)} scores=({p.m},{p.n},{p.g},{p.e},{p.q},{p.c}) banded={p.banded}"
        assert_block_equal(res[b], g_, sc, cells, label=label)
        assert (res[b].consensus == g_.consensus()).all(), label
