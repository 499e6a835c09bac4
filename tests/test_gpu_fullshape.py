"""GPU parity at BASELINE's full shapes: every sequence of whole blocks against committed oracle output.

tests/golden/fullshape_oracle.json (made by tests/golden/make_fullshape.py in the build container)
holds, per block, the oracle's scores and SHA-256 digests of node codes / ranks / groups / edges /
weights / all sequence paths / consensus.  The HIP path runs the same seeded blocks through the C ABI
and must reproduce every one of them: north-star headline local + global (64 x 5 kbp, convex), config 3
(64 x 5 kbp, affine 1,4,8,2), config 2 (16 x 1 kbp), config 4's extremes (128 x 10 kbp, 8 x 0.5 kbp), and the headline shape in
GLOBAL mode with the four-parameter affine scores 1,4,6,2, whose all-gap corner leaves int16 (the packed sweep's clamped form).
Round 6: the same shapes in spoa's node order (decree S7', the default) and config 3's blocks on the banded (`-A`) path.
"""
import json
import os

import numpy as np
import pytest

from smoothxg_amd import Params, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with open(os.path.join(HERE, "golden", "fullshape_oracle.json")) as f:
        return json.load(f)["cases"]


def _digests(r):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_fullshape import digest_block
    return digest_block(r.node_code, r.node_rank, r.node_group, r.edge_tail, r.edge_head, r.edge_weight, r.paths,
                        r.consensus)


# (order: "spoa" = decree S7', the engine's and the bench's default since round 6; "s7" = the incrementally kept order of rounds 1-5.
#  c3b / c3a: config 3's blocks on the `-A` path -- static band / abPOA's adaptive band -- which keeps the order s7.)
@pytest.mark.parametrize("group,order", [("ns_sw", "spoa"), ("ns_nw", "spoa"), ("ns_nw_affine", "spoa"), ("c3", "spoa"), ("c2+c4_min", "spoa"),
                                         ("ns_sw", "s7"), ("ns_nw", "s7"), ("ns_nw_affine", "s7"), ("c3", "s7"), ("c2+c4_min", "s7"), ("c4_max", "s7"),
                                         ("c3b+c3a", "s7")])
def test_full_shape_blocks_match_committed_oracle_output(engine, group, order):
    names = group.split("+")
    cases = [c for c in _cases() if c["name"] in names and c.get("order", "s7") == order]
    assert cases, "fixture has no case for " + group
    by_param = {}
    for c in cases:
        by_param.setdefault((tuple(c["params"]), c["mode"], c.get("banded", 0)), []).append(c)
    for (prm, mode, banded), cs in by_param.items():
        blocks = [synth.make_block(c["block_id"], c["n_seqs"], c["length"]) for c in cs]
        for c, seqs in zip(cs, blocks):
            assert [len(s) for s in seqs] == c["seq_lens"], "generator drifted from the fixture inputs"
        res = engine.run_blocks(blocks, Params(*prm, mode | (0x10 if order == "spoa" else 0), banded), want_consensus=True)
        for c, r in zip(cs, res):
            label = "%s block %d, order %s, banded %d" % (c["name"], c["block_id"], order, banded)
            assert r.status == 0, label
            assert r.scores.tolist() == c["scores"], label + ": scores of all %d sequences" % c["n_seqs"]
            assert int(r.cells.sum()) == c["cells"], label
            assert len(r.node_code) == c["n_nodes"] and len(r.edge_tail) == c["n_edges"], label
            got = _digests(r)
            for k, v in c["digests"].items():
                assert got[k] == v, "%s: %s differs from the oracle" % (label, k)
