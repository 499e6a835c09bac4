"""GPU parity at BASELINE's full shapes: every sequence of whole blocks against committed oracle output.

tests/golden/fullshape_oracle.json (made by tests/golden/make_fullshape.py in the build container)
holds, per block, the oracle's scores and SHA-256 digests of node codes / ranks / groups / edges /
weights / all sequence paths / consensus.  The HIP path runs the same seeded blocks through the C ABI
and must reproduce every one of them: north-star headline local + global (64 x 5 kbp, convex), config 3
(64 x 5 kbp, affine 1,4,8,2), config 2 (16 x 1 kbp), config 4's extremes (128 x 10 kbp, 8 x 0.5 kbp), and the headline shape in
GLOBAL mode with the four-parameter affine scores 1,4,6,2, whose all-gap corner leaves int16 (the packed sweep's clamped form).
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


@pytest.mark.parametrize("group", ["ns_sw", "ns_nw", "ns_nw_affine", "c3", "c2+c4_min", "c4_max"])
def test_full_shape_blocks_match_committed_oracle_output(engine, group):
    names = group.split("+")
    cases = [c for c in _cases() if c["name"] in names]
    assert cases, "fixture has no case for " + group
    by_param = {}
    for c in cases:
        by_param.setdefault((tuple(c["params"]), c["mode"]), []).append(c)
    for (prm, mode), cs in by_param.items():
        blocks = [synth.make_block(c["block_id"], c["n_seqs"], c["length"]) for c in cs]
        for c, seqs in zip(cs, blocks):
            assert [len(s) for s in seqs] == c["seq_lens"], "generator drifted from the fixture inputs"
        res = engine.run_blocks(blocks, Params(*prm, mode, 0), want_consensus=True)
        for c, r in zip(cs, res):
            label = "%s block %d" % (c["name"], c["block_id"])
            assert r.status == 0, label
            assert r.scores.tolist() == c["scores"], label + ": scores of all %d sequences" % c["n_seqs"]
            assert int(r.cells.sum()) == c["cells"], label
            assert len(r.node_code) == c["n_nodes"] and len(r.edge_tail) == c["n_edges"], label
            got = _digests(r)
            for k, v in c["digests"].items():
                assert got[k] == v, "%s: %s differs from the oracle" % (label, k)
