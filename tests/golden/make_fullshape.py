"""Generates tests/golden/fullshape_oracle.json: oracle results for whole blocks at BASELINE's shapes.

SELF-ORACLE fixtures (oracle/poa_oracle.c, scalar C; NOT reference-derived -- see the header of
tests/golden/make_golden.py).  Every BASELINE.json config shape is covered with ALL sequences of
the block (the chain of src/smooth.cpp:760-769, not a prefix):

  ns_sw / ns_nw   north-star headline, 64 x 5 kbp, convex 1,4,6,2,26,1, local / global   2 blocks each
  c3              config 3, 64 x 5 kbp, affine 1,4,8,2 (abPOA o+k*e convention), local   2 blocks
  c4_max / c4_min config 4 extremes, 128 x 10 kbp and 8 x 0.5 kbp, convex, local         1 block each
  c2              config 2, 16 x 1 kbp, convex, local                                    2 blocks
  ns_nw_affine    headline shape, GLOBAL, affine 1,4,6,2 (corner score -20 006 < int16)  2 blocks

Round 6: every case names its node ORDER -- "spoa" (decree S7': the depth-first re-sort after every AddAlignment, what the engine
and bench.py run by default since round 6) or "s7" (the incrementally kept order, the default of rounds 1-5) -- and its band mode
(0 = full matrix, 1 = static band, 2 = abPOA's adaptive band: the `-A` path's c3b / c3a bench workloads, which keep the order s7:
spoa's sort is not abPOA's).  The spoa-order cases cover ns_sw, ns_nw, ns_nw_affine, c3, c2 and c4_min.

The blocks come from smoothxg_amd.synth.make_block (seeded), so the GPU test regenerates the inputs
and only scores + SHA-256 digests of the oracle's outputs are committed.  A 64 x 5 kbp block costs the
scalar oracle ~35 s, the 128 x 10 kbp block ~8 min; cases run in parallel processes.

    python tests/golden/make_fullshape.py          (from the repo root, ~10 min on 8 cores)
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CONVEX = (1, -4, -6, -2, -26, -1)
AFFINE_C3 = (1, -4, -8, -2, -8, -2)
AFFINE_4P = (1, -4, -6, -2, -6, -2)   # smoothxg's four-parameter form 1,4,6,2 (spoa: q = g, c = e; src/main.cpp:348-359)
# name, block id, sequences, length, scores, mode
CASES = [
    ("c4_max", 7001, 128, 10000, CONVEX, 0),
    ("ns_sw", 0, 64, 5000, CONVEX, 0),
    ("ns_sw", 999, 64, 5000, CONVEX, 0),
    ("ns_nw", 0, 64, 5000, CONVEX, 1),
    ("ns_nw", 999, 64, 5000, CONVEX, 1),
    ("c3", 0, 64, 5000, AFFINE_C3, 0),
    ("c3", 4999, 64, 5000, AFFINE_C3, 0),
    ("c2", 0, 16, 1000, CONVEX, 0),
    ("c2", 999, 16, 1000, CONVEX, 0),
    ("c4_min", 7002, 8, 500, CONVEX, 0),
    # global alignment whose all-gap corner (-20 006) leaves int16: the packed sweep's clamped form (round 5)
    ("ns_nw_affine", 0, 64, 5000, AFFINE_4P, 1),
    ("ns_nw_affine", 999, 64, 5000, AFFINE_4P, 1),
    # round 6 -- spoa's order (S7'): name, block, sequences, length, scores, mode, order, banded
    ("ns_sw", 0, 64, 5000, CONVEX, 0, "spoa", 0),
    ("ns_sw", 999, 64, 5000, CONVEX, 0, "spoa", 0),
    ("ns_nw", 0, 64, 5000, CONVEX, 1, "spoa", 0),
    ("ns_nw", 999, 64, 5000, CONVEX, 1, "spoa", 0),
    ("ns_nw_affine", 0, 64, 5000, AFFINE_4P, 1, "spoa", 0),
    ("ns_nw_affine", 999, 64, 5000, AFFINE_4P, 1, "spoa", 0),
    ("c3", 0, 64, 5000, AFFINE_C3, 0, "spoa", 0),
    ("c3", 4999, 64, 5000, AFFINE_C3, 0, "spoa", 0),
    ("c2", 0, 16, 1000, CONVEX, 0, "spoa", 0),
    ("c2", 999, 16, 1000, CONVEX, 0, "spoa", 0),
    ("c4_min", 7002, 8, 500, CONVEX, 0, "spoa", 0),
    # round 6 -- the `-A` path: config 3's blocks under the static band (c3b) and abPOA's adaptive band (c3a)
    ("c3b", 0, 64, 5000, AFFINE_C3, 0, "s7", 1),
    ("c3b", 4999, 64, 5000, AFFINE_C3, 0, "s7", 1),
    ("c3a", 0, 64, 5000, AFFINE_C3, 0, "s7", 2),
    ("c3a", 4999, 64, 5000, AFFINE_C3, 0, "s7", 2),
]


def case_key(c):
    """(name, block, order, banded) of a CASES tuple or of a fixture entry (entries of rounds 1-5 carry neither: s7, full matrix)"""
    if isinstance(c, dict):
        return (c["name"], c["block_id"], c.get("order", "s7"), c.get("banded", 0))
    return (c[0], c[1], c[6] if len(c) > 6 else "s7", c[7] if len(c) > 7 else 0)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digest_block(code, rank, group, tail, head, weight, paths, consensus):
    """The digests both sides compute: oracle here, the HIP path in tests/test_gpu_fullshape.py."""
    return {
        "node_code": sha(np.asarray(code, np.uint8)), "node_rank": sha(np.asarray(rank, np.int32)),
        "node_group": sha(np.asarray(group, np.int32)), "edge_tail": sha(np.asarray(tail, np.int32)),
        "edge_head": sha(np.asarray(head, np.int32)), "edge_weight": sha(np.asarray(weight, np.uint32)),
        "paths": sha(np.concatenate([np.asarray(p, np.int32) for p in paths])),
        "consensus": sha(np.asarray(consensus, np.int32)),
    }


def run_case(case):
    from oracle import oracle_py as O
    from smoothxg_amd import synth
    name, bid, ns, ln, prm, mode = case[:6]
    _, _, order, banded = case_key(case)
    t0 = time.time()
    seqs = synth.make_block(bid, ns, ln)
    g, sc, cells = O.block_run(seqs, None, O.mkparams(*prm, mode=mode | (0x10 if order == "spoa" else 0), banded=banded))
    code, rank, grp = g.nodes()
    t, h, w = g.edges()
    out = {"name": name, "block_id": bid, "n_seqs": ns, "length": ln, "params": list(prm), "mode": mode, "order": order, "banded": banded,
           "seq_lens": [int(len(s)) for s in seqs], "scores": [int(x) for x in sc],
           "cells": int(cells.sum()), "n_nodes": int(g.n_nodes), "n_edges": int(g.n_edges),
           "digests": digest_block(code, rank, grp, t, h, w, [g.seq_path(s) for s in range(g.n_seqs)],
                                   g.consensus()),
           "oracle_seconds": round(time.time() - t0, 1)}
    print("done", name, bid, out["oracle_seconds"], "s", flush=True)
    return out


def main():
    from oracle import oracle_py as O
    O.build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullshape_oracle.json")
    cases, keep = CASES, []
    if len(sys.argv) > 1 and sys.argv[1] == "--missing" and os.path.exists(path):   # only the cases the file does not hold yet
        keep = json.load(open(path))["cases"]
        have = {case_key(c) for c in keep}
        cases = [c for c in CASES if case_key(c) not in have]
    with mp.Pool(min(max(len(cases), 1), max(1, (os.cpu_count() or 2) - 1))) as pool:
        res = keep + pool.map(run_case, cases, chunksize=1)
    out = {"provenance": "self-oracle (oracle/poa_oracle.c, scalar), generated by tests/golden/make_fullshape.py in the "
                         "build container; NOT reference-derived (spoa/abPOA/odgi absent from /root/reference)",
           "cases": res}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullshape_oracle.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(res), "cases")


if __name__ == "__main__":
    main()
