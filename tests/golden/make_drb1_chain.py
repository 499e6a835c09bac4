#!/usr/bin/env python3
"""Golden fixture of the reference's own test configuration as a CHAIN: CMakeLists.txt:565 runs smoothxg on
test/data/DRB1-3123.seqwish.gfa with -l 700,900,1100 -j 5k -e 5k -r 12, i.e. three iterations, each on the GFA the one
before wrote (src/main.cpp:374-1065), consensus paths only in the last (src/main.cpp:404).  This script runs the chain
through the ORACLE stack (oracle/smooth_oracle.py + oracle/poa_oracle.c: self-oracle, not reference-derived -- the
reference binary cannot be built here) and records size and SHA-256 of every iteration's GFA, once per node order:
"iterations" = spoa's order (decree S7', the default since round 6: sxg_smooth_params::poa_spoa_order = 1),
"iterations_s7" = the incrementally kept order of rounds 1-5 (poa_spoa_order = 0).  ~3 minutes per order.
    python tests/golden/make_drb1_chain.py   ->  tests/golden/drb1_chain.json"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import smooth_oracle as SO  # noqa: E402

out = {"input": "DRB1-3123.seqwish.gfa", "flags": "-l 700,900,1100 -j 5000 -e 5000 -r 12 (consensus paths in the last iteration)",
       "made_by": "tests/golden/make_drb1_chain.py (oracle stack)",
       "orders": {"iterations": "spoa (S7'): poa_spoa_order = 1, the default", "iterations_s7": "incremental (S7): poa_spoa_order = 0"}}
for key, spoa in (("iterations", True), ("iterations_s7", False)):
    text = open(os.path.join(HERE, "DRB1-3123.seqwish.gfa")).read()
    out[key] = []
    for it, tl in enumerate((700, 900, 1100)):
        g = SO.Graph(text)
        blocks = SO.break_blocks(g, SO.smoothable_blocks(g, tl * 12, tl, 5000, 5000), 2 * tl)
        text = SO.smooth(g, blocks, add_consensus=(it == 2), spoa_order=spoa)
        o = SO.Graph(text)
        out[key].append({"target_poa_length": tl, "blocks": len(blocks), "gfa_bytes": len(text), "nodes": len(o.seq),
                         "paths": len(o.pname), "sha256": hashlib.sha256(text.encode()).hexdigest()})
        print(key, out[key][-1], flush=True)
json.dump(out, open(os.path.join(HERE, "drb1_chain.json"), "w"), indent=1)
