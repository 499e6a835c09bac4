"""Generates tests/golden/self_oracle_blocks.json from the CPU oracle.

SELF-ORACLE fixtures: the reference (pangenome/smoothxg) commits no expected output for the
POA path and its spoa/abPOA dependencies are absent, so these vectors pin the oracle against
itself (regression guard) -- they are NOT reference-derived.  Hand-made small cases plus
three generator blocks.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle_py as O  # noqa: E402
from smoothxg_amd import synth  # noqa: E402

CASES = [
    (["ACGTACGTACGTACGTACGT", "ACGTACGAACGTACGTACGT", "ACGTACGTACGTCGTACGT"], [1, 1, 1]),
    (["GATTACAGATTACAGATTACA", "GATTACAGATACAGATTACA", "GATTACAGATTACAGATTTACA", "GATTACAGATTACAGATTACA"], [1, 2, 1, 3]),
    (["AAAAAAAAAACCCCCCCCCC", "AAAAAAAAAAGGGGCCCCCCCCCC", "AAAAAAAAAACCCCCCCCCC"], [1, 1, 1]),
    (["ACGTN", "ACGTN", "ANGTN"], [1, 1, 1]),
]
PARAMS = [((1, -4, -6, -2, -26, -1), 0), ((1, -4, -6, -2, -26, -1), 1), ((1, -4, -6, -2, -6, -2), 0),
          ((2, -3, -5, -5, -5, -5), 1)]


def main():
    cases = []
    blocks = [(c, w) for c, w in CASES]
    for b in range(3):
        seqs = synth.make_block(b, 6, 120, sub=0.03, ins=0.01, dele=0.01)
        blocks.append(([synth.decode(s) for s in seqs], [1] * len(seqs)))
    for seqs, w in blocks:
        for prm, mode in PARAMS:
            enc = [synth.encode(s) for s in seqs]
            g, sc, cells = O.block_run(enc, w, O.mkparams(*prm, mode=mode))
            cases.append({"seqs": seqs, "weights": w, "params": list(prm), "mode": mode,
                          "scores": sc.tolist(), "n_nodes": g.n_nodes, "n_edges": g.n_edges,
                          "consensus": synth.decode(g.nodes()[0][g.consensus()]), "msa": g.msa(True)})
    out = {"provenance": "self-oracle (oracle/poa_oracle.c); NOT reference-derived: pangenome/smoothxg "
                         "holds no golden vectors for this path and spoa/abPOA/odgi are absent",
           "cases": cases}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "self_oracle_blocks.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
