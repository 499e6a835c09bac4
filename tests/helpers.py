"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from oracle import oracle_py as O
from smoothxg_amd import Params

PARAM_SETS = {
    # name: (m, n, g, e, q, c) in spoa's sign convention
    "convex_default": (1, -4, -6, -2, -26, -1),   # smoothxg defaults, src/main.cpp:322-327
    "affine_4param": (1, -4, -6, -2, -6, -2),     # 4-parameter form: q=g, c=e (src/main.cpp:348-353)
    "linear": (2, -3, -5, -5, -5, -5),
    "convex_heavy": (3, -5, -9, -3, -30, -1),
    "adaptive_tier": (1, -19, -39, -3, -81, -1),  # src/smooth.cpp:2028-2062 style tier
}


def oparams(name, mode):
    return O.mkparams(*PARAM_SETS[name], mode=mode)


def gparams(name, mode):
    m, n, g, e, q, c = PARAM_SETS[name]
    return Params(m, n, g, e, q, c, mode, 0)


def random_block(rng, n_seqs, length, div=0.05, alphabet=4):
    """A block of related sequences (random ancestor + independent edits), longest first."""
    anc = rng.integers(0, alphabet, length, dtype=np.uint8)
    seqs = []
    for _ in range(n_seqs):
        s = []
        for b in anc:
            r = rng.random()
            if r < div / 3:
                continue
            if r < 2 * div / 3:
                s.append(int(rng.integers(0, alphabet)))
            if r < div:
                s.append(int((b + 1 + rng.integers(0, 3)) % alphabet))
            else:
                s.append(int(b))
        if not s:
            s = [int(anc[0])]
        seqs.append(np.asarray(s, np.uint8))
    seqs.sort(key=lambda x: -len(x))
    return seqs


def assert_block_equal(res, g, scores, cells, label=""):
    """res: smoothxg_amd BlockResult; g: oracle Graph."""
    code, rank, grp = g.nodes()
    t, h, w = g.edges()
    assert res.status == 0, f"{label}: status {res.status}"
    assert len(res.node_code) == len(code), f"{label}: node count {len(res.node_code)} != {len(code)}"
    assert (res.scores == scores).all(), f"{label}: scores {res.scores} != {scores}"
    assert (res.cells == cells).all(), f"{label}: cells differ"
    assert (res.node_code == code).all(), f"{label}: node codes differ"
    assert (res.node_rank == rank).all(), f"{label}: ranks differ"
    assert (res.node_group == grp).all(), f"{label}: groups differ"
    assert len(res.edge_tail) == len(t), f"{label}: edge count"
    assert (res.edge_tail == t).all() and (res.edge_head == h).all(), f"{label}: edges differ"
    assert (res.edge_weight == w).all(), f"{label}: edge weights differ"
    for s in range(g.n_seqs):
        assert (res.paths[s] == g.seq_path(s)).all(), f"{label}: path {s} differs"


def heaviest_bundle_independent(code, rank, edge_tail, edge_head, edge_weight):
    """Lee 2003 heaviest bundle with branch completion, written independently of oracle/poa_oracle.c and of the device
    code (which share their text): rank-space CSR of in-edges, the best in-edge of a node is the maximum of the key
    (weight, score of the tail, position in the node's in-list) over the tails that are still alive, scores are
    accumulated in rank order, and a consensus that ends inside the graph is completed by killing the rival tails
    of its successors and re-running the relaxation behind it.  Returns node ids in path order."""
    n = len(code)
    if n == 0:
        return np.zeros(0, np.int32)
    order = np.argsort(rank)
    ins = [[] for _ in range(n)]          # per node: (weight, tail) in edge-insertion order
    outs = [[] for _ in range(n)]
    for t, h, w in zip(edge_tail.tolist(), edge_head.tolist(), edge_weight.tolist()):
        ins[h].append((w, t))
        outs[t].append(h)
    score = [-1] * n
    back = [-1] * n
    dead = [False] * n

    def relax(v):
        cands = [(w, score[t], k, t) for k, (w, t) in enumerate(ins[v]) if not dead[t]]
        if not cands:
            score[v], back[v] = -1, -1
            return
        w, s, _, t = max(cands)
        score[v], back[v] = w + s, t

    best = None
    for v in order.tolist():
        cands = [(w, score[t], k, t) for k, (w, t) in enumerate(ins[v])]
        if cands:
            w, s, _, t = max(cands)
            score[v], back[v] = w + s, t
        else:
            score[v], back[v] = -1, -1
        if best is None or score[v] > score[best]:
            best = v
    while outs[best]:
        start = best
        for h in outs[start]:
            for _, t in ins[h]:
                if t != start:
                    dead[t] = True
                    score[t] = -1
        nxt, top = None, 0
        for v in order[rank[start] + 1:].tolist():
            dead[v] = False
            relax(v)
            if score[v] == -1:
                dead[v] = True
            if score[v] > top:
                top, nxt = score[v], v
        if nxt is None:
            break
        best = nxt
    path = []
    v = best
    while v != -1:
        path.append(v)
        v = back[v]
    return np.asarray(path[::-1], np.int32)
