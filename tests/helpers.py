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


def rerun_in_own_process(request):
    """Tests that create a real RCCL communicator run in a pytest process of their own: a long-lived process that has
    loaded librccl.so (rccl 2.27.7 of ROCm 7.2) and goes on allocating and freeing HIP memory afterwards aborts in the
    library teardown AFTER the interpreter has finalised ("double free or corruption" with every test passed: exit code
    134 for the whole session; a process that only runs the communicator test exits cleanly).  In the parent this starts
    the child on the same test and returns True (the caller returns); in the child it returns False and the body runs.
    The child's verdict is pytest's own summary line."""
    import os
    import subprocess
    import sys
    if os.environ.get("SXG_TEST_OWN_PROCESS"):
        return False
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    node = "%s::%s" % (str(request.node.fspath), request.node.name)
    r = subprocess.run([sys.executable, "-m", "pytest", node, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=root,
                       env=dict(os.environ, SXG_TEST_OWN_PROCESS="1"), capture_output=True, text=True, timeout=1200)
    if "1 skipped" in r.stdout and "passed" not in r.stdout:   # (no GPU / no RCCL in the child's environment: say so, do not fail)
        import pytest
        pytest.skip("the test skipped in its own process: " + r.stdout[-400:])
    # the child's return code counts: 0, or 134 (SIGABRT in rccl's teardown AFTER pytest printed its summary -- the reason this
    # helper exists); any other abort or crash behind the summary line is a failure
    assert r.returncode in (0, 134, -6), "child exited with %d\n" % r.returncode + r.stdout[-4000:] + r.stderr[-4000:]
    assert "1 passed" in r.stdout and "failed" not in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]
    return True
