"""Independent invariants of the GFAs this repository writes (test infrastructure).

The host rows are checked elsewhere by comparing the C++ with a Python restatement of the same decrees; two bugs of the
DECREES (a merged link left behind as a self loop, a "topological" order that ignored half of the edges) passed those
comparisons for three rounds because both sides shared them.  The checks below share no code with either side: they parse
the GFA text themselves and test properties that follow from what the reference does, whatever the tie-breaks:

  laced output (src/main.cpp:599-1061)
    L1  the L lines are exactly the consecutive step pairs of the P lines, each once (edges are only ever created between
        consecutive steps: src/main.cpp:743-750, 1002-1016) -- so no self loop unless a path walks one;
    L2  S ids are 1..n without holes and every node is on some path;
    L3  the graph is unchopped (src/main.cpp:1021): no u+ -> v+ with that link the only edge on u's right and v's left side
        and no path end in between is left to merge;
    L4  every input path is there and spells its input sequence, the other paths are consensus paths
        (src/main.cpp:770-810);
  block graph (src/smooth.cpp:931-1010)
    B1  L1 + L2 + L3 (only path-supported edges: src/smooth.cpp:980-994; unchop :935);
    B2  all L lines are + +, from a lower to a higher id: the numbering is a topological order of a DAG (:947);
    B3  every path walks forward through ascending ids (or, collected in reverse, entirely flipped through descending ids:
        src/smooth.cpp:2612-2615).
"""


class GfaInvariantError(AssertionError):
    pass


def _comp(s):
    return s[::-1].translate(str.maketrans("ACGTNacgtn", "TGCANtgcan"))


def parse(text):
    seqs, links, paths = {}, [], []
    for ln in text.split("\n"):
        if not ln:
            continue
        f = ln.split("\t")
        if f[0] == "S":
            if int(f[1]) in seqs:
                raise GfaInvariantError("node %s defined twice" % f[1])
            seqs[int(f[1])] = f[2]
        elif f[0] == "L":
            if f[2] not in "+-" or f[4] not in "+-":
                raise GfaInvariantError("bad L line: " + ln)
            links.append((int(f[1]), f[2] == "-", int(f[3]), f[4] == "-"))
        elif f[0] == "P":
            steps = [(int(x[:-1]), x[-1] == "-") for x in f[2].split(",")] if f[2] else []
            paths.append((f[1], steps))
        elif f[0] != "H":
            raise GfaInvariantError("unexpected line: " + ln[:60])
    return seqs, links, paths


def _canon(a, ar, b, br):
    """An edge and its reverse complement are the same edge: the smaller of the two spellings."""
    x, y = (a, ar, b, br), (b, not br, a, not ar)
    return min(x, y)


def check_edges_are_the_walked_pairs(seqs, links, paths, what):
    walked = set()
    for _, st in paths:
        for (a, ar), (b, br) in zip(st, st[1:]):
            walked.add(_canon(a, ar, b, br))
    have = [_canon(*e) for e in links]
    if len(set(have)) != len(have):
        raise GfaInvariantError("%s: an edge is written twice" % what)
    have = set(have)
    if have - walked:
        e = sorted(have - walked)[0]
        raise GfaInvariantError("%s: edge %r is walked by no path (%d such edges)" % (what, e, len(have - walked)))
    if walked - have:
        e = sorted(walked - have)[0]
        raise GfaInvariantError("%s: consecutive steps %r have no edge (%d such pairs)" % (what, e, len(walked - have)))


def check_nodes(seqs, paths, what):
    n = len(seqs)
    if n and sorted(seqs) != list(range(1, n + 1)):
        raise GfaInvariantError("%s: node ids are not 1..%d" % (what, n))
    on_path = set(v for _, st in paths for v, _ in st)
    if on_path - set(seqs):
        raise GfaInvariantError("%s: a path steps on an undefined node" % what)
    if set(seqs) - on_path:
        raise GfaInvariantError("%s: node %d is on no path" % (what, min(set(seqs) - on_path)))
    for v, s in seqs.items():
        if not s:
            raise GfaInvariantError("%s: node %d is empty" % (what, v))


def check_unchopped(seqs, links, paths, what):
    # ends: (node, side), side 0 = left, 1 = right.  An edge leaves a+ by its right end, a- by its left end; it enters b+ at its
    # left end, b- at its right end.
    deg = {}
    right_to_left = []
    for a, ar, b, br in links:
        ea, eb = (a, 0 if ar else 1), (b, 1 if br else 0)
        deg[ea] = deg.get(ea, 0) + 1
        if ea != eb:
            deg[eb] = deg.get(eb, 0) + 1
        if ea[1] == 1 and eb[1] == 0:
            right_to_left.append((a, b))      # a+ -> b+
        elif ea[1] == 0 and eb[1] == 1:
            right_to_left.append((b, a))      # a- -> b- is b+ -> a+
    ends = set()   # ends of the graph where a path begins or stops
    for _, st in paths:
        if st:
            v, r = st[0]
            ends.add((v, 1 if r else 0))      # the path enters its first node here
            v, r = st[-1]
            ends.add((v, 0 if r else 1))      # ... and leaves its last node here
    for u, v in right_to_left:
        if u != v and deg.get((u, 1)) == 1 and deg.get((v, 0)) == 1 and (u, 1) not in ends and (v, 0) not in ends:
            raise GfaInvariantError("%s: nodes %d and %d can still be merged (not unchopped)" % (what, u, v))


def path_sequence(seqs, steps):
    return "".join(_comp(seqs[v]) if r else seqs[v] for v, r in steps)


def check_laced(text, input_text=None, what="laced GFA", consensus_prefix="Consensus_"):
    seqs, links, paths = parse(text)
    check_nodes(seqs, paths, what)
    check_edges_are_the_walked_pairs(seqs, links, paths, what)
    check_unchopped(seqs, links, paths, what)
    names = [nm for nm, _ in paths]
    if len(set(names)) != len(names):
        raise GfaInvariantError("%s: a path name appears twice" % what)
    if input_text is not None:
        iseqs, _, ipaths = parse(input_text)
        mine = dict(paths)
        for nm, st in ipaths:
            if not st:
                continue
            if nm not in mine:
                raise GfaInvariantError("%s: input path %s is missing" % (what, nm))
            if path_sequence(seqs, mine[nm]) != path_sequence(iseqs, st):
                raise GfaInvariantError("%s: path %s no longer spells its input sequence" % (what, nm))
        extra = [nm for nm in names if nm not in set(n for n, s in ipaths if s)]
        for nm in extra:
            if not nm.startswith(consensus_prefix) and "Consensus" not in nm:
                raise GfaInvariantError("%s: unexpected path %s" % (what, nm))
    return seqs, links, paths


def check_block_graph(text, what="block graph"):
    seqs, links, paths = parse(text)
    if not seqs and not paths:
        return seqs, links, paths          # an empty block yields an empty graph (src/smooth.cpp:748-750)
    check_nodes(seqs, paths, what)
    check_edges_are_the_walked_pairs(seqs, links, paths, what)
    check_unchopped(seqs, links, paths, what)
    for a, ar, b, br in links:
        if ar or br or not a < b:
            raise GfaInvariantError("%s: edge %d%s -> %d%s does not run forward to a higher id" % (what, a, "-" if ar else "+", b, "-" if br else "+"))
    for nm, st in paths:
        revs = set(r for _, r in st)
        if len(revs) > 1:
            raise GfaInvariantError("%s: path %s mixes orientations" % (what, nm))
        ids = [v for v, _ in st]
        want = sorted(ids, reverse=(revs == {True}))
        if ids != want or len(set(ids)) != len(ids):
            raise GfaInvariantError("%s: path %s does not walk the topological order" % (what, nm))
    return seqs, links, paths
