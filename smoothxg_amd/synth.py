"""Deterministic synthetic block generator.

Per block: seed = 0xC0FFEE + block_id; ancestor uniform over {A,C,G,T}; each sequence gets
independent per-base substitutions (1.0 %), insertions (0.25 %) and deletions (0.25 %) with
geometric(p=0.5) length capped at 16, plus one shared 50-300 bp structural variant carried
by 25 % of the sequences; sequences are ordered longest first, mirroring
src/blocks.cpp:206-219 of the reference.  Codes: A=0 C=1 G=2 T=3 (N=4 never generated).
"""
import numpy as np

SEED0 = 0xC0FFEE


def _mutate(rng, anc, sub, ins, dele, sv):
    L = len(anc)
    seq = anc.copy()
    m = rng.random(L) < sub
    k = int(m.sum())
    if k:
        seq[m] = (seq[m] + rng.integers(1, 4, k, dtype=np.uint8)) & 3
    count = np.ones(L, np.int64)                      # copies of each ancestral base kept
    d = np.flatnonzero(rng.random(L) < dele)
    if len(d):
        dl = np.minimum(rng.geometric(0.5, len(d)), 16)
        for p, n in zip(d, dl):
            count[p:p + n] = 0
    ins_len = np.zeros(L, np.int64)
    i = np.flatnonzero(rng.random(L) < ins)
    if len(i):
        ins_len[i] = np.minimum(rng.geometric(0.5, len(i)), 16)
    if sv is not None:
        kind, pos, n, payload = sv
        if kind == 0:
            count[pos:pos + n] = 0
            ins_len[pos:pos + n] = 0
        else:
            ins_len[pos] += n
    rep = count + ins_len
    idx = np.repeat(np.arange(L), rep)
    out = seq[idx]
    # positions that are insertions: within each run of `rep[p]` copies, the last ins_len[p]
    start = np.cumsum(rep) - rep
    within = np.arange(len(idx)) - start[idx]
    is_ins = within >= count[idx]
    n_ins = int(is_ins.sum())
    if n_ins:
        out[is_ins] = rng.integers(0, 4, n_ins, dtype=np.uint8)
        if sv is not None and sv[0] == 1:
            kind, pos, n, payload = sv
            # the shared SV payload is identical across carriers: overwrite its span
            s0 = start[pos] + count[pos] + (ins_len[pos] - n)
            out[s0:s0 + n] = payload
    return out


def make_block(block_id, n_seqs, length, sub=0.01, ins=0.0025, dele=0.0025, sv_frac=0.25):
    """Returns a list of uint8 code arrays, longest first."""
    rng = np.random.Generator(np.random.PCG64(SEED0 + int(block_id)))
    anc = rng.integers(0, 4, int(length), dtype=np.uint8)
    sv = None
    if length >= 400:
        n = int(rng.integers(50, 301))
        n = min(n, length // 4)
        pos = int(rng.integers(0, length - n))
        kind = int(rng.integers(0, 2))
        sv = (kind, pos, n, rng.integers(0, 4, n, dtype=np.uint8))
    seqs = []
    for _ in range(int(n_seqs)):
        carries = sv is not None and rng.random() < sv_frac
        seqs.append(_mutate(rng, anc, sub, ins, dele, sv if carries else None))
    order = sorted(range(len(seqs)), key=lambda k: -len(seqs[k]))  # stable, longest first
    return [seqs[k] for k in order]


def make_batch(n_blocks, n_seqs, length, first_block=0, mixed=False):
    """Flat batch in the C-ABI layout: bases u8, seq_off i64[n_seqs+1], blk_off i32[n_blocks+1].

    mixed=True draws per-block S ~ U{8..128} and L ~ U[500,10000] (BASELINE config C4)."""
    bases, seq_off, blk_off = [], [0], [0]
    for b in range(first_block, first_block + n_blocks):
        S, L = n_seqs, length
        if mixed:
            r = np.random.Generator(np.random.PCG64(SEED0 * 31 + b))
            S = int(r.integers(8, 129))
            L = int(r.integers(500, 10001))
        for s in make_block(b, S, L):
            bases.append(s)
            seq_off.append(seq_off[-1] + len(s))
        blk_off.append(len(seq_off) - 1)
    return (np.concatenate(bases) if bases else np.zeros(0, np.uint8),
            np.asarray(seq_off, np.int64), np.asarray(blk_off, np.int32))


def decode(codes):
    return "".join("ACGTN"[c] for c in codes)


def encode(s):
    lut = np.full(256, 4, np.uint8)
    for i, ch in enumerate("ACGT"):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    return lut[np.frombuffer(s.encode(), np.uint8)]
