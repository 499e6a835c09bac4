#!/usr/bin/env python3
"""Host phases of sxg_smooth_gfa (include/sxg_smooth.h) timed WITHOUT a GPU: the POA results of K distinct blocks
of the headline shape are computed once by the CPU oracle (checker code used as a data generator for a host-side
profile -- nothing here is a product path), cached in /tmp, tiled to NB blocks and handed to sxg_smooth_gfa through
a provider that only copies pointers.  With SXG_SMOOTH_TIMING=1 the library prints its phase times.

    python profiles/tools/host_phases.py [NB=1000] [K=16] [depth=64] [len=5000]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O          # noqa: E402
from smoothxg_amd import smooth as SM      # noqa: E402
from smoothxg_amd import synth, poa        # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 64
LEN = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
PRM = (1, -4, -6, -2, -26, -1)   # spoa convention, as bench.py's WORKLOADS


def one_block(args):
    bases, so = args
    par = O.mkparams(*PRM, mode=0)
    seqs = [bases[so[s]:so[s + 1]] for s in range(len(so) - 1)]
    impl = O.IMPL_AVX2 if O.simd_available() else O.IMPL_SCALAR
    g, _, _ = O.block_run(seqs, None, par, impl=impl)
    return g.nodes()[0], [g.seq_path(k) for k in range(len(seqs))], g.consensus()


def distinct_blocks():
    cache = "/tmp/host_phases_%d_%d_%d.npz" % (K, DEPTH, LEN)
    bases, seq_off, blk_off = synth.make_batch(K, DEPTH, LEN)
    if os.path.exists(cache):
        z = np.load(cache, allow_pickle=True)
        return bases, seq_off, blk_off, list(z["codes"]), list(z["paths"]), list(z["cons"])
    import multiprocessing as mp
    jobs = []
    for b in range(K):
        s0, s1 = int(blk_off[b]), int(blk_off[b + 1])
        so = seq_off[s0:s1 + 1] - seq_off[s0]
        jobs.append((bases[seq_off[s0]:seq_off[s1]].copy(), so.copy()))
    t0 = time.time()
    with mp.Pool(min(8, K)) as pool:
        res = pool.map(one_block, jobs)
    print("oracle: %d blocks in %.1f s" % (K, time.time() - t0), file=sys.stderr)
    codes = [r[0] for r in res]
    paths = [np.concatenate(r[1]).astype(np.int32) for r in res]
    cons = [r[2] for r in res]
    np.savez(cache, codes=np.array(codes, dtype=object), paths=np.array(paths, dtype=object), cons=np.array(cons, dtype=object))
    return bases, seq_off, blk_off, codes, paths, cons


def main():
    bases, seq_off, blk_off, codes, paths, cons = distinct_blocks()
    lut = np.frombuffer(b"ACGTN", np.uint8)
    text = lut[bases].tobytes()
    lines = []
    for b in range(NB):
        kb = b % K
        for k in range(DEPTH):
            s = int(blk_off[kb]) + k
            lines.append(b"S\t%d\t%s\n" % (b * DEPTH + k + 1, text[int(seq_off[s]):int(seq_off[s + 1])]))
    for k in range(DEPTH):
        lines.append(b"P\thap%d\t%s\t*\n" % (k, b",".join(b"%d+" % (b * DEPTH + k + 1) for b in range(NB))))
    gfa = b"".join(lines)
    blocks = [[(k, b, b + 1) for k in range(DEPTH)] for b in range(NB)]
    sm = SM.Smoother(gfa, blocks=blocks)
    del gfa, lines
    p = SM.default_params(poa_m=PRM[0], poa_n=-PRM[1], poa_g=-PRM[2], poa_e=-PRM[3], poa_q=-PRM[4], poa_c=-PRM[5],
                          local_alignment=1, poa_padding_fraction=0.0)
    # tiled POA output
    node_off = np.zeros(NB + 1, np.int64)
    cons_off = np.zeros(NB + 1, np.int64)
    for b in range(NB):
        node_off[b + 1] = node_off[b] + len(codes[b % K])
        cons_off[b + 1] = cons_off[b] + len(cons[b % K])
    reps = (NB + K - 1) // K
    node_code = np.ascontiguousarray(np.concatenate((list(codes) * reps)[:NB]), np.uint8)
    seq_paths = np.ascontiguousarray(np.concatenate((list(paths) * reps)[:NB]), np.int32)
    cons_nodes = np.ascontiguousarray(np.concatenate((list(cons) * reps)[:NB]), np.int32)
    status = np.zeros(NB, np.int32)
    keep = dict(node_off=node_off, cons_off=cons_off, node_code=node_code, seq_paths=seq_paths, cons_nodes=cons_nodes, status=status)

    RUN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(poa.BatchIn), C.POINTER(poa.BatchOut))
    FREE = C.CFUNCTYPE(None, C.POINTER(poa.BatchOut))

    def run(ctx, pin, pout):
        i, o = pin.contents, pout.contents
        assert i.n_blocks == NB, (i.n_blocks, NB)
        o.n_blocks, o.n_seqs = NB, NB * DEPTH
        o.status = keep["status"].ctypes.data_as(C.POINTER(C.c_int32))
        o.node_off = keep["node_off"].ctypes.data_as(C.POINTER(C.c_int64))
        o.node_code = keep["node_code"].ctypes.data_as(C.POINTER(C.c_uint8))
        o.seq_path_nodes = keep["seq_paths"].ctypes.data_as(C.POINTER(C.c_int32))
        o.cons_off = keep["cons_off"].ctypes.data_as(C.POINTER(C.c_int64))
        o.cons_nodes = keep["cons_nodes"].ctypes.data_as(C.POINTER(C.c_int32))
        return 0

    def fre(pout):
        pass

    runp, frep = RUN(run), FREE(fre)
    for rep in range(int(os.environ.get("REPS", "2"))):
        out = C.c_void_p()
        t0 = time.perf_counter()
        rc = sm.L.sxg_smooth_gfa(sm.g, sm.b, C.byref(p), C.cast(runp, C.c_void_p), C.cast(frep, C.c_void_p), None, C.byref(out))
        dt = time.perf_counter() - t0
        if rc:
            raise RuntimeError(sm.L.sxg_smooth_last_error().decode())
        libc = C.CDLL("libc.so.6")
        libc.strlen.restype = C.c_size_t
        libc.strlen.argtypes = [C.c_void_p]
        n = libc.strlen(out)
        import hashlib
        h = hashlib.sha256(C.string_at(out, n)).hexdigest()[:16] if n < (1 << 31) else "-"
        sm.L.sxg_smooth_free(out)
        print("sxg_smooth_gfa host phases (no POA): %.3f s, %d bytes of GFA, sha %s" % (dt, n, h), file=sys.stderr)
    sm.close()


if __name__ == "__main__":
    main()
