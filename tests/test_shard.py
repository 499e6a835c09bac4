"""CPU: block sharding + the reassembly all-gather, world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smoothxg_amd import shard, synth


def test_lpt_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    costs = rng.random(200) ** 3 * 1e9
    for world in (1, 2, 4, 8):
        parts = shard.partition_blocks(costs, world)
        assert sorted(b for p in parts for b in p) == list(range(200))
        loads = [costs[p].sum() for p in parts]
        assert max(loads) <= min(loads) + costs.max() + 1e-6


def test_shard_batch_round_trip():
    bases, seq_off, blk_off = synth.make_batch(7, 3, 50)
    seen = {}
    for r in range(2):
        ids, b, so, bo = shard.shard_batch(bases, seq_off, blk_off, r, 2)
        for k, bid in enumerate(ids):
            got = [b[so[s]:so[s + 1]] for s in range(bo[k], bo[k + 1])]
            ref = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[bid], blk_off[bid + 1])]
            assert all((x == y).all() for x, y in zip(got, ref))
            seen[bid] = True
    assert len(seen) == 7


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.arange(3 + 4 * rank, dtype=torch.int32) + 100 * rank
    outs = shard.all_gather_v(t)
    ok = all((o == torch.arange(3 + 4 * r, dtype=torch.int32) + 100 * r).all() for r, o in enumerate(outs))

    # point-to-point reassembly: rank 0 gets every rank's tensors at their exact sizes, twice through the same buffers
    rg = shard.RootGather()
    for rep in range(2):
        mine = [torch.arange(5 + 3 * rank + rep, dtype=torch.int32) + 1000 * rank, torch.zeros(0, dtype=torch.int32) if rank else torch.ones(2, dtype=torch.int32)]
        got = rg(mine)
        if rank == 0:
            ok = ok and len(got) == world and all((got[r][0] == torch.arange(5 + 3 * r + rep, dtype=torch.int32) + 1000 * r).all() for r in range(world))
            ok = ok and got[1][1].numel() == 0 and got[0][1].numel() == 2
        else:
            ok = ok and got is None

    class R:
        pass
    res = []
    ids = [rank, rank + 2]
    for b in ids:
        r = R()
        r.status = 0
        for k in ("node_code", "node_rank", "node_group", "edge_tail", "edge_head", "edge_weight", "scores"):
            setattr(r, k, np.full(2, b))
        r.paths = [np.full(3, b)]
        res.append(r)
    merged = shard.gather_results_host(ids, res)
    ok = ok and sorted(merged) == [0, 1, 2, 3] and all((merged[b]["node_code"] == b).all() for b in merged)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_all_gather_v_and_host_reassembly_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == [(0, True), (1, True)]


def _lace_worker(rank, world, port, q, chunk_blocks=0):
    """shard -> (mock engine: the C oracle, own blocks only) -> gather -> rank 0 laces -> GFA.
    chunk_blocks > 0: the iteration runs as a pipeline of chunks of that many blocks (SXG_SMOOTH_CHUNK_BLOCKS): every chunk is
    one collective provider call on every rank, and rank 0 indexes and validates chunk k while the ranks align chunk k + 1."""
    import ctypes as C
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_smooth_host as H
    from oracle import oracle_py as O
    from smoothxg_amd import smooth as S

    class ShardedOracleProvider(H.OracleProvider):
        """What sxg_poa_batch_run_sharded does, with the oracle as every rank's engine and gloo as the transport: every
        rank is handed the same batch, aligns its LPT share, the results meet on rank 0, the others answer NOT_ROOT."""

        def _run(self, ctx, pin, pout):
            i = pin.contents
            nb = i.n_blocks
            blk = np.ctypeslib.as_array(i.blk_off, (nb + 1,)).copy()
            ns = int(blk[-1])
            so = np.ctypeslib.as_array(i.seq_off, (ns + 1,)).copy()
            bases = np.ctypeslib.as_array(i.bases, (max(int(so[-1]), 1),)).copy()
            w = np.ctypeslib.as_array(i.weights, (max(ns, 1),)).copy()
            costs = [shard.block_cost(np.diff(so[blk[b]:blk[b + 1] + 1])) for b in range(nb)]
            mine = shard.partition_blocks(costs, world)[rank]
            pr = i.params[0]
            par = O.mkparams(pr.m, pr.n, pr.g, pr.e, pr.q, pr.c, pr.mode)
            local = {}
            for b in mine:
                seqs = [bases[so[s]:so[s + 1]] for s in range(blk[b], blk[b + 1])]
                if seqs:
                    g, _, _ = O.block_run(seqs, w[blk[b]:blk[b + 1]], par)
                    local[b] = (g.nodes()[0], [g.seq_path(k) for k in range(len(seqs))], g.consensus())
            parts = [None] * world
            dist.all_gather_object(parts, local)
            if rank != 0:
                return 1   # SXG_NOT_ROOT
            merged = {}
            for part in parts:
                merged.update(part)
            self.merged = merged
            self.blk = blk
            self.calls = getattr(self, "calls", 0) + 1
            self.blocks_seen = getattr(self, "blocks_seen", 0) + len(merged)
            return H.OracleProvider._run(self, ctx, pin, pout)

    text = H.haplotype_gfa(5, n_paths=5, length=900)
    prov = ShardedOracleProvider()
    # rank 0's lacing must see exactly what the oracle computes for EVERY block: replace block_run with a lookup
    real = O.block_run

    def looked_up(seqs, weights, params, impl=0):
        if rank == 0 and hasattr(prov, "merged"):
            for b, (code, paths, cons) in prov.merged.items():
                if len(paths) == len(seqs) and all(len(p) == len(s) for p, s in zip(paths, seqs)) and \
                        all((code[p] == s).all() for p, s in zip(paths, seqs)):
                    class G:
                        n_seqs = len(seqs)
                        def nodes(self_): return code, None, None
                        def seq_path(self_, k): return paths[k]
                        def consensus(self_): return cons
                        def msa(self_, c): raise NotImplementedError
                    return G(), None, None
        return real(seqs, weights, params, impl)
    O.block_run = looked_up
    sm = S.Smoother(text, 250)
    if chunk_blocks:
        os.environ["SXG_SMOOTH_CHUNK_BLOCKS"] = str(chunk_blocks)
    got = sm.smooth_gfa(S.default_params(add_consensus=1), prov.provider())
    os.environ.pop("SXG_SMOOTH_CHUNK_BLOCKS", None)
    O.block_run = real
    n_chunks = max(1, sm.n_blocks // chunk_blocks) if chunk_blocks else 1
    if rank == 0:
        single = S.Smoother(text, 250).smooth_gfa(S.default_params(add_consensus=1), H.OracleProvider().provider())
        ok = got == single and got is not None and prov.blocks_seen == sm.n_blocks and prov.calls == n_chunks and (n_chunks >= 3 or not chunk_blocks)
        if not ok:
            print("MISMATCH", got == single, prov.blocks_seen, sm.n_blocks, prov.calls, n_chunks, flush=True)
        q.put((rank, ok))
    else:
        q.put((rank, got is None))
    dist.destroy_process_group()


def test_shard_gather_lace_equals_single_rank_gloo_world2():
    """Config 5's shape on CPU: two ranks run the same smoothing iteration through a sharded provider (LPT share each,
    results gathered on rank 0 over gloo); rank 0's laced GFA equals the single-rank GFA, rank 1 gets NOT_ROOT."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lace_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == [(0, True), (1, True)]


def test_sharded_iteration_in_three_chunks_equals_single_rank_gloo_world2():
    """The sharded run through the chunk pipeline (verdict item 6): three chunks, each a collective provider call on both
    ranks; rank 0 builds its view of chunk k's block graphs and validates their ranges while both ranks align chunk k + 1,
    laces at the end; the GFA equals the single-rank, single-chunk one; rank 1 gets NOT_ROOT after taking part in every chunk."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lace_worker, args=(r, 2, port, q, 1)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == [(0, True), (1, True)]


def _bg_lace_worker(rank, world, port, q):
    """Every rank builds the COMPACT block graphs of its LPT share (the product's block-graph code through the CPU
    emulation harness, on the oracle's POA), the graphs meet on rank 0 over gloo, and rank 0 laces graphs it did not build
    (sxg_poa_batch_out::bg_*: what sxg_poa_batch_run_sharded hands it when block graphs are asked for)."""
    import ctypes as C
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_graph_emul as GE
    import test_smooth_host as H
    from oracle import oracle_py as O
    from smoothxg_amd import smooth as S
    emul = C.CDLL(os.path.join(here, "csrc", "libgraph_emul.so"))
    stats = {"built": 0, "asked": 0}

    class ShardedBgProvider(H.OracleProvider):
        def _run(self, ctx, pin, pout):
            i, o = pin.contents, pout.contents
            nb = i.n_blocks
            blk = np.ctypeslib.as_array(i.blk_off, (nb + 1,)).copy()
            ns = int(blk[-1])
            so = np.ctypeslib.as_array(i.seq_off, (ns + 1,)).copy()
            bases = np.ctypeslib.as_array(i.bases, (max(int(so[-1]), 1),)).copy()
            w = np.ctypeslib.as_array(i.weights, (max(ns, 1),)).copy()
            trim = np.ctypeslib.as_array(i.bg_trim, (max(nb, 1),)).copy() if i.bg_trim else np.zeros(max(nb, 1), np.int32)
            stats["asked"] = i.want_block_graph
            costs = [shard.block_cost(np.diff(so[blk[b]:blk[b + 1] + 1])) for b in range(nb)]
            mine = shard.partition_blocks(costs, world)[rank]
            pr = i.params[0]
            par = O.mkparams(pr.m, pr.n, pr.g, pr.e, pr.q, pr.c, pr.mode)
            local = {}
            for b in mine:
                seqs = [bases[so[s]:so[s + 1]] for s in range(blk[b], blk[b + 1])]
                if seqs:
                    g, _, _ = O.block_run(seqs, w[blk[b]:blk[b + 1]], par)
                    B = GE.run_block_graph_emul(emul, g, seqs, int(trim[b]), (2 if i.bg_consensus_visited_only else 1) if i.want_consensus else 0)
                    local[b] = (B.node_seq, B.node_indeg, B.edges, [np.asarray(p) for p in B.paths], np.asarray(B.consensus))
                    stats["built"] += 1
            parts = [None] * world
            dist.all_gather_object(parts, local)
            if rank != 0:
                return 1   # SXG_NOT_ROOT
            merged = {}
            for part in parts:
                merged.update(part)
            node_off, seq_off, edge_off, cons_off, step_off = [0], [0], [0], [0], [0]
            nlen, nod, nid_, seq, eto, steps, cons = [], [], [], b"", [], [], []
            for b in range(nb):
                if b in merged:
                    ns_, idg, edges, paths, cn = merged[b]
                    od = np.zeros(len(ns_), np.int32)
                    for a, _ in edges:
                        od[a] += 1
                    nlen += [len(x) for x in ns_]; nod += od.tolist(); nid_ += list(idg)
                    seq += "".join(ns_).encode(); eto += [h for _, h in edges]; cons += cn.tolist()
                    for p in paths:
                        steps += p.tolist()
                        step_off.append(len(steps))
                else:
                    step_off += [step_off[-1]] * int(blk[b + 1] - blk[b])
                node_off.append(len(nlen)); seq_off.append(len(seq)); edge_off.append(len(eto)); cons_off.append(len(cons))
            arr = lambda x, dt: np.ascontiguousarray(np.asarray(x if len(x) else [0], dt))
            A = dict(node_off=arr(node_off, np.int64), seq_off=arr(seq_off, np.int64), edge_off=arr(edge_off, np.int64), cons_off=arr(cons_off, np.int64),
                     step_off=arr(step_off, np.int64), nlen=arr(nlen, np.int32), nod=arr(nod, np.int32), nid=arr(nid_, np.uint8),
                     seq=np.frombuffer(seq + b"\0", np.uint8).copy(), eto=arr(eto, np.int32), steps=arr(steps, np.int32), cons=arr(cons, np.int32),
                     status=np.zeros(max(nb, 1), np.int32))
            self.keep.append(A)
            P32, P64, P8 = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
            o.n_blocks, o.n_seqs = nb, ns
            o.status = A["status"].ctypes.data_as(P32)
            o.bg_node_off = A["node_off"].ctypes.data_as(P64); o.bg_node_len = A["nlen"].ctypes.data_as(P32)
            o.bg_node_outdeg = A["nod"].ctypes.data_as(P32); o.bg_node_indeg = A["nid"].ctypes.data_as(P8)
            o.bg_seq_off = A["seq_off"].ctypes.data_as(P64); o.bg_seq = A["seq"].ctypes.data
            o.bg_edge_off = A["edge_off"].ctypes.data_as(P64); o.bg_edge_to = A["eto"].ctypes.data_as(P32)
            o.bg_step_off = A["step_off"].ctypes.data_as(P64); o.bg_steps = A["steps"].ctypes.data_as(P32)
            if i.want_consensus:
                o.bg_cons_off = A["cons_off"].ctypes.data_as(P64); o.bg_cons_steps = A["cons"].ctypes.data_as(P32)
            return 0

    ok = True
    for text, tb, kw in ((H.haplotype_gfa(5, n_paths=5, length=900), 250, dict(add_consensus=1)),
                         (H.synthetic_gfa(2), 120, dict()),
                         (H.haplotype_gfa(7, n_paths=4, length=700), 200, dict(add_consensus=1, use_abpoa=1, poa_padding_fraction=0.0))):
        prov = ShardedBgProvider()
        stats["built"] = 0
        sm = S.Smoother(text, tb)
        got = sm.smooth_gfa(S.default_params(**kw), prov.provider())
        if rank == 0:
            single = S.Smoother(text, tb).smooth_gfa(S.default_params(**kw), H.OracleProvider().provider())
            if not (got is not None and got == single and stats["asked"] == 3 and 0 < stats["built"] < sm.n_blocks):
                print("MISMATCH", tb, kw, got == single, stats, sm.n_blocks, flush=True)
            ok = ok and got is not None and got == single and stats["asked"] == 3 and 0 < stats["built"] < sm.n_blocks
        else:
            ok = ok and got is None
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_root_laces_compact_block_graphs_built_on_other_ranks_gloo_world2():
    """Verdict item: in the sharded run the block graphs are built on the owning rank; rank 0 receives compact graphs and
    only laces.  Two gloo ranks, the product's block-graph code (emulated on the CPU) as every rank's builder: rank 0's GFA
    equals the single-rank GFA built from per-base paths on the host."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-C", os.path.join(here, "csrc"), "-s"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bg_lace_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == [(0, True), (1, True)]
