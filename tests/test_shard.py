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
