"""Block sharding across the GPUs of one node.

Blocks are independent (the reference's only data-parallel axis is the OpenMP loop over
blocks, src/smooth.cpp:1931), so ranks never exchange data while aligning.  The one real
exchange is the reassembly of per-block results before lacing (src/main.cpp:599+): an
all-gather-v over RCCL/xGMI (backend "nccl" on ROCm), or gloo in the CPU tests.
"""
import numpy as np


def block_cost(seq_lens):
    """cost_b = sum_k L_k * (L_1 + 0.05 * sum_{j<k} L_j)."""
    seq_lens = np.asarray(seq_lens, np.float64)
    if len(seq_lens) < 2:
        return 0.0
    prev = np.concatenate([[0.0], np.cumsum(seq_lens)[:-1]])
    return float((seq_lens[1:] * (seq_lens[0] + 0.05 * prev[1:])).sum())


def partition_blocks(costs, world):
    """Longest-processing-time greedy: returns world lists of block ids (each ascending)."""
    costs = np.asarray(costs, np.float64)
    order = sorted(range(len(costs)), key=lambda b: (-costs[b], b))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for b in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(b)
        load[r] += costs[b]
    return [sorted(p) for p in parts]


def shard_batch(bases, seq_off, blk_off, rank, world):
    """Returns (block_ids, bases, seq_off, blk_off) of this rank's share of a flat batch."""
    nb = len(blk_off) - 1
    costs = [block_cost(np.diff(seq_off[blk_off[b]:blk_off[b + 1] + 1])) for b in range(nb)]
    mine = partition_blocks(costs, world)[rank]
    sb, so, bo = [], [0], [0]
    for b in mine:
        for s in range(blk_off[b], blk_off[b + 1]):
            sb.append(bases[seq_off[s]:seq_off[s + 1]])
            so.append(so[-1] + int(seq_off[s + 1] - seq_off[s]))
        bo.append(len(so) - 1)
    return (mine, np.concatenate(sb) if sb else np.zeros(0, np.uint8), np.asarray(so, np.int64),
            np.asarray(bo, np.int32))


def all_gather_v(t, group=None):
    """all-gather of 1-D tensors of different lengths (RCCL has no AllGatherv: lengths first,
    then one padded all-gather).  Returns the list of every rank's tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    outs = [torch.empty(m, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return [o[:k] for o, k in zip(outs, sizes)]


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def device_tensor(ptr, n, typestr):
    """Zero-copy torch view of an HBM array owned by the engine (sxg_poa_device_view)."""
    import torch
    if not ptr or n == 0:
        return torch.zeros(0, dtype={"<i4": torch.int32, "|u1": torch.uint8}[typestr], device="cuda")
    return torch.as_tensor(_DevArr(ptr, n, typestr), device="cuda")


def all_gather_block_summaries(eng, n_blocks, group=None, with_paths=True):
    """The reassembly hand-off of one executed batch, device to device over RCCL: per-block
    (status, #nodes, #edges) and -- the dominant payload, SURVEY 8(e) -- the per-base node
    paths.  Returns (summaries[list of (3, n_blocks_r) tensors], paths[list] | None)."""
    import torch
    v = eng.device_view()
    st = device_tensor(v.status, v.n_blocks, "<i4")
    nn = device_tensor(v.n_nodes, v.n_blocks, "<i4")
    ne = device_tensor(v.n_edges, v.n_blocks, "<i4")
    summ = all_gather_v(torch.cat([st, nn, ne]), group)
    summ = [s.view(3, -1) for s in summ]
    paths = None
    if with_paths:
        paths = all_gather_v(device_tensor(v.seq_path_nodes, v.n_bases, "<i4"), group)
    return summ, paths


class RootGather:
    """The reassembly hand-off as the lacing needs it: only rank 0 laces, so every peer sends its results to rank 0 and
    nobody else receives anything.  Point to point (grouped ncclSend / ncclRecv: all 7 xGMI links into rank 0 carry one
    peer each), one exact-size message per tensor, receive buffers allocated once and reused -- the device-side twin
    of sxg_poa_batch_run_sharded's host-side blobs.  `__call__` takes the local tensors (any device / backend) and
    returns, on rank 0, one list of tensors per rank (rank 0's own are the inputs themselves); None elsewhere."""

    def __init__(self, group=None):
        self.group = group
        self.bufs = {}

    def __call__(self, tensors):
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        dev = tensors[0].device
        n = torch.tensor([t.numel() for t in tensors], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=self.group)     # a few integers per rank
        ops, out = [], None
        if rank == 0:
            out = [list(tensors)]
            for r in range(1, world):
                got = []
                for k, t in enumerate(tensors):
                    m = int(sizes[r][k].item())
                    key = (r, k)
                    if key not in self.bufs or self.bufs[key].numel() < m:
                        self.bufs[key] = torch.empty(max(m, 1), dtype=t.dtype, device=dev)
                    buf = self.bufs[key][:m]
                    if m:
                        ops.append(dist.P2POp(dist.irecv, buf, r, group=self.group))
                    got.append(buf)
                out.append(got)
        else:
            for t in tensors:
                if t.numel():
                    ops.append(dist.P2POp(dist.isend, t.contiguous(), 0, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out


def engine_result_tensors(eng):
    """Zero-copy views of what the lacing needs of an executed batch: (status | #nodes | #edges) per block, and the
    per-base node paths -- the dominant payload, SURVEY 8(e)."""
    import torch
    v = eng.device_view()
    st = device_tensor(v.status, v.n_blocks, "<i4")
    nn = device_tensor(v.n_nodes, v.n_blocks, "<i4")
    ne = device_tensor(v.n_edges, v.n_blocks, "<i4")
    return [torch.cat([st, nn, ne]), device_tensor(v.seq_path_nodes, v.n_bases, "<i4")]


def gather_results_host(local_block_ids, local_results, group=None):
    """Host-side reassembly (any backend): every rank ends up with {block_id: result-dict}."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    payload = [(b, {k: getattr(r, k) for k in ("status", "node_code", "node_rank", "node_group", "edge_tail",
                                               "edge_head", "edge_weight", "paths", "scores")})
               for b, r in zip(local_block_ids, local_results)]
    out = [None] * world
    dist.all_gather_object(out, payload, group=group)
    merged = {}
    for part in out:
        for b, r in part:
            merged[b] = r
    return merged
