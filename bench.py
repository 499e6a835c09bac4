#!/usr/bin/env python3
"""bench.py -- blocked-POA throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (every alignment of every block of the batch:
DP fill, traceback, graph fusion, re-rank -- src/smooth.cpp:760-769 per block) over one
synthetic batch whose inputs are already resident in HBM.  Default workload = the north-star
headline: 1000 synthetic blocks x 64 sequences x 5 kbp, default convex scores 1,4,6,2,26,1,
local alignment (the reference's defaults, src/main.cpp:322-327,487).

    python bench.py --gpus N --steps K --warmup W [--workload ns|c2|c3|tiny] [--mode sw|nw]

N>1: launched by torch.distributed.run, one rank per GPU; blocks are independent, so each
rank owns its own 1000 blocks (weak scaling) and only the per-block result summaries are
all-gathered over RCCL at the end of every step (reassembly hand-off for lacing).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch / HIP initialise (mixed batches: one stream per geometry)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (blocks, seqs, length, params (m,n,g,e,q,c spoa convention), description)
    "ns": (1000, 64, 5000, (1, -4, -6, -2, -26, -1), "north-star: 1000 blocks x 64 seqs x 5 kbp, convex 1,4,6,2,26,1"),
    "c2": (1000, 16, 1000, (1, -4, -6, -2, -26, -1), "config 2: 1000 blocks x 16 seqs x 1 kbp, convex 1,4,6,2,26,1"),
    "c3": (5000, 64, 5000, (1, -4, -8, -2, -8, -2), "config 3: 5000 blocks x 64 seqs x 5 kbp, affine (abPOA o+k*e => g=-(o+e))"),
    # config 4 is 50 000 mixed blocks over 8 GPUs: 6 250 per GPU; seqs/length are drawn per block (synth mixed=True)
    "c4": (6250, 0, 0, (1, -4, -6, -2, -26, -1), "config 4: mixed blocks, 8-128 seqs x 0.5-10 kbp, 6250 per GPU, convex 1,4,6,2,26,1"),
    "tiny": (64, 8, 400, (1, -4, -6, -2, -26, -1), "smoke: 64 blocks x 8 seqs x 400 bp"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(workload, mode, seconds_hint=20.0):
    """Oracle ("port") timed on the host cores on a bounded sample of the same workload."""
    from oracle import oracle_py as O
    from smoothxg_amd import synth
    nb, ns, ln, prm, _ = WORKLOADS[workload]
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # memory guard: ~ (tb + ordinals + live rows) per thread
    per_thread = 8.0 * ln * ln * 2.2 + 64e6
    try:
        with open("/proc/meminfo") as f:
            avail = [int(l.split()[1]) * 1024 for l in f if l.startswith("MemAvailable")][0]
        cores = max(1, min(cores, int(avail * 0.5 / per_thread)))
    except Exception:
        pass
    # sample: `cores` blocks of the workload (one per thread), each cut to its first k sequences;
    # k is sized from a short calibration so the timed sample is ~seconds_hint of CPU work
    def cut(bases, seq_off, blk_off, k):
        keep, so, bo = [], [0], [0]
        for b in range(len(blk_off) - 1):
            for s in range(blk_off[b], blk_off[b] + k):
                keep.append(bases[seq_off[s]:seq_off[s + 1]])
                so.append(so[-1] + int(seq_off[s + 1] - seq_off[s]))
            bo.append(len(so) - 1)
        return np.concatenate(keep), np.asarray(so, np.int64), np.asarray(bo, np.int32)
    bases, seq_off, blk_off = synth.make_batch(cores, ns, ln, first_block=10_000_000)
    p = O.mkparams(*prm, mode=mode)
    kc = min(4, ns)
    cb, cso, cbo = cut(bases, seq_off, blk_off, kc)
    t0 = time.time()
    _, ccells, _, _ = O.blocks_run_omp(cb, cso, cbo, None, p, cores)
    rate = ccells / max(time.time() - t0, 1e-3)            # cells/s of the whole machine
    k = kc
    while k < ns and 0.55 * cores * ln * ln * (2 * k) * (2 * k) * (1 + 0.006 * 2 * k) / rate < seconds_hint:
        k *= 2
    k = min(k, ns)
    sb, so, bo = cut(bases, seq_off, blk_off, k)
    t0 = time.time()
    _, cells, _, _ = O.blocks_run_omp(sb, so, bo, None, p, cores)
    dt = time.time() - t0
    return {"cells_per_s": cells / dt, "cores": cores, "seconds": dt,
            "sample": "%d blocks of the workload (one per thread), first %d of %d sequences each, "
                      "oracle/poa_oracle.c scalar C, OpenMP schedule(dynamic,1); %.1f s, %.3g cells"
                      % (cores, k, ns, dt, cells)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ns", choices=list(WORKLOADS))
    ap.add_argument("--mode", default="sw", choices=["sw", "nw"])
    ap.add_argument("--blocks", type=int, default=0, help="override the number of blocks per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also verify 2 blocks against the oracle")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import smoothxg_amd as S
    from smoothxg_amd import synth
    from smoothxg_amd import shard

    nb, ns, ln, prm, desc = WORKLOADS[a.workload]
    if a.blocks:
        nb = a.blocks
    mode = 0 if a.mode == "sw" else 1
    params = S.Params(*prm, mode, 0)
    bases, seq_off, blk_off = synth.make_batch(nb, ns, ln, first_block=rank * nb, mixed=(a.workload == "c4"))
    eng = S.PoaEngine(local_rank)
    eng.upload(bases, seq_off, blk_off, None, params)  # inputs resident in HBM from here on

    def step():
        eng.execute()
        if world > 1:
            shard.all_gather_block_summaries(eng, nb)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, cells, algo_bytes, launches = 0.0, 0, 0, 0
    for _ in range(a.steps):
        step()
        st = eng.stats()
        kernel_ms += st["kernel_ms"]
        cells += st["cells"]
        algo_bytes += st["algo_bytes"]
        launches += st["dp_launches"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor([float(cells)], device="cuda", dtype=torch.float64)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        total_cells = float(cc.item())
    else:
        total_cells = float(cells)
    st = eng.stats()

    if a.check and rank == 0:
        from oracle import oracle_py as O
        res = eng.download()
        for b in (0, nb - 1):
            seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[b], blk_off[b + 1])]
            g, sc, _ = O.block_run(seqs, None, O.mkparams(*prm, mode=mode))
            assert (res[b].scores == sc).all() and len(res[b].node_code) == g.n_nodes, "bench check failed"

    if rank == 0:
        blocks_total = nb * world * a.steps
        value = blocks_total / dt
        ach = (algo_bytes / 1e9) / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("%s_%s" % (a.workload, a.mode), {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "POA blocks/sec (+ DP cells/sec) on 1000-block synthetic",
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16" if st["dom_row_mode"] == 2 else "int32", "data": "synthetic",
            "config": {"workload": desc, "blocks_per_gpu": nb, "mode": a.mode,
                       "cells_per_step_per_gpu": cells / a.steps},
            "cells_per_sec": total_cells / dt,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "poa_block_kernel<T=%d, cols/lane=%d, %s>" % (
                             st["dom_threads"], st["dom_cols_per_lane"],
                             "packed int16 sweep" if st["dom_row_mode"] == 2 else "32-bit sweep"),
                         "kernel_ms_per_launch": kernel_ms / max(launches, 1),
                         "algo_bytes_per_launch": algo_bytes / max(launches, 1),
                         "bytes_per_cell": algo_bytes / max(cells, 1)},
            "engine": {"slots": st["n_slots"], "retries": st["retries"], "arena_bytes": st["device_bytes"]},
        }
        if world == 1 and not a.no_cpu_baseline and a.workload != "c4":  # (no fixed shape to sample for c4)
            cb = cpu_baseline(a.workload, mode)
            cells_per_block = cells / a.steps / nb
            out["cpu_baseline"] = {"value": cb["cells_per_s"] / cells_per_block, "unit": "blocks/s",
                                   "cores": cb["cores"], "kind": "port", "sample": cb["sample"],
                                   "cells_per_sec": cb["cells_per_s"]}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
