#!/usr/bin/env python3
"""bench.py -- blocked-POA throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (every alignment of every block of the batch:
DP fill, traceback, graph fusion, re-rank -- src/smooth.cpp:760-769 per block) over one
synthetic batch whose inputs are already resident in HBM.  Default workload = the north-star
headline: 1000 synthetic blocks x 64 sequences x 5 kbp, default convex scores 1,4,6,2,26,1,
local alignment (the reference's defaults, src/main.cpp:322-327,487).

    python bench.py --gpus N --steps K --warmup W [--workload ns|c2|c3|tiny] [--mode sw|nw]

N>1: launched by torch.distributed.run, one rank per GPU; blocks are independent, so each
rank owns its own 1000 blocks (weak scaling; --scaling strong deals ONE workload by LPT) and the
only exchange is the reassembly hand-off for lacing at the end of every step: every peer sends its
per-block summaries and per-base node paths to rank 0, device to device, one exact-size message
each (grouped ncclSend/ncclRecv over xGMI, shard.RootGather).  Rank 0 prints ONE JSON line.
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
    # config 2's shape in the batch size real runs have (DRB1 at -l 700: 2 161 blocks per iteration; a human chromosome: 10^5): the
    # batch fills the chip with one wave per block
    "c2x8": (8000, 16, 1000, (1, -4, -6, -2, -26, -1), "config 2's blocks, 8000 of them: 8000 blocks x 16 seqs x 1 kbp, convex 1,4,6,2,26,1"),
    # the headline shape with smoothxg's four-parameter scores (spoa: q = g, c = e): in --mode nw the all-gap corner is -20 006
    "ns4": (1000, 64, 5000, (1, -4, -6, -2, -6, -2), "headline shape, affine 1,4,6,2 (four-parameter form): 1000 blocks x 64 seqs x 5 kbp"),
    "c3": (5000, 64, 5000, (1, -4, -8, -2, -8, -2), "config 3: 5000 blocks x 64 seqs x 5 kbp, affine (abPOA o+k*e => g=-(o+e)), full matrix"),
    # config 3 as the reference's -A (abPOA) path runs it: banded, wb=311 wf=0.03 (src/smooth.cpp:266-271); cells = band cells
    "c3b": (5000, 64, 5000, (1, -4, -8, -2, -8, -2), "config 3 banded: 5000 blocks x 64 seqs x 5 kbp, affine, band w = 311 + 0.03 L (abPOA path)"),
    # ... and with abPOA's ADAPTIVE band (params.banded = 2, decree B4): what smooth_abpoa really asks for (src/smooth.cpp:2090)
    "c3a": (5000, 64, 5000, (1, -4, -8, -2, -8, -2), "config 3 adaptive band: 5000 blocks x 64 seqs x 5 kbp, affine, abPOA's adaptive band w = 311 + 0.03 L"),
    # config 4 is 50 000 mixed blocks over 8 GPUs: 6 250 per GPU; seqs/length are drawn per block (synth mixed=True)
    "c4": (6250, 0, 0, (1, -4, -6, -2, -26, -1), "config 4: mixed blocks, 8-128 seqs x 0.5-10 kbp, 6250 per GPU, convex 1,4,6,2,26,1"),
    "tiny": (64, 8, 400, (1, -4, -6, -2, -26, -1), "smoke: 64 blocks x 8 seqs x 400 bp"),
    # configs 1 / 5: the reference's own ctest on its DRB1 input (CMakeLists.txt:565; timed whole in test/performance/check.md)
    "drb1": (0, 0, 0, (1, -4, -6, -2, -26, -1), "DRB1-3123 seqwish GFA, three chained iterations -l 700,900,1100 -j 5k -e 5k -r 12 (the reference's ctest)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU issue roof of the packed sweep.  256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD's issue port for
# 4 cycles (64 lanes over a 16-lane SIMD), except the full-rate opcodes (v_add_u32, v_sub_u32, v_and/or/xor, v_mov, shifts
# right ...) issued back to back, which take 2.  Measured on the box, cycles counted on the device
# (profiles/ubench/op_rate.hip -> profiles/r03/op_rate.json): 4.13-4.51 and 2.25-2.47 per wave instruction per SIMD.  The
# roof is priced with the ARCHITECTURAL 4 and 2 (a lower bound on the issue time of the instructions executed), the
# dual-rate instructions are counted by the hardware: SQ_ACTIVE_INST_VALU2 reads 0.429 per full-rate instruction issued
# back to back and 0.000 per half-rate one (profiles/r03/class_counters_summary.txt, one opcode per kernel).
VALU_SIMDS, VALU_CLOCK_GHZ = 1024, 2.395
VALU_CYCLES_HALF_RATE, VALU_CYCLES_FULL_RATE = 4.0, 2.0


def issue_costs():
    """Calibration of the VALU roof from the committed micro-benchmark outputs: measured cycles per wave instruction per
    SIMD of the two opcode classes (profiles/r03/op_rate.json) and SQ_ACTIVE_INST_VALU2 per full-rate instruction
    (profiles/r03/class_counters_summary.txt).  Falls back to the figures of the committed run."""
    out = {"half_rate_cycles_measured": 4.263, "full_rate_cycles_measured": 2.469, "valu2_per_full_rate_inst": 0.429,
           "source": "defaults (figures of profiles/r03)"}
    try:
        ops = {(o["op"], o["waves_per_simd"]): o["cycles_per_wave_instruction"]
               for o in json.load(open(os.path.join(ROOT, "profiles", "r03", "op_rate.json")))}
        out["half_rate_cycles_measured"] = ops[("k_pk_max_i16", 4)]
        out["full_rate_cycles_measured"] = ops[("k_add_u32", 4)]
        out["source"] = "profiles/r03/op_rate.json (4 waves per SIMD: k_pk_max_i16, k_add_u32)"
        import re
        for line in open(os.path.join(ROOT, "profiles", "r03", "class_counters_summary.txt")):
            if line.startswith("class_ubench k_sub_u32 "):
                out["valu2_per_full_rate_inst"] = float(re.search(r"SQ_ACTIVE_INST_VALU2=[0-9.e+]+\(([0-9.]+)\)", line).group(1))
                out["source"] += " + profiles/r03/class_counters_summary.txt (k_sub_u32)"
    except Exception:
        pass
    return out


def physical_cores():
    """Physical cores this process may run on (SMT siblings counted once)."""
    try:
        allowed = os.sched_getaffinity(0)
    except Exception:
        allowed = set(range(os.cpu_count() or 1))
    cores, cpu, phys, core = set(), None, 0, 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
                if cpu in allowed:
                    cores.add((phys, core))
    except Exception:
        pass
    return max(1, len(cores) if cores else len(allowed))


def cgroup_cpu_limit():
    """CPUs the container's cgroup allows (cpu.max quota / period), None if unlimited.  Measured on the GPU box:
    cpu.max = "1600000 100000" -- 16 CPUs of a 2 x 64-core host; with more runnable threads than that the CFS
    quota throttles all of them and the aggregate rate FALLS (22 -> 9 Gcells/s from 32 to 128 threads)."""
    for f in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(f).read().split()[:2]
            if q != "max":
                return max(1, int(int(q) / int(per)))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except Exception:
        pass
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(workload, mode, spoa_order=True):
    """The oracle ("port") timed on the host's physical cores on WHOLE blocks of the workload (all sequences of
    every sampled block), one block per thread per round, per-thread workspaces, OpenMP schedule(dynamic,1) as
    src/smooth.cpp:1931.  Quoted: the AVX2 int16 row sweep (oracle/poa_simd.c) -- the reference's spoa is an int16
    SIMD row sweep too, so the scalar oracle would be a strawman.  Also reported: the scalar oracle's
    single-thread and all-thread rates on a smaller sample (the contention check)."""
    from oracle import oracle_py as O
    from smoothxg_amd import synth
    nb, ns, ln, prm, _ = WORKLOADS[workload]
    phys = physical_cores()
    quota = cgroup_cpu_limit()
    cores = min(phys, quota) if quota else phys
    p = O.mkparams(*prm, mode=mode | (0x10 if spoa_order else 0))   # (the same node order as the timed GPU batch)
    impl = O.IMPL_AVX2 if O.simd_available() else O.IMPL_SCALAR
    # memory guard: the vectorised variant keeps H, oF, oO of every cell (6 B) of one alignment per thread
    rows = ln * max(2.0, 1.0 + 0.0165 * ns) + 1024
    per_thread = 6.0 * rows * (ln + 64) * 1.2 + 64e6
    threads = cores
    try:
        with open("/proc/meminfo") as f:
            avail = [int(l.split()[1]) * 1024 for l in f if l.startswith("MemAvailable")][0]
        threads = max(1, min(cores, int(avail * 0.5 / per_thread)))
    except Exception:
        pass
    # calibration: one whole block on one thread
    cb, cso, cbo = synth.make_batch(1, ns, ln, first_block=10_000_000)
    t0 = time.time()
    _, c1, _, _ = O.blocks_run_omp(cb, cso, cbo, None, p, 1, impl=impl)
    t_one = time.time() - t0
    rate_one = c1 / max(t_one, 1e-6)
    rounds = int(max(1, min(8, 15.0 / max(t_one * 1.5, 1e-3))))   # ~15 s of wall time if threads scale
    n_blk = threads * rounds
    bases, seq_off, blk_off = synth.make_batch(n_blk, ns, ln, first_block=10_000_001)
    t0 = time.time()
    _, cells, _, _ = O.blocks_run_omp(bases, seq_off, blk_off, None, p, threads, impl=impl)
    dt = time.time() - t0
    # scalar oracle: single-thread and all-thread rate on blocks cut to their first 8 sequences
    def cut(bases, seq_off, blk_off, k):
        keep, so, bo = [], [0], [0]
        for b in range(len(blk_off) - 1):
            for s in range(blk_off[b], min(blk_off[b] + k, blk_off[b + 1])):
                keep.append(bases[seq_off[s]:seq_off[s + 1]])
                so.append(so[-1] + int(seq_off[s + 1] - seq_off[s]))
            bo.append(len(so) - 1)
        return np.concatenate(keep), np.asarray(so, np.int64), np.asarray(bo, np.int32)
    k = min(8, ns)
    sb, sso, sbo = cut(bases, seq_off, blk_off[:threads + 1], k)
    t0 = time.time()
    _, sc1, _, _ = O.blocks_run_omp(*cut(cb, cso, cbo, k), None, p, 1, impl=O.IMPL_SCALAR)
    s_one = sc1 / max(time.time() - t0, 1e-6)
    t0 = time.time()
    _, sca, _, _ = O.blocks_run_omp(sb, sso, sbo, None, p, threads, impl=O.IMPL_SCALAR)
    s_all = sca / max(time.time() - t0, 1e-6)
    name = "AVX2 int16 row sweep (oracle/poa_simd.c)" if impl == O.IMPL_AVX2 else "scalar oracle (no AVX2 on this host)"
    return {"cells_per_s": cells / dt, "cores": threads, "seconds": dt,
            "per_thread_cells_per_s": cells / dt / threads, "single_thread_cells_per_s": rate_one,
            "scalar_single_thread_cells_per_s": s_one, "scalar_all_threads_cells_per_s": s_all,
            "cpu": cpu_model(), "physical_cores": phys, "cgroup_cpu_limit": quota, "implementation": name,
            "sample": "%d WHOLE blocks of the workload (%d sequences each, %d threads = min(physical cores, cgroup CPU quota) x %d rounds), %s, "
                      "per-thread workspaces, OpenMP schedule(dynamic,1); %.1f s, %.3g cells; single thread %.3g cells/s, "
                      "per thread under load %.3g cells/s; scalar oracle %.3g cells/s single / %.3g cells/s on %d threads"
                      % (n_blk, ns, threads, rounds, name, dt, cells, rate_one, cells / dt / threads, s_one, s_all, threads)}


def end_to_end(eng, bases, seq_off, blk_off, prm, mode, spoa_order=True):
    """One whole smoothing iteration through sxg_smooth_gfa (include/sxg_smooth.h) on the SAME blocks the kernel
    bench runs: host collection (A2-A4) -> upload -> POA kernels -> download -> block graphs (A9/A10) -> lacing ->
    validation -> unchop -> GFA text.  The input graph has one node per (block, sequence) and one path per
    sequence rank, the blockset comes in through sxg_blockset_from_ranges, padding is off (-O 0) so that the blocks
    reach the engine exactly as in the kernel-only measurement.  Returns the seconds of the second of two calls, the output
    size and the seconds of the first call."""
    import ctypes as C
    from smoothxg_amd import smooth as SM
    nb = len(blk_off) - 1
    depth = int(blk_off[1] - blk_off[0])
    lut = np.frombuffer(b"ACGTN", np.uint8)
    text = lut[bases].tobytes()
    lines = []
    for b in range(nb):
        for k in range(depth):
            s = int(blk_off[b]) + k
            lines.append(b"S\t%d\t%s\n" % (b * depth + k + 1, text[int(seq_off[s]):int(seq_off[s + 1])]))
    for k in range(depth):
        lines.append(b"P\thap%d\t%s\t*\n" % (k, b",".join(b"%d+" % (b * depth + k + 1) for b in range(nb))))
    gfa = b"".join(lines)
    blocks = [[(k, b, b + 1) for k in range(depth)] for b in range(nb)]
    sm = SM.Smoother(gfa, blocks=blocks)
    del gfa, lines, text
    p = SM.default_params(poa_m=prm[0], poa_n=-prm[1], poa_g=-prm[2], poa_e=-prm[3], poa_q=-prm[4], poa_c=-prm[5],
                          local_alignment=1 if mode == 0 else 0, poa_padding_fraction=0.0, poa_spoa_order=1 if spoa_order else 0)
    run, fre, ctx = SM.gpu_provider(eng)
    libc = C.CDLL("libc.so.6")
    libc.strlen.restype = C.c_size_t
    libc.strlen.argtypes = [C.c_void_p]
    times = []
    for _ in range(2):   # smoothxg runs its iterations (-l 700,900,1100) on one engine: the second call is the steady state
        out = C.c_void_p()
        t0 = time.perf_counter()
        rc = sm.L.sxg_smooth_gfa(sm.g, sm.b, C.byref(p), run, fre, ctx, C.byref(out))
        times.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError("sxg_smooth_gfa: " + sm.L.sxg_smooth_last_error().decode())
        n = libc.strlen(out)
        head = C.string_at(out, 12)
        sm.L.sxg_smooth_free(out)
        assert head.startswith(b"H\tVN:Z:1.0"), head
    sm.close()
    return times[1], int(n), times[0]


def source_hash():
    """SHA-256 of the kernel sources: profiles/<round>/counters.json records the hash of the build it measured."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smoothxg_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "sxg_poa.h"), "rb").read())
    return h.hexdigest()


def profile_counters(key):
    """Hardware-counter figures of the dominant kernel (per launch) from the newest profiles/rNN/counters.json, and
    whether they were measured on the sources this run was built from."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "counters.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if key in j.get("workloads", {}):
            best = (f, j)
    if not best:
        return None
    f, j = best
    w = dict(j["workloads"][key])
    w["file"] = os.path.relpath(f, ROOT)
    w["matches_build"] = j.get("source_sha256") == source_hash()
    return w

# fixture names of the blocks tests/golden/fullshape_oracle.json holds for a bench workload: (workload, mode) -> case name
FIXTURE_CASE = {("c2x8", "sw"): "c2", ("ns", "sw"): "ns_sw", ("ns", "nw"): "ns_nw", ("ns4", "nw"): "ns_nw_affine", ("c3", "sw"): "c3", ("c2", "sw"): "c2",
                ("c3a", "sw"): "c3a", ("c3b", "sw"): "c3b"}
BANDED_OF = {"c3b": 1, "c3a": 2}   # the `-A` path's workloads: static band / abPOA's adaptive band (sxg_poa_params::banded)


def digest_block(r):
    """SHA-256 digests of one downloaded block, as tests/golden/make_fullshape.py::digest_block computes them."""
    import hashlib

    def sha(a, dt):
        return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dt)).tobytes()).hexdigest()
    return {"node_code": sha(r.node_code, np.uint8), "node_rank": sha(r.node_rank, np.int32), "node_group": sha(r.node_group, np.int32),
            "edge_tail": sha(r.edge_tail, np.int32), "edge_head": sha(r.edge_head, np.int32), "edge_weight": sha(r.edge_weight, np.uint32),
            "paths": sha(np.concatenate([np.asarray(q, np.int32) for q in r.paths]) if len(r.paths) else np.zeros(0, np.int32), np.int32)}


def verify_against_fixture(res, workload, mode, first_block, prm, order="spoa"):
    """The timed batch checked against COMMITTED oracle output (tests/golden/fullshape_oracle.json: scores of every sequence,
    node / edge counts, cells, SHA-256 of nodes, ranks, groups, edges, weights, per-base paths) for the blocks of the batch
    the fixture holds -- blocks 0 and 999 of the headline and of config 2, 0 and 4999 of config 3.  No oracle code runs.
    Returns (verified, block ids, note); a mismatch raises."""
    name = FIXTURE_CASE.get((workload, mode))
    if name is None:
        return None, [], "no committed full-shape fixture for this workload / mode"
    cases = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "fullshape_oracle.json")))["cases"]
             if c["name"] == name and list(c["params"]) == list(prm) and c.get("order", "s7") == order and c.get("banded", 0) == BANDED_OF.get(workload, 0)]
    done = []
    for c in cases:
        k = c["block_id"] - first_block
        if k < 0 or k >= len(res):
            continue
        r = res[k]
        label = "bench verification failed: %s block %d: " % (name, c["block_id"])
        if r.status != 0:
            raise SystemExit(label + "status %d" % r.status)
        if [int(x) for x in r.scores] != c["scores"]:
            raise SystemExit(label + "scores differ from the committed oracle output")
        if int(np.asarray(r.cells, np.uint64).sum()) != c["cells"] or len(r.node_code) != c["n_nodes"] or len(r.edge_tail) != c["n_edges"]:
            raise SystemExit(label + "cells / node count / edge count differ")
        got = digest_block(r)
        for key, want in c["digests"].items():
            if key in got and got[key] != want:
                raise SystemExit(label + key + " differs from the committed oracle output")
        done.append(c["block_id"])
    if not done:
        return None, [], "none of the fixture's blocks is in this batch"
    return True, done, "scores of all sequences, cells, node/edge counts and SHA-256 of nodes, ranks, groups, edges, weights, paths == tests/golden/fullshape_oracle.json"


def bench_drb1(a, local_rank):
    """Configs 1 / 5 as the reference's ctest runs them (CMakeLists.txt:565): THREE chained smoothing iterations
    (-l 700,900,1100 -j 5k -e 5k -r 12) on test/data's DRB1-3123 seqwish GFA (committed as tests/golden/DRB1-3123.seqwish.gfa),
    every iteration = block discovery on the previous GFA + collection + ONE batched POA call + lacing + GFA text, consensus
    paths in the last.  A step is one whole chain; the value is its wall time in seconds (the only number the reference
    publishes: 23-25 s for its whole run, test/performance/check.md:7-28 -- that run also writes a MAF and builds the
    consensus graph, which this path does not, so vs_baseline stays null).  Every iteration's GFA is compared with the SHA-256
    pinned in tests/golden/drb1_chain.json."""
    import hashlib
    import smoothxg_amd as S
    from smoothxg_amd import smooth as SM
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "drb1_chain.json")))["iterations"]
    text0 = open(os.path.join(ROOT, "tests", "golden", "DRB1-3123.seqwish.gfa")).read()
    eng = S.PoaEngine(local_rank)
    prov = SM.gpu_provider(eng)

    def chain():
        text, per, shas, kms, blocks = text0, [], [], [], []
        for it, tl in enumerate((700, 900, 1100)):
            t0 = time.perf_counter()
            sm = SM.Smoother(text, discover=dict(target_poa_length=tl, n_haps=12, max_path_jump=5000, max_edge_jump=5000))
            blocks.append(sm.n_blocks)
            text = sm.smooth_gfa(SM.default_params(add_consensus=1 if it == 2 else 0), prov)
            sm.close()
            per.append(time.perf_counter() - t0)
            kms.append(eng.stats()["kernel_ms"])
            shas.append(hashlib.sha256(text.encode()).hexdigest())
        return per, shas, kms, blocks
    for _ in range(a.warmup):
        chain()
    t0 = time.perf_counter()
    runs = [chain() for _ in range(a.steps)]
    dt = time.perf_counter() - t0
    ok = all(r[1] == [g["sha256"] for g in gold] for r in runs)
    if not ok:
        raise SystemExit("bench verification failed: a DRB1 iteration's GFA differs from tests/golden/drb1_chain.json")
    per = [sum(r[0][k] for r in runs) / len(runs) for k in range(3)]
    kms = [sum(r[2][k] for r in runs) / len(runs) for k in range(3)]
    out = {"metric": "wall seconds of the reference's ctest chain on DRB1-3123 (three smoothing iterations)", "value": dt / a.steps, "unit": "s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": False, "scaling": "weak",
           "vs_baseline": None, "dtype": "int16", "data": "tests/golden/DRB1-3123.seqwish.gfa (the reference's test/data input)",
           "config": {"workload": WORKLOADS["drb1"][4], "blocks_per_iteration": runs[0][3], "mode": "sw"},
           "iteration_seconds": per, "iteration_kernel_ms": kms,
           "verified": True, "verified_what": "SHA-256 of every iteration's GFA == tests/golden/drb1_chain.json (oracle stack)",
           "reference_published": "23.4-25.5 s wall for the reference's whole run of this input on a Ryzen 7 3700X (test/performance/check.md:7-28; "
                                  "includes MAF output and the consensus graph, which are outside this path)",
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ns", choices=list(WORKLOADS))
    ap.add_argument("--mode", default="sw", choices=["sw", "nw"])
    ap.add_argument("--s7-order", action="store_true",
                    help="the node order of rounds 1-5 (decree S7: kept incrementally) instead of spoa's depth-first re-sort after every "
                         "sequence (decree S7', sxg_poa_params::mode | SXG_ORDER_SPOA: what src/smooth.cpp:764 does, the default since round 6); "
                         "prices the default against the cheaper order.  The `-A` workloads (c3a, c3b) always keep S7: abPOA does not call spoa's sort")
    ap.add_argument("--blocks", type=int, default=0, help="override the number of blocks per rank")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: ONE batch of --blocks blocks for all ranks, dealt by cost (shard.shard_batch, LPT) -- config 4's mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end sxg_smooth_gfa measurement")
    ap.add_argument("--check", action="store_true", help="also verify 2 blocks against the oracle")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed batch's fixture blocks with tests/golden/fullshape_oracle.json")
    ap.add_argument("--exchange", default="cabi", choices=["cabi", "torch"],
                    help="N > 1: the hand-off of every step's results to rank 0 -- cabi: sxg_poa_batch_execute_sharded (the C ABI's own RCCL "
                         "communicator: counts all-gathered, one grouped ncclSend/ncclRecv per peer); torch: shard.RootGather "
                         "(torch.distributed batch_isend_irecv on zero-copy views of the results)")
    ap.add_argument("--launch", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (what --gpus N > 1 does by itself when no launcher "
                         "started this process): the one-rank job then runs the multi-rank path -- RCCL communicator on the engine handle, "
                         "sxg_poa_batch_execute_sharded per step, `exchange` on the line")
    a = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # --gpus N is a PROMISE about what the line measures (round 5 parsed it and never read it: `python bench.py --gpus 8` timed one
    # GPU and said n_gpus 1).  Started without a launcher, N > 1 starts its own N ranks (one process per GPU, rendezvous on 127.0.0.1);
    # started by one (the driver: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N), the world must be N.
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.launch):
        n_dev = torch.cuda.device_count()
        if a.gpus > n_dev:
            raise SystemExit("bench.py --gpus %d: this box has %d GPU(s) visible -- refusing to time fewer GPUs than the line would claim" % (a.gpus, n_dev))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, SXG_BENCH_LAUNCHED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + [x for x in sys.argv[1:] if x != "--launch"]
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, env)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the line would not measure what it says (start it as "
                         "python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d, or without a launcher)" % (a.gpus, world, a.gpus, a.gpus))
    # multi: the multi-rank path (process group, communicator, sharded steps) -- also a ONE-rank job that came through the launcher
    multi = world > 1 or os.environ.get("SXG_BENCH_LAUNCHED") == "1"
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if multi:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import smoothxg_amd as S
    from smoothxg_amd import synth
    from smoothxg_amd import shard

    if a.workload == "drb1":
        if multi:
            raise SystemExit("--workload drb1 is a single-GPU measurement (2000 small blocks per iteration)")
        return bench_drb1(a, local_rank)
    nb, ns, ln, prm, desc = WORKLOADS[a.workload]
    if a.blocks:
        nb = a.blocks
    mode = 0 if a.mode == "sw" else 1
    spoa_order = not a.s7_order and a.workload not in BANDED_OF
    params = S.Params(*prm, mode | (0x10 if spoa_order else 0), BANDED_OF.get(a.workload, 0))
    strong = a.scaling == "strong" and world > 1
    eng = S.PoaEngine(local_rank)
    # N > 1: the engine's own RCCL communicator (sxg_poa_comm_init): rank 0 draws the id, torch.distributed carries it
    exchange = "none"
    if multi:
        exchange = a.exchange
        if exchange == "cabi":
            os.environ.setdefault("SXG_POA_COMM_TIMEOUT_S", "180")
            ok = 1
            try:
                ids = [eng.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                eng.comm_init(ids[0], world, rank)
            except Exception as e:   # (every rank must learn of it: the ranks switch together)
                print("[bench] rank %d: C-ABI communicator failed: %s" % (rank, e), file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                exchange = "torch (C-ABI communicator could not be created)"
    if strong and exchange == "cabi":
        # every rank hands the SAME batch to the C ABI, which deals it by cost (LPT) and uploads this rank's share
        bases, seq_off, blk_off = synth.make_batch(nb, ns, ln, first_block=0, mixed=(a.workload == "c4"))
        eng.upload_sharded(bases, seq_off, blk_off, None, params)
    else:
        if strong:
            # every rank builds the same batch and keeps its LPT share: block costs span four orders of magnitude
            # on config 4, so the deal -- not the count -- balances the GPUs
            bases, seq_off, blk_off = synth.make_batch(nb, ns, ln, first_block=0, mixed=(a.workload == "c4"))
            _, bases, seq_off, blk_off = shard.shard_batch(bases, seq_off, blk_off, rank, world)
        else:
            bases, seq_off, blk_off = synth.make_batch(nb, ns, ln, first_block=rank * nb, mixed=(a.workload == "c4"))
        eng.upload(bases, seq_off, blk_off, None, params)  # inputs resident in HBM from here on
    n_local = len(blk_off) - 1

    # The C-ABI exchange has never run with more than one real rank in the environment this was built in: the first
    # exchange is a probe.  A failure is returned by ALL ranks (the error code travels in the count exchange; a peer that
    # never answers ends in a timeout on every rank), so the ranks switch to the torch hand-off together.
    if exchange == "cabi":
        ok = 1
        try:
            eng.execute_sharded()
        except Exception as e:
            print("[bench] rank %d: C-ABI exchange failed: %s" % (rank, e), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            exchange = "torch (the C-ABI exchange failed in the probe step)"
            a.check = False
    to_root = shard.RootGather() if multi and exchange != "cabi" else None

    def step():
        if exchange == "cabi":
            # align the resident share, pack it into one device blob, sizes all-gathered, every blob straight to rank 0:
            # the one real exchange of the path (results meet on the rank that laces), through the C ABI's own communicator
            eng.execute_sharded()
            return
        eng.execute()
        if multi:
            # the same hand-off through torch.distributed: device to device, one exact-size message per peer over its
            # own xGMI link; nothing is all-gathered
            to_root(shard.engine_result_tensors(eng))

    def fence():
        torch.cuda.synchronize()
        if multi:
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
    if multi:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor([float(cells)], device="cuda", dtype=torch.float64)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        total_cells = float(cc.item())
    else:
        total_cells = float(cells)
    st = eng.stats()

    # The number printed below is worth something only if the batch it timed is right: the blocks of the batch that the
    # committed full-shape fixture holds are downloaded and compared (outside the timed region; no oracle code involved).
    verified, verified_blocks, verified_note = None, [], "skipped (--no-verify)"
    if rank == 0 and not a.no_verify:
        if strong and exchange == "cabi":
            verified_note = "strong scaling deals the blocks over the ranks: not checked here"
        elif FIXTURE_CASE.get((a.workload, a.mode)) is None:
            verified_note = "no committed full-shape fixture for this workload / mode"
        else:
            vres = eng.download()
            verified, verified_blocks, verified_note = verify_against_fixture(vres, a.workload, a.mode, 0, prm, "spoa" if spoa_order else "s7")
            del vres
    if a.check and rank == 0:
        from oracle import oracle_py as O
        res = eng.download_sharded() if (strong and exchange == "cabi") else eng.download()
        for b in (0, len(res) - 1):
            seqs = [bases[seq_off[s]:seq_off[s + 1]] for s in range(blk_off[b], blk_off[b + 1])]
            g, sc, _ = O.block_run(seqs, None, O.mkparams(*prm, mode=mode | (0x10 if spoa_order else 0), banded=BANDED_OF.get(a.workload, 0)))
            assert (res[b].scores == sc).all() and len(res[b].node_code) == g.n_nodes, "bench check failed"

    # What rank 0 does with a step's exchange before it can lace: the blobs of all ranks reassembled in batch order
    # (sxg_poa_batch_download_sharded).  Outside the timed region; only while the reassembled batch stays below ~12 GB of host memory.
    root_ms, root_note = None, "not measured"
    if multi and exchange == "cabi" and not strong:
        root_note = "weak scaling: every rank brought its own blocks, there is no common batch to put back in order (--scaling strong measures it)"
    elif multi and exchange == "cabi":
        est_bytes = 5.0 * float(len(bases))
        if est_bytes > 12e9 and not os.environ.get("SXG_BENCH_ROOT_DOWNLOAD"):
            root_note = "skipped: ~%.0f GB of reassembled results on rank 0 (SXG_BENCH_ROOT_DOWNLOAD=1 forces it)" % (est_bytes / 1e9)
        else:
            try:
                t_r = time.perf_counter()
                rres = eng.download_sharded()
                if rank == 0:
                    root_ms = (time.perf_counter() - t_r) * 1e3
                    root_note = "sxg_poa_batch_download_sharded of one step's results: %d blocks of %d ranks in batch order" % (len(rres), world)
                del rres
            except Exception as e:   # (a measurement beside the headline: never fails the line)
                root_note = "failed: %s" % e
    copy_gbps = None
    if rank == 0:
        try:
            copy_gbps = eng.measure_copy(1 << 30, 5)
        except Exception as e:   # (a figure beside the roofline: never fails the line)
            print("[bench] copy bandwidth not measured: %s" % e, file=sys.stderr)
    if rank == 0:
        blocks_total = (nb if strong else nb * world) * a.steps
        value = blocks_total / dt
        k_s = kernel_ms / 1e3
        algo_gbs = (algo_bytes / 1e9) / k_s if k_s > 0 else 0.0
        pc = profile_counters("%s_%s" % (a.workload, a.mode))
        # Round 4: a step may be SEVERAL launches side by side (the headline's two geometries are no longer merged into
        # the wider one; the mixed batch always was several).  Everything "per launch" below is per STEP: all dispatches of
        # one pass over the batch, timed by the HIP events around them (kernel_ms) -- the counters are collected the same way.
        launch_cells = cells / max(a.steps, 1)
        launches_per_step = launches / max(a.steps, 1)
        # the shader clock the dominant launch actually ran at (sampled by the engine from its slots: core-clock cycles per
        # 100 MHz wall tick); the boxes of the pool sustain 2.1-2.4 GHz under this kernel, the nominal value is the fallback
        clock_ghz = st["dom_clock_mhz"] / 1000.0 if st.get("dom_clock_mhz", 0) > 0 else VALU_CLOCK_GHZ
        # VALU roof: issue cycles the launch's instructions need at least / SIMD cycles the launch had.
        #   needed = 4 * (instructions - dual) + 2 * dual,  dual = SQ_ACTIVE_INST_VALU2 / 0.429 (see issue_costs)
        #   had    = 1024 SIMDs * measured clock * kernel time
        ic = issue_costs()
        simd_cycles_per_s = VALU_SIMDS * clock_ghz * 1e9
        traffic = valu_frac = valu_rate = None
        hbm = {"model": "SURVEY 8(d): 2*n_cross*sizeof(score)+1 bytes per cell", "algo_bytes_per_cell": algo_bytes / max(cells, 1),
               "algo_GBps": algo_gbs, "algo_frac_of_peak": algo_gbs / HBM_PEAK_GBS, "peak_GBps": HBM_PEAK_GBS,
               # what a plain streaming copy reaches on THIS box (sxg_poa_measure_copy: 1 GiB read + 1 GiB written per launch, HIP events)
               "copy_GBps_measured": copy_gbps,
               "copy_what": "device-to-device streaming copy kernel of the engine (16 B per lane, non-temporal), bytes read + written / HIP-event time"}
        valu = {"simds": VALU_SIMDS, "clock_GHz": clock_ghz,
                "clock_source": "measured in the launch (s_memtime cycles / s_memrealtime ticks)" if st.get("dom_clock_mhz", 0) > 0 else "nominal",
                "issue_cycles_half_rate": VALU_CYCLES_HALF_RATE, "issue_cycles_full_rate": VALU_CYCLES_FULL_RATE,
                "calibration": ic, "peak_issue_cycles_per_s": simd_cycles_per_s}
        if pc:
            # per-cell figures from the counter passes (one launch of the same workload), scaled to THIS run's cells
            ipc = pc["SQ_INSTS_VALU"] / pc["cells_per_launch"]
            dual = (pc.get("SQ_ACTIVE_INST_VALU2") or 0.0) / ic["valu2_per_full_rate_inst"] / pc["cells_per_launch"]   # dual-rate instructions per cell
            cyc_per_cell = VALU_CYCLES_HALF_RATE * (ipc - dual) + VALU_CYCLES_FULL_RATE * dual
            valu_rate = cyc_per_cell * cells / k_s                      # issue cycles needed per second of kernel time
            valu_frac = valu_rate / simd_cycles_per_s
            cyc_meas = ic["half_rate_cycles_measured"] * (ipc - dual) + ic["full_rate_cycles_measured"] * dual
            # columns the sweep's geometry covers against the columns the sequences use: padded columns are swept too
            swept = st["dom_threads"] * st["dom_cols_per_lane"]
            used = min(swept, (ln + 1) if ln else swept)
            valu.update({"wave_insts_per_cell": ipc, "wave_insts_per_launch": ipc * launch_cells,
                         "dual_rate_insts_per_launch": dual * launch_cells, "dual_rate_share": dual / ipc if ipc else 0.0,
                         "issue_cycles_per_cell": cyc_per_cell,
                         "frac_at_measured_opcode_costs": cyc_meas * cells / k_s / simd_cycles_per_s,
                         "frac_if_every_instruction_took_4_cycles": VALU_CYCLES_HALF_RATE * ipc * cells / k_s / simd_cycles_per_s,
                         "swept_columns": swept, "used_columns": used,
                         "useful_frac": valu_frac * used / swept if a.workload != "c4" and st["dom_row_mode"] == 2 else None,
                         "counter_file": pc["file"], "counters_match_build": pc["matches_build"]})
            if pc.get("hbm_bytes_per_launch"):
                bpc = pc["hbm_bytes_per_launch"] / pc["cells_per_launch"]
                traffic = bpc * launch_cells if pc["matches_build"] else None
                hbm.update({"counter_bytes_per_cell": bpc, "counter_GBps": bpc * cells / k_s / 1e9,
                            "counter_frac_of_peak": bpc * cells / k_s / 1e9 / HBM_PEAK_GBS,
                            "counter_correction": pc.get("correction")})
        # The recurrence's own instruction count: convex local alignment, two-pass in-row scan, per
        # packed column of a lane -- pass 1 8.5 (score look-up 1.5, diagonal add, max with F and O, two carry steps of add +
        # max), pass 2 9 (max with E, Q and 0; E and Q updates of two adds + max), outgoing candidates 6, the wave scans
        # 40 per row / W columns -- against the wave instructions executed per cell (a packed wave instruction touches
        # 128 cells, so "instructions per cell" = wave instructions x 128 / cells, the unit of the counters above).
        # Round 6: the floor follows the GAP MODEL of the score set (spoa's Create rule, poa_kernels.hip.h::normalise): an affine
        # recurrence has no O / Q states -- pass 1 5.5 (look-up 1.5, diagonal add, max with F, one carry step), pass 2 5 (max with
        # E and 0, E update), outgoing candidate 3, one wave scan of 20 per row / W --, a linear one (g = e) also folds its gap
        # states into H + g: pass 2 4, outgoing candidate 1.  (Rounds 1-5 priced every model with the convex count.)
        if pc and st["dom_row_mode"] == 2:
            wl_w = max(st["dom_cols_per_lane"] // 2, 1)
            g_, e_, q_, c_ = prm[2], prm[3], prm[4], prm[5]
            gap_model = "linear" if g_ >= e_ else ("affine" if (g_ <= q_ or e_ >= c_) else "convex")
            min_ipc = {"convex": 23.5 + 40.0 / wl_w, "affine": 13.5 + 20.0 / wl_w, "linear": 10.5 + 20.0 / wl_w}[gap_model]
            exe_ipc = valu["wave_insts_per_cell"] * 128.0
            valu.update({"gap_model": gap_model, "min_insts_per_cell": min_ipc, "executed_insts_per_cell": exe_ipc,
                         "algorithmic_frac": (min_ipc / exe_ipc) * valu_frac if exe_ipc else None,
                         "insts_per_cell_unit": "wave instructions x 128 cells / cells of the step"})
        valu_peak = simd_cycles_per_s
        # Which roof binds: the one the launch sits closest to.  The packed full-matrix sweeps of the headline and of config 2
        # are bound by VALU issue; config 3's full matrix and the banded (-A) sweeps move more bytes per instruction and sit
        # closer to the HBM roof (counter bytes: FETCH_SIZE / WRITE_SIZE in their own passes, corrected as the guide prescribes).
        hbm_counter_frac = hbm.get("counter_frac_of_peak")
        hbm_bound = hbm_counter_frac is not None and (valu_frac is None or hbm_counter_frac > valu_frac)
        roof = {"bound": "hbm" if (hbm_bound or valu_frac is None) else "valu"}
        if roof["bound"] == "valu":
            roof.update({"achieved": valu_rate / 1e9, "peak": valu_peak / 1e9, "unit": "G VALU issue cycles/s", "frac": valu_frac})
        elif hbm_bound:
            roof.update({"achieved": hbm["counter_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_counter_frac,
                         "achieved_is": "HBM bytes the counters measured per second of kernel time (the 8(d) model's bytes are in hbm_8d_frac)"})
        else:
            roof.update({"achieved": algo_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo_gbs / HBM_PEAK_GBS,
                         "achieved_is": "SURVEY 8(d) algorithmic bytes per second (no counters committed for this workload)"})
        roof.update({
            # every roof side by side, whatever binds:
            "valu_frac": valu_frac,                                   # issue cycles of the executed VALU instructions / SIMD cycles
            "algorithmic_frac": valu.get("algorithmic_frac"),         # ... counting only the recurrence's own instructions
            "hbm_counter_frac": hbm_counter_frac,                     # measured HBM bytes / s over the 8 TB/s peak
            "hbm_8d_frac": algo_gbs / HBM_PEAK_GBS,                   # SURVEY 8(d)'s 13 (9) B per cell / s over the peak
            "hbm_8d_frac_is": "NOT an achieved fraction and not binding: rows whose only predecessor is the previous rank stay in "
                              "registers and short-lived rows in LDS, so the sweep moves fewer bytes than the 8(d) model charges",
            "traffic": traffic,
            "kernel": "poa_block_kernel<T=%d, cols/lane=%d, %s>" % (
                st["dom_threads"], st["dom_cols_per_lane"],
                {2: "packed int16 sweep", 3: "banded packed int16 sweep (one wave, sliding window)"}.get(st["dom_row_mode"], "32-bit sweep")),
            "kernel_ms_per_launch": kernel_ms / max(a.steps, 1), "kernel_ms_total": kernel_ms,
            "launches_per_step": launches_per_step,
            "per_launch_means": "per step: every dispatch of one pass over the batch (launches of different geometries run side by side)",
            "algo_bytes_per_launch": algo_bytes / max(a.steps, 1),
            "bytes_per_cell": algo_bytes / max(cells, 1),
            "valu": valu, "hbm": hbm})
        out = {
            "metric": "POA blocks/sec (+ DP cells/sec) on 1000-block synthetic",
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "int16" if st["dom_row_mode"] >= 2 else "int32", "data": "synthetic",
            "config": {"workload": desc, "blocks_per_gpu": nb, "mode": a.mode, "order": "spoa (S7': depth-first re-sort after every sequence, src/smooth.cpp:764)" if spoa_order else "incremental (S7)",
                       "cells_per_step_per_gpu": cells / a.steps},
            "cells_per_sec": total_cells / dt,
            "verified": verified, "verified_blocks": verified_blocks, "verified_what": verified_note,
            # The binding roof of this kernel is VALU issue, not HBM: the sweep keeps adjacent-rank rows in
            # registers and moves about half of SURVEY 8(d)'s 13 B/cell, so the 8(d) figure alone exceeds the HBM
            # peak (hbm.algo_frac_of_peak > 1 is NOT an achieved fraction).  frac = architectural issue cycles of the
            # VALU instructions executed per second / SIMD cycles per second (1024 SIMDs x measured clock);
            # hbm.counter_* is what FETCH_SIZE/WRITE_SIZE measured.
            "roofline": roof,
            "engine": {"slots": st["n_slots"], "retries": st["retries"], "arena_bytes": st["device_bytes"],
                       "roctx_ranges": bool(eng.roctx_available())},
            "exchange": ({"path": "C ABI: sxg_poa_batch_execute_sharded (RCCL all-gather of sizes + grouped ncclSend/ncclRecv to rank 0)"
                          if exchange == "cabi" else exchange, **(eng.sharded_info() if exchange == "cabi" else {}),
                          "rank0_reassembly_ms": root_ms, "rank0_reassembly_what": root_note,
                          "launcher": "bench.py started its own ranks" if os.environ.get("SXG_BENCH_LAUNCHED") == "1" else "started by a launcher"} if multi else None),
        }
        if world == 1 and not a.no_e2e and a.workload in ("ns", "c2", "c2x8", "tiny"):
            e_s, e_bytes, e_first = end_to_end(eng, bases, seq_off, blk_off, prm, mode, spoa_order)
            out["end_to_end"] = {"what": "sxg_smooth_gfa on the same %d blocks: host collection + upload + POA kernels + "
                                         "download + block graphs + lacing + validation + unchop + GFA text (padding off)" % nb,
                                 "seconds": e_s, "first_call_seconds": e_first,
                                 "calls": "two calls on one engine, `seconds` is the second (the first also pins the download buffer and "
                                          "faults in the host arenas)",
                                 "blocks_per_sec": nb / e_s, "gfa_bytes": e_bytes,
                                 "kernel_only_blocks_per_sec": nb * a.steps / (kernel_ms / 1e3),
                                 "ratio_to_kernel_only": (nb / e_s) / (nb * a.steps / (kernel_ms / 1e3))}
        if world == 1 and not a.no_cpu_baseline and a.workload not in ("c4", "c3b", "c3a"):  # (no fixed shape to sample for c4)
            cb = cpu_baseline(a.workload, mode, spoa_order)
            cells_per_block = cells / a.steps / nb
            out["cpu_baseline"] = {"value": cb["cells_per_s"] / cells_per_block, "unit": "blocks/s",
                                   "cores": cb["cores"], "kind": "port", "sample": cb["sample"],
                                   "cells_per_sec": cb["cells_per_s"], "cpu": cb["cpu"],
                                   "physical_cores": cb["physical_cores"], "cgroup_cpu_limit": cb["cgroup_cpu_limit"],
                                   "implementation": cb["implementation"],
                                   "per_thread_cells_per_sec": cb["per_thread_cells_per_s"],
                                   "single_thread_cells_per_sec": cb["single_thread_cells_per_s"],
                                   "scalar_single_thread_cells_per_sec": cb["scalar_single_thread_cells_per_s"],
                                   "scalar_all_threads_cells_per_sec": cb["scalar_all_threads_cells_per_s"],
                                   "gpu_over_cpu": (total_cells / dt) / cb["cells_per_s"]}
        print(json.dumps(out))
    if multi:
        # Every rank has printed / handed over what it had: give the communicators back explicitly and leave without the
        # libraries' own teardown -- a process that created an RCCL communicator through the C ABI can abort in it after main
        # has returned ("Observed with rccl 2.27.7"), which torchrun would report as a failed rank.
        eng.close()
        dist.barrier()
        dist.destroy_process_group()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
