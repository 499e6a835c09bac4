#!/usr/bin/env python3
"""End-to-end phase times of sxg_smooth_gfa on a bench workload without the kernel-only steps:
    SXG_SMOOTH_TIMING=2 SXG_POA_DEBUG=1 python profiles/tools/e2e_only.py --workload ns --runs 2
prints the wall time of every run (the phase laps go to stderr)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ns")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--runs", type=int, default=2)
    a = ap.parse_args()
    import smoothxg_amd as S
    from smoothxg_amd import synth
    nb, ns, ln, prm, desc = bench.WORKLOADS[a.workload]
    nb = a.blocks or nb
    bases, seq_off, blk_off = synth.make_batch(nb, ns, ln)
    eng = S.PoaEngine(0)
    for r in range(a.runs):
        dt, n, first = bench.end_to_end(eng, bases, seq_off, blk_off, prm, 0)   # (two calls each: first, second)
        st = eng.stats()
        print("run %d: %.3f s end to end (first call %.3f s), kernels %.3f s, block-graph kernel %.1f ms, ratio %.3f, %d bytes" %
              (r, dt, first, st["kernel_ms"] / 1e3, st["bg_ms"], st["kernel_ms"] / 1e3 / dt, n), flush=True)
        print("---", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
