import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle_py as O
from smoothxg_amd import synth
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("affinity", len(os.sched_getaffinity(0)))
p = O.mkparams()
for shape in ((16, 1000), (32, 3000)):
    nmax = 128
    bases, so, bo = synth.make_batch(nmax, shape[0], shape[1], first_block=777)
    for impl in (1, 0):
        for th in (1, 2, 4, 8, 16, 32, 64, 128):
            nb = th
            t = time.time()
            _, cells, _, _ = O.blocks_run_omp(bases, so[:bo[nb] + 1], bo[:nb + 1], None, p, th, impl=impl)
            dt = time.time() - t
            print("shape", shape, "impl", impl, "threads", th, "per-thread Mcells/s %.0f total %.1f G" % (cells / dt / th / 1e6, cells / dt / 1e9), flush=True)
