"""Development aid: one block per packed geometry, prints the geometry that ran and whether the result equals the oracle
(no asserts) -- the loop of tests/test_gpu_parity.py::test_every_packed_strip_width_and_wave_count without stopping."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ.setdefault("SXG_POA_NO_SPREAD", "1")
import numpy as np
from helpers import assert_block_equal, gparams, oparams, random_block
from test_gpu_parity import _packed_geometry
import smoothxg_amd as S
from oracle import oracle_py
oracle_py.lib()
eng = S.PoaEngine(0)
rng = np.random.default_rng(41)
for W in [int(x) for x in os.environ.get('DBG_W', '4,5,6,7,8,9,10,11,12').split(',')]:
    for NW in [int(x) for x in os.environ.get('DBG_NW', '1,2,3,4,8').split(',')]:
        L = 128 * NW * W - 3
        cols, T, cpl = _packed_geometry(L)
        if (T, cpl) != (64 * NW, 2 * W):
            continue
        seqs = [s[:L] for s in random_block(rng, int(os.environ.get('DBG_SEQS', '3')), L, div=0.02)]
        res = eng.run_blocks([seqs], gparams("convex_default", 0))
        st = eng.stats()
        g, sc, cells = oracle_py.block_run(seqs, None, oparams("convex_default", 0))
        try:
            assert_block_equal(res[0], g, sc, cells, label="x")
            ok = "equal"
        except AssertionError as e:
            ok = "DIFFERENT " + str(e)[:80]
        print(f"W={W} NW={NW} L={L}: ran rm={st['dom_row_mode']} T={st['dom_threads']} cpl={st['dom_cols_per_lane']} launches={st['dp_launches']} retries={st['retries']} {ok}", flush=True)
