import sys, numpy as np
sys.path.insert(0, '.')
from smoothxg_amd import synth, poa
bases, seq_off, blk_off = synth.make_batch(1000, 64, 5000)
eng = poa.PoaEngine(0)
P = poa.params_from_cli(1, 4, 6, 2, 26, 1, local=True)
res = eng.run_flat(bases, seq_off, blk_off, None, P) if hasattr(eng, 'run_flat') else None
print(type(res))
cells = None
for name in ('cells',):
    if hasattr(res, name): cells = getattr(res, name)
if cells is None and isinstance(res, (list, tuple)):
    cells = np.concatenate([np.asarray(r.cells) for r in res])
cells = np.asarray(cells, np.float64)
per = np.add.reduceat(cells, blk_off[:-1])
print('per-block cells: min %.3g p10 %.3g med %.3g p90 %.3g max %.3g  min/max %.3f mean/max %.3f' % (per.min(), np.percentile(per,10), np.median(per), np.percentile(per,90), per.max(), per.min()/per.max(), per.mean()/per.max()))
