import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

def run(pname, mode, batch):
    from helpers import PARAM_SETS, oparams, gparams, random_block
    from oracle import oracle_py as O
    import smoothxg_amd as S
    eng = S.PoaEngine(0)
    rng = np.random.default_rng(7 + mode)
    probs, exp = [], []
    for trial in range(12):
        Sn = int(rng.integers(2, 9)); L = int(rng.integers(8, 300))
        seqs = random_block(rng, Sn, L, div=0.08)
        p = oparams(pname, mode)
        g, _, _ = O.block_run(seqs[:-1], None, p)
        codes, off, pred, sink, row_node = g.rows()
        q = seqs[-1]
        an, ap, sc = O.align_csr(codes, off, pred, sink, q, p)
        probs.append((codes, off, pred, sink, q)); exp.append((an, ap, sc))
    if batch:
        got = eng.align(probs, gparams(pname, mode))
    else:
        got = [eng.align([pb], gparams(pname, mode))[0] for pb in probs]
    for k, ((pr, pp, gsc, st), (an, ap, sc)) in enumerate(zip(got, exp)):
        ok = gsc == sc and len(pr) == len(an) and (pr == an).all() and (pp == ap).all()
        if not ok:
            codes, off, pred, sink, q = probs[k]
            npred = np.diff(off)
            print("  MISMATCH", pname, mode, "prob", k, "N", len(codes), "L", len(q), "score", gsc, sc, "npairs", len(pr), len(an), "maxnp", npred.max())
            n = min(len(pr), len(an))
            bad = [x for x in range(n) if pr[x] != an[x] or pp[x] != ap[x]]
            if bad:
                x = bad[0]
                print("    first diff at pair", x, "gpu", pr[max(0,x-2):x+3], pp[max(0,x-2):x+3], "ora", an[max(0,x-2):x+3], ap[max(0,x-2):x+3])
                r = an[x] if an[x] >= 0 else an[x-1]
                print("    row", r, "np", npred[r], "preds", pred[off[r]:off[r+1]], "rank-1?", r)
    print("ran", pname, mode, "batch" if batch else "single", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    else:
        for pname in ["convex_default", "affine_4param", "linear", "convex_heavy", "adaptive_tier"]:
            for mode in (0, 1):
                for batch in (1,):
                    r = subprocess.run([sys.executable, __file__, pname, str(mode), str(batch)], capture_output=True, text=True, env=dict(os.environ, SXG_POA_DEBUG='1'))
                    out = (r.stdout + r.stderr)
                    lines = [l for l in out.split("\n") if l and "GPU core" not in l and "Failed to write" not in l and "amdgpu.ids" not in l]
                    print("\n".join([l for l in lines if "MISMATCH" in l or l.startswith("ran") or "rror" in l]))
