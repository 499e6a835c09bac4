"""Summarise SXG_POA_SLOT_CSV: where the hardware put the workgroups and how long each ran."""
import sys, collections
rows = [l.strip().split(',') for l in open(sys.argv[1]) if l.strip()]
t0 = min(int(r[1]) for r in rows)
tend = max(int(r[2]) for r in rows)
span = tend - t0
percu = collections.defaultdict(list)
simd_patterns = collections.Counter()
for r in rows:
    st, en = int(r[1]) - t0, int(r[2]) - t0
    waves = [int(x, 16) for x in r[3:]]
    smid = waves[0] >> 32
    percu[smid].append((st, en))
    simds = tuple(sorted(((w & 0xffffffff) >> 4) & 3 for w in waves))
    simd_patterns[simds] += 1
print("slots", len(rows), "CUs used", len(percu), "span(100MHz ticks)", span)
cnt = collections.Counter(len(v) for v in percu.values())
print("workgroups per CU histogram:", sorted(cnt.items()))
print("SIMD placement of a workgroup's waves:", simd_patterns.most_common(6))
for k in sorted(cnt):
    ends = [en / span for v in percu.values() if len(v) == k for (_, en) in v]
    starts = [st / span for v in percu.values() if len(v) == k for (st, _) in v]
    print(f"  CUs with {k} WGs: mean start {sum(starts)/len(starts):.3f} mean end {sum(ends)/len(ends):.3f} min end {min(ends):.3f} max end {max(ends):.3f}")
