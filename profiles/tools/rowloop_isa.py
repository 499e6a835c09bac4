#!/usr/bin/env python3
"""Static look at the row loop of a packed-sweep block kernel: compiles sxg_poa.hip for ONE kernel class to
ISA, finds the row loop (the deepest loop holding two s_barrier; without barriers -- round 3 -- the outermost loop), and prints per basic block the number of
VALU / SALU / LDS / VMEM instructions and every scratch access inside the loop.

    python profiles/tools/rowloop_isa.py [W=11] [TMAX=256] [kernel substring]
"""
import re
import subprocess
import sys

W = sys.argv[1] if len(sys.argv) > 1 else "11"
TM = sys.argv[2] if len(sys.argv) > 2 else "256"
KSUB = sys.argv[3] if len(sys.argv) > 3 else "dp_fill_p16ILi%sELb1ELb1E" % W
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                       "-DSXG_DEV_ONLY_W=" + W, "-DSXG_DEV_ONLY_TMAX=" + TM, "-S", "--cuda-device-only",
                       "-o", "/tmp/rowloop.s", "smoothxg_amd/csrc/sxg_poa.hip"] + sys.argv[4:],
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/rowloop.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and KSUB in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i] or "s_setpc_b64 s[30:31]" in lines[i])
body = lines[start:end]
# blocks
blocks, cur = [], None
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"name": m.group(1), "depth": 0, "ins": [], "hdr": ""}
        blocks.append(cur)
        continue
    if cur is None:
        continue
    m = re.search(r"Depth=(\d+)", l)
    if m and l.strip().startswith(";"):
        cur["depth"] = max(cur["depth"], int(m.group(1)))
        cur["hdr"] += l.strip() + " "
        continue
    t = l.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ins"].append(t)
# the row loop: deepest depth that contains >= 2 barriers at exactly that depth
by_depth = {}
for b in blocks:
    by_depth.setdefault(b["depth"], []).append(b)
cands = [d for d, bs in by_depth.items() if sum(i.startswith("s_barrier") for b in bs for i in b["ins"]) >= 2]
# (round 3: the packed sweep's waves no longer meet at barriers inside the row loop, and the one-wave banded sweep never
#  did: the row loop is then the outermost loop of the function)
row_depth = max(cands) if cands else 1
tot = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "scratch": 0}
print("row loop depth", row_depth)
for b in blocks:
    if b["depth"] < row_depth:
        continue
    c = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "scratch": 0}
    for i in b["ins"]:
        op = i.split()[0]
        if op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
            c["vmem"] += 1
    for k in tot:
        tot[k] += c[k]
    flag = " BARRIER" if any(i.startswith("s_barrier") for i in b["ins"]) else ""
    print("%-12s d%d valu %4d salu %4d lds %3d vmem %3d scratch %2d%s" % (b["name"], b["depth"], c["valu"], c["salu"], c["lds"], c["vmem"], c["scratch"], flag))
    for i in b["ins"]:
        if i.startswith("scratch_"):
            print("      ", i)
print("total in loop", tot)
import collections
hist = collections.Counter(i.split()[0] for b in blocks if b["depth"] >= row_depth for i in b["ins"])
print("opcodes:", ", ".join("%s %d" % kv for kv in hist.most_common(14)))
