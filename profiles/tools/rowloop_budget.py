#!/usr/bin/env python3
"""Static instruction budget of the packed sweep's row loop, per basic block in program order (round 6; replaces the loop
detection of rowloop_isa.py, which went stale when the waves stopped meeting at barriers).

    python profiles/tools/rowloop_budget.py [W=11] [TMAX=256] [function-name substring]

Compiles sxg_poa.hip for ONE packed class (-DSXG_DEV_ONLY_W / _TMAX) to ISA, takes dp_fill_p16 of the class the headline runs
(convex, local, 2-byte cells, exact thread count, default scores), and prints every basic block between the loop header and
the back edge: VALU / SALU / LDS / VMEM instructions, s_nop, v_readlane + v_writelane (SGPR spills to VGPR lanes and the wave
scans' lane reads), v_mov, and where the block branches to.  The blocks of a row's path add up to its instruction budget."""
import re
import subprocess
import sys

W = sys.argv[1] if len(sys.argv) > 1 else "11"
TM = sys.argv[2] if len(sys.argv) > 2 else "256"
KSUB = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "dp_fill_p16ILi%sELb1ELb1ELi2ELi%sELi%sELb1E" % (W, "2" if TM == "64" and int(W) <= 11 else ("1" if int(TM) <= 128 else "0"), TM)
out = "/tmp/rowloop_budget_%s_%s.s" % (W, TM)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DSXG_DEV_ONLY_W=" + W,
                       "-DSXG_DEV_ONLY_TMAX=" + TM, "-S", "--cuda-device-only", "-o", out, "smoothxg_amd/csrc/sxg_poa.hip"] + sys.argv[4:],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and KSUB in l and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], {"name": "entry", "ins": [], "depth": 0}
blocks.append(cur)
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"name": m.group(1), "ins": [], "depth": 0}
        blocks.append(cur)
        continue
    t = l.strip()
    m = re.search(r"Depth=(\d+)", t)
    if t.startswith(";") and m:
        cur["depth"] = max(cur["depth"], int(m.group(1)))
        continue
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ins"].append(t.split(";")[0].strip())


def mix(ins):
    c = dict(valu=0, salu=0, lds=0, vmem=0, nop=0, lane=0, mov=0, wait=0)
    for i in ins:
        op = i.split()[0]
        if op == "s_nop":
            c["nop"] += 1
        elif op.startswith("s_waitcnt"):
            c["wait"] += 1
        elif op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
            c["lane"] += 1
            c["valu"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if op.startswith("v_mov"):
                c["mov"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    return c


tot = None
# the row loop: from the first block LLVM marks as inside a loop (the header) to the last block that branches back to it
hdr = next(i for i, b in enumerate(blocks) if b["depth"] >= 1)
back = max(i for i, b in enumerate(blocks) if any(x.startswith(("s_cbranch", "s_branch")) and x.split()[-1] == blocks[hdr]["name"] for x in b["ins"]))
print("row loop: %s .. %s" % (blocks[hdr]["name"], blocks[back]["name"]))
print("%-12s %2s %5s %5s %4s %4s %4s %5s %4s  %s" % ("block", "d", "valu", "salu", "lds", "vmem", "nop", "lane", "mov", "ends with"))
for b in blocks[hdr:back + 1]:
    c = mix(b["ins"])
    br = [i for i in b["ins"] if i.startswith(("s_cbranch", "s_branch"))]
    print("%-12s %2d %5d %5d %4d %4d %4d %5d %4d  %s" % (b["name"], b["depth"], c["valu"], c["salu"], c["lds"], c["vmem"], c["nop"], c["lane"], c["mov"],
                                                        "; ".join(x.replace("s_cbranch_", "").replace("s_branch", "->") for x in br)))
    tot = c if tot is None else {k: tot[k] + c[k] for k in c}
print("all blocks inside loops:", tot)
# register footprint of the function (a spill inside the row loop shows up as scratch_ instructions above)
for l in lines[end:end + 60]:
    if "NumVgprs" in l or "ScratchSize" in l or "NumSgprs" in l or "Occupancy" in l:
        print(l.strip())
