#!/usr/bin/env python3
"""Static instruction budget of one kernel from the device assembly (developer aid).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --cuda-device-only -S vegasafterglow_amd/csrc/vag_capi.hip -o /tmp/dev.s
  python profiles/isa_budget.py /tmp/dev.s 'vag_flux_grid_kernelILb0ELi0ELb0ELi512ELb0E'

Prints every basic block (label) of the kernel with its instruction counts by class -- VALU (FP64 FMA-class, other), LDS, global /
scratch memory, scalar, waits, barriers, branches -- and the label it loops back to, so that a loop body's cost can be read off.
"""
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:.*", l))
name = lines[start][:-1]
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], None
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        d = re.search(r"Depth=(\d+)", l)
        cur = {"label": m.group(1), "ins": [], "depth": d.group(1) if d else "-"}
        blocks.append(cur)
        continue
    t = l.strip()
    if cur is not None and cur["depth"] == "-" and "Depth=" in t and t.startswith(";"):
        d = re.search(r"Depth=(\d+)", t)
        cur["depth"] = d.group(1)
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur is None:
        cur = {"label": "entry", "ins": [], "depth": "0"}
        blocks.append(cur)
    cur["ins"].append(t.split(";")[0].strip())


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64"):
        return "f64"
    if op.startswith("v_") and "f64" in op:
        return "f64x"  # rcp / rndne / ldexp / cmp / min / max / cvt on doubles
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


print(name)
keys = ["f64", "f64x", "valu", "lds", "vmem", "scratch", "salu", "wait", "barrier", "branch"]
print(f"{'block':>12} dep " + " ".join(f"{k:>7}" for k in keys) + "   total  -> branch targets")
tot = {k: 0 for k in keys}
for b in blocks:
    c = {k: 0 for k in keys}
    targets = []
    for ins in b["ins"]:
        op = ins.split()[0]
        k = classify(op)
        if k in c:
            c[k] += 1
            tot[k] += 1
        if k == "branch":
            targets.append(ins.split()[-1])
    n = sum(c.values())
    if n:
        print(f"{b['label']:>12} {b['depth']:>3} " + " ".join(f"{c[k]:7d}" for k in keys) + f" {n:7d}  {' '.join(targets)}")
print(f"{'TOTAL':>12}     " + " ".join(f"{tot[k]:7d}" for k in keys) + f" {sum(tot.values()):7d}")
