"""Kernel resource table (SGPRs, VGPRs, scratch, occupancy, static LDS) from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python3 profiles/resource_table.py > profiles/rNN_kernel_resource_usage.txt   (runs in the dev container: no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
       "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "vegasafterglow_amd", "csrc", "vag_capi.hip"), "-o", "/tmp/vag_resource_probe.so"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(\w[\w \[\]/]*?): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
print("kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)")
print("%-90s %6s %6s %8s %5s %8s" % ("kernel", "SGPRs", "VGPRs", "scratch", "occ", "LDS"))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)[:88]
    print("%-90s %6s %6s %8s %5s %8s" % (name, r.get("TotalSGPRs", "?"), r.get("VGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                          r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
