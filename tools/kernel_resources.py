#!/usr/bin/env python3
"""Per-kernel register / spill table from `hipcc ... -Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile):
   hipcc --offload-arch=gfx950 -O3 ... -c k.hip -o k.o -Rpass-analysis=kernel-resource-usage 2> res.txt ; python tools/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    rows.append((name, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%-90s %5s %5s %5s %7s %6s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "Vspill", "Sspill", "occ"))
for n, r in zip(names, rows):
    n = re.sub(r"\(.*$", "", n.replace("void ", ""))
    if flt in n:
        print("%-90s %5s %5s %5s %7s %6s %6s %4s" % ((n[:90],) + r[1:]))
