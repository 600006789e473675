"""Per-kernel register / scratch / LDS usage from a `hipcc -Rpass-analysis=kernel-resource-usage` log.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> log; python tools/kernel_resources.py log [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    try:
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        d = name
    d = d.replace("ddcn::", "").replace("(anonymous namespace)::", "")
    if flt and flt not in d:
        continue
    print("%-100s vgpr %3d agpr %3d scratch %4d occ %d lds %6d" % (d[:100], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                                  g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
