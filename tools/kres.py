"""Per-kernel register / scratch summary of a hipcc -Rpass-analysis=kernel-resource-usage log (developer tool).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip -o /tmp/x.o 2> res.txt ; python tools/kres.py res.txt"""
import re
import sys

t = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-64s VGPR %3s AGPR %3s scratch %4s spill %3s occ %s" % (name[:64], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                                   g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
