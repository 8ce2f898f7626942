#!/usr/bin/env python
"""Registers / scratch / LDS / occupancy of every kernel, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py file.res [...]   (file.res = stderr of the compile)"""
import re
import sys

for path in sys.argv[1:]:
    text = open(path).read()
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = blk.split("\n")[0].strip()

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", blk)
            return m.group(1) if m else "?"
        print("%-58s vgpr %3s agpr %3s scratch %4s occ %s lds %6s" % (
            name[:58], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
            g("LDS Size [bytes/block]")))
