#!/usr/bin/env python3
"""Print VGPR / SGPR / spill / scratch / LDS per kernel from a hipcc -S device assembly file."""
import re, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print("%-90s vgpr=%s agpr=%s sgpr=%s vspill=%s sspill=%s scratch=%s lds=%s" % (name[:90], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
