"""Dev tool: registers / spills / LDS of every kernel in a `hipcc -save-temps` gfx950 .s file:  python tools/kmeta.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
    name, blk = m.group(1), m.group(2)
    if flt not in name:
        continue
    def g(k):
        r = re.search(k + r":\s+(\d+)", blk)
        return r.group(1) if r else "-"
    print(name[:70], "vgpr", g(r"\.vgpr_count"), "spill", g(r"\.vgpr_spill_count"), "lds", g(r"\.group_segment_fixed_size"), "scratch", g(r"\.private_segment_fixed_size"))
