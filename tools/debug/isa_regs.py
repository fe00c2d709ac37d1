#!/usr/bin/env python3
"""registers / spills of the kernels of an --save-temps .s file whose mangled name contains every given substring:
   tools/debug/isa_regs.py file.s render_fast_kernel Lb1ELb1E"""
import re
import sys

txt = open(sys.argv[1]).read()
subs = sys.argv[2:]
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    if all(s in name for s in subs):
        f = dict(re.findall(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", body))
        print(name[-70:], f)
