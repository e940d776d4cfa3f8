#!/usr/bin/env python3
"""Scan gfx950 assembly for MFMAs whose destination registers overlap their A or B operand.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s kernel.hip && python tools/mfma_overlap_check.py k.s

On MI355X a multi-pass MFMA writes its result while later passes still read B: `v_mfma_f32_16x16x4_f32 v[46:49], v33, v49,
v[58:61]` (the register allocator re-used the B operand's register for the result) gave wrong columns 12-15.  The compiler
does not forbid the overlap for every variant, so kernels keep the B operands alive past the MFMA chain and this script
checks the generated code."""
import re
import sys


def rng(op):
    m = re.match(r"([va])\[(\d+):(\d+)\]", op)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r"([va])(\d+)$", op)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None


bad = 0
for path in sys.argv[1:]:
    kernel = "?"
    for ln, line in enumerate(open(path), 1):
        if line.startswith("_Z") and line.rstrip().endswith(":"):
            kernel = line.strip()[:-1]
        m = re.match(r"\s+(v_mfma\S+)\s+(.*)", line)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        d, a, b = rng(ops[0]), rng(ops[1]), rng(ops[2])
        for name, s in (("A", a), ("B", b)):
            if d and s and d[0] == s[0] and not (d[2] < s[1] or s[2] < d[1]):
                bad += 1
                print(f"{path}:{ln}: {kernel[:70]}: dst {ops[0]} overlaps src{name} {ops[1] if name == 'A' else ops[2]}: {line.strip()}")
print(f"{bad} overlapping MFMA(s)")
sys.exit(1 if bad else 0)
