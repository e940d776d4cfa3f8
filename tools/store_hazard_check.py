#!/usr/bin/env python3
"""Scan gfx950 assembly for vector-ALU writes to the data registers of a wide store issued just before.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s kernel.hip && python tools/store_hazard_check.py k.s

Measured on MI355X (bn_f32_strip.hip): `buffer_store_dwordx4 v[4:7], v95, s[8:11], s6 offen` followed at once by
`v_med3_f32 v4, v16, s34, v28` stored the NEW v4 for lanes 12-15 of every 16-lane row — the store reads its data registers
a few cycles after issue, later still when the memory pipeline is busy (1 launch in 50 up to every launch).  The compiler
inserts a wait state for this hazard only when the store has no SGPR offset.  A hit = a 12/16-byte memory store (buffer / global /
flat / scratch) whose data registers are the destination of a vector-ALU or MFMA instruction within the next WINDOW
instructions; `s_nop n` counts as n + 1.  LDS writes are exempt: `ds_write_b128` followed at once by a rewrite of its data
occurs 39 times in bn_ingest.hip, whose results are bit-identical to numpy/scipy in every test (--strict lists them).  Exit status 1 on a hit: the Makefile gates
the build on it.  Fix in source: keep the stored value an in/out operand of `asm volatile("s_nop 1" : "+v"(value))` placed
right after the store (store16() in bn_f32_strip.hip)."""
import re
import sys

WINDOW = 2
STRICT = "--strict" in sys.argv
WIDE = re.compile(r"^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)$" + (r"|^ds_write_b(96|128)$" if STRICT else ""))


def rng(op):
    m = re.match(r"v\[(\d+):(\d+)\]", op)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", op)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


hits = 0
for path in [p for p in sys.argv[1:] if not p.startswith('--')]:
    kernel, pending = "?", []  # pending: (age in wait states, data range, line number, text)
    for ln, line in enumerate(open(path), 1):
        if line.startswith("_Z") and ":" in line:
            kernel, pending = line.split(":")[0], []
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
        if not m or line.lstrip().startswith((";", ".")):
            continue
        opc, ops = m.group(1), [o.strip() for o in m.group(2).split(";")[0].split(",")]
        if opc.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
            pending = []
            continue
        cost = int(ops[0]) + 1 if opc == "s_nop" and ops and ops[0].isdigit() else 1
        if opc.startswith("v_") and not opc.startswith("v_cmp") and ops:
            w = rng(ops[0])
            for age, d, l, text in pending:
                if w and d and not (w[1] < d[0] or d[1] < w[0]):
                    hits += 1
                    print(f"{path}:{ln}: {kernel[:70]}: `{line.strip()}` rewrites the data of the store at line {l} ({text}) {age} wait state(s) after it")
        pending = [(a + cost, d, l, t) for a, d, l, t in pending if a + cost < WINDOW]
        if WIDE.match(opc):
            data = rng(ops[1]) if opc.startswith(("ds_", "global", "flat")) else rng(ops[0])
            pending.append((0, data, ln, line.strip()[:60]))
print(f"{hits} store-data hazard(s)")
sys.exit(1 if hits else 0)
