#!/usr/bin/env python3
"""Scan gfx950 assembly for code that touches the A/B operand registers of an MFMA while it may still be reading them.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s kernel.hip && python tools/mfma_overlap_check.py k.s

Measured on MI355X: a multi-pass MFMA (v_mfma_f32_16x16x4_f32, 8 passes) reads its B operand pass by pass while it already
writes results, so the values B holds ~30 cycles after issue decide output columns 12-15.  The compiler assumes operands are
consumed at issue.  Two patterns gave wrong results in bn_f32_strip.hip:

  1. destination overlapping an operand:   v_mfma_f32_16x16x4_f32 v[46:49], v33, v49, v[58:61]        (every run)
  2. vector-ALU write right behind the MFMA: v_mfma_f32_16x16x4_f32 ..., v30, v6, ... ; v_mov_b32 v6, s33  (1 run in 100)

  3. two interleaved dependent chains:     v_mfma D0, .., .., D0 ; v_mfma D1, .., .., D1 ; v_mfma D0, .., .., D0       (1 launch in 50:
     an accumulator register of the first chain short of one term in columns 12-15; back-to-back dependent MFMAs and chains
     interleaved four or more deep have never failed)

Pattern 3 is an error for the f32 MFMAs when exactly one or two other MFMAs and nothing else separate the dependent pair.
Pattern 1 is an error for every MFMA and either operand; pattern 2 for the B operand of the f32 MFMAs within a window of ~32
cycles (one vector-ALU instruction = 4 cycles, an MFMA = its passes x 4, s_nop n = n + 1).  Exit status 1 on either: the Makefile
gates the build on it.  The A operand is consumed at issue (the tile kernels of bn_f32_fused.hip overwrite A registers right
behind the last MFMA of a chain and have always matched the oracle); the int8 MFMAs (16x16x32 / 16x16x64) showed neither
problem in 100-launch stress runs.  --strict reports those cases too, as notes.
Memory loads into an operand register are ignored: their data arrives hundreds of cycles later."""
import re
import sys

WINDOW = 32
PASSES = {"v_mfma_f32_16x16x4_f32": 8, "v_mfma_i32_16x16x32_i8": 4, "v_mfma_i32_16x16x64_i8": 8}


def rng(op):
    m = re.match(r"([va])\[(\d+):(\d+)\]", op)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r"([va])(\d+)$", op)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None


def hit(a, b):
    return a and b and a[0] == b[0] and not (a[2] < b[1] or b[2] < a[1])


strict = "--strict" in sys.argv
errors = warnings = 0
for path in [p for p in sys.argv[1:] if not p.startswith("--")]:
    kernel = "?"
    recent = []  # (cycles since issue, opcode, srcA, srcB, line number)
    chain = []   # (destination, opcode, line number) of the MFMAs seen so far
    for ln, line in enumerate(open(path), 1):
        if line.startswith("_Z") and ":" in line:
            kernel, recent, chain = line.split(":")[0], [], []
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
        if not m or line.lstrip().startswith((";", ".")):
            continue
        opc, ops = m.group(1), [o.strip() for o in m.group(2).split(";")[0].split(",")]
        if opc.startswith("s_cbranch") or opc in ("s_branch", "s_barrier", "s_endpgm"):
            recent = []  # other paths / long waits: not followed
            continue
        cost = 4
        if opc.startswith("v_mfma"):
            d, a, b = rng(ops[0]), rng(ops[1]), rng(ops[2])
            for name, s in (("A", a), ("B", b)):
                if hit(d, s):
                    errors += 1
                    print(f"{path}:{ln}: {kernel[:60]}: destination overlaps src{name}: {line.strip()}")
            c_src = rng(ops[3]) if len(ops) > 3 else None
            if "f32" in opc:
                for back in (2, 3):  # dependent on the MFMA issued `back` MFMAs ago with only MFMAs in between
                    if len(chain) >= back and chain[-back][2] == ln - back and hit(c_src, chain[-back][0]) and not any(hit(c_src, chain[-k][0]) for k in range(1, back)):
                        errors += 1
                        print(f"{path}:{ln}: {kernel[:60]}: depends on the MFMA {back} instructions back with only MFMAs between (interleaved chains): {line.strip()}")
            chain.append((d, opc, ln))
            cost = 4 * PASSES.get(opc, 8)
            recent = [(c + cost, o, x, y, l) for c, o, x, y, l in recent]
            recent.append((0, opc, a, b, ln))
            recent = [r for r in recent if r[0] < WINDOW]
            continue
        if opc == "s_nop":
            cost = int(ops[0]) + 1 if ops and ops[0].isdigit() else 1
        elif opc.startswith("v_") and not opc.startswith("v_cmp") and ops:
            w = rng(ops[0])
            for c, o, a, b, l in recent:
                if (strict or "f32" in o) and hit(w, b):
                    warnings += 1
                    print(f"{path}:{ln}: {kernel[:60]}: writes the B operand of the {o} at line {l}, {c} cycles after its issue: {line.strip()}")
                elif strict and hit(w, a):
                    print(f"{path}:{ln}: note: writes the A operand of the {o} at line {l}, {c} cycles after its issue: {line.strip()}")
        recent = [(c + cost, o, x, y, l) for c, o, x, y, l in recent if c + cost < WINDOW]
print(f"{errors} MFMA operand-overlap or interleaved-chain error(s), {warnings} early write(s) to a B operand")
sys.exit(1 if errors or warnings else 0)
