#!/usr/bin/env python3
"""Build gate: no kernel of a PRODUCTION plan may spill registers to scratch.

    python tools/spill_check.py <file.s> [...]

Reads the kernel metadata the compiler writes into the device assembly (.vgpr_spill_count, .private_segment_fixed_size).  A spill in
a kernel the shipped float32 / INT8 plans, the configs[4] plans or the audio path launch by default fails the build; spills in the
A/B and fallback instantiations (kernels only non-default options or other topologies reach) are listed as warnings.  (Round 2 shipped
the dominant INT8 kernel with 16 spilled registers, 1.7 x its algorithmic HBM traffic, and nothing in the build said so.)
"""
import re
import subprocess
import sys

# kernels that are NOT on a default path (measurement switches kept behind an option that is off): listed instead of failing the build
NON_PRODUCTION = [
    r"i8_pwdw_kernel",  # option i8_pwdw (off: measured slower)
    r"i8_mel_mfma_kernel<true, 3>",  # option stft_audit (off: the guarded mixer with the audit of its near misses)
    r"rocprim::",  # bn_sort.hip: rocPRIM's radix sorts behind the ranking metrics (once per evaluation, not on the per-chunk path); their
                   # onesweep kernel keeps a 48-byte local array in scratch by design
]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    bad, warn = [], []
    for path in sys.argv[1:]:
        text = open(path).read()
        recs = []
        for blk in text.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            if spill or scratch:
                recs.append((name, spill, scratch))
        if not recs:
            continue
        dm = demangle([r[0] for r in recs])
        for name, spill, scratch in recs:
            pretty = re.sub(r"\(anonymous namespace\)::", "", dm[name])
            line = f"{path}: {pretty.split('(')[0]}: {spill} spilled VGPRs, {scratch} B of scratch per lane"
            (warn if any(re.search(p, pretty) for p in NON_PRODUCTION) else bad).append(line)
    for w in warn:
        print("spill (non-production instantiation):", w)
    for b in bad:
        print("SPILL IN A PRODUCTION KERNEL:", b)
    print(f"{len(bad)} production kernel(s) with register spills")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
