#!/usr/bin/env python3
"""Digest rocprofv3 outputs of a bench.py run into the small tables kept under profiles/.

    python tools/profile_digest.py <stats_dir> <fetch_dir> <write_dir> <sq_dir> <steps_in_pmc_runs> <out_md>

* kernel-trace --stats: calls, total / average duration per kernel (product kernels only);
* --pmc FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes): HBM bytes per launch.
  FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads); both are in KiB.
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?bn::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def pmc(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    stats_dir, fetch_dir, write_dir, sq_dir, out = sys.argv[1:6]
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for f in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                lines.append(f"| {k} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    fe, wr, sq = pmc(fetch_dir), pmc(write_dir), pmc(sq_dir)
    lines += ["", "| kernel | grid | HBM read MiB/launch (2 x FETCH_SIZE) | HBM write MiB/launch | L2 hit % | VALU insts/wave | MFMA insts/wave | LDS conflict cyc/wave |",
              "|---|---|---|---|---|---|---|---|"]
    for key in sorted(set(fe) | set(wr)):
        f = fe.get(key, {}).get("FETCH_SIZE", [0.0])
        w = wr.get(key, {}).get("WRITE_SIZE", [0.0])
        h, ms = wr.get(key, {}).get("TCC_HIT_sum", [0.0]), wr.get(key, {}).get("TCC_MISS_sum", [0.0])
        s = sq.get(key, {})
        waves = (sum(s.get("SQ_WAVES", [1])) / max(len(s.get("SQ_WAVES", [1])), 1)) or 1
        mean = lambda v: sum(v) / max(len(v), 1)  # noqa: E731
        hit = 100 * mean(h) / max(mean(h) + mean(ms), 1)
        lines.append(f"| {key[0]} | {key[1]} | {2*mean(f)/1024:.1f} | {mean(w)/1024:.1f} | {hit:.0f} | {mean(s.get('SQ_INSTS_VALU',[0]))/waves:.0f} | "
                     f"{mean(s.get('SQ_INSTS_MFMA',[0]))/waves:.1f} | {mean(s.get('SQ_LDS_BANK_CONFLICT',[0]))/waves:.0f} |")
    # machine-readable HBM traffic per launch (bytes), read back by bench.py for roofline.traffic
    import json

    traffic = []
    for key in sorted(set(fe) | set(wr)):
        f = fe.get(key, {}).get("FETCH_SIZE", [0.0])
        w = wr.get(key, {}).get("WRITE_SIZE", [0.0])
        mean = lambda v: sum(v) / max(len(v), 1)  # noqa: E731
        traffic.append({"kernel": key[0], "grid_threads": int(key[1]), "read_bytes": int(2 * mean(f) * 1024), "write_bytes": int(mean(w) * 1024)})
    open(re.sub(r"_digest\.md$|\.md$", "", out) + "_traffic.json", "w").write(json.dumps(traffic, indent=1) + "\n")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
