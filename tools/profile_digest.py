#!/usr/bin/env python3
"""Digest rocprofv3 outputs of a bench.py run into the small tables kept under profiles/.

    python tools/profile_digest.py <stats_dir> <fetch_dir> <write_dir> <sq_dir> <out_md> [<sq2_dir> [<bench_json>]]

* kernel-trace --stats: calls, total / average duration per kernel (product kernels only);
* SQ passes: instruction mix per wave; with the second pass (<sq2_dir>) the wave-cycle breakdown (parked in s_waitcnt / barrier,
  issue-stalled, issuing) and the matrix-pipe busy share: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), both in cycles;
* with <bench_json> (the JSON line of the same bench.py command): each operator's 1x1-convolution OP/s as a fraction of the
  dense matrix-core peak (5 POP/s int8, 157.3 TFLOP/s float32), from its HIP-event time;

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
    sq2_dir = sys.argv[6] if len(sys.argv) > 6 else None
    bench_json = sys.argv[7] if len(sys.argv) > 7 else None
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
    import json

    if sq2_dir:
        s2 = pmc(sq2_dir)
        lines += ["", "| kernel | grid | wave cycles/wave | parked (s_waitcnt, barrier) % | issue-stalled % | issuing % | LDS insts/wave | matrix pipe busy % of CU-busy time |",
                  "|---|---|---|---|---|---|---|---|"]
        mean = lambda v: sum(v) / max(len(v), 1)  # noqa: E731
        for key in sorted(s2):
            c = s2[key]
            a = sq.get(key, {})
            waves = mean(c.get("SQ_WAVES", [1])) or 1
            wc = mean(c.get("SQ_WAVE_CYCLES", [0])) or 1
            busy_cu = mean(c.get("SQ_BUSY_CU_CYCLES", [0]))
            mfma_busy = mean(a.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
            # SQ_VALU_MFMA_BUSY_CYCLES = cycles, summed over the SIMDs (N_mfma x 32 for v_mfma_f32_16x16x4_f32: checked on f32_pw_ws_kernel, 9.437e6
            # instructions -> 3.020e8); SQ_BUSY_CU_CYCLES = plain cycles per CU summed over the CUs (kernel time x clock x CUs), NOT quad-cycles —
            # the round-2 / early round-3 digests divided by 16 and under-reported this column four-fold
            pipe = 100 * mfma_busy / max(4 * busy_cu, 1)
            lines.append(f"| {key[0]} | {key[1]} | {4 * wc / waves:.0f} | {100 * mean(c.get('SQ_WAIT_ANY', [0])) / wc:.0f} | "
                         f"{100 * mean(c.get('SQ_WAIT_INST_ANY', [0])) / wc:.0f} | {100 * mean(c.get('SQ_ACTIVE_INST_ANY', [0])) / wc:.0f} | "
                         f"{mean(c.get('SQ_INSTS_LDS', [0])) / waves:.0f} | {pipe:.1f} |")
    if bench_json:
        try:
            b = json.loads([ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")][-1])
            lines += ["", f"Per-operator times of the same command ({b['dtype']}, batch {b['config']['batch_per_gpu']}; HIP events on the launch stream): "
                          "algorithmic HBM rate and 1x1-convolution matrix-core rate as fractions of peak (8 TB/s; 5 POP/s int8 / 157.3 TFLOP/s f32)", "",
                      "| operator | kernel | avg ms | GB/s | % of HBM peak | TOP/s (all ops) | 1x1 OP/s as % of matrix-core peak |", "|---|---|---|---|---|---|---|"]
            for st in b["stages"]:
                lines.append(f"| {st['layer']} | {st.get('symbol', st['kernel'])} | {st['avg_ms']} | {st['GBps']} | {100 * st.get('hbm_frac', 0):.1f} | {st['Tops']} | {100 * st.get('mfma_frac', 0):.2f} |")
            lines += ["", f"whole path: {b['value']} chunks/s, {b['ms_per_step']} ms/step, all matrix-core work (53.08 MOP/chunk) at "
                          f"{100 * b.get('whole_path_mfma_frac', 0):.2f} % of peak"]
        except Exception as e:  # pragma: no cover
            lines += ["", f"(bench JSON not digested: {e})"]

    # machine-readable HBM traffic per launch (bytes), read back by bench.py for roofline.traffic

    traffic = []
    for key in sorted(set(fe) | set(wr)):
        f = fe.get(key, {}).get("FETCH_SIZE", [0.0])
        w = wr.get(key, {}).get("WRITE_SIZE", [0.0])
        mean = lambda v: sum(v) / max(len(v), 1)  # noqa: E731
        traffic.append({"kernel": key[0], "grid_threads": int(key[1]), "read_bytes": int(2 * mean(f) * 1024), "write_bytes": int(mean(w) * 1024)})
    open(re.sub(r"_digest\.md$|\.md$", "", out) + "_traffic.json", "w").write(json.dumps(traffic, indent=1) + "\n")
    # machine-readable SQ counters per launch, read back by bench.py for roofline.valu_issue_frac / lds_wait
    s2 = pmc(sq2_dir) if sq2_dir else {}
    sqrows = []
    mean = lambda v: sum(v) / max(len(v), 1)  # noqa: E731
    for key in sorted(sq):
        c, d = sq[key], s2.get(key, {})
        waves = mean(c.get("SQ_WAVES", [1])) or 1
        wc = mean(d.get("SQ_WAVE_CYCLES", [0])) or 0
        row = {"kernel": key[0], "grid_threads": int(key[1]), "waves": int(waves), "valu_insts": int(mean(c.get("SQ_INSTS_VALU", [0]))),
               "valu_insts_per_wave": round(mean(c.get("SQ_INSTS_VALU", [0])) / waves, 1), "mfma_insts_per_wave": round(mean(c.get("SQ_INSTS_MFMA", [0])) / waves, 1),
               "lds_conflict_cycles_per_wave": round(mean(c.get("SQ_LDS_BANK_CONFLICT", [0])) / waves, 1)}
        if wc:
            row.update({"parked_frac": round(mean(d.get("SQ_WAIT_ANY", [0])) / wc, 3), "issue_stalled_frac": round(mean(d.get("SQ_WAIT_INST_ANY", [0])) / wc, 3),
                        "issuing_frac": round(mean(d.get("SQ_ACTIVE_INST_ANY", [0])) / wc, 3), "lds_insts_per_wave": round(mean(d.get("SQ_INSTS_LDS", [0])) / waves, 1)})
        sqrows.append(row)
    open(re.sub(r"_digest\.md$|\.md$", "", out) + "_sq.json", "w").write(json.dumps(sqrows, indent=1) + "\n")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
