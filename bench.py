#!/usr/bin/env python3
"""Throughput benchmark of the per-chunk inference hot path on MI355X.

One *step* = one pass of the whole path over one batch of synthetic 3 s @ 24 kHz chunks that are
already resident in HBM:  windowed STFT magnitude -> hybrid mel mixer -> PWL -> DS-CNN -> scores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype i8|f32] [--batch B]

* N = 1 (default): BASELINE.json configs[2] — the shipped birdnet_stm32n6_100 INT8 DS-CNN (int8 MFMA 1x1 convolutions)
  with the hybrid+pwl frontend at batch 4096, the largest single-GPU configuration (``--dtype f32`` gives configs[1],
  float32 at batch 1024; the default run reports it under ``also_measured``).
* N > 1: BASELINE.json configs[3] — the INT8 chunk stream sharded over N GPUs, one process per GPU.  Started bare
  (``python bench.py --gpus N`` with WORLD_SIZE unset) the parent launches ``python -m torch.distributed.run
  --nproc-per-node N`` on itself BEFORE anything touches a GPU and relays the workers' output; started by the driver's
  torchrun line it reads RANK / LOCAL_RANK / WORLD_SIZE.  The global stream has N x K x 4096 chunks (K = 8 on 8 GPUs is
  the 262 144 chunks of configs[3]); rank r scores its contiguous block of K x 4096 chunks in K batches of 4096 into a
  ``[K * 4096, 100]`` buffer (``evaluation/sharding.py: run_sharded`` — the same function ``evaluate`` shards with) and
  the job ends with ONE RCCL all-gather of those scores (12.8 MB per rank at K = 8), inside the timed region.

The timed job (K steps between barrier + device synchronisation on both sides, MAX over ranks) is REPEATED ``--repeats`` R times
(default 25: more than a second of timed work at the default sizes); ``value`` / ``ms_per_step`` are the MEDIAN repeat, ``value_min`` /
``value_max`` the slowest / fastest one.  ``--collective`` initialises RCCL even on one GPU and runs the all-gather inside the timed
region at world size 1 (the collective code path on hardware without a second GPU); without it a one-GPU run still probes RCCL after
the measurement and reports what it saw under ``collective``.

Every rank keeps ``min(K, 8)`` distinct synthetic batches in HBM (9.4 GB) and walks them round-robin, so no step re-reads a
cache-warm input.  Rank 0 prints ONE JSON line: the throughput contract fields plus ``roofline`` (dominant kernel — named by
the fully profiled warm-up steps, then timed with HIP events on the launch stream during the timed region, where it is the
only bracketed operator; ``stages`` are the warm-up measurements of the other operators) and ``cpu_baseline`` (the C/OpenMP
restatement of the reference path under ``oracle/`` timed on this host on a bounded sample — a reported baseline, not a
target).
"""

from __future__ import annotations

import argparse
import glob
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "birdnet-stm32_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

SR, SECONDS, W, NFFT = 24000, 3.0, 256, 512
T = int(SR * SECONDS)
HOP = T // W

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 / 16x16x4 = vector FP32 rate
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 = 2x bf16
MOP_PER_CHUNK = 53.08  # 26 539 008 MAC per chunk (mel mixer + backbone), 2 ops per MAC
STFT_MFLOP_PER_CHUNK = 3.35
MAX_DISTINCT_BATCHES = 8
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9  # 39.3 T lane-op/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md)
PWDW_HEAD, PWDW_COVERED = 0x7A110009, 0x7A11000A  # expand 1x1 + depthwise 3x3 of an inverted-residual block as one kernel (csrc/bn_blob.h)


def synth_audio_device(torch, batch: int, first_chunk: int, device, seed: int):
    """peaknorm(0.3 N(0,1) + sin(2 pi f_g t)), f_g = 500 + 37 (g mod 200), g = global chunk index."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    t = torch.arange(T, device=device, dtype=torch.float64) / SR
    g = torch.arange(batch, device=device, dtype=torch.float64) + first_chunk
    f = 500.0 + 37.0 * torch.remainder(g, 200.0)
    tone = torch.sin(2.0 * np.pi * f[:, None] * t[None, :]).to(torch.float32)
    x = 0.3 * torch.randn((batch, T), generator=gen, device=device, dtype=torch.float32) + tone
    x = x / x.abs().amax(dim=1, keepdim=True)
    return x.contiguous()


FRONT2_HEAD, FRONT2_COVERED = 0x7A110003, 0x7A110004  # OpRec.p[38] tags of a front block / residual block pair (csrc/bn_blob.h)


def kernel_symbol(kind: str, p: list) -> str:
    """The device kernel an operator launches (the launchers' choices: strip kernels for the wide early blocks)."""
    if kind in ("stft512", "f32_stftmel"):
        return "stft512_mag_kernel"
    if kind == "i8_tail":
        return "i8_tail2_kernel"   # (the default form where the plan carries its constants: the shipped graph does)
    if kind == "i8_mid":
        return "i8_mid2_kernel"
    if kind == "i8_dwpw" and p[35]:
        # (32 -> 32 channels, stride 1, residual ADD — stage1_ds2 of the shipped graph: the strip kernel with the depthwise stage on the matrix cores)
        return "i8_strip_mf_kernel" if (p[2] == 32 and p[14] == 32 and p[3] == 1 and p[18]) else "i8_strip_kernel"
    if kind == "i8_dwpw" and p[30] and p[14] == 64:
        return "i8_mel_mfma_kernel"
    if kind == "i8_front" and p[16]:
        return "i8_front_strip_kernel"
    if kind == "f32_dwpw" and p[15]:
        cin, cout, ow, stride = p[2], p[10], p[7], p[3]
        if ow % 16 == 0 and stride in (1, 2) and ((cin == 32 and cout in (32, 64)) or (cin == 64 and cout in (64, 128)) or (cin, cout) == (128, 128)):
            return "f32_strip_kernel"
        if cin <= 64 and cout <= 64:
            return "f32_dwpw_wave_kernel"
    return kind + "_kernel"


def algorithmic_work(row: dict, batch: int, dtype: str) -> tuple[float, float, float]:
    """(bytes, ops, matrix-core ops) one launch of plan operator ``row`` must move / execute for ``batch`` chunks.

    Bytes are what the kernel that runs the operator has to move once: a residual that the strip kernels take from the centre
    tap they already hold is NOT counted (only the generic tile kernels read it again).  Matrix-core ops are the 1x1
    convolutions' (and the mel mixer's) multiply-accumulates x 2 — the numerator of ``mfma_frac``.
    """
    k, p = row["kind"], row["p"]
    e = 4 if dtype == "f32" else 1
    sym = kernel_symbol(k, p)
    if k == "stft512":
        return batch * (T * 4 + (NFFT // 2 + 1) * W * 4), batch * 3.35e6, 0.0
    if k == "f32_stftmel":
        return batch * (T * 4 + p[2] * p[1] * 4), batch * 3.35e6, 0.0
    if k == "f32_melfin":
        return batch * 2.0 * p[0] * p[1] * 4, batch * 10.0 * p[0] * p[1], 0.0
    if k in ("f32_front", "i8_front"):
        macs = 9 * p[0] * (p[1] // 2) * p[2] + p[4] * p[5] * p[2] * (9 + p[3])
        mm = 2.0 * (9 * p[0] * (p[1] // 2) * p[2] + p[4] * p[5] * p[2] * p[3]) if "strip" in sym else 2.0 * p[4] * p[5] * p[2] * p[3]
        return batch * float(e) * (p[0] * p[1] + p[4] * p[5] * p[3]), batch * 2.0 * macs, batch * mm
    if k == "f32_mel":
        return batch * (p[0] * p[1] * 4 + p[2] * p[1] * 4), batch * 2.0 * p[0] * p[1] * p[2], 0.0
    if k == "i8_quant":
        return batch * (p[0] * p[1] * 4 + p[1] * p[2]), batch * 2.0 * p[0] * p[1], 0.0
    if k == "i8_mel":
        return batch * (p[0] * p[1] + p[2] * p[0]), batch * 2.0 * p[0] * p[1] * p[2], 0.0
    if k in ("f32_stem", "i8_stem"):
        return batch * (p[0] * p[1] * e + p[6] * p[7] * p[2] * e), batch * 2.0 * 9 * p[6] * p[7] * p[2], 0.0
    if k in ("f32_dw", "i8_dw"):
        return batch * (p[0] * p[1] * p[2] * e + p[6] * p[7] * p[2] * e), batch * 2.0 * 9 * p[6] * p[7] * p[2], 0.0
    if k == "f32_dwpw":
        n_in, n_out = p[0] * p[1] * p[2], p[6] * p[7] * p[10]
        macs = p[6] * p[7] * p[2] * (p[10] + (9 if p[15] else 0))
        res = 1 if (p[12] and sym != "f32_strip_kernel") else 0
        return batch * 4.0 * (n_in + n_out * (1 + res)), batch * 2.0 * macs, batch * 2.0 * p[6] * p[7] * p[2] * p[10]
    if k == "i8_dwpw" and p[36]:  # mel mixer with QUANTIZE fused into its load: float32 spectrogram in, int8 [M][W] out
        return batch * (p[5] * p[1] * 4.0 + p[14] * p[1]), batch * 2.0 * p[1] * p[2] * p[14], batch * 2.0 * p[1] * p[2] * p[14]
    if k == "i8_dwpw":
        n_in, n_out = p[0] * p[1] * p[2], p[6] * p[7] * p[14]
        macs = p[6] * p[7] * p[2] * (p[14] + (9 if p[29] else 0))
        res = 1 if (p[18] and sym not in ("i8_strip_kernel", "i8_strip_mf_kernel")) else 0
        return batch * 1.0 * (n_in + n_out * (1 + res)), batch * 2.0 * macs, batch * 2.0 * p[6] * p[7] * p[2] * p[14]
    if k == "i8_mid":  # p: in_bytes pw_macs dw_macs 0 0 n_layers H0 W0 C0 P_last C_last
        return batch * 1.0 * (p[0] + p[9] * p[10]), batch * 2.0 * (p[1] + p[2]), batch * 2.0 * p[1]
    if k == "i8_tail":  # p: in_bytes pw_macs dw_macs other_macs n_classes
        return batch * (p[0] + 4.0 * p[4]), batch * 2.0 * (p[1] + p[2] + p[3]), batch * 2.0 * p[1]
    if k == "f32_pw":
        return batch * (p[0] * p[1] * 4 + p[0] * p[2] * 4 * (2 if p[4] else 1)), batch * 2.0 * p[0] * p[1] * p[2], batch * 2.0 * p[0] * p[1] * p[2]
    if k == "i8_pw":
        return batch * (p[0] * p[1] + p[0] * p[2] * (2 if p[6] else 1)), batch * 2.0 * p[0] * p[1] * p[2], 0.0
    if k in ("f32_gap", "i8_mean"):
        return batch * (p[0] * p[1] * e + p[1] * e), batch * 1.0 * p[0] * p[1], 0.0
    if k == "f32_gapdense":
        return batch * (p[0] * p[1] * 4 + p[2] * 4), batch * (1.0 * p[0] * p[1] + 2.0 * p[1] * p[2]), 0.0
    if k in ("f32_dense", "i8_fc"):
        return batch * (p[0] * e + p[1] * 4), batch * 2.0 * p[0] * p[1], 0.0
    return 0.0, 0.0, 0.0


def output_bytes(kind: str, p: list, batch: int, dtype: str):
    """Bytes one launch of the operator writes (what the PMC WRITE_SIZE of that launch shows), or None if not modelled."""
    e = 4 if dtype == "f32" else 1
    if kind == "stft512":
        return batch * (NFFT // 2 + 1) * W * 4
    if kind == "f32_stftmel":
        return batch * p[2] * p[1] * 4
    if kind in ("f32_front", "i8_front"):
        return batch * p[4] * p[5] * p[3] * e
    if kind == "f32_dwpw":
        return batch * p[6] * p[7] * p[10] * 4
    if kind == "i8_dwpw":
        return batch * p[6] * p[7] * p[14]
    return None


def digests(dtype: str, batch: int, what: str) -> list[str]:
    """Committed digests ``profiles/rNN_<dtype>_b<batch>_<what>.json`` of THIS workload, newest round first (the name is matched as a
    whole: ``r02_config5_f32_b1024_traffic.json`` is another network's digest)."""
    pat = re.compile(rf"^r(\d+)_{dtype}_b{batch}_{what}\.json$")
    found = [(int(m.group(1)), q) for q in glob.glob(os.path.join(REPO, "profiles", "*.json")) if (m := pat.match(os.path.basename(q)))]
    return [q for _, q in sorted(found, reverse=True)]


def sq_counters(dom: dict, batch: int, dtype: str) -> dict:
    """The roof that binds the integer kernels is vector-ALU issue, not the matrix pipe: ``valu_issue_frac`` = VALU lane-ops of one
    launch (SQ_INSTS_VALU x 64, committed ``rocprofv3 --pmc`` SQ pass of this workload, replayed) / 39.3 T lane-op/s / this run's launch
    time; ``lds_wait`` = the share of the kernel's wave cycles parked in s_waitcnt / barriers and its LDS bank-conflict cycles per wave
    from the same pass.  Empty when no digest of the workload names the kernel."""
    base = kernel_symbol(dom["kernel"], dom["p"])
    for path in digests(dtype, batch, "sq"):
        rows = [r for r in json.load(open(path)) if r["kernel"].split("<")[0] == base]
        if not rows:
            continue
        r = max(rows, key=lambda q: q["valu_insts"])  # several instantiations: the heaviest launch is the one bench names dominant
        out = {"valu_issue_frac": round(r["valu_insts"] * 64 / VALU_PEAK_LANE_OPS / (dom["avg_ms"] * 1e-3), 4),
               "valu_insts_per_wave": r["valu_insts_per_wave"], "mfma_insts_per_wave": r["mfma_insts_per_wave"],
               "lds_wait": {"parked_frac": r.get("parked_frac"), "issue_stalled_frac": r.get("issue_stalled_frac"),
                            "lds_bank_conflict_cycles_per_wave": r.get("lds_conflict_cycles_per_wave"), "lds_insts_per_wave": r.get("lds_insts_per_wave")},
               "sq_source": "profiles/" + os.path.basename(path) + " (committed rocprofv3 --pmc SQ passes of this workload, replayed)"}
        return out
    return {}


def pmc_traffic(dom: dict, batch: int, dtype: str):
    """(HBM bytes per launch of the dominant kernel, source file) from the newest committed PMC digest of this workload
    (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950; produced by tools/profile_digest.py).  Counters cannot be read from inside the benchmark, so this is a REPLAY of
    the committed profile — ``traffic_source`` names the file — and null when no digest of the same workload (dtype, batch,
    kernel, output size) is on disk."""
    want = output_bytes(dom["kernel"], dom["p"], batch, dtype)
    paths = digests(dtype, batch, "traffic")
    base = kernel_symbol(dom["kernel"], dom["p"])
    for path in paths:
        rows = [row for row in json.load(open(path)) if row["kernel"].split("<")[0] == base]
        best = None
        if want is not None:  # several launches of one kernel template: the one whose written bytes match this operator's output
            for row in rows:
                if abs(row["write_bytes"] - want) <= 0.05 * want and (best is None or abs(row["write_bytes"] - want) < abs(best["write_bytes"] - want)):
                    best = row
        elif len(rows) == 1:  # a kernel that is launched once per step (the fused tail)
            best = rows[0]
        if best is not None:
            return int(best["read_bytes"] + best["write_bytes"]), "profiles/" + os.path.basename(path) + " (committed rocprofv3 --pmc run of this workload, replayed)"
    return None, None


def roofline_of(rows: list[dict], batch: int, dtype: str, dom_op: int = -1) -> tuple[dict, list[dict]]:
    stages = []
    peak_compute = F32_MFMA_PEAK_TFLOPS if dtype == "f32" else I8_MFMA_PEAK_TOPS
    by_op = {r["op"]: r for r in rows}
    for r in rows:
        if not r["launches"]:
            continue
        avg_ms = r["ms"] / r["launches"]
        nbytes, nops, mops = algorithmic_work(r, batch, dtype)
        symbol, layer = kernel_symbol(r["kind"], r["p"]), r["name"]
        # front block + the residual block behind it as one kernel (f32_front2_kernel): the second operator did not launch, its
        # arithmetic belongs to this launch; the bytes stay input map + ONE 32-channel map (the one between them stays in LDS)
        nxt = by_op.get(r["op"] + r["p"][37]) if r["kind"] == "f32_front" and r["p"] and r["p"][38] == FRONT2_HEAD else None
        if nxt is not None and not nxt["launches"] and nxt["p"][38] == FRONT2_COVERED:
            _, o2, m2 = algorithmic_work(nxt, batch, dtype)
            nops, mops, symbol, layer = nops + o2, mops + m2, "f32_front2_kernel", r["name"] + " + " + nxt["name"]
        stages.append({"kernel": r["kind"], "symbol": symbol, "layer": layer, "avg_ms": round(avg_ms, 4),
                       "GBps": round(nbytes / avg_ms / 1e6, 1), "hbm_frac": round(nbytes / avg_ms / 1e6 / HBM_PEAK_GBS, 4),
                       "Tops": round(nops / avg_ms / 1e9, 2), "mfma_Tops": round(mops / avg_ms / 1e9, 2),
                       "mfma_frac": round(mops / avg_ms / 1e9 / peak_compute, 4),
                       "bytes": nbytes, "ops": nops, "p": r["p"], "op": r["op"]})
    picked = [s for s in stages if s["op"] == dom_op]
    dom = picked[0] if picked else max(stages, key=lambda s: s["avg_ms"])
    ridge = peak_compute * 1e12 / (HBM_PEAK_GBS * 1e9)
    intensity = dom["ops"] / max(dom["bytes"], 1.0)
    if intensity > ridge and (dom["kernel"].endswith("pw") or dom["kernel"] in ("i8_tail", "i8_mid")):
        # matrix-core work only (the 1x1 convolutions): depthwise / dense arithmetic runs on the vector ALU and does not count against this roof
        roof = {"bound": "mfma", "achieved": dom["mfma_Tops"], "peak": peak_compute, "unit": "TFLOP/s" if dtype == "f32" else "TOP/s",
                "all_ops_Tops": dom["Tops"]}
    else:
        roof = {"bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["traffic"], roof["traffic_source"] = pmc_traffic(dom, batch, dtype)
    roof["kernel"] = dom["symbol"]
    roof["layer"] = dom["layer"]
    roof["avg_launch_ms"] = dom["avg_ms"]
    roof["algorithmic_bytes_per_launch"] = dom["bytes"]
    roof["timing"] = "HIP events on the launch stream around this kernel only, averaged over the timed region"
    roof.update(sq_counters(dom, batch, dtype))
    # the two longest kernels of the INT8 path (STFT, fused tail) are within a few per cent of each other and swap places from run to
    # run: the runner-up is named with its own roof (from the warm-up profile) so that the line reads the same either way
    rest = sorted((s for s in stages if s is not dom), key=lambda s: -s["avg_ms"])
    if rest:
        ru = rest[0]
        ru_mfma = ru["ops"] / max(ru["bytes"], 1.0) > ridge and (ru["kernel"].endswith("pw") or ru["kernel"] in ("i8_tail", "i8_mid"))
        roof["runner_up"] = {"kernel": ru["symbol"], "avg_launch_ms": ru["avg_ms"], "bound": "mfma" if ru_mfma else "hbm",
                             "frac": ru["mfma_frac"] if ru_mfma else ru["hbm_frac"],
                             "timing": "warm-up profile (every operator bracketed by events)"}
    for s in stages:
        s.pop("bytes"), s.pop("ops"), s.pop("p"), s.pop("op")
    return roof, stages


def physical_cores() -> int:
    """Physical cores of this host (``cpu_baseline.cores``; the OpenMP thread count is reported beside it as ``threads``)."""
    try:
        import psutil

        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:  # pragma: no cover
        pass
    return os.cpu_count() or 1


def host_peaks(used_threads: int | None = None) -> dict:
    """What the host could do at best, for scale beside ``cpu_baseline.gops``: cores x max clock x int8 multiply-add lanes per cycle (two
    512-bit VNNI pipes = 256 ops per cycle and core with avx512_vnni, 128 with avx_vnni only, 64 with AVX2's vpmaddubsw) — an upper bound from
    the flags and the clock the kernel reports, not a measurement."""
    flags, mhz = "", 0.0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags") and not flags:
                flags = line
            if line.startswith("cpu MHz"):
                mhz = max(mhz, float(line.split(":")[1]))
        try:
            mhz = max(mhz, float(open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq").read()) / 1e3)
        except OSError:
            pass
    except OSError:
        pass
    per_cycle = 256 if " avx512_vnni" in flags else 128 if " avx_vnni" in flags else 64 if " avx2" in flags else 16
    cores = physical_cores()
    share = min(cores, used_threads) if used_threads else cores   # (the baseline ran on the process's CPU share, not on the whole host)
    return {"cores": cores, "max_mhz": round(mhz, 1), "int8_ops_per_cycle_per_core": per_cycle,
            "theoretical_int8_peak_gops": round(cores * mhz * 1e6 * per_cycle / 1e9, 1),
            "cores_available_to_this_process": share, "theoretical_int8_peak_gops_of_that_share": round(share * mhz * 1e6 * per_cycle / 1e9, 1), "isa": "avx512_vnni" if per_cycle == 256 else "avx_vnni" if per_cycle == 128 else "avx2" if per_cycle == 64 else "scalar"}


def cpu_baseline(dtype: str, seconds_budget: float = 15.0) -> dict:
    """Time the CPU restatement of the reference path (``oracle/``) on this host, on a bounded sample.

    float32: the plain-C + OpenMP port (``oracle/c/oracle_cpu.c``) on all host threads when it has been built, otherwise the
    numpy oracle on one thread.  INT8: the C + OpenMP port of the TFLite int8 reference kernels (``oracle/c/oracle_i8.c``) when
    built, otherwise the numpy interpreter on one thread.  ``gops`` is the arithmetic rate that throughput amounts to
    (53.08 MOP + 3.35 MFLOP of STFT per chunk), ``host`` the machine's theoretical int8 peak beside it.  A port, not
    a tuned CPU library and not the reference's TFLite: a reported baseline, never a target.
    """
    import contextlib

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    from oracle import cport, float_graph, stft
    from oracle.int8_graph import Int8Interpreter

    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._tflite_reader import load_tflite

    ckpt = os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100")
    rng = np.random.default_rng(42)
    t = np.arange(T) / SR
    gops = lambda rate: round(rate * (MOP_PER_CHUNK + STFT_MFLOP_PER_CHUNK) / 1e3, 2)  # noqa: E731

    def chunks(n):
        x = 0.3 * rng.standard_normal((n, T)) + np.sin(2 * np.pi * (500 + 37 * (np.arange(n) % 200))[:, None] * t[None, :])
        return (x / np.abs(x).max(axis=1, keepdims=True)).astype(np.float32)

    def timed(fn, threads, what):
        nb = 1024   # chunks per call: eight per thread on a 128-thread host
        x = chunks(nb)
        fn(x[:64])  # warm up the OpenMP pool
        t0 = time.perf_counter()
        fn(x)
        per = (time.perf_counter() - t0) / nb
        n = int(max(nb, min(1 << 20, seconds_budget / per // nb * nb)))
        reps, done, t0 = n // nb, 0, time.perf_counter()
        for _ in range(reps):
            fn(x)
            done += nb
        dt = time.perf_counter() - t0
        # `cores` = the threads actually used (the process's CPU share: affinity mask capped by the cgroup quota — oracle/cport.py: cpu_share), not the host's count
        return {"value": round(done / dt, 1), "unit": "chunks/s", "cores": threads, "threads": threads, "host_physical_cores": physical_cores(), "kind": "port",
                "gops": gops(done / dt),
                "sample": f"{done} synthetic 3 s @ 24 kHz chunks, {what}, {threads} threads, {dt:.1f} s"}

    # the C ports are rebuilt -march=native on THIS host first (oracle/Makefile: native; a few seconds): on AVX-512 / VNNI hosts the INT8 port then
    # runs its vpdpbusd / sixteen-lane requantisation paths (bit-identical to the numpy interpreter: tests/test_oracle_pinning.py)
    native = cport.build_native()
    if dtype == "f32" and (native or os.path.isfile(cport.CPU_LIB)):
        path = cport.CpuFloatPath(load_keras_archive(ckpt + ".keras"), native=native)
        out = timed(path, path.threads, "plain-C + OpenMP port of the float path (oracle/c/oracle_cpu.c" + (", -march=native)" if native else ")"))
        out["host"] = host_peaks()
        return out
    if dtype == "i8" and (native or (os.path.isfile(cport.I8_LIB) and os.path.isfile(cport.CPU_LIB))):
        # the whole graph per chunk inside the C port (oi_program_run: the chunks dealt to the OpenMP threads, every thread's activations in its
        # cache, 1x1 weights packed once) — no numpy graph walk between the operators (round 4: 0.8 k chunks/s on 128 threads under the walk)
        path = cport.CpuInt8Program(load_tflite(ckpt + ".tflite"), native=native)
        sbuf = {}

        def run_i8(x):   # (the 270 MB spectrogram buffer of a 1024-chunk call is allocated once, not per call)
            key = x.shape[0]
            if key not in sbuf:
                sbuf.clear()
                sbuf[key] = np.empty((key, 257, W, 1), np.float32)
            return path.invoke(path.spectrogram(x, HOP, W, out=sbuf[key]))

        out = timed(run_i8, path.threads,
                    "C + OpenMP port of the TFLite int8 reference kernels, whole graph per chunk in C (oracle/c/oracle_i8.c: oi_program_run"
                    + (", -march=native" + (", AVX-512 VNNI paths" if path.vectorised else "") if native else "") + ") + C STFT")
        out["host"] = host_peaks(path.threads)
        out["int8_vector_paths"] = bool(path.vectorised)
        return out

    if dtype == "f32":
        spec = load_keras_archive(ckpt + ".keras")
        run = lambda S: float_graph.forward(spec, S, np.float32)  # noqa: E731
    else:
        interp = Int8Interpreter(load_tflite(ckpt + ".tflite"))
        run = lambda S: interp.invoke(S)  # noqa: E731

    def work(xs):
        t0 = time.perf_counter()
        for i in range(0, len(xs), 64):
            S = np.stack([stft.hybrid_spectrogram(a, NFFT, W) for a in xs[i : i + 64]])[..., None]
            run(S)
        return time.perf_counter() - t0

    with threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext():
        per = work(chunks(16)) / 16
        n = int(max(64, min(4096, seconds_budget / per // 64 * 64)))
        dt = work(chunks(n))
    return {"value": round(n / dt, 2), "unit": "chunks/s", "cores": 1, "threads": 1, "kind": "port", "gops": gops(n / dt),
            "sample": f"{n} synthetic 3 s @ 24 kHz chunks, numpy oracle (oracle/stft.py + "
                      f"{'float_graph' if dtype == 'f32' else 'int8_graph'}.py), 1 thread, {dt:.1f} s"}


def side_measure(torch, runner, audio, dtype: str, batch: int, device, steps: int = 10, repeats: int = 5, hop=None) -> dict:
    """Whole-path throughput of another configuration on this GPU (input resident in HBM), reported beside the main measurement:
    two warm-up steps, one step with every operator bracketed by HIP events (names the dominant operator and gives its time), then
    ``repeats`` x ``steps`` timed steps between device synchronisations (median repeat)."""
    scores = torch.empty((batch, runner.num_classes), dtype=torch.float32, device=device)
    kw = {"hop": hop} if hop else {}
    for _ in range(2):
        runner.infer_audio_device(audio, out=scores, **kw)
    torch.cuda.synchronize(device)
    runner.profile(True)
    runner.infer_audio_device(audio, out=scores, **kw)
    torch.cuda.synchronize(device)
    rows = runner.profile_collect()
    runner.profile(False)
    times = []
    for _ in range(repeats):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.infer_audio_device(audio, out=scores, **kw)
        torch.cuda.synchronize(device)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    finite = bool(torch.isfinite(scores).all().item())
    return {"dtype": dtype, "batch_per_gpu": batch, "value": round(batch * steps / dt, 1), "unit": "chunks/s", "steps": steps, "repeats": repeats,
            "ms_per_step": round(dt / steps * 1e3, 4), "scores_finite": finite, "_rows": rows}


def quick_rate(torch, dtype: str, batch: int, device, local_rank: int, steps: int = 10) -> dict:
    """The other single-GPU BASELINE configuration of the shipped network, with the roofline of ITS dominant kernel."""
    from birdnet_stm32.models.runners import load_model_runner

    ckpt = os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100" + (".keras" if dtype == "f32" else ".tflite"))
    runner = load_model_runner(ckpt, device=local_rank, max_batch=batch)
    audio = synth_audio_device(torch, batch, 0, device, 42)
    out = side_measure(torch, runner, audio, dtype, batch, device, steps, hop=HOP)
    runner.close()
    roof, _ = roofline_of(out.pop("_rows"), batch, dtype)
    roof["timing"] = "one profiled step (every operator bracketed by HIP events)"
    out["workload"] = ("birdnet_stm32n6_100 float32 DS-CNN" if dtype == "f32" else "birdnet_stm32n6_100 INT8 DS-CNN") + ", hybrid+pwl frontend"
    out["roofline"] = roof
    return out


def hard_inputs(torch, device, local_rank: int, headline: float, batch: int = 4096, steps: int = 5) -> dict:
    """Throughput of the headline path (INT8 from audio, ``batch`` chunks per step) per SIGNAL FAMILY (tools/signal_families.py: the twelve
    families of tools/exact_soak.py, random parameters and levels per chunk), with the share of each family's chunks on every route of the
    exactness pass (csrc/bn_stft_exact.hip): elements re-evaluated one by one inside the mel mixer, chunks recomputed as whole float64
    spectrograms (noise-free and flat inputs: the guard's bound is useless there).  The benchmark's own input is family 0."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from signal_families import FAMILIES, family_batch

    from birdnet_stm32.models.runners import load_model_runner

    from birdnet_stm32._hip import options as _hip_options

    runner = load_model_runner(os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100.tflite"), device=local_rank, max_batch=batch)
    g = torch.Generator(device=device).manual_seed(1234)
    scores = torch.empty((batch, runner.num_classes), dtype=torch.float32, device=device)
    rows = []
    for kind, name in enumerate(FAMILIES):
        x = family_batch(torch, kind, batch, g, device)
        for _ in range(2):
            runner.infer_audio_device(x, hop=HOP, out=scores)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.infer_audio_device(x, hop=HOP, out=scores)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        st = runner.guard_stats(batch)
        row = {"family": name, "chunks_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 4), "vs_headline": round(batch / dt / headline, 3),
               "elements_reevaluated_frac": round(st["listed"] / (batch * 257.0 * 256.0), 7),
               "whole_float64_chunks_frac": round((st["whole_minmax"] + st["whole_fix"]) / batch, 4),
               "scores_finite": bool(torch.isfinite(scores).all().item())}
        # the same family under the PROVEN worst-case bound (option stft_guard = 1, docs/exactness.md): identical scores, what the guarantee costs
        base = scores.clone()
        with _hip_options(stft_guard=1):
            runner.infer_audio_device(x, hop=HOP, out=scores)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(max(2, steps // 2)):
                runner.infer_audio_device(x, hop=HOP, out=scores)
            torch.cuda.synchronize(device)
            dtp = (time.perf_counter() - t0) / max(2, steps // 2)
            stp = runner.guard_stats(batch)
        row["proven_bound"] = {"chunks_per_s": round(batch / dtp, 1), "vs_empirical": round(dt / dtp, 3), "scores_identical": bool(torch.equal(scores, base)),
                               "elements_reevaluated_frac": round(stp["listed"] / (batch * 257.0 * 256.0), 7),
                               "whole_float64_chunks_frac": round((stp["whole_minmax"] + stp["whole_fix"]) / batch, 4)}
        rows.append(row)
        del x
    runner.close()
    worst = min(rows, key=lambda r: r["chunks_per_s"])
    worst_p = min(rows, key=lambda r: r["proven_bound"]["chunks_per_s"])
    return {"batch": batch, "steps": steps, "worst_family": worst["family"], "worst_vs_headline": worst["vs_headline"],
            "proven_bound_worst_family": worst_p["family"], "proven_bound_worst_vs_headline": round(worst_p["proven_bound"]["chunks_per_s"] / headline, 3),
            "proven_bound_scores_identical": all(r["proven_bound"]["scores_identical"] for r in rows), "families": rows,
            "note": "same kernels and options as the headline; only the input family changes (random frequencies, phases, levels over four decades); "
                    "proven_bound = the same run with option stft_guard = 1 (worst-case float32 FFT error bound instead of the empirical one)"}


def evaluate_leg(n_files: int = 1024) -> dict:
    """``evaluate`` end to end (files on tmpfs -> metrics) as a child process running tools/evaluate_bench.py on ``n_files`` synthetic 30 s stereo
    PCM16 WAVs: reader pool -> pinned slabs -> H2D on a copy stream -> ingest + inference -> pooling -> metrics (reference flow:
    birdnet_stm32/evaluation/metrics.py:117-153).  PCIe is this path's roof; the JSON carries both bounds and the per-stage split."""
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "evaluate_bench.py"), "--files", str(n_files), "--repeats", "3"],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stderr or p.stdout)[-400:]}
    d = json.loads(lines[-1])
    keep = ("files", "seconds", "channels", "native_rate", "dataset_gb", "value", "unit", "files_per_s", "pinned_copy_gbps", "h2d_gbps", "h2d_frac_of_pinned_copy",
            "pcm_bytes_per_chunk", "pcie_bound_chunks_per_s", "kernel_bound_chunks_per_s", "frac_of_min_bound", "best_warm_run", "cold_run")
    return {"workload": "python -m birdnet_stm32 evaluate's function on synthetic WAV files on tmpfs, INT8 shipped checkpoint (tools/evaluate_bench.py)",
            **{k: d[k] for k in keep if k in d}}


C4_PW_MAC, C4_MMAC = 191_889_408, 200.8  # configs[4] pointwise multiply-accumulates / all MACs per chunk


def config4_rate(torch, device, local_rank: int, seconds: int = 2, int8: bool = False, batch: int = 1024, steps: int = 10) -> dict:
    """BASELINE configs[4] on this GPU, beside the main measurement: raw-waveform learned filterbank + PCEN + alpha = 1.5 DS-CNN with
    squeeze-excite and inverted residuals, seeded random weights (there is no checkpoint of it).  ``seconds`` = 2: the geometry the
    reference's raw frontend builds at (its T < 65536 guard, models/dscnn.py:144-151); 3: the metric's chunk length, with the guard lifted
    (``raw_length_limit=None``).  float32, or INT8 through this build's own exporter.  One step = per-chunk peak normalisation + the
    whole plan over ``batch`` waveform chunks resident in HBM.  The roofline is the whole step's: all 1x1-convolution work (95.6 % of the
    network's MACs) against the dense matrix-core peak of the dtype, plus the slowest operator of one profiled step.  The N-GPU form of
    the config is the same plan on every rank over its own chunks (no collective until the scores)."""
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models.runners import HipRunner

    T4 = 24000 * seconds
    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=seconds, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42,
                       **({"raw_length_limit": None} if seconds != 2 else {}))
    if int8:
        from birdnet_stm32.conversion.export import convert_netspec_to_int8
        from birdnet_stm32.models._lower_i8 import lower_i8
        from birdnet_stm32.models._tflite_reader import parse_tflite
        from birdnet_stm32.models._tflite_writer import write_tflite

        rng = np.random.default_rng(0)
        cal = [rng.standard_normal((1, T4, 1)).astype(np.float32) for _ in range(8)]
        cal = [c / (np.abs(c).max() + 1e-6) for c in cal]
        plan = lower_i8(parse_tflite(write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal)))))
    else:
        from birdnet_stm32.models._lower_f32 import lower_f32

        plan = lower_f32(spec)
    runner = HipRunner(plan, device=local_rank, max_batch=batch)
    g = torch.Generator(device=device).manual_seed(4)
    x = torch.randn((batch, T4), device=device, generator=g)
    dtype = "i8" if int8 else "f32"
    out = side_measure(torch, runner, x, dtype, batch, device, steps)
    runner.close()
    rows = [r for r in out.pop("_rows") if r["launches"]]
    slow = max(rows, key=lambda r: r["ms"])
    peak = I8_MFMA_PEAK_TOPS if int8 else F32_MFMA_PEAK_TFLOPS
    rate = 2.0 * C4_PW_MAC * batch / (out["ms_per_step"] * 1e-3) / 1e12
    out.update({"workload": f"BASELINE configs[4] topology: raw frontend + PCEN + alpha=1.5 IR/SE DS-CNN, seeded random weights, {seconds} s @ 24 kHz"
                            + (", INT8 (own PTQ exporter)" if int8 else ""),
                "MMAC_per_chunk": C4_MMAC, ("TOP_per_s" if int8 else "TFLOP_per_s"): round(2 * C4_MMAC * 1e6 * batch / (out["ms_per_step"] * 1e-3) / 1e12, 1),
                "roofline": {"bound": "mfma", "achieved": round(rate, 2), "peak": peak, "unit": "TOP/s" if int8 else "TFLOP/s", "frac": round(rate / peak, 4),
                             "scope": "whole step: the 1x1 convolutions' 2 x 191.9 MMAC per chunk / step time (the network has no single dominant kernel)",
                             "slowest_operator": {"kind": slow["kind"], "layer": slow["name"], "ms": round(slow["ms"], 4),
                                                  "share_of_step": round(slow["ms"] / out["ms_per_step"], 3)},
                             "timing": "wall clock between device synchronisations; slowest operator from one profiled step"}})
    return out


def timed_job(score_batch, steps: int, batch: int, n_classes: int, device, barrier, sync):
    """The timed region: this rank's ``steps`` batches through ``evaluation/sharding.py: run_sharded`` (one process: all of
    them) and the ONE all-gather of the scores; bracketed by barrier + device synchronisation on both sides.

    ``score_batch(k, out_rows)`` scores the rank's k-th batch into ``out_rows`` ([batch, n_classes]).  Returns
    ``(seconds on this rank, gathered scores [world * steps * batch, n_classes])``.
    """
    import torch

    from birdnet_stm32.evaluation.sharding import run_sharded, shard_bounds, world_info

    rank, world = world_info()
    n_items = world * steps * batch
    lo, _hi = shard_bounds(n_items, rank, world)
    local = torch.empty((steps * batch, n_classes), dtype=torch.float32, device=device)

    def score(start, stop, out_rows):
        score_batch((start - lo) // batch, out_rows)

    barrier()
    sync()
    t0 = time.perf_counter()
    gathered = run_sharded(score, n_items, batch, into=local)
    sync()
    barrier()
    return time.perf_counter() - t0, gathered


def self_launch(args, argv: list[str]) -> int:
    """``python bench.py --gpus N`` with WORLD_SIZE unset: start N workers (one per GPU) with torch.distributed.run as a CHILD
    process before this process has touched a GPU, relay their output, return their exit code."""
    import socket

    import torch  # device_count() does not initialise the GPU on this image

    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its communicator comes up; rank 0's stdout must carry the JSON line and nothing else.
    While active, file descriptor 1 points at stderr (the library writes with C stdio, so ``sys.stdout`` redirection would not catch it)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)  # C stdio buffers of the libraries that printed
        except Exception:  # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


def collective_block(torch, dist, device, rows: int, n_classes: int, reps: int = 5) -> dict:
    """What RCCL saw: backend, ranks, library version, and the all-gather of ``[rows, n_classes]`` float32 per rank on its own
    (one warm-up, then ``reps`` timed calls between device synchronisations + barriers; MAX over ranks is not taken: rank 0's clock)."""
    world = dist.get_world_size()
    local = torch.zeros((rows, n_classes), dtype=torch.float32, device=device)
    out = torch.empty((world * rows, n_classes), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, local)
    torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_gather_into_tensor(out, local)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / reps
    ver = torch.cuda.nccl.version()
    return {"backend": dist.get_backend(), "ranks_seen": world, "nccl_version": ".".join(str(v) for v in ver) if isinstance(ver, tuple) else str(ver),
            "all_gather_ms": round(dt * 1e3, 4), "bytes_per_rank": rows * n_classes * 4, "op": "all_gather_into_tensor"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps per repeat (default 20 on one GPU, 8 = BASELINE configs[3] on several)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=25, help="how often the timed job of --steps steps is repeated (value = the median repeat)")
    ap.add_argument("--dtype", choices=["f32", "i8"], default="i8")
    ap.add_argument("--batch", type=int, default=0, help="chunks per GPU per step (default 4096 i8, 1024 f32)")
    ap.add_argument("--collective", action="store_true", help="initialise RCCL even on one GPU and run the all-gather inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs beside the main measurement (other configurations, hard inputs, evaluate end to end)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    steps = args.steps or (20 if args.gpus == 1 else 8)
    repeats = max(1, args.repeats)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    def init_rccl():
        import socket

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            probe = torch.zeros(8, device=device)
            dist.all_reduce(probe)  # the communicator (and its banner) comes up with the first collective
            torch.cuda.synchronize(device)

    use_dist = world > 1 or args.collective
    if use_dist:
        init_rccl()
        if world == 1:  # run the collective at world size 1 instead of returning the local tensor (evaluation/sharding.py)
            from birdnet_stm32.evaluation import sharding

            sharding.COLLECTIVE_AT_WORLD_1 = True

    from birdnet_stm32.models.runners import load_model_runner

    batch = args.batch or (1024 if args.dtype == "f32" else 4096)
    ckpt = os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100" + (".keras" if args.dtype == "f32" else ".tflite"))
    runner = load_model_runner(ckpt, device=local_rank, max_batch=batch)
    n_distinct = min(steps, MAX_DISTINCT_BATCHES)
    # this rank's block of the global chunk stream starts at chunk rank * steps * batch; batch k of it is pool[k % n_distinct]
    pool = [synth_audio_device(torch, batch, (rank * steps + k) * batch, device, 42 + rank * 1000 + k) for k in range(n_distinct)]
    scratch = torch.empty((batch, runner.num_classes), dtype=torch.float32, device=device)

    def barrier():
        if use_dist:
            dist.barrier()

    # Warm-up steps carry an event pair around EVERY operator: they give the per-stage table and name the dominant kernel.
    # In the timed region only that kernel is bracketed (26 event records per step would cost ~6 % of it).
    runner.profile(True)
    for w in range(args.warmup):
        if w == args.warmup - 1 and w > 0:  # first launches carry module loading: the table comes from the last warm-up step
            torch.cuda.synchronize(device)
            runner.profile_collect()
        runner.infer_audio_device(pool[w % n_distinct], hop=HOP, out=scratch)
    if use_dist:  # warm the RCCL communicator (and its buffers for this message size) outside the timed region
        warm = torch.empty((world * steps * batch, runner.num_classes), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(warm, warm[rank * steps * batch : (rank + 1) * steps * batch].clone())
        del warm
    torch.cuda.synchronize(device)
    warm_rows = runner.profile_collect()
    dom_op = max((r for r in warm_rows if r["launches"]), key=lambda r: r["ms"] / r["launches"])["op"] if args.warmup else -1
    runner.profile_only(dom_op)

    elapsed_all, finite = [], True
    for _ in range(repeats):
        elapsed, gathered = timed_job(lambda k, out: runner.infer_audio_device(pool[k % n_distinct], hop=HOP, out=out), steps, batch,
                                      runner.num_classes, device, barrier, lambda: torch.cuda.synchronize(device))
        assert gathered.shape == (world * steps * batch, runner.num_classes)
        elapsed_all.append(elapsed)
    finite = bool(torch.isfinite(gathered).all().item())
    del gathered
    runner.profile(False)
    rows = runner.profile_collect()
    runner.profile_only(-1)
    if args.warmup:  # stages from the warm-up profile, the dominant kernel's entry replaced by its timed-region measurement
        timed = {r["op"]: r for r in rows if r["launches"]}
        rows = [timed.get(r["op"], r) if r["op"] == dom_op else r for r in warm_rows]

    if world > 1:  # every repeat's time is the MAX over the ranks
        tt = torch.tensor(elapsed_all, dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_all = [float(v) for v in tt.cpu()]
    elapsed = float(np.median(elapsed_all))
    coll = collective_block(torch, dist, device, steps * batch, runner.num_classes) if use_dist else None

    if rank == 0:
        roof, stages = roofline_of(rows, batch, args.dtype, dom_op)
        total_chunks = world * batch * steps
        name = "birdnet_stm32n6_100 " + ("float32" if args.dtype == "f32" else "INT8") + " DS-CNN, hybrid+pwl frontend"
        if world > 1:
            name += f": {total_chunks} synthetic chunks sharded over {world} GPUs (BASELINE configs[3] is 262144 over 8), one RCCL all-gather of the scores"
        gather_note = (f" -> ONE RCCL all-gather of [{steps * batch}, {runner.num_classes}] f32 per rank "
                       f"({steps * batch * runner.num_classes * 4 / 1e6:.1f} MB), inside the timed region" if use_dist else "")
        out = {
            "metric": "audio chunks/sec (3 s @ 24 kHz)",
            "value": round(total_chunks / elapsed, 1),
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (peak-normalised tone + gaussian noise, generated on device); shipped birdnet_stm32n6_100 weights",
            "repeats": repeats,
            "value_min": round(total_chunks / max(elapsed_all), 1),
            "value_max": round(total_chunks / min(elapsed_all), 1),
            "timed_s_total": round(sum(elapsed_all), 3),
            "timed_region": f"{repeats} repeats of {steps} steps, each between barrier + device synchronisation (MAX over ranks); value = the median repeat",
            "config": {
                "workload": name,
                "batch_per_gpu": batch,
                "global_batch": world * batch,
                "chunks_total": total_chunks,
                "chunk": "3 s @ 24 kHz (72000 samples), n_fft 512, hop 281, 257x256 spectrogram",
                "path": "audio in HBM -> STFT -> mel+PWL -> DS-CNN -> scores in HBM" + gather_note,
                "distinct_input_batches_per_gpu": n_distinct,
                "int8_input_bytes": ("float32 STFT + float64 pass over every element whose byte the float32 error could change, inside the timed region: the reference's "
                                     "bytes on every input checked (0 differing bytes on 3e6 soaked chunks, profiles/r03_exact_soak.txt; again on 1.4e6 chunks of 12 families with round 4's kernels, profiles/r04_exact_soak.txt). The guard's error bound is "
                                     "EMPIRICAL (4 x the largest error seen; largest observed |S' - S| / eps = 0.343 over 2.2e11 elements), not a worst-case proof; "
                                     "bn_set_option('stft_exact', 1) computes every bin in float64 and needs no bound") if args.dtype == "i8" else None,
            },
            "scores_finite": finite,
            "whole_path_mfma_frac": round(total_chunks / world * MOP_PER_CHUNK * 1e6 / elapsed / 1e12 /
                                          (I8_MFMA_PEAK_TOPS if args.dtype == "i8" else F32_MFMA_PEAK_TFLOPS), 4),
            "roofline": roof,
            "stages": stages,
        }
        if coll is not None:
            coll["in_timed_region"] = True
            out["collective"] = coll
        if world == 1 and not args.batch and not args.no_extras:  # the other single-GPU BASELINE configurations, for reference (not the reported value)
            runner.close()
            runner = None
            del pool
            torch.cuda.empty_cache()
            if args.dtype == "i8":
                try:  # what the headline costs on inputs the benchmark's own generator never produces
                    out["hard_inputs"] = hard_inputs(torch, device, local_rank, out["value"])
                except Exception as e:  # noqa: BLE001
                    out["hard_inputs"] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            other = "i8" if args.dtype == "f32" else "f32"
            out["also_measured"] = quick_rate(torch, other, 4096 if other == "i8" else 1024, device, local_rank)
            for key, kw in (("also_measured_configs4", {}), ("also_measured_configs4_3s", {"seconds": 3}), ("also_measured_configs4_int8", {"int8": True})):
                try:  # reported extras: a failure here must not cost the main line
                    out[key] = config4_rate(torch, device, local_rank, **kw)
                except Exception as e:  # noqa: BLE001
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not use_dist:
            # one GPU, no collective in the path: still show that RCCL comes up on this box and moves the score tensor (outside the timed region)
            try:
                init_rccl()
                out["collective"] = dict(collective_block(torch, dist, device, steps * batch, 100), in_timed_region=False,
                                         note="probe behind the measurement; `--collective` puts the all-gather inside the timed region")
            except Exception as e:  # noqa: BLE001
                out["collective"] = {"backend": None, "ranks_seen": 1, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.batch and not args.no_extras and not use_dist:
            try:
                out["also_measured_evaluate"] = evaluate_leg()
            except Exception as e:  # noqa: BLE001
                out["also_measured_evaluate"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.dtype)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
    if runner is not None:
        runner.close()


if __name__ == "__main__":
    main()
