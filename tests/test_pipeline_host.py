"""Host side of the pipelined ``evaluate`` (birdnet_stm32/audio/pipeline.py, csrc/host/bn_pcmio.c): probing, planning, slab layout and
offset tables — everything in front of the first GPU call — checked on the CPU against the per-file functions of
``birdnet_stm32.audio.io`` / ``audio.ingest`` (themselves pinned to numpy / scipy in tests/test_oracle_pinning.py).

Reference behaviour being matched: birdnet_stm32/audio/io.py:63-130 (read window), :133-174 (chunk starts), evaluation/metrics.py:117-141
(files in order, chunks of a file contiguous).
"""

from __future__ import annotations

import ctypes
import os
import struct

import numpy as np
import pytest

from birdnet_stm32.audio import _pcmio, ingest
from birdnet_stm32.audio import io as aio
from birdnet_stm32.audio import pipeline as pl


def _write_wav(path, x, sr, bits=16, code=1, extra=b"", extensible=False, data_size=None):
    ch = x.shape[1]
    if code == 3:
        payload = x.astype("<f4" if bits == 32 else "<f8").tobytes()
    elif bits == 8:
        payload = x.astype(np.uint8).tobytes()
    elif bits == 16:
        payload = x.astype("<i2").tobytes()
    elif bits == 24:
        v = x.astype(np.int32)
        payload = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], axis=-1).astype(np.uint8).tobytes()
    else:
        payload = x.astype("<i4").tobytes()
    if extensible:
        fmt = struct.pack("<4sIHHIIHHHHIH14s", b"fmt ", 40, 0xFFFE, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits, 22, bits, 0, code,
                          b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    else:
        fmt = struct.pack("<4sIHHIIHH", b"fmt ", 16, code, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits)
    hdr = struct.pack("<4sI4s", b"RIFF", 0, b"WAVE") + extra + fmt + struct.pack("<4sI", b"data", len(payload) if data_size is None else data_size)
    with open(path, "wb") as fh:
        fh.write(hdr + payload)


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipe")
    rng = np.random.default_rng(3)
    paths = []
    for i in range(41):
        sr = [24000, 48000, 44100, 22050, 16000][i % 5]
        n = int(sr * [0.5, 3.0, 7.3, 61.0, 2.99, 3.01][i % 6])
        ch = 1 + i % 3
        p = str(d / f"f{i:02d}.wav")
        odd_chunk = (b"LIST" + struct.pack("<I", 5) + b"abcde\0") if i % 2 else b""
        if i % 7 == 0:
            _write_wav(p, rng.standard_normal((n, ch)).astype(np.float32) * 0.3, sr, 32, 3, extra=odd_chunk)
        elif i % 7 == 1:
            _write_wav(p, rng.integers(-(1 << 23), 1 << 23, (n, ch)), sr, 24, extra=odd_chunk, extensible=True)
        elif i % 7 == 2:
            _write_wav(p, rng.integers(-(1 << 31), 1 << 31, (n, ch)), sr, 32)
        elif i % 7 == 3 and i % 6 != 3:
            _write_wav(p, rng.integers(0, 256, (n, ch)), sr, 8)  # 8-bit: decoded on the host (kind 1)
        else:
            _write_wav(p, rng.integers(-20000, 20000, (n, ch)), sr, 16, extra=odd_chunk, data_size=(1 << 31) if i % 11 == 4 else None)
        paths.append(p)
    for tag, name in ((7, "g711_mu.wav"), (6, "g711_a.wav")):   # G.711 (libsndfile reads them): decoded on the host like 8-bit PCM
        _write_wav(str(d / name), rng.integers(0, 256, (8000 * 7, 1 + tag % 2)), 8000, 8, code=tag)
        paths.append(str(d / name))
    (d / "bad.wav").write_bytes(b"nope")
    (d / "empty.wav").write_bytes(b"")
    _write_wav(str(d / "nodata.wav"), np.zeros((0, 1)), 24000)
    paths.insert(5, str(d / "bad.wav"))
    paths.insert(9, str(d / "empty.wav"))
    paths.insert(11, str(d / "missing.wav"))
    paths.insert(20, str(d / "nodata.wav"))
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import flac_writer as fw  # the tests' own encoder (RFC 9639)

    n = 44100 + 123
    for name, bps, known in (("s16.flac", 16, True), ("s24.flac", 24, True), ("nolen.flac", 16, False)):
        x = rng.integers(-9000, 9000, (n, 2)).astype(np.int64) << (bps - 16)
        frames = [{"n": 4096, "mode": "indep", "sub": [dict(kind="fixed", order=2, po=3)] * 2} for _ in range(n // 4096)]
        frames.append({"n": n % 4096, "mode": "indep", "sub": [dict(kind="fixed", order=1, po=0)] * 2})
        (d / name).write_bytes(fw.encode(x, 44100, bps, frames, total_known=known))
        paths.insert(14, str(d / name))
    return paths


def test_probe_matches_the_python_header_walk(dataset, tmp_path):
    rng = np.random.default_rng(5)
    paths = list(dataset)
    # fuzz: truncated files, chunk sizes running past the end, junk chunks
    base = open(dataset[0], "rb").read()[:4096]
    for k in range(60):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(12, 80))] = int(rng.integers(0, 256))
        p = tmp_path / f"z{k}.wav"
        p.write_bytes(bytes(b[: int(rng.integers(8, len(b)))]))
        paths.append(str(p))
    lay = _pcmio.probe_wavs(paths, 3)
    for p, row in zip(paths, lay):
        try:
            want = aio._wav_layout(open(p, "rb").read())
        except Exception:
            want = None
        if want is None:
            assert row["status"] != 0, p
        else:
            assert row["status"] == 0, p
            got = (int(row["format_tag"]), int(row["channels"]), int(row["sample_rate"]), int(row["bits"]), int(row["data_offset"]), int(row["data_bytes"]))
            assert got == tuple(int(v) for v in want), p


def test_vectorised_tables_match_the_per_file_functions():
    rng = np.random.default_rng(1)
    total = rng.integers(0, 4_000_000, 500)
    sr = rng.choice([8000, 16000, 22050, 24000, 44100, 48000, 96000], 500)
    for md in (60, 30, 0.5, None):
        got = pl.window_counts(total, sr, md)
        want = [ingest._window_frames(int(t), int(s), md, 3.0, False)[1] for t, s in zip(total, sr)]
        assert got.tolist() == [max(0, w) for w in want]
    n_out = pl.resampled_lengths(total, sr, 24000)
    from math import gcd

    assert n_out.tolist() == [ingest.resampled_length(int(t), 24000 // gcd(int(s), 24000), int(s) // gcd(int(s), 24000)) if s != 24000 else int(t)
                              for t, s in zip(total, sr)]
    for cd, ov in ((3.0, 0.0), (3.0, 1.5), (2.0, 5.0), (0.7, 0.3)):
        lens = np.concatenate([rng.integers(0, 300_000, 200), [0, 1, int(24000 * cd) - 1, int(24000 * cd), int(24000 * cd) + 1]])
        s0, v0, o0, c0, size0 = ingest.chunk_table(lens.tolist(), 24000, cd, ov)
        s1, v1, o1, c1, size1 = pl.chunk_table_arrays(lens, 24000, cd, ov)
        assert size0 == size1 and c1.tolist() == c0 and np.array_equal(s0, s1) and np.array_equal(v0, v1) and np.array_equal(o0, o1)
        assert pl.chunk_counts(lens, 24000, cd, ov).tolist() == [aio.estimate_num_chunks(int(n), 24000, cd, ov) for n in lens]


def _decode(buf, fmt, ch):
    if fmt == ingest.PCM_S16:
        x = np.frombuffer(buf, "<i2").astype(np.float32) / 32768.0
    elif fmt == ingest.PCM_S24:
        b = np.frombuffer(buf, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif fmt == ingest.PCM_S32:
        x = (np.frombuffer(buf, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        x = np.frombuffer(buf, "<f4").astype(np.float32)
    return x.reshape(-1, ch)


@pytest.mark.parametrize("overlap", [0.0, 1.0])
def test_slab_layout_and_tables_reproduce_load_audio_file(dataset, overlap):
    """Emulate bn_ingest_resample + bn_ingest_chunks on the host from the slab and the table exactly as the kernels read them; the
    chunks must be the ones ``load_audio_file`` returns for every file, in file order."""
    sr, cd = 24000, 3.0
    tab = pl.plan_files(dataset, sr, cd, overlap, 60, 4)
    for i, p in enumerate(dataset):
        w = ingest.read_pcm_window(p, 60, cd, False)
        if w is None or w.frames <= 0:
            assert tab.kind[i] == -1, p
        else:
            assert tab.kind[i] in (0, 1) and (tab.fmt[i], tab.channels[i], tab.sr0[i], tab.frames[i], tab.nbytes[i]) == (w.fmt, w.channels, w.sample_rate, w.frames, w.payload.nbytes), p
    assert (tab.kind == 1).sum() >= 1  # the 8-bit files (and the FLAC stream) take the host-decoded route
    groups = pl.cut_groups(tab.nbytes, tab.n_chunks, 12 << 20, 48)
    assert groups[0][0] == 0 and groups[-1][1] == len(dataset) and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
    slab = np.zeros(40 << 20, np.uint8)
    table = np.zeros(1 << 15, np.int64)
    seen = 0
    for lo, hi in groups:
        lay = pl.layout_group(tab, lo, hi, slab.ctypes.data, slab.size, table, sr, cd, overlap, 60, 4)
        assert lay.counts.tolist() == tab.n_chunks[lo:hi].tolist()
        mono = np.zeros(lay.total_out, np.float32)
        peak = np.zeros(lay.n_windows, np.float32)
        out_off = table[lay.off_out : lay.off_out + lay.n_windows + 1]
        for pos, fmt, ch, sr0, w0, nw, in_off, max_in, max_out in lay.subs:
            assert pos % 256 == 0
            ino = table[in_off : in_off + nw + 1]
            fb = ingest._BYTES[fmt] * ch
            assert max_in == int(np.diff(ino).max()) and max_out == int(np.diff(out_off[w0 : w0 + nw + 1]).max())
            for j in range(nw):
                frames = _decode(slab[pos + ino[j] * fb : pos + ino[j + 1] * fb].tobytes(), fmt, ch)
                y = frames.mean(axis=1).astype(np.float32, copy=False)
                y = aio.fast_resample(y, sr0, sr)
                assert y.shape[0] == out_off[w0 + j + 1] - out_off[w0 + j]
                mono[out_off[w0 + j] : out_off[w0 + j + 1]] = y
                peak[w0 + j] = np.abs(y).max()
        src = table[lay.off_src : lay.off_src + lay.n_chunks]
        valid = table[lay.off_valid : lay.off_valid + (lay.n_chunks + 1) // 2].view(np.int32)[: lay.n_chunks]
        owner = table[lay.off_owner : lay.off_owner + (lay.n_chunks + 1) // 2].view(np.int32)[: lay.n_chunks]
        size = int(sr * cd)
        chunks = np.zeros((lay.n_chunks, size), np.float32)
        for c in range(lay.n_chunks):
            pk = peak[owner[c]]
            seg = mono[src[c] : src[c] + valid[c]]
            chunks[c, : valid[c]] = seg / pk if pk > 0 else seg
        row = 0
        for i in range(lo, hi):
            want = aio.load_audio_file(dataset[i], sr, 60, cd, overlap)
            n = len(want)
            assert n == lay.counts[i - lo], dataset[i]
            if n:
                assert np.array_equal(chunks[row : row + n], np.asarray(want)), dataset[i]
            row += n
        assert row == lay.n_chunks
        seen += row
    assert seen == int(tab.n_chunks.sum())


def test_a_file_that_vanishes_after_probing_drops_out(dataset, tmp_path):
    import shutil

    paths = []
    for i, p in enumerate(dataset[:8]):
        q = str(tmp_path / os.path.basename(p))
        if os.path.isfile(p):
            shutil.copy(p, q)
        paths.append(q)
    tab = pl.plan_files(paths, 24000, 3.0, 0.0, 60, 2)
    victim = int(np.flatnonzero(tab.kind == 0)[1])
    os.remove(paths[victim])
    slab = np.zeros(40 << 20, np.uint8)
    table = np.zeros(1 << 15, np.int64)
    lay = pl.layout_group(tab, 0, len(paths), slab.ctypes.data, slab.size, table, 24000, 3.0, 0.0, 60, 2)
    assert victim not in lay.files.tolist() and lay.counts[victim] == 0
    keep = [i for i in range(len(paths)) if tab.kind[i] >= 0 and i != victim]
    assert lay.files.tolist() == keep and lay.n_chunks == int(tab.n_chunks[keep].sum())


def test_read_windows_and_copy_into_fill_exact_ranges(tmp_path):
    rng = np.random.default_rng(0)
    blobs = [rng.integers(0, 256, int(n), dtype=np.uint8) for n in (1, 4097, 70000, 0, 333)]
    paths = []
    for i, b in enumerate(blobs):
        p = tmp_path / f"b{i}.bin"
        p.write_bytes(b.tobytes())
        paths.append(str(p))
    buf = np.full(200000, 0xEE, np.uint8)
    off = np.array([7, 100, 5000, 80000, 90000], np.int64)
    foff = np.array([0, 10, 100, 0, 3], np.int64)
    nb = np.array([1, 4000, 60000, 0, 330], np.int64)
    st = _pcmio.read_windows(paths, foff, nb, buf.ctypes.data, off, 3)
    assert st.tolist() == [0, 0, 0, 0, 0]
    want = np.full(200000, 0xEE, np.uint8)
    for b, o, f, n in zip(blobs, off, foff, nb):
        want[o : o + n] = b[f : f + n]
    assert np.array_equal(buf, want)
    st = _pcmio.read_windows(paths[:2] + [str(tmp_path / "nope")], np.array([0, 4090, 0]), np.array([1, 100, 5]), buf.ctypes.data, np.array([0, 10, 200]), 2)
    assert st.tolist() == [0, _pcmio.IO_SHORT, _pcmio.IO_OPEN]
    dst = np.zeros(1000, np.uint8)
    _pcmio.copy_into([blobs[4], blobs[0]], dst.ctypes.data, np.array([10, 500], np.int64), 2)
    assert np.array_equal(dst[10:343], blobs[4]) and dst[500] == blobs[0][0] and dst[:10].sum() == 0


def test_g711_wave_files_decode_like_the_standards_tables(tmp_path):
    """WAVE format tags 6 (A-law) and 7 (mu-law), which libsndfile — the reference's reader, audio/io.py:90,114-116 — opens as 16-bit samples: the
    byte -> sample tables against ITU-T G.711's known answers and an independent bit-level decoder written here, a stereo file through
    ``load_audio_window`` (mono mean, scaling by 1 / 32768), and the evaluate pipeline's planner taking such files as host-decoded windows."""
    alaw, ulaw = aio._g711_tables()
    assert [int(ulaw[b]) for b in (0xFF, 0x7F, 0x00, 0x80, 0x0F, 0x8F)] == [0, 0, -32124, 32124, -16764, 16764]
    assert [int(alaw[b]) for b in (0xD5, 0x55, 0xAA, 0x2A, 0x80, 0x00)] == [8, -8, 32256, -32256, 5504, -5504]

    def ulaw_ref(b):   # G.711 section 3: invert, then sign | 3-bit exponent | 4-bit mantissa with the bias of 33 (<< 2 in 16-bit terms)
        u = ~b & 0xFF
        t = (((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7)
        return (0x84 - t) if u & 0x80 else (t - 0x84)

    def alaw_ref(b):   # G.711 section 2: toggle the even bits, then sign | exponent | mantissa, the lowest segment linear
        a = b ^ 0x55
        e, m = (a >> 4) & 7, a & 0x0F
        t = (m << 4) + 8 if e == 0 else ((m << 4) + 0x108) << (e - 1)
        return t if a & 0x80 else -t

    assert [ulaw_ref(b) for b in range(256)] == ulaw.tolist() and [alaw_ref(b) for b in range(256)] == alaw.tolist()
    assert sorted(set(ulaw.tolist())) == sorted(set((-ulaw).tolist())) and len(set(alaw.tolist())) == 256   # symmetric; A-law has no zero
    rng = np.random.default_rng(8)
    sr, n = 8000, 8000 * 4
    codes = rng.integers(0, 256, (n, 2))
    for tag, table, name in ((7, ulaw, "mu.wav"), (6, alaw, "a.wav")):
        path = str(tmp_path / name)
        _write_wav(path, codes, sr, 8, code=tag)
        y = aio.load_audio_window(path, sample_rate=sr, max_duration=60, chunk_duration=3.0, random_offset=False)
        want = table[codes].astype(np.float32).mean(axis=1) / np.float32(32768.0)
        want = want / (np.abs(want).max() + 1e-12)
        assert y.shape == want.shape and np.allclose(y, want, atol=1e-6), name
        tab = pl.plan_files([path], sr, 3.0, 0.0, 60, 2)
        assert tab.kind[0] == 1 and tab.frames[0] == n and tab.channels[0] == 2 and tab.n_chunks[0] > 0   # decoded on the host, like 8-bit PCM


def test_a_file_truncated_under_the_mapped_reader_is_a_short_read_not_a_signal(tmp_path):
    """mmap mode: a file that shrinks while it is being copied out of its mapping raises SIGBUS in the copying thread.  The reader installs a
    guard for the duration of a call (csrc/host/bn_pcmio.c: bus_guard): the fault unwinds the copy, the window goes to pread, which reports it
    short.  The library's self-test provokes exactly that (map, truncate through a second descriptor, copy); afterwards ordinary reads work and
    the process's SIGBUS disposition is what it was."""
    import signal

    before = signal.getsignal(signal.SIGBUS)
    p = tmp_path / "shrinks.bin"
    p.write_bytes(os.urandom(1 << 20))
    assert _pcmio._load().bn_host_selftest_truncated_map(str(p).encode()) == -1 and p.stat().st_size == 0
    assert _pcmio._load().bn_host_selftest_truncated_map(str(p).encode()) == 0          # (nothing to map any more: set-up fails, no signal)
    assert signal.getsignal(signal.SIGBUS) == before
    blob = np.frombuffer(os.urandom(300000), np.uint8)
    q = tmp_path / "fine.bin"
    q.write_bytes(blob.tobytes())
    buf = np.zeros(300000, np.uint8)
    prev = _pcmio.set_read_mode("mmap")
    try:
        st = _pcmio.read_windows([str(q), str(p)], np.array([0, 0]), np.array([300000, 70000]), buf.ctypes.data, np.array([0, 0]), 2)
    finally:
        _pcmio.set_read_mode(prev)
    assert st.tolist() == [0, _pcmio.IO_SHORT] and np.array_equal(buf, blob)


def test_reader_share_honours_the_cgroup_cpu_quota(tmp_path, monkeypatch):
    """A container's CPU quota does not show in the affinity mask (the timing box: 256 CPUs listed, 16 granted): ``cgroup_cpu_quota`` reads it from
    cgroup v2's ``cpu.max`` or v1's ``cpu.cfs_*`` (rounded up; no quota = None) and ``default_threads`` divides min(affinity, quota) by the ranks
    on the host."""
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert _pcmio.cgroup_cpu_quota(str(v2)) == 16
    (v2 / "cpu.max").write_text("150000 100000\n")
    assert _pcmio.cgroup_cpu_quota(str(v2)) == 2
    (v2 / "cpu.max").write_text("max 100000\n")
    assert _pcmio.cgroup_cpu_quota(str(v2)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("800000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert _pcmio.cgroup_cpu_quota(str(v1)) == 8
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert _pcmio.cgroup_cpu_quota(str(v1)) is None
    assert _pcmio.cgroup_cpu_quota(str(tmp_path / "none")) is None
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    for quota, world, want in ((16, 1, 16), (16, 8, 2), (64, 8, 8), (None, 8, 16), (None, 64, 4), (4, 1, 4)):
        monkeypatch.setattr(_pcmio, "cgroup_cpu_quota", lambda root="": quota)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(world))
        assert _pcmio.default_threads() == want, (quota, world)


def test_read_modes_give_the_same_bytes_and_status(tmp_path):
    """The reader's two ways out of the page cache (pread | mmap + MADV_SEQUENTIAL, csrc/host/bn_pcmio.c) fill the same bytes and report the same
    status: page-unaligned offsets, windows ending exactly at / running past the end of the file, windows below the mmap threshold."""
    rng = np.random.default_rng(5)
    sizes = (300000, 65536 + 4097, 1 << 20, 70000, 100)
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in sizes]
    paths = []
    for i, b in enumerate(blobs):
        (tmp_path / f"m{i}.bin").write_bytes(b.tobytes())
        paths.append(str(tmp_path / f"m{i}.bin"))
    foff = np.array([4097, 1, 123457, 3, 0], np.int64)
    nb = np.array([290000, 65536 + 4096, (1 << 20) - 123457, 70000, 100], np.int64)   # [3] runs 3 bytes past the end: short
    off = np.concatenate([[0], np.cumsum(nb + 64)[:-1]]).astype(np.int64)
    prev = _pcmio.set_read_mode(None)
    try:
        out = {}
        for mode in ("pread", "mmap"):
            assert _pcmio.set_read_mode(mode) in ("pread", "mmap") and _pcmio.set_read_mode(None) == mode
            buf = np.full(int(off[-1] + nb[-1] + 64), 0xEE, np.uint8)
            st = _pcmio.read_windows(paths, foff, nb, buf.ctypes.data, off, 3)
            out[mode] = (st.tolist(), buf)
        assert out["pread"][0] == out["mmap"][0] == [0, 0, 0, _pcmio.IO_SHORT, 0]
        assert np.array_equal(out["pread"][1], out["mmap"][1])
        for i in (0, 1, 2, 4):
            assert np.array_equal(out["mmap"][1][off[i] : off[i] + nb[i]], blobs[i][foff[i] : foff[i] + nb[i]])
        with pytest.raises(ValueError):
            _pcmio.set_read_mode("direct")
    finally:
        _pcmio.set_read_mode(prev)


def test_cut_groups_limits_and_ramp():
    """Groups are contiguous, respect the slab / chunk limits, and the first three are cut at 1/8, 1/4, 1/2 of a slab (the copy stream starts on a
    small slab); ``ramp=()`` gives full slabs from the start; a file larger than a limit is a group of its own."""
    rng = np.random.default_rng(3)
    nbytes = rng.integers(1 << 20, 3 << 20, 400).astype(np.int64)
    n_chunks = rng.integers(1, 12, 400).astype(np.int64)
    slab = 64 << 20
    for ramp in ((8, 4, 2), ()):
        groups = pl.cut_groups(nbytes, n_chunks, slab, 4096, ramp)
        assert groups[0][0] == 0 and groups[-1][1] == 400 and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
        for i, (a, b) in enumerate(groups):
            lim = slab // (ramp[i] if i < len(ramp) else 1)
            assert b - a == 1 or int((nbytes[a:b] + 256).sum()) <= lim + 256 * (b - a), (i, a, b)
        if ramp:
            sizes = [int(nbytes[a:b].sum()) for a, b in groups[:4]]
            assert sizes[0] < sizes[1] < sizes[2] < sizes[3]
    big = np.array([1 << 20, 200 << 20, 1 << 20], np.int64)
    assert pl.cut_groups(big, np.array([1, 900, 1], np.int64), slab, 4096, ()) == [(0, 1), (1, 2), (2, 3)]
    assert [b - a for a, b in pl.cut_groups(np.full(10, 1 << 10, np.int64), np.full(10, 5, np.int64), slab, 12, ())] == [2, 2, 2, 2, 2]


def test_balanced_bounds_are_contiguous_and_even():
    rng = np.random.default_rng(2)
    w = rng.integers(0, 21, 1000)
    for world in (1, 2, 3, 8):
        b = pl.balanced_bounds(w, world)
        assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:])) and len(b) == world + 1
        loads = [int(w[b[r] : b[r + 1]].sum()) for r in range(world)]
        assert max(loads) - min(loads) <= 2 * 21 + world, loads
    assert pl.balanced_bounds([], 4) == [0, 0, 0, 0, 0]
    assert pl.balanced_bounds([5], 4)[-1] == 1
    # a few long files among many short ones: by-count dealing would give rank 0 ten times the work
    w = np.array([20] * 100 + [1] * 900)
    b = pl.balanced_bounds(w, 2)
    loads = [int(w[b[r] : b[r + 1]].sum()) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= 60  # (within a file of 20 chunks of the weighted target on each side: the per-file host cost counts too)
    # the cut nearest to the target, not the first one past it: 20 | 28 chunks instead of 40 | 8
    assert pl.balanced_bounds([20, 20, 1, 1, 1, 1, 1, 1, 1, 1], 2) == [0, 1, 10]
    # eight ranks, 262 144 files' worth of metadata: disjoint, covering, and no rank more than one file's chunks away from the mean
    w = rng.integers(1, 31, 262144)
    b = pl.balanced_bounds(w, 8)
    loads = np.array([int(w[b[r] : b[r + 1]].sum()) for r in range(8)])
    assert b[0] == 0 and b[-1] == len(w) and all(x < y for x, y in zip(b, b[1:]))
    assert np.abs(loads - w.sum() / 8).max() <= 30, loads


def test_host_library_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "bn_host.h")).read()
    import re

    declared = set(re.findall(r"\b(bn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_pcmio.EXPORTS)
    lib = ctypes.CDLL(_pcmio._LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_reader_share_and_numa_placement_for_ranks_sharing_a_host(monkeypatch, tmp_path):
    """Eight ranks on one host must not start 8 x 16 reader threads: a rank's readers are its share of the CPUs it may use (at most 16);
    a rank's CPUs are those of ITS GPU's NUMA node, dealt evenly among the local ranks on that node (sysfs: numa_node / local_cpulist)."""
    from birdnet_stm32.audio import _pcmio

    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert _pcmio.local_world_size() == 1 and _pcmio.default_threads() == 16
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert _pcmio.local_world_size() == 8 and _pcmio.default_threads() == 16          # 128 / 8
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    assert _pcmio.default_threads() == 8                                                  # 64 / 8
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    assert _pcmio.default_threads() == 2                                                  # never below two
    assert _pcmio._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    # sysfs of a two-socket host with four GPUs per socket
    for i, (node, cpus) in enumerate([(0, "0-31,64-95")] * 4 + [(1, "32-63,96-127")] * 4):
        d = tmp_path / f"0000:{0xc1 + i:02x}:00.0"
        d.mkdir()
        (d / "numa_node").write_text(f"{node}\n")
        (d / "local_cpulist").write_text(cpus + "\n")
    info = [_pcmio.gpu_numa_cpus(f"0000:{0xc1 + i:02X}:00.0", sysfs=str(tmp_path)) for i in range(8)]
    assert [n for n, _ in info] == [0] * 4 + [1] * 4 and len(info[0][1]) == 64
    assert _pcmio.gpu_numa_cpus("0000:ff:00.0", sysfs=str(tmp_path)) == (-1, set())
    allowed = set(range(128))
    shares = [_pcmio.rank_cpu_share(r, [n for n, _ in info], [c for _, c in info], allowed) for r in range(8)]
    assert all(len(s) == 16 for s in shares)                                             # 64 CPUs of a node over its four ranks
    assert all(shares[a].isdisjoint(shares[b]) for a in range(8) for b in range(a + 1, 8))
    assert set().union(*shares[:4]) == info[0][1] and set().union(*shares[4:]) == info[4][1]
    # a cgroup that leaves the process half of every node: the share shrinks with it; an unknown layout leaves the affinity alone
    half = {c for c in allowed if c % 2 == 0}
    assert len(_pcmio.rank_cpu_share(5, [n for n, _ in info], [c for _, c in info], half)) == 8
    assert _pcmio.rank_cpu_share(0, [-1] * 8, [set()] * 8, allowed) == set()
