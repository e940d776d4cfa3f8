"""ctypes binding of the threaded file reader in ``lib/libbn_host.so`` (``csrc/host/bn_pcmio.c``, C ABI ``include/bn_host.h``).

The reference reads its test files one at a time through libsndfile (reference: birdnet_stm32/audio/io.py:89-117, called per file
from evaluation/metrics.py:117-125).  The device pipeline of ``evaluate`` needs only the container layout on the host — the samples
travel to the GPU as they lie in the file — so this module offers exactly two bulk operations, both running on a pool of POSIX
threads outside the GIL: ``probe_wavs`` (header walks of many files) and ``read_windows`` (``pread`` of many byte ranges straight into
one page-locked slab).
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "lib", "libbn_host.so")
_lib = None

IO_OK, IO_OPEN, IO_NOT_WAVE, IO_NO_CHUNK, IO_SHORT = 0, -1, -2, -3, -4

# every symbol include/bn_host.h declares
EXPORTS = ("bn_wav_probe", "bn_wav_probe_many", "bn_file_read_many", "bn_file_read_many_mode", "bn_host_set_read_mode", "bn_host_selftest_truncated_map", "bn_copy_many", "bn_flac_info", "bn_flac_md5", "bn_flac_decode")

LAYOUT_DTYPE = np.dtype([("status", "<i4"), ("format_tag", "<i4"), ("channels", "<i4"), ("sample_rate", "<i4"), ("bits", "<i4"),
                         ("reserved", "<i4"), ("data_offset", "<i8"), ("data_bytes", "<i8")])


def _load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: run `make -C birdnet-stm32_amd/csrc` (or __graft_entry__.build())")
        lib = ctypes.CDLL(_LIB_PATH)
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError(f"{_LIB_PATH} does not export {name}")
        vp, i64p = ctypes.c_void_p, ctypes.c_void_p
        lib.bn_wav_probe.argtypes = [ctypes.c_char_p, vp]
        lib.bn_wav_probe_many.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, vp, ctypes.c_int]
        lib.bn_file_read_many.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, i64p, i64p, vp, i64p, vp, ctypes.c_int]
        lib.bn_file_read_many_mode.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, i64p, i64p, vp, i64p, vp, ctypes.c_int, ctypes.c_int]
        lib.bn_host_set_read_mode.argtypes = [ctypes.c_int]
        lib.bn_host_selftest_truncated_map.argtypes = [ctypes.c_char_p]
        lib.bn_copy_many.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, i64p, vp, i64p, ctypes.c_int]
        _lib = lib
    return _lib


def local_world_size() -> int:
    """Ranks of this job on THIS host (``LOCAL_WORLD_SIZE`` of torch.distributed.run; 1 outside it)."""
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        return 1


def cgroup_cpu_quota(root: str = "/sys/fs/cgroup") -> int | None:
    """CPUs the container's cgroup lets this process use at once (``cpu.max`` of cgroup v2, ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us`` of v1,
    rounded up), or None without a quota.  The affinity mask does not show it: the timing box of round 5 lists 256 CPUs and grants 16."""
    try:
        quota, period = open(os.path.join(root, "cpu.max")).read().split()[:2]
        return None if quota == "max" else max(1, -(-int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open(os.path.join(root, "cpu", "cpu.cfs_quota_us")).read())
        period = int(open(os.path.join(root, "cpu", "cpu.cfs_period_us")).read())
        return max(1, -(-quota // period)) if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def default_threads() -> int:
    """Reader threads of this process: its share of the CPUs it may run on — the affinity mask, capped by the cgroup's CPU quota, divided by the
    ranks on this host — at most 16 (what one GPU's PCIe link can use), at least 2.  Eight ranks on a 128-thread host get 16 each, on a
    64-thread host (or under a 64-CPU quota) 8: the reader pools of a node never oversubscribe it."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 2
    quota = cgroup_cpu_quota()
    if quota is not None:
        n = min(n, quota)
    return max(2, min(16, n // local_world_size()))


def _parse_cpulist(text: str) -> set[int]:
    cpus: set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(pci_bus_id: str, sysfs: str = "/sys/bus/pci/devices") -> tuple[int, set[int]]:
    """(NUMA node, CPUs local to it) of the PCI device ``pci_bus_id`` ("0000:c1:00.0") from sysfs; (-1, empty) when the host does not say."""
    base = os.path.join(sysfs, pci_bus_id.lower())
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
        cpus = _parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except (OSError, ValueError):
        return -1, set()
    return node, cpus


def rank_cpu_share(local_rank: int, nodes: list[int], node_cpus: list[set[int]], allowed: set[int]) -> set[int]:
    """The CPUs local rank ``local_rank`` should run on: those of ITS GPU's NUMA node that the process may use, dealt evenly (in CPU order)
    among the local ranks whose GPUs sit on the same node.  ``nodes[r]`` / ``node_cpus[r]`` = node and local CPUs of local rank r's GPU.
    Empty = no information (leave the affinity alone)."""
    if not (0 <= local_rank < len(nodes)) or nodes[local_rank] < 0:
        return set()
    mine = sorted(node_cpus[local_rank] & allowed)
    peers = [r for r in range(len(nodes)) if nodes[r] == nodes[local_rank]]
    if not mine or local_rank not in peers:
        return set()
    k, n = peers.index(local_rank), len(peers)
    share = mine[k * len(mine) // n:(k + 1) * len(mine) // n]
    return set(share)


def pin_process_to_gpu_node(torch, local_rank: int, n_local: int) -> dict:
    """Restrict this process (the reader threads it starts later inherit the mask, and the page-locked slabs it allocates are first touched
    under it) to the CPUs of its GPU's NUMA node, shared fairly with the other local ranks on that node.  Called by the evaluate pipeline
    when several ranks share a host; returns what it did (for the run's stats)."""
    try:
        allowed = set(os.sched_getaffinity(0))
        ids = []
        for r in range(n_local):
            pr = torch.cuda.get_device_properties(r)
            ids.append(f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
        info = [gpu_numa_cpus(i) for i in ids]
        share = rank_cpu_share(local_rank, [n for n, _ in info], [c for _, c in info], allowed)
        if share:
            os.sched_setaffinity(0, share)
        return {"numa_node": info[local_rank][0], "cpus": len(share), "pinned": bool(share)}
    except Exception as e:  # pragma: no cover - best effort: an unknown host layout must not stop the run
        return {"numa_node": -1, "cpus": 0, "pinned": False, "error": str(e)}


def _path_array(paths):
    arr = (ctypes.c_char_p * len(paths))()
    arr[:] = [os.fsencode(p) for p in paths]
    return arr


def probe_wavs(paths: list[str], n_threads: int | None = None) -> np.ndarray:
    """Structured array (``LAYOUT_DTYPE``) with the RIFF/WAVE layout of every path; ``status`` != 0 marks what is not a WAV file."""
    out = np.zeros(len(paths), LAYOUT_DTYPE)
    if paths:
        _load().bn_wav_probe_many(_path_array(paths), len(paths), out.ctypes.data, int(n_threads or default_threads()))
    return out


READ_MODES = {"pread": 0, "mmap": 1}


def set_read_mode(mode: str | None) -> str:
    """Process-wide way ``read_windows`` takes bytes out of the page cache: ``"mmap"`` (default: mmap + MADV_SEQUENTIAL + memcpy — the first
    read of freshly written files is not slowed by LRU activation) or ``"pread"``; ``None`` only queries.  Returns the previous mode."""
    if mode is not None and mode not in READ_MODES:
        raise ValueError(f"read mode {mode!r}: expected one of {sorted(READ_MODES)}")
    prev = _load().bn_host_set_read_mode(-1 if mode is None else READ_MODES[mode])
    return "pread" if prev == 0 else "mmap"


def read_windows(paths: list[str], file_off: np.ndarray, nbytes: np.ndarray, base_ptr: int, dst_off: np.ndarray,
                 n_threads: int | None = None, mode: str | None = None) -> np.ndarray:
    """``nbytes[i]`` bytes from offset ``file_off[i]`` of ``paths[i]`` into ``base_ptr + dst_off[i]``; returns the per-file status.

    The caller owns the destination (a page-locked slab), guarantees ``dst_off[i] + nbytes[i]`` stays inside it and that ranges
    do not overlap.
    """
    n = len(paths)
    status = np.zeros(n, np.int32)
    if n:
        fo = np.ascontiguousarray(file_off, np.int64)
        nb = np.ascontiguousarray(nbytes, np.int64)
        do = np.ascontiguousarray(dst_off, np.int64)
        if not (fo.shape == nb.shape == do.shape == (n,)):
            raise ValueError("file_off, nbytes and dst_off need one entry per path")
        _load().bn_file_read_many_mode(_path_array(paths), n, fo.ctypes.data, nb.ctypes.data, ctypes.c_void_p(int(base_ptr)), do.ctypes.data,
                                       status.ctypes.data, int(n_threads or default_threads()), -1 if mode is None else READ_MODES[mode])
    return status


def copy_into(arrays: list[np.ndarray], base_ptr: int, dst_off: np.ndarray, n_threads: int | None = None) -> None:
    """``memcpy`` every (contiguous) array's bytes to ``base_ptr + dst_off[i]`` on the pool."""
    n = len(arrays)
    if not n:
        return
    srcs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrays])
    nb = np.array([a.nbytes for a in arrays], np.int64)
    do = np.ascontiguousarray(dst_off, np.int64)
    _load().bn_copy_many(srcs, n, nb.ctypes.data, ctypes.c_void_p(int(base_ptr)), do.ctypes.data, int(n_threads or default_threads()))
