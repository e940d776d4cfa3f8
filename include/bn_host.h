/* C ABI of lib/libbn_host.so — the HOST side of the evaluate pipeline (plain C, no GPU code, no torch types).
 *
 * The reference reaches its audio files one at a time through libsndfile from Python
 * (reference: birdnet_stm32/audio/io.py:89-117 `sf.info`, `sf.SoundFile.read(dtype='float32', always_2d=True)`; called once per
 * file by birdnet_stm32/evaluation/metrics.py:117-125).  These entry points replace that host work for the device pipeline:
 * containers are parsed here, samples are handed to the GPU as they lie in the file (bn_ingest_resample of birdnet_hip.h decodes
 * them), and many files are read at once by a pool of threads into one caller-owned, page-locked buffer.
 *
 * Bound with ctypes in birdnet_stm32/audio/_pcmio.py and birdnet_stm32/audio/_flac.py; INTEGRATION.md shows the call sequence. */
#ifndef BN_HOST_H
#define BN_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RIFF/WAVE probing and window reads (csrc/host/bn_pcmio.c) -------------------------------------------------------------- */

#define BN_IO_OK 0
#define BN_IO_OPEN (-1)     /* open / fstat failed */
#define BN_IO_NOT_WAVE (-2) /* no RIFF....WAVE signature */
#define BN_IO_NO_CHUNK (-3) /* 'fmt ' or 'data' chunk missing */
#define BN_IO_SHORT (-4)    /* the file holds fewer bytes than asked for */

/* What `sf.info` tells the reference about a WAV file (io.py:90-95): sample format, channel count, native rate, and where the
 * samples lie.  frames = data_bytes / (bits / 8 * channels). */
typedef struct {
    int32_t status;      /* BN_IO_* */
    int32_t format_tag;  /* 1 = PCM, 3 = IEEE float (WAVE_FORMAT_EXTENSIBLE resolved through its sub-format GUID) */
    int32_t channels;
    int32_t sample_rate;
    int32_t bits;
    int32_t reserved;
    int64_t data_offset; /* byte offset of the first sample in the file */
    int64_t data_bytes;  /* min('data' chunk size, bytes left in the file) */
} bn_wav_layout;

/* Header walk of one file; returns out->status. */
int bn_wav_probe(const char* path, bn_wav_layout* out);

/* Header walks of n files on n_threads POSIX threads (1..64); returns how many have status BN_IO_OK. */
int bn_wav_probe_many(const char* const* paths, int n, bn_wav_layout* out, int n_threads);

/* The read windows of n files (reference io.py:112-117: seek + read of at most max_duration seconds) in one call:
 * nbytes[i] bytes from byte offset file_off[i] of paths[i] are pread() into base + dst_off[i].  The caller owns `base` (a
 * page-locked slab the GPU copies from), sizes it, and keeps the destination ranges disjoint.  status[i] = BN_IO_*.
 * Returns the number of items that failed. */
int bn_file_read_many(const char* const* paths, int n, const int64_t* file_off, const int64_t* nbytes, void* base,
                      const int64_t* dst_off, int32_t* status, int n_threads);

/* How bn_file_read_many takes a window out of the page cache: 0 = pread(), 1 = mmap + MADV_SEQUENTIAL + memcpy (default; the first read of
 * freshly written files runs 3-4 x faster: no LRU activation under the shared lru lock, csrc/host/bn_pcmio.c).  Process-wide; seeded from
 * BN_READ_MODE (pread | mmap).  Returns the previous mode; any other argument only queries.  (The reference has one path: libsndfile's read.) */
int bn_host_set_read_mode(int mode);
/* bn_file_read_many with the mode of THIS call (0 / 1; anything else = the process default).  The evaluate pipeline reads with pread while
 * its page-locked slabs are still being registered (that holds the process's mmap lock, which a mapping needs and pread does not). */
int bn_file_read_many_mode(const char* const* paths, int n, const int64_t* file_off, const int64_t* nbytes, void* base,
                           const int64_t* dst_off, int32_t* status, int n_threads, int mode);

/* Self-test of the mapped reader's SIGBUS guard (a file truncated while it is being copied out of a mapping must end as a short read, not as a
 * signal): maps `path` (>= 64 KB), truncates it to nothing, copies under the guard.  -1 = the fault was caught (expected), 1 = the copy went
 * through, 0 = the set-up failed.  The file is left empty. */
int bn_host_selftest_truncated_map(const char* path);

/* n memcpy()s into the same kind of slab on the same pool (windows that had to be decoded on the host first: FLAC). */
int bn_copy_many(const void* const* src, int n, const int64_t* nbytes, void* base, const int64_t* dst_off, int n_threads);

/* ---- FLAC stream decoder (csrc/host/bn_flac.c; reference: libsndfile behind soundfile.read, io.py:90,114-116) ------------- */

#define BN_FLAC_ERR_FORMAT (-1)
#define BN_FLAC_ERR_CRC (-2)
#define BN_FLAC_ERR_UNSUPPORTED (-3)
#define BN_FLAC_ERR_NOMEM (-4)

/* STREAMINFO of a FLAC byte string: rate, channels, bits per sample, inter-channel frames (0 = not recorded). */
int bn_flac_info(const uint8_t* data, size_t n, int* sample_rate, int* channels, int* bps, int64_t* total_frames);

/* The 16-byte MD5 of the decoded audio as STREAMINFO records it (all zero = not recorded). */
int bn_flac_md5(const uint8_t* data, size_t n, uint8_t* md5_out);

/* Frames [first, first + max_frames) as interleaved int32 (the stream's integers, not scaled); out == NULL only counts.
 * Returns the number of frames written / counted, or a negative BN_FLAC_ERR_*. */
int64_t bn_flac_decode(const uint8_t* data, size_t n, int64_t first, int64_t max_frames, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* BN_HOST_H */
