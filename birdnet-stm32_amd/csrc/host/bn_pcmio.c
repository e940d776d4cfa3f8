/* Host side of the evaluate pipeline: RIFF/WAVE header probing and window reads for MANY files at once, on a pool of
 * POSIX threads, straight into a caller-owned (pinned) buffer.  Plain C, no dependencies; part of lib/libbn_host.so
 * (declared in include/bn_host.h, bound with ctypes in birdnet_stm32/audio/_pcmio.py).
 *
 * What it replaces: the reference reads one file after another through libsndfile inside its per-file loop
 * (reference: birdnet_stm32/audio/io.py:89-117 `sf.info` + `sf.SoundFile.read`, called from evaluation/metrics.py:117-125).
 * Here only the container is parsed on the host; the samples travel as they lie in the file and are decoded on the GPU
 * (bn_ingest_resample), so the host's work per file is one header walk and one pread() of the window.
 *
 * The chunk walk is the one of birdnet_stm32/audio/io.py:_wav_layout (same rules: 'fmt ' may come anywhere before 'data',
 * WAVE_FORMAT_EXTENSIBLE takes its code from the sub-format GUID, odd chunk sizes are padded, a 'data' size running past the end
 * of the file is cut to the file); tests/test_host_logic.py holds the two against each other on fuzzed files. */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <setjmp.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define BN_HOST_API __attribute__((visibility("default")))

/* status codes of bn_wav_layout.status / bn_file_read_many's per-item status */
#define BN_IO_OK 0
#define BN_IO_OPEN -1     /* open / fstat failed */
#define BN_IO_NOT_WAVE -2 /* no RIFF....WAVE signature */
#define BN_IO_NO_CHUNK -3 /* 'fmt ' or 'data' missing */
#define BN_IO_SHORT -4    /* fewer bytes than asked for */

typedef struct {
    int32_t status;
    int32_t format_tag; /* 1 PCM, 3 IEEE float (after WAVE_FORMAT_EXTENSIBLE resolution) */
    int32_t channels;
    int32_t sample_rate;
    int32_t bits;
    int32_t reserved;
    int64_t data_offset; /* byte offset of the first sample */
    int64_t data_bytes;  /* min('data' size, bytes left in the file) */
} bn_wav_layout;

static uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t rd16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

static ssize_t pread_full(int fd, void* dst, size_t n, off_t off) {
    size_t got = 0;
    while (got < n) {
        ssize_t r = pread(fd, (char*)dst + got, n - got, off + (off_t)got);
        if (r < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (r == 0) break;
        got += (size_t)r;
    }
    return (ssize_t)got;
}

static int probe_fd(int fd, bn_wav_layout* out) {
    struct stat st;
    unsigned char h[40];
    memset(out, 0, sizeof(*out));
    if (fstat(fd, &st) != 0) return out->status = BN_IO_OPEN;
    int64_t size = (int64_t)st.st_size;
    if (size < 12 || pread_full(fd, h, 12, 0) != 12 || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) return out->status = BN_IO_NOT_WAVE;
    int64_t pos = 12;
    int have_fmt = 0;
    while (pos + 8 <= size) {
        if (pread_full(fd, h, 8, (off_t)pos) != 8) break;
        int64_t csize = (int64_t)rd32(h + 4), body = pos + 8;
        if (memcmp(h, "fmt ", 4) == 0) {
            unsigned char f[26];
            size_t want = csize >= 26 ? 26 : 16;
            if (pread_full(fd, f, want, (off_t)body) != (ssize_t)want) return out->status = BN_IO_NO_CHUNK; /* the Python walk raises struct.error here */
            uint32_t code = rd16(f);
            if (code == 0xFFFE && csize >= 26) code = rd16(f + 24);
            out->format_tag = (int32_t)code;
            out->channels = (int32_t)rd16(f + 2);
            out->sample_rate = (int32_t)rd32(f + 4);
            out->bits = (int32_t)rd16(f + 14);
            have_fmt = 1;
        } else if (memcmp(h, "data", 4) == 0) {
            if (!have_fmt) return out->status = BN_IO_NO_CHUNK;
            out->data_offset = body;
            out->data_bytes = csize < size - body ? csize : size - body;
            return out->status = BN_IO_OK;
        }
        pos = body + csize + (csize & 1);
    }
    return out->status = BN_IO_NO_CHUNK;
}

BN_HOST_API int bn_wav_probe(const char* path, bn_wav_layout* out) {
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        memset(out, 0, sizeof(*out));
        return out->status = BN_IO_OPEN;
    }
    int rc = probe_fd(fd, out);
    close(fd);
    return rc;
}

/* ---- a tiny work-sharing pool: n items, an atomic cursor, up to 64 threads ------------------------------------ */
typedef struct {
    void (*fn)(void* arg, int item);
    void* arg;
    int n;
    int next; /* __atomic cursor */
} pool_job;

static void* pool_worker(void* p) {
    pool_job* job = (pool_job*)p;
    for (;;) {
        int i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (i >= job->n) break;
        job->fn(job->arg, i);
    }
    return NULL;
}

static void pool_run(void (*fn)(void*, int), void* arg, int n, int n_threads) {
    pool_job job = {fn, arg, n, 0};
    pthread_t th[64];
    if (n_threads > 64) n_threads = 64;
    if (n_threads > n) n_threads = n;
    int started = 0;
    for (int t = 1; t < n_threads; t++) /* the caller is worker 0 */
        if (pthread_create(&th[started], NULL, pool_worker, &job) == 0) started++;
    pool_worker(&job);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

typedef struct {
    const char* const* paths;
    bn_wav_layout* out;
} probe_args;

static void probe_item(void* a, int i) {
    probe_args* p = (probe_args*)a;
    bn_wav_probe(p->paths[i], &p->out[i]);
}

/* Headers of n files; out[i].status says which ones are RIFF/WAVE.  Returns the number of files with status OK. */
BN_HOST_API int bn_wav_probe_many(const char* const* paths, int n, bn_wav_layout* out, int n_threads) {
    if (n <= 0) return 0;
    probe_args a = {paths, out};
    pool_run(probe_item, &a, n, n_threads < 1 ? 1 : n_threads);
    int ok = 0;
    for (int i = 0; i < n; i++) ok += out[i].status == BN_IO_OK;
    return ok;
}

typedef struct {
    const char* const* paths;
    const int64_t* file_off;
    const int64_t* nbytes;
    unsigned char* base;
    const int64_t* dst_off;
    int32_t* status;
    int mode; /* BN_READ_* of this call */
} read_args;

/* How a window leaves the page cache (bn_host_set_read_mode):
 *   BN_READ_PREAD  pread() into the slab;
 *   BN_READ_MMAP   (default) mmap(MAP_SHARED) + madvise(MADV_SEQUENTIAL) + memcpy + munmap.
 * Why the second is the default: pread (and an ordinary mapping, when it is torn down) calls folio_mark_accessed on every page, and the SECOND
 * touch of a page — the first read of a freshly written file, a data set just unpacked or generated — moves it to the active LRU list under the
 * one lru lock all reader threads share: 17-20 GB/s on 16 threads against 75+ GB/s once the pages are active (tools/probe/cold_read_probe.c,
 * profiles/r05_evaluate_cold_probe.md).  A VM_SEQ_READ mapping is exempt from that bookkeeping (mm: vma_has_recency): 65-70 GB/s on the first read
 * and on every later one, and a one-pass read of a data set no longer pushes it onto the active list.  Windows below 64 KB, files that cannot be
 * mapped and windows running past the end of the file take pread (same status codes).  A file truncated by someone else WHILE it is being copied
 * raises SIGBUS under mmap where pread would return short: caught (bus_guard below) and handed to pread. */
#define BN_READ_PREAD 0
#define BN_READ_MMAP 1
static int g_read_mode = -1; /* -1: not yet seeded from the environment */

static int read_mode(void) {
    int m = __atomic_load_n(&g_read_mode, __ATOMIC_RELAXED);
    if (m < 0) {
        const char* e = getenv("BN_READ_MODE");
        m = (e && (strcmp(e, "pread") == 0 || strcmp(e, "0") == 0)) ? BN_READ_PREAD : BN_READ_MMAP;
        __atomic_store_n(&g_read_mode, m, __ATOMIC_RELAXED);
    }
    return m;
}

/* Sets the process-wide read mode (BN_READ_*), returns the previous one; any other value only queries. */
BN_HOST_API int bn_host_set_read_mode(int mode) {
    int prev = read_mode();
    if (mode == BN_READ_PREAD || mode == BN_READ_MMAP) __atomic_store_n(&g_read_mode, mode, __ATOMIC_RELAXED);
    return prev;
}

/* A file that shrinks while it is mapped turns the copy's page fault into SIGBUS.  While bn_file_read_many runs in mmap mode a SIGBUS handler is
 * installed (reference-counted across concurrent calls, the previous disposition restored afterwards): a fault INSIDE one of our copies jumps back
 * into mmap_window, which reports the window as not copied — the caller's pread then returns short like it always did; any other SIGBUS goes to the
 * handler that was there before (or the default action). */
static __thread sigjmp_buf* tl_copy_jmp = NULL;
static struct sigaction g_old_bus;
static int g_bus_users = 0;
static pthread_mutex_t g_bus_mu = PTHREAD_MUTEX_INITIALIZER;

static void bus_handler(int sig, siginfo_t* si, void* uc) {
    if (tl_copy_jmp) siglongjmp(*tl_copy_jmp, 1);
    if (g_old_bus.sa_flags & SA_SIGINFO) {
        if (g_old_bus.sa_sigaction) g_old_bus.sa_sigaction(sig, si, uc);
    } else if (g_old_bus.sa_handler == SIG_DFL || g_old_bus.sa_handler == SIG_IGN) {
        signal(SIGBUS, SIG_DFL);
        raise(SIGBUS);
    } else {
        g_old_bus.sa_handler(sig);
    }
}

static void bus_guard(int on) {
    pthread_mutex_lock(&g_bus_mu);
    if (on) {
        if (g_bus_users++ == 0) {
            struct sigaction sa;
            memset(&sa, 0, sizeof sa);
            sa.sa_sigaction = bus_handler;
            sa.sa_flags = SA_SIGINFO | SA_NODEFER;
            sigemptyset(&sa.sa_mask);
            sigaction(SIGBUS, &sa, &g_old_bus);
        }
    } else if (--g_bus_users == 0) {
        sigaction(SIGBUS, &g_old_bus, NULL);
    }
    pthread_mutex_unlock(&g_bus_mu);
}

/* memcpy out of a mapping under the guard: 1 = copied, -1 = the mapping faulted (file truncated under us) */
static int guarded_copy(void* dst, const void* src, size_t n) {
    sigjmp_buf jb;
    int ok = -1;
    tl_copy_jmp = &jb;
    if (sigsetjmp(jb, 1) == 0) {
        memcpy(dst, src, n);
        ok = 1;
    }
    tl_copy_jmp = NULL;
    return ok;
}

/* 1 = copied, 0 = not applicable here or the mapping faulted (the caller takes pread, which reports a short file as it always did) */
static int mmap_window(int fd, void* dst, size_t n, off_t off) {
    static long page = 0;
    if (!page) page = sysconf(_SC_PAGESIZE);
    struct stat st;
    if (n < 65536 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (int64_t)off + (int64_t)n > (int64_t)st.st_size) return 0;
    const off_t lead = off & (off_t)(page - 1);
    void* m = mmap(NULL, n + (size_t)lead, PROT_READ, MAP_SHARED, fd, off - lead);
    if (m == MAP_FAILED) return 0;
    (void)madvise(m, n + (size_t)lead, MADV_SEQUENTIAL);
    const int ok = guarded_copy(dst, (const char*)m + lead, n);
    munmap(m, n + (size_t)lead);
    return ok == 1;
}

/* Self-test of the guard (tests/test_pipeline_host.py): maps `path` (at least 64 KB), truncates it to nothing through a second descriptor and
 * copies out of the mapping under the guard.  Returns -1 when the fault was caught (expected), 1 if the copy went through, 0 if the set-up failed. */
BN_HOST_API int bn_host_selftest_truncated_map(const char* path) {
    int fd = open(path, O_RDWR | O_CLOEXEC);
    if (fd < 0) return 0;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 65536) {
        close(fd);
        return 0;
    }
    const size_t n = (size_t)st.st_size;
    void* m = mmap(NULL, n, PROT_READ, MAP_SHARED, fd, 0);
    char* dst = (char*)malloc(n);
    int rc = 0;
    if (m != MAP_FAILED && dst && ftruncate(fd, 0) == 0) {
        bus_guard(1);
        rc = guarded_copy(dst, m, n);
        bus_guard(0);
    }
    if (m != MAP_FAILED) munmap(m, n);
    free(dst);
    close(fd);
    return rc;
}

static void read_item(void* a, int i) {
    read_args* r = (read_args*)a;
    if (r->nbytes[i] <= 0) {
        r->status[i] = BN_IO_OK;
        return;
    }
    int fd = open(r->paths[i], O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        r->status[i] = BN_IO_OPEN;
        return;
    }
    if (r->mode == BN_READ_MMAP && mmap_window(fd, r->base + r->dst_off[i], (size_t)r->nbytes[i], (off_t)r->file_off[i])) {
        close(fd);
        r->status[i] = BN_IO_OK;
        return;
    }
    ssize_t got = pread_full(fd, r->base + r->dst_off[i], (size_t)r->nbytes[i], (off_t)r->file_off[i]);
    close(fd);
    r->status[i] = got == (ssize_t)r->nbytes[i] ? BN_IO_OK : (got < 0 ? BN_IO_OPEN : BN_IO_SHORT);
}

/* nbytes[i] bytes from offset file_off[i] of paths[i] into base + dst_off[i], for n files on n_threads threads.
 * The destination ranges must not overlap and must lie inside the caller's buffer (the caller sized it).
 * Returns the number of items whose status is not OK. */
BN_HOST_API int bn_file_read_many_mode(const char* const* paths, int n, const int64_t* file_off, const int64_t* nbytes, void* base,
                                       const int64_t* dst_off, int32_t* status, int n_threads, int mode) {
    if (n <= 0) return 0;
    read_args a = {paths, file_off, nbytes, (unsigned char*)base, dst_off, status, (mode == BN_READ_PREAD || mode == BN_READ_MMAP) ? mode : read_mode()};
    if (a.mode == BN_READ_MMAP) bus_guard(1);
    pool_run(read_item, &a, n, n_threads < 1 ? 1 : n_threads);
    if (a.mode == BN_READ_MMAP) bus_guard(0);
    int bad = 0;
    for (int i = 0; i < n; i++) bad += status[i] != BN_IO_OK;
    return bad;
}

BN_HOST_API int bn_file_read_many(const char* const* paths, int n, const int64_t* file_off, const int64_t* nbytes, void* base,
                                  const int64_t* dst_off, int32_t* status, int n_threads) {
    return bn_file_read_many_mode(paths, n, file_off, nbytes, base, dst_off, status, n_threads, -1);
}

/* memcpy on the pool (pinned <- pageable staging of decoded FLAC windows and the like): n ranges. */
typedef struct {
    const void* const* src;
    const int64_t* nbytes;
    unsigned char* base;
    const int64_t* dst_off;
} copy_args;

static void copy_item(void* a, int i) {
    copy_args* c = (copy_args*)a;
    if (c->nbytes[i] > 0) memcpy(c->base + c->dst_off[i], c->src[i], (size_t)c->nbytes[i]);
}

BN_HOST_API int bn_copy_many(const void* const* src, int n, const int64_t* nbytes, void* base, const int64_t* dst_off, int n_threads) {
    if (n <= 0) return 0;
    copy_args a = {src, nbytes, (unsigned char*)base, dst_off};
    pool_run(copy_item, &a, n, n_threads < 1 ? 1 : n_threads);
    return 0;
}
