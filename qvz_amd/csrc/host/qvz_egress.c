/* qvz_egress.c - how the products of a whole-file encode or decode leave the device: slab by slab through two pinned
 * staging buffers, the next slab's kernel and PCIe transfer running while the host writes the previous one into the file
 * with a few threads. The reference writes its -u text with one fputs per line (qv_compressor.c:100-125) and sums the
 * per-read errors as it goes (:127); here both are N-sized device arrays (15.1 GB of text and 0.8 GB of doubles at 100M x 150
 * reads), and no N-sized host buffer, pageable copy or second N-sized device buffer is made for either.
 * Every function may be called from a host thread of its own (it makes its own streams); the caller has selected the device. */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "qvz_job.h"

#define HIPE(call)                                                                                  \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_) { rc = qvz_job_fail(rc_, "%s: %s", #call, qvz_hip_last_error()); goto done; }      \
    } while (0)

/* page cache -> pinned staging with a few threads: one core copies ~4 GB/s, PCIe takes several times that */
typedef struct { uint8_t *dst; const uint8_t *src; size_t n; } copy_part_t;
static void *copy_part(void *arg) {
    const copy_part_t *p = (const copy_part_t *)arg;
    memcpy(p->dst, p->src, p->n);
    return NULL;
}
void qvz_parallel_memcpy(uint8_t *dst, const uint8_t *src, size_t n) {
    enum { TMAX = 16 };
    pthread_t th[TMAX];
    copy_part_t part[TMAX];
    int started[TMAX] = {0};
    if (n < (8u << 20)) { memcpy(dst, src, n); return; }
    /* page cache -> pinned memory at one core's speed (~2 GB/s) would hold the PCIe link back: up to 16 threads, as
     * many as the machine has, QVZ_COPY_THREADS to override */
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    int T = getenv("QVZ_COPY_THREADS") ? atoi(getenv("QVZ_COPY_THREADS")) : 8;   /* 100M x 150 in: 0.61 s with 4, 0.53 with 8, 0.51 with 16 */
    if (cores > 0 && T > cores) T = (int)cores;
    if (T < 1) T = 1;
    if (T > TMAX) T = TMAX;
    const size_t each = (n / (size_t)T + 4095) & ~(size_t)4095;
    for (int i = 0; i < T; ++i) {
        const size_t a = (size_t)i * each, b = a + each < n ? a + each : n;
        part[i].dst = dst + a; part[i].src = src + a; part[i].n = a < n ? b - a : 0;
        if (i + 1 < T && part[i].n) started[i] = pthread_create(&th[i], NULL, copy_part, &part[i]) == 0;
        if (!started[i] && part[i].n) memcpy(part[i].dst, part[i].src, part[i].n);   /* the last part, or no thread */
    }
    for (int i = 0; i < T; ++i) if (started[i]) pthread_join(th[i], NULL);
}

/* ---- a buffer into a file at a given offset, with a few threads (one core writes ~2-3 GB/s into the page cache) ---- */
typedef struct { int fd; const uint8_t *buf; size_t n; off_t off; int failed; } wpart_t;

static void *write_part(void *arg) {
    wpart_t *p = (wpart_t *)arg;
    int zero_writes = 0;
    while (p->n) {
        const ssize_t w = pwrite(p->fd, p->buf, p->n, p->off);
        if (w < 0 && errno == EINTR) continue;                                  /* a signal (a profiler's timer) is not a failed write */
        if (w < 0 || (w == 0 && ++zero_writes > 1)) { p->failed = 1; return NULL; }
        p->buf += w; p->off += w; p->n -= (size_t)w;
    }
    return NULL;
}

int qvz_parallel_pwrite(int fd, const uint8_t *buf, size_t n, int64_t off) {
    enum { TMAX = 16 };
    pthread_t th[TMAX];
    wpart_t part[TMAX];
    int started[TMAX] = {0}, failed = 0;
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    int T = getenv("QVZ_COPY_THREADS") ? atoi(getenv("QVZ_COPY_THREADS")) : 8;   /* the decoder's 15 GB of text: 1.87 s with 4, 1.30 with 8, 1.40 with 12 */
    if (cores > 0 && T > cores) T = (int)cores;
    if (T < 1 || n < (8u << 20)) T = 1;
    if (T > TMAX) T = TMAX;
    const size_t each = (n / (size_t)T + 4095) & ~(size_t)4095;
    for (int i = 0; i < T; ++i) {
        const size_t a = (size_t)i * each, b = a + each < n ? a + each : n;
        part[i].fd = fd; part[i].buf = buf + a; part[i].n = a < n ? b - a : 0; part[i].off = (off_t)off + (off_t)a; part[i].failed = 0;
        if (i + 1 < T && part[i].n) started[i] = pthread_create(&th[i], NULL, write_part, &part[i]) == 0;
        if (!started[i] && part[i].n) write_part(&part[i]);                    /* the last part, or no thread */
    }
    for (int i = 0; i < T; ++i) { if (started[i]) pthread_join(th[i], NULL); failed |= part[i].failed; }
    return failed ? qvz_job_fail(QVZ_E_ARG, "write failed") : QVZ_OK;
}

/* ---- pages ahead of the writer -------------------------------------------------------------------------------------
 * A big output whose size is known early (the -u text: N (C + 1) bytes once the first line has been seen; the text of a
 * decode: from the header) is mostly the cost of MAKING its pages in the page cache: measured on the MI355X host's tmpfs,
 * pwrite into new pages 4-5 GB/s whatever the thread count (writers of one file take turns on its inode lock), page
 * faults on a fresh mapping 4 GB/s, fallocate 7 GB/s, and any two of these at once less than either alone (2.6-3.8 GB/s) -
 * while copying into pages that EXIST through a mapping runs at 14-17 GB/s (tools/io_bench.c, profiles/r03_host_io.txt).
 * So one thread makes the pages with fallocate from the moment the size is known - beside the upload and the kernels,
 * when nothing else wants the file - and the writer (put, below) copies through a mapping and never gets ahead of it.
 * Anything that does not work out (no O_RDWR twin of the descriptor, fallocate or mmap refused) falls back to pwrite. */
typedef struct {
    int fd, fd2;                         /* the FILE's descriptor (the key) and an O_RDWR twin of it */
    int64_t base;                        /* file offset the output starts at */
    uint64_t bytes;
    uint8_t *map;                        /* mapping of [base_page, base + bytes) */
    size_t map_len;
    int64_t base_page;
    int64_t sized_at;                    /* the length the file was given for the mapping */
    volatile uint64_t done;              /* bytes from base whose pages exist */
    volatile int stop, failed;
    pthread_t th;
} prealloc_t;
static prealloc_t *g_pre[8];
static pthread_mutex_t g_pre_mu = PTHREAD_MUTEX_INITIALIZER;

static pthread_mutex_t g_alloc_mu = PTHREAD_MUTEX_INITIALIZER;   /* one file's pages at a time (two allocators are slower than one): first come, first served */

static void *prealloc_main(void *arg) {
    prealloc_t *p = (prealloc_t *)arg;
    const uint64_t slab = 32ull << 20;
    while (pthread_mutex_trylock(&g_alloc_mu)) {
        if (p->stop) return NULL;
        usleep(500);
    }
    for (uint64_t at = 0; at < p->bytes && !p->stop; at += slab) {
        const uint64_t m = p->bytes - at < slab ? p->bytes - at : slab;
        if (fallocate(p->fd2, 0, (off_t)(p->base + (int64_t)at), (off_t)m)) { p->failed = 1; break; }
        __atomic_store_n(&p->done, at + m, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&g_alloc_mu);
    return NULL;
}

int qvz_prealloc_start(FILE *f, uint64_t bytes) {
    /* QVZ_PREALLOC: "0" turns the page helper off; test hooks: "min=<bytes>" smallest output it is started for (default 256 MiB),
     * "fail" the failure path behind the ftruncate (comma-separated) */
    const char *pe = getenv("QVZ_PREALLOC"), *pm = pe ? strstr(pe, "min=") : NULL;
    const uint64_t least = pm ? strtoull(pm + 4, NULL, 10) : 256ull << 20;
    if (!f || bytes < least || (pe && strcmp(pe, "0") == 0)) return 0;
    fflush(f);
    const int64_t base = (int64_t)ftello(f);
    const int fd = fileno(f);
    struct stat st;
    if (base < 0 || fd < 0 || fstat(fd, &st) || !S_ISREG(st.st_mode)) return 0;
    char link[64];
    snprintf(link, sizeof(link), "/proc/self/fd/%d", fd);
    const int fd2 = open(link, O_RDWR);                                        /* fopen(.., "w") is write-only: no shared mapping of it */
    if (fd2 < 0) return 0;
    prealloc_t *p = (prealloc_t *)calloc(1, sizeof(*p));
    const long page = sysconf(_SC_PAGESIZE);
    if (!p || page <= 0) { free(p); close(fd2); return 0; }
    /* the file is made long enough for the mapping; whatever fails from here on gives it its old length back (`bytes` may be an
     * estimate above the real size - the shard set's streams -, and nobody would shrink the file again: qvz_prealloc_finish
     * finds no entry for it) */
    const int grown = (uint64_t)st.st_size < (uint64_t)base + bytes;
    if (grown && ftruncate(fd2, (off_t)((uint64_t)base + bytes))) { free(p); close(fd2); return 0; }
    p->fd = fd; p->fd2 = fd2; p->base = base; p->bytes = bytes;
    p->sized_at = grown ? (int64_t)((uint64_t)base + bytes) : -1;
    p->base_page = base - base % page;
    p->map_len = (size_t)((uint64_t)(base - p->base_page) + bytes);
    p->map = (uint8_t *)mmap(NULL, p->map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd2, (off_t)p->base_page);
    int slot = -1;
    if (p->map != MAP_FAILED) {
        pthread_mutex_lock(&g_pre_mu);
        for (int i = 0; i < 8 && slot < 0; ++i) if (!g_pre[i]) slot = i;
        if (pe && strstr(pe, "fail")) slot = -1;                               /* tests: the failure path behind the ftruncate */
        if (slot >= 0 && pthread_create(&p->th, NULL, prealloc_main, p) == 0) g_pre[slot] = p; else slot = -1;
        pthread_mutex_unlock(&g_pre_mu);
    }
    if (slot < 0) {
        if (p->map != MAP_FAILED) munmap(p->map, p->map_len);
        if (grown && ftruncate(fd2, st.st_size)) { /* best effort */ }
        close(fd2); free(p);
        return 0;
    }
    return 1;
}

static prealloc_t *prealloc_of(int fd) {
    prealloc_t *p = NULL;
    pthread_mutex_lock(&g_pre_mu);
    for (int i = 0; i < 8; ++i) if (g_pre[i] && g_pre[i]->fd == fd) p = g_pre[i];
    pthread_mutex_unlock(&g_pre_mu);
    return p;
}

void qvz_prealloc_finish(FILE *f) {
    if (!f) return;
    const int fd = fileno(f);
    prealloc_t *p = NULL;
    pthread_mutex_lock(&g_pre_mu);
    for (int i = 0; i < 8; ++i) if (g_pre[i] && g_pre[i]->fd == fd) { p = g_pre[i]; g_pre[i] = NULL; }
    pthread_mutex_unlock(&g_pre_mu);
    if (!p) return;
    p->stop = 1;
    pthread_join(p->th, NULL);
    munmap(p->map, p->map_len);
    /* the file was sized for the mapping (bytes may have been an estimate: the shard set's streams); it ends where the FILE's
     * position is - everything written so far - and pages made beyond that go back */
    fflush(f);
    const int64_t end = (int64_t)ftello(f);
    struct stat st;
    if (end >= 0 && fstat(p->fd2, &st) == 0 && (int64_t)st.st_size > end && p->sized_at == (int64_t)st.st_size) {
        if (ftruncate(p->fd2, (off_t)end)) { /* best effort: the caller's own writes define the file */ }
    }
    close(p->fd2);
    free(p);
}

/* n bytes at file offset off through the mapping once their pages exist; 0: not this file's business (use pwrite) */
static int prealloc_put(int fd, const uint8_t *buf, size_t n, int64_t off) {
    prealloc_t *p = prealloc_of(fd);
    if (!p || off < p->base || (uint64_t)(off - p->base) + n > p->bytes) return 0;
    const uint64_t need = (uint64_t)(off - p->base) + n;
    while (__atomic_load_n(&p->done, __ATOMIC_ACQUIRE) < need) {
        if (p->failed) return 0;
        usleep(100);
    }
    qvz_parallel_memcpy(p->map + (off - p->base_page), buf, n);
    return 1;
}

/* where a FILE's next byte goes, as (descriptor, offset), with its buffer flushed; *off < 0: not seekable, use fwrite */
static int file_cursor(FILE *f, int64_t *off) {
    fflush(f);
    *off = (int64_t)ftello(f);
    return fileno(f);
}

static int put(FILE *f, int fd, int64_t *off, const uint8_t *buf, size_t n) {
    if (*off < 0) return fwrite(buf, 1, n, f) == n ? QVZ_OK : qvz_job_fail(QVZ_E_ARG, "write failed");
    const int rc = prealloc_put(fd, buf, n, *off) ? QVZ_OK : qvz_parallel_pwrite(fd, buf, n, *off);
    *off += (int64_t)n;
    return rc;
}

/* A host buffer into a file at its position the way the slab pipelines below write theirs: with `estimate` > 0 the pages are
 * made ahead (qvz_prealloc_start) and the slabs go through the mapping - beyond the estimate through pwrite -, and the file ends
 * where the last byte was written. The pipelines' host half on its own: what the CPU tests (and the sanitizer build) exercise. */
int qvz_host_write_slabs(FILE *f, const uint8_t *buf, uint64_t bytes, uint64_t estimate, uint64_t slab) {
    if (!f || (!buf && bytes) || !slab) return qvz_job_fail(QVZ_E_ARG, "bad write arguments");
    int rc = QVZ_OK;
    if (estimate) qvz_prealloc_start(f, estimate);
    int64_t off;
    const int fd = file_cursor(f, &off);
    for (uint64_t at = 0; at < bytes && !rc; at += slab) rc = put(f, fd, &off, buf + at, (size_t)(bytes - at < slab ? bytes - at : slab));
    if (off >= 0) fseeko(f, (off_t)off, SEEK_SET);
    qvz_prealloc_finish(f);
    return rc;
}

/* ---- the pipeline: slab i+1 is produced (kernel + D2H on its own stream) while slab i is written ---- */
typedef struct {
    void *st[2];
    uint8_t *h[2], *d[2];
} stage_t;

static void stage_free(stage_t *s) {
    for (int b = 0; b < 2; ++b) {
        qvz_hip_free(s->d[b]); qvz_hip_free_host(s->h[b]);
        if (s->st[b]) qvz_hip_stream_destroy(s->st[b]);
    }
    memset(s, 0, sizeof(*s));
}

static int stage_make(stage_t *s, size_t slab, int want_device) {
    int rc = QVZ_OK;
    memset(s, 0, sizeof(*s));
    for (int b = 0; b < 2; ++b) {
        HIPE(qvz_hip_stream_create(&s->st[b]));
        HIPE(qvz_hip_malloc_host((void **)&s->h[b], slab));
        if (want_device) HIPE(qvz_hip_malloc((void **)&s->d[b], slab));
    }
done:
    if (rc) stage_free(s);
    return rc;
}

int qvz_egress_text(FILE *fu, const uint8_t *d_qv, uint64_t N, uint32_t C, uint64_t ld, void *after) {
    int rc = QVZ_OK;
    const uint64_t stride = (uint64_t)C + 1;
    uint64_t slab_reads = (48ull << 20) / stride & ~255ull;                     /* ~48 MB of text per slab */
    if (slab_reads < 256) slab_reads = 256;
    if (slab_reads > ((N + 255) & ~255ull)) slab_reads = (N + 255) & ~255ull;
    const uint64_t slabs = (N + slab_reads - 1) / slab_reads;
    stage_t s;
    int64_t off;
    const int fd = file_cursor(fu, &off);
    if ((rc = stage_make(&s, (size_t)(slab_reads * stride), 1))) return rc;
    if (after) { HIPE(qvz_hip_stream_wait(s.st[0], after)); HIPE(qvz_hip_stream_wait(s.st[1], after)); }
#define ISSUE(i, b)                                                                                                    \
    do {                                                                                                               \
        const uint64_t r0_ = (i) * slab_reads, n_ = N - r0_ < slab_reads ? N - r0_ : slab_reads;                       \
        HIPE(qvz_hip_untranspose_rows(d_qv, r0_, n_, C, ld, s.d[b], s.st[b]));                                         \
        HIPE(qvz_hip_copy_d2h(s.h[b], s.d[b], (size_t)(n_ * stride), s.st[b]));                                        \
    } while (0)
    ISSUE(0, 0);
    for (uint64_t i = 0, b = 0; i < slabs; ++i, b ^= 1) {
        if (i + 1 < slabs) ISSUE(i + 1, b ^ 1);                                /* its buffer was written out an iteration ago */
        HIPE(qvz_hip_stream_sync(s.st[b]));
        const uint64_t r0 = i * slab_reads, n = N - r0 < slab_reads ? N - r0 : slab_reads;
        if ((rc = put(fu, fd, &off, s.h[b], (size_t)(n * stride)))) goto done;
    }
#undef ISSUE
    if (off >= 0) fseeko(fu, (off_t)off, SEEK_SET);
done:
    stage_free(&s);
    return rc;
}

int qvz_egress_bytes(FILE *f, const uint8_t *d_src, uint64_t bytes, void *after) {
    int rc = QVZ_OK;
    const uint64_t slab = 64ull << 20;
    const uint64_t slabs = (bytes + slab - 1) / slab;
    stage_t s;
    int64_t off;
    if (!bytes) return QVZ_OK;
    const int fd = file_cursor(f, &off);
    if ((rc = stage_make(&s, (size_t)(bytes < slab ? bytes : slab), 0))) return rc;
    if (after) { HIPE(qvz_hip_stream_wait(s.st[0], after)); HIPE(qvz_hip_stream_wait(s.st[1], after)); }
#define ISSUE(i, b) HIPE(qvz_hip_copy_d2h(s.h[b], d_src + (i) * slab, (size_t)(bytes - (i) * slab < slab ? bytes - (i) * slab : slab), s.st[b]))
    ISSUE(0, 0);
    for (uint64_t i = 0, b = 0; i < slabs; ++i, b ^= 1) {
        if (i + 1 < slabs) ISSUE(i + 1, b ^ 1);
        HIPE(qvz_hip_stream_sync(s.st[b]));
        if ((rc = put(f, fd, &off, s.h[b], (size_t)(bytes - i * slab < slab ? bytes - i * slab : slab)))) goto done;
    }
#undef ISSUE
    if (off >= 0) fseeko(f, (off_t)off, SEEK_SET);
done:
    stage_free(&s);
    return rc;
}

/* sum over the reads, in file order, of err[i] / C (qv_compressor.c:127: `distortion += error / columns` per line): one
 * dependent chain of additions, so one host core - ~0.15 s for 100M reads, beside the coder kernels. keep (may be NULL)
 * receives the N values (a part of a multi-device encode hands them to the caller, who adds all parts in file order). */
int qvz_sum_errors(const double *d_err, uint64_t N, uint32_t C, double *sum_out, double *keep, void *after) {
    int rc = QVZ_OK;
    const uint64_t slab = 1ull << 22;                                           /* doubles per slab: 32 MB */
    const uint64_t slabs = (N + slab - 1) / slab;
    const double cols = (double)C;
    double sum = 0.0;
    stage_t s;
    if ((rc = stage_make(&s, (size_t)((N < slab ? N : slab) * sizeof(double)), 0))) return rc;
    if (after) { HIPE(qvz_hip_stream_wait(s.st[0], after)); HIPE(qvz_hip_stream_wait(s.st[1], after)); }
#define ISSUE(i, b) HIPE(qvz_hip_copy_d2h(s.h[b], d_err + (i) * slab, (size_t)((N - (i) * slab < slab ? N - (i) * slab : slab) * sizeof(double)), s.st[b]))
    ISSUE(0, 0);
    for (uint64_t i = 0, b = 0; i < slabs; ++i, b ^= 1) {
        if (i + 1 < slabs) ISSUE(i + 1, b ^ 1);
        HIPE(qvz_hip_stream_sync(s.st[b]));
        const uint64_t n = N - i * slab < slab ? N - i * slab : slab;
        const double *e = (const double *)s.h[b];
        for (uint64_t j = 0; j < n; ++j) sum += e[j] / cols;
        if (keep) memcpy(keep + i * slab, e, (size_t)(n * sizeof(double)));
    }
#undef ISSUE
    *sum_out = sum;
done:
    stage_free(&s);
    return rc;
}

/* host bytes -> device, the way the input text goes in (qvz_ingest_rows): threaded copy into pinned slabs, PCIe behind it */
int qvz_ingress_bytes(uint8_t *d_dst, const uint8_t *h_src, uint64_t bytes) {
    int rc = QVZ_OK;
    const uint64_t slab = 64ull << 20;
    const uint64_t slabs = (bytes + slab - 1) / slab;
    stage_t s;
    if (!bytes) return QVZ_OK;
    if ((rc = stage_make(&s, (size_t)(bytes < slab ? bytes : slab), 0))) return rc;
    for (uint64_t i = 0, b = 0; i < slabs; ++i, b ^= 1) {
        const uint64_t n = bytes - i * slab < slab ? bytes - i * slab : slab;
        HIPE(qvz_hip_stream_sync(s.st[b]));                                    /* this slab's copy of two slabs ago */
        qvz_parallel_memcpy(s.h[b], h_src + i * slab, (size_t)n);
        HIPE(qvz_hip_copy_h2d(d_dst + i * slab, s.h[b], (size_t)n, s.st[b]));
    }
    HIPE(qvz_hip_stream_sync(s.st[0]));
    HIPE(qvz_hip_stream_sync(s.st[1]));
done:
    stage_free(&s);
    return rc;
}

int qvz_egress_rows(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, const char *path) {
    if (!d_sym || !n || !c || ld < n || !path) return qvz_job_fail(QVZ_E_ARG, "bad egress arguments");
    FILE *f = fopen(path, "w");
    if (!f) return qvz_job_fail(QVZ_E_ARG, "Unable to open output file");
    /* (no pages made ahead here: without a head start the helper and the writer work on the same pages at the same time,
     * which is slower than the writer alone - 4.6 s against 3.6 s for 15.1 GB) */
    const int rc = qvz_egress_text(f, d_sym, n, c, ld, NULL);
    if (fclose(f) && !rc) return qvz_job_fail(QVZ_E_ARG, "write failed");
    return rc;
}
