/* qvz_encode.c - host C driver of the encode path (see include/qvz_encode.h, qvz_job.h). Counterpart of encode()
 * in the reference's src/main.c:18-127; prints the same -v / -s lines. All heavy phases are HIP kernel calls
 * through the C-ABI; this file owns file I/O, the k-means loop condition, libc rand() semantics and the serial
 * coder. */
#define _GNU_SOURCE
#include "qvz_encode.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "qvz_codebook.h"
#include "qvz_coder.h"
#include "qvz_job.h"
#include "qvz_shardset.h"

#define A QVZ_A
#define MAX_COLUMNS 1022u            /* MAX_READS_PER_LINE, reference include/lines.h:13 */

static __thread char g_err[512] = "";
const char *qvz_encode_last_error(void) { return g_err; }

int qvz_job_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP(call)                                                                                   \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_) { rc = qvz_job_fail(rc_, "%s: %s", #call, qvz_hip_last_error()); goto done; }      \
    } while (0)

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);   /* util.c:11-78 uses CLOCK_REALTIME */
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void qvz_options_default(qvz_options_t *o) {
    memset(o, 0, sizeof(*o));
    o->ratio = 0.5;               /* main.c:200 */
    o->clusters = 1;              /* main.c:201 */
    o->distortion = QVZ_DIST_MSE; /* main.c:203 */
    o->cluster_threshold = 4;     /* main.c:204 */
    o->mode = QVZ_MODE_RATIO;     /* uninitialised in the reference (main.c:198-204); README documents -f 0.5 */
    o->seed_mode = QVZ_SEED_TIME;
}

/* a launch timed with events on its stream, for a job that reports kernel times (qvz_resident_*): TIMED(job, &job->k_x_ms, call) */
typedef struct { void *a, *b; } span_t;
static void span_begin(qvz_job_t *job, span_t *sp) {
    sp->a = sp->b = NULL;
    if (!job->keep_device) return;
    if (qvz_hip_event_create(&sp->a) || qvz_hip_event_create(&sp->b) || qvz_hip_event_record(sp->a, NULL)) { qvz_hip_event_destroy(sp->a); qvz_hip_event_destroy(sp->b); sp->a = sp->b = NULL; }
}
static void span_end(span_t *sp) { if (sp->a) (void)qvz_hip_event_record(sp->b, NULL); }
static void span_add(span_t *sp, double *ms) {                                  /* after the stream has been synchronised */
    double d = 0.0;
    if (sp->a && qvz_hip_event_elapsed_ms(sp->a, sp->b, &d) == QVZ_OK) *ms += d;
    qvz_hip_event_destroy(sp->a); qvz_hip_event_destroy(sp->b);
    sp->a = sp->b = NULL;
}

/* A big device buffer of a job: from the job's pool when it has one (qvz_resident_*: the block of that role is kept between
 * encodes, replaced when too small), else a plain allocation. job_free gives it back the same way. */
static int job_malloc(qvz_job_t *job, void **p, size_t bytes, int role) {
    if (!job->pool) return qvz_hip_malloc(p, bytes);
    qvz_pool_t *pl = job->pool;
    if (pl->ptr[role] && pl->bytes[role] < bytes) { qvz_hip_free(pl->ptr[role]); pl->ptr[role] = NULL; pl->bytes[role] = 0; }
    if (!pl->ptr[role]) {
        const int rc = qvz_hip_malloc(&pl->ptr[role], bytes);
        if (rc) { pl->ptr[role] = NULL; return rc; }
        pl->bytes[role] = bytes;
    }
    *p = pl->ptr[role];
    return QVZ_OK;
}

static void job_free(qvz_job_t *job, void *p) {
    if (!p) return;
    if (job->pool) for (int r = 1; r < QVZ_POOL_ROLES; ++r) if (job->pool->ptr[r] == p) return;   /* stays with the pool */
    qvz_hip_free(p);
}

/* ------------------------------------------------------------------------------------------------------- */
static int job_init(qvz_job_t *job, const uint8_t *text, uint64_t bytes, uint64_t lines, uint32_t columns,
                    const qvz_options_t *opts, int draw_centroids);

int qvz_job_init(qvz_job_t *job, const uint8_t *text, uint64_t bytes, uint64_t lines, uint32_t columns,
                 const qvz_options_t *opts) {
    return job_init(job, text, bytes, lines, columns, opts, 1);
}

int qvz_job_init_part(qvz_job_t *job, const uint8_t *text, uint64_t lines, uint32_t columns, const qvz_options_t *opts,
                      uint64_t first_read, const uint8_t *centroid_rows) {
    const int rc = job_init(job, text, 0, lines, columns, opts, 0);
    if (rc) return rc;
    if (first_read & 3) return qvz_job_fail(QVZ_E_ARG, "a part must start at a multiple of 4 reads (a WELL word boundary)");
    job->first_read = first_read;
    job->centroid_rows = centroid_rows;
    return 0;
}

static int job_init(qvz_job_t *job, const uint8_t *text, uint64_t bytes, uint64_t lines, uint32_t columns,
                    const qvz_options_t *opts, int draw_centroids) {
    memset(job, 0, sizeof(*job));
    job->opts = *opts;
    job->text = text;
    if (bytes) {   /* load_file (lines.c:44-54): columns from the first line, lines = size / (columns+1) */
        const uint8_t *nl = (const uint8_t *)memchr(text, '\n', bytes < MAX_COLUMNS + 2 ? bytes : MAX_COLUMNS + 2);
        if (!nl) return qvz_job_fail(QVZ_E_DATA, "first line longer than %u symbols or missing newline (LF_ERROR_TOO_LONG)", MAX_COLUMNS);
        columns = (uint32_t)(nl - text);
        lines = columns ? bytes / ((uint64_t)columns + 1) : 0;
    }
    if (columns == 0 || columns > MAX_COLUMNS) return qvz_job_fail(QVZ_E_DATA, "bad column count %u", columns);
    if (lines == 0) return qvz_job_fail(QVZ_E_DATA, "no complete line in the input");
    if (lines >= (1ull << 32)) return qvz_job_fail(QVZ_E_ARG, "more than 2^32-1 lines: the .q header and the u32 counters cannot hold it (codebook.c:482)");
    if (opts->clusters < 1 || opts->clusters > 255) return qvz_job_fail(QVZ_E_ARG, "clusters must be 1..255");
    job->N = lines;
    job->C = columns;
    job->K = (uint32_t)opts->clusters;
    job->ld = qvz_hip_ld(lines);
    if (opts->distortion == QVZ_DIST_CUSTOM) {                                /* main.c:30-32 */
        if (qvz_distortion_matrix_file(opts->dist_file, job->h_dist)) return qvz_job_fail(QVZ_E_ARG, "%s", qvz_hip_last_error());
    } else if (qvz_distortion_matrix(opts->distortion, job->h_dist)) {
        return qvz_job_fail(QVZ_E_ARG, "%s", qvz_hip_last_error());
    }

    if (!draw_centroids) return 0;      /* a part of a file (qvz_multi.c): the whole file's centroid reads were drawn by the caller */
    /* Centroid picks (cluster.c:198-204) use libc rand() in whatever state the process is in. They are drawn here,
     * BEFORE the first HIP call: initialising the HIP runtime consumes rand() values of its own, which would shift
     * the sequence away from what the reference binary sees in a fresh process. */
    uint32_t block_count = (uint32_t)(lines / QVZ_LINES_PER_BLOCK);            /* lines.c:93-96 */
    if ((uint64_t)block_count * QVZ_LINES_PER_BLOCK != lines) block_count += 1;
    for (uint32_t j = 0; j < job->K; ++j) {
        const uint32_t block_id = (uint32_t)rand() % block_count;
        const uint64_t left = lines - (uint64_t)block_id * QVZ_LINES_PER_BLOCK;
        const uint32_t count = left > QVZ_LINES_PER_BLOCK ? (uint32_t)QVZ_LINES_PER_BLOCK : (uint32_t)left;
        const uint32_t line_id = (uint32_t)rand() % count;
        job->centroid_lines[j] = (uint64_t)block_id * QVZ_LINES_PER_BLOCK + line_id;
        if (opts->verbose) printf("Chose block %d, line %d.\n", block_id, line_id);
    }
    return 0;
}

/* Streaming, transposing ingest (replaces load_file's row-pointer table, lines.c:27-82): the text goes through two
 * pinned staging buffers so that the host copy of chunk k+1 (page cache -> pinned), the PCIe transfer of chunk k and its
 * transposition into the column-major matrix overlap. */
static uint8_t *job_plane(qvz_job_t *job, int i);
static uint8_t *early_draws_plane(qvz_job_t *job);
static void *side_stream(void *const *others, int nothers);

int qvz_ingest_rows(const uint8_t *text, uint64_t N, uint32_t C, uint8_t *d_sym, uint64_t ld, uint32_t *bad_out) {
    int rc = 0;
    const uint64_t stride = (uint64_t)C + 1;
    uint8_t *d_up[2] = {NULL, NULL}, *h_up[2] = {NULL, NULL};
    void *st[2] = {NULL, NULL};
    uint32_t *d_bad = NULL;
    const uint64_t up_reads = 1u << 18;   /* reads per staging buffer; a multiple of 256 */
    const uint64_t up_n = N < up_reads ? N : up_reads;
    if (!text || !N || !C || !d_sym || ld < N) return qvz_job_fail(QVZ_E_ARG, "bad ingest arguments");
    HIP(qvz_hip_malloc((void **)&d_bad, sizeof(uint32_t)));
    HIP(qvz_hip_memset(d_bad, 0, sizeof(uint32_t), NULL));
    HIP(qvz_hip_stream_sync(NULL));
    for (int b = 0; b < 2; ++b) {
        HIP(qvz_hip_malloc((void **)&d_up[b], (size_t)up_n * stride));
        HIP(qvz_hip_malloc_host((void **)&h_up[b], (size_t)up_n * stride));
        HIP(qvz_hip_stream_create(&st[b]));
    }
    {
        int b = 0;
        for (uint64_t r0 = 0; r0 < N; r0 += up_reads, b ^= 1) {
            const uint64_t n = N - r0 < up_reads ? N - r0 : up_reads;
            HIP(qvz_hip_stream_sync(st[b]));                       /* this staging pair was used two chunks ago */
            qvz_parallel_memcpy(h_up[b], text + r0 * stride, n * stride);
            HIP(qvz_hip_copy_h2d(d_up[b], h_up[b], n * stride, st[b]));
            HIP(qvz_hip_transpose_text(d_up[b], n, C, d_sym + r0, ld, d_bad, st[b]));
        }
        HIP(qvz_hip_stream_sync(st[0]));
        HIP(qvz_hip_stream_sync(st[1]));
    }
    {
        uint32_t bad = 0;
        HIP(qvz_hip_copy_d2h(&bad, d_bad, sizeof(bad), NULL));
        HIP(qvz_hip_stream_sync(NULL));
        if (bad_out) *bad_out = bad;
    }
done:
    for (int b = 0; b < 2; ++b) {
        qvz_hip_free(d_up[b]);
        qvz_hip_free_host(h_up[b]);
        if (st[b]) qvz_hip_stream_destroy(st[b]);
    }
    qvz_hip_free(d_bad);
    return rc;
}

static void *coder_warmup_main(void *arg) {
    if (qvz_hip_set_device(*(const int *)arg) == QVZ_OK) (void)qvz_hip_encode_planes_warmup();
    return NULL;
}

int qvz_job_upload(qvz_job_t *job) {
    int rc = 0;
    const double t0 = now();
    const uint64_t ld = job->ld;
    uint32_t bad = 0;
    pthread_t th_warm;
    int warm_running = 0;
    if (job->uploaded) return 0;
    HIP(qvz_hip_set_device(job->opts.device));
    (void)qvz_hip_codebook_warmup();                               /* the one-time log2 check of the codebook generation, beside the upload */
    /* ... and the coder's stream pair, code object and reciprocal check (a sharded encode's first coder call) */
    if (getenv("QVZ_SHARDS") || getenv("QVZ_DEVICES")) warm_running = pthread_create(&th_warm, NULL, coder_warmup_main, &job->opts.device) == 0;
    HIP(qvz_hip_malloc((void **)&job->d_sym, (size_t)job->C * ld));
    HIP(qvz_hip_malloc((void **)&job->d_labels, ld));
    HIP(qvz_hip_memset(job->d_labels, 0, ld, NULL));               /* calloc'd line_t.cluster, lines.c:116 */
    if ((rc = qvz_ingest_rows(job->text, job->N, job->C, job->d_sym, ld, &bad))) goto done;
    if (bad) { rc = qvz_job_fail(QVZ_E_DATA, "%u input bytes outside '!'..'h' or lines without '\\n' (undefined behaviour in the reference, pmf.c:372-381)", bad); goto done; }
    job->uploaded = 1;
    job->t_upload = now() - t0;
done:
    if (warm_running) pthread_join(th_warm, NULL);
    return rc;
}

int qvz_job_kmeans(qvz_job_t *job, int want_host_labels) {
    int rc = 0;
    const double t0 = now();
    const uint32_t K = job->K, C = job->C;
    const uint64_t N = job->N, ld = job->ld, stride = (uint64_t)C + 1;
    double *h_moved = NULL;
    if ((rc = qvz_job_upload(job))) return rc;
    job->h_means = (uint8_t *)malloc((size_t)K * C);
    job->h_counts = (uint32_t *)calloc(K, sizeof(uint32_t));
    job->h_acc = (uint64_t *)calloc((size_t)K * C, sizeof(uint64_t));
    h_moved = (double *)malloc((K + 1) * sizeof(double));
    if (!job->h_means || !job->h_counts || !job->h_acc || !h_moved) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    for (uint32_t j = 0; j < K; ++j) {                                          /* memcpy of the read, cluster.c:201 */
        const uint8_t *row = job->centroid_rows ? job->centroid_rows + (size_t)j * stride : job->text + job->centroid_lines[j] * stride;
        for (uint32_t s = 0; s < C; ++s) job->h_means[(size_t)j * C + s] = (uint8_t)(row[s] - 33);
    }
    HIP(qvz_hip_malloc((void **)&job->d_means, (size_t)K * C));
    HIP(qvz_hip_malloc((void **)&job->d_counts, K * sizeof(uint32_t)));
    HIP(qvz_hip_malloc((void **)&job->d_acc, (size_t)K * C * sizeof(uint64_t)));
    HIP(qvz_hip_malloc((void **)&job->d_moved, (K + 1) * sizeof(double)));
    HIP(qvz_hip_malloc((void **)&job->d_kstat, sizeof(int32_t)));
    if (job->allreduce) HIP(qvz_hip_malloc((void **)&job->d_counts64, K * sizeof(uint64_t)));
    HIP(qvz_hip_copy_h2d(job->d_means, job->h_means, (size_t)K * C, NULL));
    HIP(qvz_hip_memset(job->d_kstat, 0, sizeof(int32_t), NULL));
    uint32_t iter = 0;
    int loop = 1;
    if (K == 1) {
        /* One cluster: every read is in it whatever the centroid is (cluster.c:75-98 finds the nearest of one), so there is
         * no assignment pass, and the column sums the reference's loop accumulates (cluster.c:100-131) are read off the PMF
         * counters - sum over reads of X_s = sum over x of x * #(X_s = x), exact integers - instead of costing passes over
         * the matrix of their own. The second iteration the reference runs when the first moved the centroid by more than
         * the threshold recomputes the same sums: it moves by 0. */
        const size_t words = (size_t)C * A * A;
        uint64_t *h_col = NULL;
        if (!job->counted && (rc = qvz_job_stats(job))) goto done;
        h_col = (uint64_t *)malloc(words * sizeof(uint64_t));
        if (!h_col) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        if (job->d_cnt64) {
            if (qvz_hip_copy_d2h(h_col, job->d_cnt64, words * sizeof(uint64_t), NULL) || qvz_hip_stream_sync(NULL)) { free(h_col); rc = qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); goto done; }
        } else {
            uint32_t *h32 = (uint32_t *)h_col;                                   /* widened in place, from the top down */
            if (qvz_hip_copy_d2h(h32, job->d_cnt, words * sizeof(uint32_t), NULL) || qvz_hip_stream_sync(NULL)) { free(h_col); rc = qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); goto done; }
            for (size_t i = words; i-- > 0;) h_col[i] = h32[i];
        }
        uint64_t total = 0;
        for (uint32_t s = 0; s < C; ++s) {
            uint64_t sum = 0;
            for (uint32_t p = 0; p < A; ++p)
                for (uint32_t x = 0; x < A; ++x) sum += (uint64_t)x * h_col[((size_t)s * A + p) * A + x];
            job->h_acc[s] = sum;
        }
        for (size_t i = 0; i < (size_t)A * A; ++i) total += h_col[i];           /* column 0's tile: one count per read */
        free(h_col);
        job->h_counts[0] = (uint32_t)N;
        HIP(qvz_hip_copy_h2d(job->d_acc, job->h_acc, (size_t)C * sizeof(uint64_t), NULL));
        HIP(qvz_hip_copy_h2d(job->d_counts, job->h_counts, sizeof(uint32_t), NULL));
        if (job->allreduce) {
            HIP(qvz_hip_copy_h2d(job->d_counts64, &total, sizeof(uint64_t), NULL));
            HIP(qvz_hip_kmeans_update64(job->d_acc, job->d_counts64, K, C, job->d_means, job->d_moved, job->d_kstat, NULL));
        } else
        HIP(qvz_hip_kmeans_update(job->d_acc, job->d_counts, K, C, job->d_means, job->d_moved, job->d_kstat, NULL));
        HIP(qvz_hip_copy_d2h(h_moved, job->d_moved, (K + 1) * sizeof(double), NULL));
        HIP(qvz_hip_stream_sync(NULL));
        iter = h_moved[0] > job->opts.cluster_threshold ? 2 : 1;                /* cluster.c:230-233 */
        if (job->opts.verbose) {
            printf("Cluster 0 moved %f.\n\n", h_moved[1]);                      /* cluster.c:126-127, :236-238 */
            if (iter == 2) printf("Cluster 0 moved %f.\n\n", 0.0);
        }
        loop = 0;
    }
    while (iter < 1000 && loop) {                                               /* cluster.c:221, cluster.h:9 */
        int32_t kstat = 0;
        HIP(qvz_hip_memset(job->d_counts, 0, K * sizeof(uint32_t), NULL));
        HIP(qvz_hip_memset(job->d_acc, 0, (size_t)K * C * sizeof(uint64_t), NULL));
        span_t sp;
        span_begin(job, &sp);
        HIP(qvz_hip_kmeans_step(job->d_sym, N, C, ld, K, job->d_means, job->d_labels, job->d_counts, job->d_acc, NULL));
        span_end(&sp);
        if (job->allreduce) {                                                   /* the sums of every part of the file, 64 bits wide */
            HIP(qvz_hip_widen_u32(job->d_counts, job->d_counts64, K, NULL));
            HIP(job->allreduce(job->allreduce_ctx, job->d_acc, (uint64_t)K * C, 8, NULL));
            HIP(job->allreduce(job->allreduce_ctx, job->d_counts64, K, 8, NULL));
            HIP(qvz_hip_kmeans_update64(job->d_acc, job->d_counts64, K, C, job->d_means, job->d_moved, job->d_kstat, NULL));
        } else
        HIP(qvz_hip_kmeans_update(job->d_acc, job->d_counts, K, C, job->d_means, job->d_moved, job->d_kstat, NULL));
        HIP(qvz_hip_copy_d2h(h_moved, job->d_moved, (K + 1) * sizeof(double), NULL));
        HIP(qvz_hip_copy_d2h(&kstat, job->d_kstat, sizeof(kstat), NULL));
        HIP(qvz_hip_stream_sync(NULL));
        span_add(&sp, &job->k_kmeans_ms);
        if (kstat) { rc = qvz_job_fail(QVZ_E_EMPTY, "a cluster went empty during k-means (the reference divides by zero here, cluster.c:113)"); goto done; }
        if (job->opts.verbose) {
            for (uint32_t j = 0; j < K; ++j) printf("Cluster %d moved %f.\n", j, h_moved[1 + j]);   /* cluster.c:126-127 */
            printf("\n");                                                       /* cluster.c:236-238 */
        }
        loop = h_moved[0] > job->opts.cluster_threshold;                        /* cluster.c:230-233 */
        iter += 1;
    }
    if (job->opts.verbose) printf("\nTotal number of iterations: %d.\n", iter); /* cluster.c:241-243 */
    job->kmeans_iterations = iter;
    HIP(qvz_hip_copy_d2h(job->h_means, job->d_means, (size_t)K * C, NULL));
    if (K > 1) {
        HIP(qvz_hip_copy_d2h(job->h_counts, job->d_counts, K * sizeof(uint32_t), NULL));
        HIP(qvz_hip_copy_d2h(job->h_acc, job->d_acc, (size_t)K * C * sizeof(uint64_t), NULL));
    }
    if (want_host_labels) {
        job->h_labels = (uint8_t *)malloc(N);
        if (!job->h_labels) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        HIP(qvz_hip_copy_d2h(job->h_labels, job->d_labels, N, NULL));
    }
    HIP(qvz_hip_stream_sync(NULL));
    job->clustered = 1;
    job->t_cluster = now() - t0;
done:
    free(h_moved);
    return rc;
}

int qvz_job_stats(qvz_job_t *job) {
    int rc = 0;
    const double t0 = now();
    const size_t words = (size_t)job->K * job->C * A * A;
    if (job->counted) return 0;
    if ((rc = qvz_job_upload(job))) return rc;
    HIP(qvz_hip_malloc((void **)&job->d_cnt, words * sizeof(uint32_t)));
    HIP(qvz_hip_memset(job->d_cnt, 0, words * sizeof(uint32_t), NULL));
    span_t sp;
    span_begin(job, &sp);
    HIP(qvz_hip_pmf_hist(job->d_sym, job->K > 1 ? job->d_labels : NULL, job->N, job->C, job->ld, job->K, job->d_cnt, NULL));
    span_end(&sp);
    if (job->allreduce) {                                                       /* counters of the whole file, 64 bits wide */
        HIP(qvz_hip_malloc((void **)&job->d_cnt64, words * sizeof(uint64_t)));
        HIP(qvz_hip_widen_u32(job->d_cnt, job->d_cnt64, words, NULL));
        HIP(job->allreduce(job->allreduce_ctx, job->d_cnt64, words, 8, NULL));
    }
    HIP(qvz_hip_stream_sync(NULL));
    span_add(&sp, &job->k_pmf_ms);
    job->counted = 1;
    job->t_stats = now() - t0;
done:
    return rc;
}

/* One cluster's generation: a chain of small kernels and host round trips (qvz_codebook.hip). The clusters' codebooks are
 * independent, so with several clusters each gets a host thread and a stream of its own. */
typedef struct {
    qvz_job_t *job;
    uint32_t j;
    void *stream;
    int rc;
    uint64_t deferred;
    char err[256];
} cb_task_t;

static void *cb_task(void *arg) {
    cb_task_t *t = (cb_task_t *)arg;
    qvz_job_t *job = t->job;
    const size_t tile = (size_t)job->C * A * A;
    int rc = qvz_hip_set_device(job->opts.device);                              /* the device is a per-thread setting */
    if (!rc)
        rc = job->d_cnt64 ? qvz_hip_generate_codebook64(job->d_cnt64 + t->j * tile, job->C, job->h_dist, job->opts.mode, job->opts.ratio, 0.0,
                                                        job->cbs[t->j], t->stream)
                          : qvz_hip_generate_codebook(job->d_cnt + t->j * tile, job->C, job->h_dist, job->opts.mode, job->opts.ratio, 0.0,
                                                      job->cbs[t->j], t->stream);
    t->rc = rc;
    if (rc) snprintf(t->err, sizeof(t->err), "codebook of cluster %u: %s", t->j, qvz_hip_last_error());
    t->deferred = qvz_hip_codebook_deferred_decisions();
    return NULL;
}

int qvz_job_codebooks(qvz_job_t *job) {
    int rc = 0;
    const double t0 = now();
    const uint32_t K = job->K;
    cb_task_t *task = NULL;
    pthread_t *th = NULL;
    if (!job->counted && (rc = qvz_job_stats(job))) return rc;
    job->cbs = (qvz_codebook_t **)calloc(K, sizeof(*job->cbs));
    task = (cb_task_t *)calloc(K, sizeof(*task));
    th = (pthread_t *)calloc(K, sizeof(*th));
    if (!job->cbs || !task || !th) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    for (uint32_t j = 0; j < K; ++j) {
        job->cbs[j] = qvz_codebook_alloc(job->C);
        if (!job->cbs[j]) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        task[j].job = job; task[j].j = j;
    }
    HIP(qvz_hip_stream_sync(NULL));                                            /* the counters are complete */
    /* cluster 0 on the calling thread and the default stream, the others on threads and streams of their own */
    {
        /* a stream per cluster that runs beside the default stream, the WELL generator's and the other clusters' (a resident handle
         * keeps them: eight at most, the clusters beyond share in turn) */
        void *others[10];
        int nothers = 0;
        others[nothers++] = job->well_stream;
        for (uint32_t j = 1; j < K; ++j) {
            void **kept = job->pool && j <= 8 ? &job->pool->cb_stream[j - 1] : NULL;
            if (kept && *kept) task[j].stream = *kept;
            else {
                task[j].stream = j <= 8 ? side_stream(others, nothers) : NULL;  /* no stream: the default one, in turn */
                if (kept) *kept = task[j].stream;
            }
            if (task[j].stream && nothers < 10) others[nothers++] = task[j].stream;
        }
    }
    for (uint32_t j = 1; j < K; ++j)
        if (pthread_create(&th[j], NULL, cb_task, &task[j])) { th[j] = 0; cb_task(&task[j]); }
    cb_task(&task[0]);
    for (uint32_t j = 1; j < K; ++j) if (th[j]) pthread_join(th[j], NULL);
    for (uint32_t j = 0; j < K; ++j) {
        if (task[j].stream && !(job->pool && j >= 1 && j <= 8 && job->pool->cb_stream[j - 1] == task[j].stream)) qvz_hip_stream_destroy(task[j].stream);
        if (task[j].rc && !rc) rc = qvz_job_fail(task[j].rc, "%s", task[j].err);
        job->deferred_decisions += task[j].deferred;
    }
    job->t_codebook = now() - t0;
done:
    free(task); free(th);
    return rc;
}

int qvz_job_write_codebooks(qvz_job_t *job, FILE *fout, uint64_t *bytes) {   /* codebook.c:474-498 */
    uint8_t hdr[9];
    uint64_t total = 9;
    qvz_codebook_write_header(hdr, job->K, job->C, job->N);
    if (fwrite(hdr, 1, 9, fout) != 9) return qvz_job_fail(QVZ_E_ARG, "write failed");
    for (uint32_t j = 0; j < job->K; ++j) {
        const size_t sz = qvz_codebook_write(NULL, job->cbs[j]);
        uint8_t *buf = (uint8_t *)malloc(sz);
        if (!buf) return qvz_job_fail(QVZ_E_NOMEM, "host alloc");
        qvz_codebook_write(buf, job->cbs[j]);
        const size_t w = fwrite(buf, 1, sz, fout);
        free(buf);
        if (w != sz) return qvz_job_fail(QVZ_E_ARG, "write failed");
        total += sz;
    }
    if (bytes) *bytes = total;
    return 0;
}

/* One set of per-chunk buffers; two sets alternate so the GPU prepares chunk i+1 while the host codes chunk i */
typedef struct {
    void *stream;
    uint8_t *d_qv, *d_qi, *d_st, *d_text;
    uint16_t *d_feed;
    uint8_t *d_draws;
    double *d_err;
    int32_t *d_status;
    uint64_t *d_rec, *h_rec;      /* model on the GPU: 16-byte records instead of the feed (qvz_hip_model_session_records) */
    uint16_t *h_feed;
    uint8_t *h_labels, *h_text;
    double *h_err;
    int32_t *h_status;
    uint64_t r0, n;               /* reads [r0, r0+n) currently in flight in this set */
} chunk_t;

static void chunk_free(chunk_t *ck) {
    qvz_hip_free(ck->d_qv); qvz_hip_free(ck->d_qi); qvz_hip_free(ck->d_st); qvz_hip_free(ck->d_text);
    qvz_hip_free(ck->d_feed); qvz_hip_free(ck->d_draws); qvz_hip_free(ck->d_err); qvz_hip_free(ck->d_status);
    qvz_hip_free(ck->d_rec); qvz_hip_free_host(ck->h_rec);
    qvz_hip_free_host(ck->h_feed); qvz_hip_free_host(ck->h_labels); qvz_hip_free_host(ck->h_text);
    qvz_hip_free_host(ck->h_err); qvz_hip_free_host(ck->h_status);
    if (ck->stream) qvz_hip_stream_destroy(ck->stream);
    memset(ck, 0, sizeof(*ck));
}

static int issue_chunk(qvz_job_t *job, chunk_t *q, uint64_t r0, uint64_t chunk, const uint16_t *d_ct, const uint16_t *d_qs,
                       const double *d_dist, int want_text, qvz_model_session_t *model) {
    int rc = 0;
    const uint32_t C = job->C, K = job->K;
    const uint64_t stride = (uint64_t)C + 1;
    q->r0 = r0;
    q->n = job->N - r0 < chunk ? job->N - r0 : chunk;
    const uint64_t ldn = qvz_hip_ld(q->n);
    HIP(qvz_hip_memset(q->d_status, 0, sizeof(int32_t), q->stream));
    HIP(qvz_hip_well_draws(job->well_state, q->r0, q->n, C, q->d_draws, ldn, q->stream));
    HIP(qvz_hip_quantize(job->d_sym + q->r0, K > 1 ? job->d_labels + q->r0 : NULL, q->n, C, job->ld, K, d_ct, d_qs, q->d_draws,
                         ldn, d_dist, q->d_qv, q->d_qi, q->d_st, ldn, q->d_err, q->d_status, q->stream));
    if (model) {       /* the adaptive model of this chunk on the GPU: the host gets records, not (quantizer, state) pairs */
        const uint64_t rec_bytes = ((q->n + 1) & ~1ull) * qvz_hip_model_session_pitch(model) * 16;
        HIP(qvz_hip_model_session_records(model, q->d_qi, q->d_st, ldn, K > 1 ? job->d_labels + q->r0 : NULL, q->n, q->d_rec, q->stream));
        HIP(qvz_hip_copy_d2h(q->h_rec, q->d_rec, rec_bytes, q->stream));
    } else {
        HIP(qvz_hip_pack_feed(q->d_qi, q->d_st, q->n, C, ldn, q->d_feed, q->stream));
        HIP(qvz_hip_copy_d2h(q->h_feed, q->d_feed, q->n * C * sizeof(uint16_t), q->stream));
        if (K > 1) HIP(qvz_hip_copy_d2h(q->h_labels, job->d_labels + q->r0, q->n, q->stream));
    }
    HIP(qvz_hip_copy_d2h(q->h_err, q->d_err, q->n * sizeof(double), q->stream));
    HIP(qvz_hip_copy_d2h(q->h_status, q->d_status, sizeof(int32_t), q->stream));
    if (want_text) {
        HIP(qvz_hip_untranspose_text(q->d_qv, q->n, C, ldn, q->d_text, q->stream));
        HIP(qvz_hip_copy_d2h(q->h_text, q->d_text, q->n * stride, q->stream));
    }
done:
    return rc;
}

int qvz_job_compress(qvz_job_t *job, FILE *fout, FILE *fu, uint64_t *stream_bytes, double *distortion_out) {
    int rc = 0;
    const double t0 = now();
    const uint32_t C = job->C, K = job->K;
    const uint64_t N = job->N, stride = (uint64_t)C + 1;
    uint16_t *d_ct = NULL, *d_qs = NULL, *h_ct = NULL, *h_qs = NULL;
    double *d_dist = NULL;
    qvz_coder_t *coder = NULL;
    qvz_model_session_t *model = NULL;
    chunk_t ck[2];
    memset(ck, 0, sizeof(ck));

    /* WELL state (qv_stream.c:70-89): written right after the codebooks */
    if (job->opts.seed_mode == QVZ_SEED_EXPLICIT) {
        memcpy(job->well_state, job->opts.well_state, sizeof(job->well_state));
    } else {
        srand((uint32_t)time(0));                                              /* qv_stream.c:77 (also in -DDEBUG builds) */
        for (int i = 0; i < 32; ++i) job->well_state[i] = job->opts.seed_mode == QVZ_SEED_DEBUG ? 0x55555555u : (uint32_t)rand();
    }
    if (fwrite(job->well_state, sizeof(uint32_t), 32, fout) != 32) return qvz_job_fail(QVZ_E_ARG, "write failed");

    {   /* quantizer tables */
        const size_t ctn = (size_t)K * C * QVZ_CT_STRIDE, qsn = (size_t)K * C * QVZ_QS_STRIDE;
        h_ct = (uint16_t *)malloc(ctn * sizeof(uint16_t));
        h_qs = (uint16_t *)malloc(qsn * sizeof(uint16_t));
        if (!h_ct || !h_qs) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        for (uint32_t j = 0; j < K; ++j)
            qvz_codebook_tables(job->cbs[j], h_ct + (size_t)j * C * QVZ_CT_STRIDE, h_qs + (size_t)j * C * QVZ_QS_STRIDE);
        HIP(qvz_hip_malloc((void **)&d_ct, ctn * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&d_qs, qsn * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&d_dist, sizeof(job->h_dist)));
        HIP(qvz_hip_copy_h2d(d_ct, h_ct, ctn * sizeof(uint16_t), NULL));
        HIP(qvz_hip_copy_h2d(d_qs, h_qs, qsn * sizeof(uint16_t), NULL));
        HIP(qvz_hip_copy_h2d(d_dist, job->h_dist, sizeof(job->h_dist), NULL));
        HIP(qvz_hip_stream_sync(NULL));
    }

    /* quantize (GPU) + arithmetic coding (host), chunked and overlapped (qv_compressor.c:48-143) */
    coder = qvz_coder_create(K, C, (const qvz_codebook_t *const *)job->cbs);
    if (!coder) { rc = qvz_job_fail(QVZ_E_NOMEM, "coder alloc"); goto done; }
    /* Where the adaptive model runs. Default: on the GPU (a wave per column over the whole file, include/qvz_coder.h), the
     * host core keeps only the interval chain; QVZ_STRICT_MODEL=host, or tables too large for LDS: the host does both. */
    {
        const char *sm = getenv("QVZ_STRICT_MODEL");
        if (!(sm && strcmp(sm, "host") == 0) && qvz_hip_model_session_open(C, K, (const qvz_codebook_t *const *)job->cbs, &model) != QVZ_OK) model = NULL;
    }
    uint64_t chunk = job->opts.chunk_reads ? job->opts.chunk_reads : (model ? 1u << 18 : 1u << 20);   /* records are 8x the feed's bytes */
    chunk = (chunk + 255) & ~255ull;
    if (chunk > ((N + 255) & ~255ull)) chunk = (N + 255) & ~255ull;
    const uint64_t ldc = qvz_hip_ld(chunk);
    for (int b = 0; b < 2; ++b) {
        chunk_t *k_ = &ck[b];
        HIP(qvz_hip_stream_create(&k_->stream));
        HIP(qvz_hip_malloc((void **)&k_->d_qv, (size_t)C * ldc));
        HIP(qvz_hip_malloc((void **)&k_->d_qi, (size_t)C * ldc));
        HIP(qvz_hip_malloc((void **)&k_->d_st, (size_t)C * ldc));
        if (model) {
            const size_t rec_bytes = (size_t)((chunk + 1) & ~1ull) * qvz_hip_model_session_pitch(model) * 16;
            HIP(qvz_hip_malloc((void **)&k_->d_rec, rec_bytes));
            HIP(qvz_hip_malloc_host((void **)&k_->h_rec, rec_bytes));
        } else {
            HIP(qvz_hip_malloc((void **)&k_->d_feed, chunk * C * sizeof(uint16_t)));
            HIP(qvz_hip_malloc_host((void **)&k_->h_feed, chunk * C * sizeof(uint16_t)));
        }
        HIP(qvz_hip_malloc((void **)&k_->d_draws, (size_t)C * ldc));
        HIP(qvz_hip_malloc((void **)&k_->d_err, chunk * sizeof(double)));
        HIP(qvz_hip_malloc((void **)&k_->d_status, sizeof(int32_t)));
        HIP(qvz_hip_malloc_host((void **)&k_->h_labels, chunk));
        HIP(qvz_hip_malloc_host((void **)&k_->h_err, chunk * sizeof(double)));
        HIP(qvz_hip_malloc_host((void **)&k_->h_status, sizeof(int32_t)));
        if (fu) {
            HIP(qvz_hip_malloc((void **)&k_->d_text, chunk * stride));
            HIP(qvz_hip_malloc_host((void **)&k_->h_text, chunk * stride));
        }
    }
    {
        double distortion = 0.0;
        uint64_t next = 0;       /* first read not yet issued to the GPU */
        int cur = 0;
        HIP(qvz_hip_stream_sync(NULL));
        if ((rc = issue_chunk(job, &ck[cur], next, chunk, d_ct, d_qs, d_dist, fu != NULL, model))) goto done;
        next += ck[cur].n;
        while (ck[cur].n) {
            chunk_t *q = &ck[cur];
            HIP(qvz_hip_stream_sync(q->stream));
            ck[cur ^ 1].n = 0;
            if (next < N) {      /* the GPU prepares the next chunk while this one is coded */
                if ((rc = issue_chunk(job, &ck[cur ^ 1], next, chunk, d_ct, d_qs, d_dist, fu != NULL, model))) goto done;
                next += ck[cur ^ 1].n;
            }
            if (*q->h_status) { rc = qvz_job_fail(QVZ_E_DEAD_CTX, "quantizer reached a context the codebook does not define (assert in the reference, codebook.c:164)"); goto done; }
            if (job->opts.verbose) {
                /* "Line: %dM" at the first line of every 1M-line block (qv_compressor.c:79-81) */
                for (uint64_t blk = (q->r0 + QVZ_LINES_PER_BLOCK - 1) / QVZ_LINES_PER_BLOCK; blk * QVZ_LINES_PER_BLOCK < q->r0 + q->n; ++blk)
                    printf("Line: %dM\n", (int)blk);
            }
            if (model ? qvz_coder_encode_records(coder, q->h_rec, q->n, qvz_hip_model_session_pitch(model))
                      : qvz_coder_encode(coder, K > 1 ? q->h_labels : NULL, q->h_feed, q->n)) { rc = qvz_job_fail(QVZ_E_DATA, "coder feed out of range"); goto done; }
            for (uint64_t i = 0; i < q->n; ++i) distortion += q->h_err[i] / ((double)C);   /* qv_compressor.c:127 */
            if (fu && fwrite(q->h_text, 1, q->n * stride, fu) != q->n * stride) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
            {
                uint64_t sz;
                const uint8_t *data = qvz_coder_data(coder, &sz);
                if (sz && fwrite(data, 1, sz, fout) != sz) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
                qvz_coder_drop(coder, sz);
            }
            cur ^= 1;
        }
        if (model && qvz_hip_model_session_status(model, NULL)) { rc = qvz_job_fail(QVZ_E_DEAD_CTX, "%s", qvz_hip_last_error()); goto done; }
        qvz_coder_finish(coder);                                               /* encoder_last_step, arith.c:99-116 */
        {
            uint64_t sz;
            const uint8_t *data = qvz_coder_data(coder, &sz);
            if (sz && fwrite(data, 1, sz, fout) != sz) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
            qvz_coder_drop(coder, sz);
        }
        if (distortion_out) *distortion_out = distortion / ((double)N);       /* qv_compressor.c:139-140 */
    }
    if (stream_bytes) *stream_bytes = qvz_coder_bytes(coder);
    job->t_encode = now() - t0;
done:
    for (int b = 0; b < 2; ++b) chunk_free(&ck[b]);
    qvz_hip_model_session_close(model);
    qvz_coder_destroy(coder);
    free(h_ct); free(h_qs);
    qvz_hip_free(d_ct); qvz_hip_free(d_qs); qvz_hip_free(d_dist);
    return rc;
}

/* A stream whose kernels run BESIDE the default stream's and those of `others`: candidates are probed (~1 ms each,
 * qvz_hip_streams_overlap) and the first that overlaps with all of them is kept; NULL when none can be made. */
static void *side_stream(void *const *others, int nothers) {
    void *cand[8] = {NULL};
    int n = 0, good = -1;
    while (n < 8 && good < 0) {
        int ok = 0;
        if (qvz_hip_stream_create(&cand[n])) break;
        if (qvz_hip_streams_overlap(cand[n], NULL, &ok) == QVZ_OK && ok) {
            good = n;
            for (int i = 0; i < nothers && good >= 0; ++i) {
                int o = 1;
                if (others[i] && (qvz_hip_streams_overlap(cand[n], others[i], &o) != QVZ_OK || !o)) good = -1;
            }
        }
        ++n;
    }
    if (good < 0 && n > 0) good = n - 1;                                        /* none overlaps with everything: the last one, in turn */
    for (int i = 0; i < n; ++i) if (i != good) qvz_hip_stream_destroy(cand[i]);
    return good < 0 ? NULL : cand[good];
}

/* The WELL state of the file (qv_stream.c:70-89) and its draw matrix. The draws depend on the seed only, so they are
 * generated on a side stream beside the codebook generation (small latency-bound kernels and host round trips): 25 ms of
 * generator and transposition kernels at 100M x 150 that the quantizer would otherwise wait for. */
int qvz_job_start_draws(qvz_job_t *job) {
    int rc = 0;
    if (job->d_draws) return 0;
    if (!job->well_chosen) {
        if (job->opts.seed_mode == QVZ_SEED_EXPLICIT) {
            memcpy(job->well_state, job->opts.well_state, sizeof(job->well_state));
        } else {
            srand((uint32_t)time(0));                                          /* qv_stream.c:77 (also in -DDEBUG builds) */
            for (int i = 0; i < 32; ++i) job->well_state[i] = job->opts.seed_mode == QVZ_SEED_DEBUG ? 0x55555555u : (uint32_t)rand();
        }
        job->well_chosen = 1;
    }
    HIP(qvz_hip_set_device(job->opts.device));
    if (!(job->d_draws = early_draws_plane(job)))                              /* allocated ahead of time, beside the upload */
        HIP(job_malloc(job, (void **)&job->d_draws, (size_t)job->C * job->ld, QVZ_POOL_DRAWS));
    /* The runtime spreads streams over a few hardware queues and a queue runs its kernels in order: a side stream that
     * shares the default stream's queue would put the generator IN FRONT of the codebook kernels instead of beside them
     * (HISTORY.md section 8). A candidate is probed (~1 ms) and dropped for the next one if it does not overlap. */
    if (job->pool && job->pool->well_stream) job->well_stream = job->pool->well_stream;   /* a resident handle's: made and probed once */
    else {
        if (!(job->well_stream = side_stream(NULL, 0))) { rc = qvz_job_fail(QVZ_E_HIP, "no stream for the WELL draws: %s", qvz_hip_last_error()); goto done; }
        if (job->pool) job->pool->well_stream = job->well_stream;
    }
    HIP(qvz_hip_stream_wait(job->well_stream, NULL));
    HIP(qvz_hip_well_draws(job->well_state, job->first_read, job->N, job->C, job->d_draws, job->ld, job->well_stream));
done:
    return rc;
}

static void *workspace_main(void *arg) {
    qvz_job_t *job = (qvz_job_t *)arg;
    uint64_t free_b = 0, per = 0;
    const uint64_t plane = ((uint64_t)job->C * job->ld + 255) & ~255ull, errb = (job->N * sizeof(double) + 255) & ~255ull;
    job->work_rc = qvz_hip_set_device(job->opts.device);
    if (job->work_rc) { __atomic_store_n(&job->draws_early_state, 2, __ATOMIC_RELEASE); return NULL; }
    /* First the draw matrix, an allocation of its own: the WELL generator is the first to need any of this (it starts beside the
     * codebooks, qvz_job_start_draws waits for THIS allocation only), and the matrix is given back right after the quantizer - room
     * the coder's stream area and the -u staging can have. */
    const double t0 = now();
    void *d = NULL;
    if (qvz_hip_malloc(&d, (size_t)plane) != QVZ_OK) d = NULL;
    job->d_draws_early = (uint8_t *)d;
    __atomic_store_n(&job->draws_early_state, 2, __ATOMIC_RELEASE);
    {   /* then the quantizer's planes: qi | st | [qv] | err, each rounded up to 256 bytes (what job_plane hands out) */
        const uint64_t need = plane * (job->want_text ? 3 : 2) + errb;
        if (qvz_hip_malloc(&job->d_planes, (size_t)need) == QVZ_OK) job->planes_bytes = need; else job->d_planes = NULL;
        if (getenv("QVZ_TIMING")) fprintf(stderr, "qvz_timing: draws + planes of %.1f GB allocated in %.4f s\n", (plane + need) / 1e9, now() - t0);
    }
    /* last the coder's record workspace, the largest and the last to be needed: everything at once if it fits, 96 GiB at most,
     * and 24 GiB of what is free now are left alone (the matrix is being uploaded, the stream area may need an allocation) */
    if (job->shard_bounds) { for (uint32_t g = 0; g < job->work_shards; ++g) if (job->shard_bounds[g + 1] - job->shard_bounds[g] > per) per = job->shard_bounds[g + 1] - job->shard_bounds[g]; }
    else per = (job->N + job->work_shards - 1) / job->work_shards + 4;
    uint64_t want = qvz_hip_encode_planes_work_bytes(per, job->C, job->K, job->work_shards);
    if (want > (96ull << 30)) want = 96ull << 30;
    if (qvz_hip_mem_info(&free_b, NULL) == QVZ_OK) {
        const uint64_t missing = (job->d_planes ? 0 : plane * 4 + errb) + (job->d_draws_early ? 0 : plane);   /* what the stages will allocate themselves */
        const uint64_t room = free_b > missing + (24ull << 30) ? free_b - missing - (24ull << 30) : 0;
        if (want > room) want = room;
    }
    if (want < (1ull << 30)) return NULL;                                       /* little room: the coder sizes its own later */
    const double t1 = now();
    if (qvz_hip_malloc(&job->d_work, (size_t)want) == QVZ_OK) job->work_bytes = want; else job->d_work = NULL;
    if (getenv("QVZ_TIMING")) fprintf(stderr, "qvz_timing: workspace of %.1f GiB allocated in %.4f s\n", want / 1073741824.0, now() - t1);
    return NULL;
}

int qvz_job_start_workspace(qvz_job_t *job, uint32_t shards) {
    pthread_t th;
    if (job->d_work || job->work_running || !shards) return 0;
    job->work_shards = shards;
    job->draws_early_state = 1;                                                 /* 1: being allocated, 2: decided (d_draws_early or NULL) */
    if (pthread_create(&th, NULL, workspace_main, job) == 0) { job->work_thread = (unsigned long)th; job->work_running = 1; }
    else job->draws_early_state = 0;
    return 0;
}

static void workspace_join(qvz_job_t *job) {
    if (job->work_running) { pthread_join((pthread_t)job->work_thread, NULL); job->work_running = 0; }
}

/* The draw matrix of the early allocation, handed over ONCE (the caller owns and frees it), or NULL when there is none: waits for
 * that one allocation, not for the thread - the record workspace behind it is up to 96 GiB and may take a second in a young process. */
static uint8_t *early_draws_plane(qvz_job_t *job) {
    if (!job->draws_early_state) return NULL;                                   /* no early allocation was started */
    while (__atomic_load_n(&job->draws_early_state, __ATOMIC_ACQUIRE) == 1) usleep(20);
    uint8_t *d = job->d_draws_early;
    job->d_draws_early = NULL;
    return d;
}

/* plane i of the early allocation (0 qi, 1 st, 2 qv when want_text; i < 0: the errors, behind the planes), or NULL when there is
 * none (the caller then allocates its own); waits for the allocating thread */
static uint8_t *job_plane(qvz_job_t *job, int i) {
    workspace_join(job);
    if (!job->d_planes) return NULL;
    const uint64_t plane = ((uint64_t)job->C * job->ld + 255) & ~255ull;
    const int n_planes = job->want_text ? 3 : 2;
    if (i >= n_planes) return NULL;
    return (uint8_t *)job->d_planes + plane * (uint64_t)(i < 0 ? n_planes : i);
}

/* The first coder call of a process loads its code object, sizes and allocates its kept buffers and makes its stream pair:
 * 40-50 ms before the first kernel. A one-shot process does that here, on a thread beside the quantizer. */
typedef struct { qvz_job_t *job; uint32_t shards; } prepare_task_t;
static void *prepare_main(void *arg) {
    prepare_task_t *t = (prepare_task_t *)arg;
    qvz_job_t *job = t->job;
    workspace_join(job);
    if (job->d_work && job->work_shards == t->shards && qvz_hip_set_device(job->opts.device) == QVZ_OK)
        (void)qvz_hip_encode_planes_prepare(job->N, job->C, job->K, (const qvz_codebook_t *const *)job->cbs, t->shards, job->shard_bounds, job->work_bytes);
    return NULL;                                                                /* a failure here shows again, with its message, in the call that codes */
}

/* what runs beside the coder kernels, each on a host thread of its own */
typedef struct {
    int device, rc;
    FILE *fu;
    const uint8_t *d_qv;
    const double *d_err;
    uint64_t N, ld;
    uint32_t C;
    double sum, *keep;
    char err[256];
} side_task_t;

static void *side_text(void *arg) {
    side_task_t *t = (side_task_t *)arg;
    t->rc = qvz_hip_set_device(t->device);
    if (!t->rc) t->rc = qvz_egress_text(t->fu, t->d_qv, t->N, t->C, t->ld, NULL);
    if (t->rc) snprintf(t->err, sizeof(t->err), "-u text: %s", qvz_encode_last_error());
    return NULL;
}

static void *side_errors(void *arg) {
    side_task_t *t = (side_task_t *)arg;
    t->rc = qvz_hip_set_device(t->device);
    if (!t->rc) t->rc = qvz_sum_errors(t->d_err, t->N, t->C, &t->sum, t->keep, NULL);
    if (t->rc) snprintf(t->err, sizeof(t->err), "per-read errors: %s", qvz_encode_last_error());
    return NULL;
}

/* start_qv_compression for sharded output (option B): the whole matrix is quantized on the GPU,
 * `shards` contiguous read ranges are coded as independent streams by the GPU coder (qvz_hip_encode_planes), and the
 * streams are written as one shard set, or every stream as a complete reference-format file "<prefix>.<g>" (header with
 * the shard's line count, the whole file's codebooks, the WELL state advanced to the shard's first draw, the stream): the
 * unmodified reference decoder decodes each, and the decodes concatenate to the -u text.
 *
 * Schedule (the one qvz_amd/pipeline.py measures, plus the products' way out): the WELL draws were started beside the
 * codebooks (qvz_job_start_draws); quantizer; then, side by side, the coder kernels (two streams of their own inside
 * qvz_hip_encode_planes), the -u text leaving through pinned slabs into the file (qvz_egress_text) and the per-read
 * errors summed in file order on one host core (qvz_sum_errors: the reference's sequential sum, qv_compressor.c:127,
 * 139-140); at the end the streams, packed back to back, through pinned slabs into the output. No N-sized host buffer. */
static int compress_sharded_impl(qvz_job_t *job, const char *prefix, FILE *fset, FILE *fu, uint32_t shards, uint64_t *stream_bytes,
                                 uint64_t *file_bytes, double *distortion_out) {
    int rc = 0;
    const double t0 = now();
    double t1, t2;
    const uint32_t C = job->C, K = job->K;
    const uint64_t N = job->N, ld = job->ld;
    uint16_t *d_ct = NULL, *d_qs = NULL, *h_ct = NULL, *h_qs = NULL;
    double *d_dist = NULL, *d_err = NULL;
    int planes_early = 0;
    uint8_t *d_qv = NULL, *d_qi = NULL, *d_st = NULL, *d_out = NULL, *d_out_own = NULL, *d_packed = NULL, *d_packed_own = NULL, *h_buf = NULL, *image = NULL;
    int32_t *d_status = NULL, status = 0;
    uint64_t *sizes = NULL, *first = NULL;
    uint32_t *wstates = NULL;
    pthread_t th_text, th_err, th_prep;
    int text_running = 0, err_running = 0, prep_running = 0;
    side_task_t task_text, task_err;
    prepare_task_t task_prep = {job, shards};
    memset(&task_text, 0, sizeof(task_text));
    memset(&task_err, 0, sizeof(task_err));
    if (!shards || (!job->shard_bounds && shards > qvz_shard_limit(N))) return qvz_job_fail(QVZ_E_ARG, "%u shards for %llu reads: a shard needs at least 4 reads", shards, (unsigned long long)N);

    if ((rc = qvz_job_start_draws(job))) goto done;                            /* already running when the caller started them */
    prep_running = pthread_create(&th_prep, NULL, prepare_main, &task_prep) == 0;
    {
        const size_t ctn = (size_t)K * C * QVZ_CT_STRIDE, qsn = (size_t)K * C * QVZ_QS_STRIDE;
        h_ct = (uint16_t *)malloc(ctn * sizeof(uint16_t));
        h_qs = (uint16_t *)malloc(qsn * sizeof(uint16_t));
        sizes = (uint64_t *)calloc(shards, sizeof(uint64_t));
        first = (uint64_t *)calloc((size_t)shards + 1, sizeof(uint64_t));
        if (!h_ct || !h_qs || !sizes || !first) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        for (uint32_t j = 0; j < K; ++j)
            qvz_codebook_tables(job->cbs[j], h_ct + (size_t)j * C * QVZ_CT_STRIDE, h_qs + (size_t)j * C * QVZ_QS_STRIDE);
        HIP(qvz_hip_malloc((void **)&d_ct, ctn * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&d_qs, qsn * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&d_dist, sizeof(job->h_dist)));
        HIP(qvz_hip_copy_h2d(d_ct, h_ct, ctn * sizeof(uint16_t), NULL));
        HIP(qvz_hip_copy_h2d(d_qs, h_qs, qsn * sizeof(uint16_t), NULL));
        HIP(qvz_hip_copy_h2d(d_dist, job->h_dist, sizeof(job->h_dist), NULL));
    }
    if (job->shard_bounds) memcpy(first, job->shard_bounds, ((size_t)shards + 1) * sizeof(uint64_t));
    else qvz_shard_bounds(N, shards, first);
    /* quantizer: the qv plane (the -u product) only for a caller that wants the text */
    if (job_plane(job, 0) && (!fu || job_plane(job, 2))) {                     /* allocated ahead of time, beside the upload */
        planes_early = 1;
        d_qi = job_plane(job, 0); d_st = job_plane(job, 1); d_err = (double *)job_plane(job, -1);
        if (fu) d_qv = job_plane(job, 2);
    } else {
        if (fu || job->want_text) HIP(job_malloc(job, (void **)&d_qv, (size_t)C * ld, QVZ_POOL_QV));
        HIP(job_malloc(job, (void **)&d_qi, (size_t)C * ld, QVZ_POOL_QI));
        HIP(job_malloc(job, (void **)&d_st, (size_t)C * ld, QVZ_POOL_ST));
        HIP(job_malloc(job, (void **)&d_err, N * sizeof(double), QVZ_POOL_ERR));
    }
    HIP(qvz_hip_malloc((void **)&d_status, sizeof(int32_t)));
    HIP(qvz_hip_memset(d_status, 0, sizeof(int32_t), NULL));
    HIP(qvz_hip_stream_wait(NULL, job->well_stream));                          /* the draws, on the device's side */
    span_t sp_q;
    span_begin(job, &sp_q);
    HIP(qvz_hip_quantize(job->d_sym, K > 1 ? job->d_labels : NULL, N, C, ld, K, d_ct, d_qs, job->d_draws, ld, d_dist, d_qv, d_qi, d_st,
                         ld, d_err, d_status, NULL));
    span_end(&sp_q);
    HIP(qvz_hip_copy_d2h(&status, d_status, sizeof(status), NULL));
    HIP(qvz_hip_stream_sync(NULL));
    span_add(&sp_q, &job->k_quantize_ms);
    if (status) { rc = qvz_job_fail(QVZ_E_DEAD_CTX, "quantizer reached a context the codebook does not define (assert in the reference, codebook.c:164)"); goto done; }
    if (!job->keep_resident) {                                                  /* (a sweep's next point quantizes the same matrix with the same draws) */
        job_free(job, job->d_draws);                                           /* 15 GB at 100M x 150, an allocation of its own for this */
        job->d_draws = NULL;
        /* the symbols are not needed again either (this stage ends a job; qvz_job_release frees what is left): 15 GB at 100M x
         * 150 that the coder's buffers and the -u staging can have - with -u the process is otherwise near what the card holds */
        if (!job->sym_borrowed) qvz_hip_free(job->d_sym);
        job->d_sym = NULL;
    }
    t1 = now();
    job->t_quantize = t1 - t0;

    /* beside the coder kernels: the per-read errors (0.8 GB over PCIe, then the file-order sum on one host core). The -u text starts
     * when the coder has finished and runs beside the way out of the streams (started beside the coder kernels instead it made no
     * difference beyond the noise of the file writes: profiles/r03_e2e_cli.txt). */
    task_err.device = task_text.device = job->opts.device;
    task_err.d_err = d_err; task_err.N = N; task_err.C = C;
    if (job->collect) {
        job->out_err = (double *)malloc(N * sizeof(double));                    /* the caller sums all parts in file order */
        if (!job->out_err) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        task_err.keep = job->out_err;
    }
    if (job->keep_device) HIP(qvz_hip_sum_errors(d_err, N, C, job->d_errsum, NULL));   /* a resident encode: this part's sum stays on the device */
    else err_running = pthread_create(&th_err, NULL, side_errors, &task_err) == 0;
    if (fu) {
        task_text.fu = fu; task_text.d_qv = d_qv; task_text.N = N; task_text.C = C; task_text.ld = ld;
    }
    {   /* coding. Room per stream: 6 bits per symbol to begin with (real data needs 1-2); a stream that outgrows it
         * (-r above 6, near-uniform input) makes the coder report QVZ_E_NOMEM and the call is repeated with twice the room,
         * up to the 22 bits per symbol no symbol can exceed (counts >= 1 of n <= 2^19 + 8, plus the flush) */
        uint64_t per = 0, cap = 0, total = 0, ftotal = 0;
        for (uint32_t g = 0; g < shards; ++g) if (first[g + 1] - first[g] > per) per = first[g + 1] - first[g];
        per += 4;
        for (uint32_t bits = 6; ; bits *= 2) {
            cap = ((per * C * bits) / 8 + 512 + 3) & ~3ull;
            const double ta = now();
            if (prep_running) { pthread_join(th_prep, NULL); prep_running = 0; }
            const double tb = now();
            workspace_join(job);
            const double tc = now();
            const int mine = job->d_work && job->work_shards == shards;        /* allocated ahead of time for this very cut */
            /* The streams' area (11 GB at 100M x 150) comes out of the END of the record workspace when that leaves the coder room
             * for two full segments (a third of the 96 GiB): one large hipMalloc less in front of the coder
             * kernels - the place where a process that starts soon after another has freed its HBM was seen to stall for a
             * second. Otherwise, and when a stream has outgrown its allotment, an allocation of its own. */
            uint64_t work_use = mine ? job->work_bytes : 0;
            const uint64_t area = (((uint64_t)shards * cap) + 4095) & ~4095ull;
            const uint64_t two_segments = 2 * qvz_hip_encode_planes_work_bytes((per + (K > 1 ? 2 : 7)) / (K > 1 ? 3 : 8) + 64, C, K, shards);
            job_free(job, d_out_own); d_out_own = NULL;
            if (mine && bits == 6 && work_use > area && work_use - area >= two_segments) {
                work_use = (work_use - area) & ~255ull;
                d_out = (uint8_t *)job->d_work + work_use;
            } else {
                HIP(job_malloc(job, (void **)&d_out_own, (size_t)shards * cap, QVZ_POOL_OUT));
                d_out = d_out_own;
            }
            if (getenv("QVZ_TIMING")) fprintf(stderr, "qvz_timing: before the coder call: waited %.4f s for the coder's set-up thread, %.4f s for the workspace thread; stream area of %.1f GB %s in %.4f s\n",
                                              tb - ta, tc - tb, (double)shards * cap / 1e9, d_out_own ? "allocated" : "taken from the workspace", now() - tc);
            const int crc = qvz_hip_encode_planes_bounds(d_qi, d_st, ld, K > 1 ? job->d_labels : NULL, N, C, K, (const qvz_codebook_t *const *)job->cbs,
                                                         shards, job->shard_bounds, d_out, cap, sizes, mine ? job->d_work : NULL, work_use, NULL);
            if (crc == QVZ_OK) break;
            if (crc != QVZ_E_NOMEM || bits >= 24) { rc = qvz_job_fail(crc, "%s", qvz_hip_last_error()); goto done; }
        }
        t2 = now();
        job->t_coder = t2 - t1;
        for (uint32_t g = 0; g < shards; ++g) total += sizes[g];
        /* the streams leave the device packed back to back: 1-2 bits per symbol instead of the 6 the area was sized for */
        /* (into the record workspace when there is one: the coder has finished with it, and one more large hipMalloc is one more
         * place for a young process's allocation stall) */
        if (!job->keep_device) {                                                /* (a resident encode packs when it is asked to: qvz_resident_pack) */
            if (job->d_work && job->work_shards == shards && !d_out_own && total <= (uint64_t)(d_out - (uint8_t *)job->d_work)) d_packed = (uint8_t *)job->d_work;
            else { HIP(job_malloc(job, (void **)&d_packed_own, total ? total : 1, QVZ_POOL_PACKED)); d_packed = d_packed_own; }
            HIP(qvz_hip_gather_streams(d_out, cap, sizes, shards, d_packed, NULL));
        }
        if (job->keep_device) {
            HIP(qvz_hip_copy_d2h(&job->r_errsum, job->d_errsum, sizeof(double), NULL));
            HIP(qvz_hip_stream_sync(NULL));
            if (distortion_out) *distortion_out = job->r_errsum / ((double)N);
        } else {
            if (err_running) { pthread_join(th_err, NULL); err_running = 0; } else side_errors(&task_err);
            if (task_err.rc) { rc = qvz_job_fail(task_err.rc, "%s", task_err.err); goto done; }
            if (distortion_out) *distortion_out = task_err.sum / ((double)N);  /* qv_compressor.c:139-140 */
        }
        if (fu && !text_running) text_running = pthread_create(&th_text, NULL, side_text, &task_text) == 0;
        if (job->keep_device) {                                                 /* a resident encode: the products stay where they are */
            job->r_out = d_out; job->r_cap = cap; job->r_out_in_work = d_out_own == NULL;
            job->r_packed = NULL; job->r_total = total;
            job->r_sizes = sizes; sizes = NULL;
            job->r_qv = d_qv;
        } else if (job->collect) {                                                     /* a part of a file: the caller assembles the whole */
            h_buf = (uint8_t *)malloc(total ? total : 1);
            if (!h_buf) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
            HIP(qvz_hip_copy_d2h(h_buf, d_packed, total, NULL));
            HIP(qvz_hip_stream_sync(NULL));
            job->out_streams = h_buf; h_buf = NULL;
            job->out_sizes = sizes; sizes = NULL;
        } else if (fset) {                                                      /* one shard set: codebooks once (qvz_shardset.h) */
            qvz_shardset_info_t si;
            if ((rc = qvz_shardset_write_head(fset, K, C, N, shards, job->well_state, (const qvz_codebook_t *const *)job->cbs, sizes, &si))) goto done;
            if ((rc = qvz_egress_bytes(fset, d_packed, total, NULL))) goto done;
            ftotal = si.file_bytes;
        } else {                                                                /* one reference-format file per shard */
            uint64_t img_cap = 0;
            uint64_t *jump = (uint64_t *)malloc((size_t)shards * sizeof(uint64_t));
            wstates = (uint32_t *)malloc((size_t)shards * 128);
            h_buf = (uint8_t *)malloc(total ? total : 1);
            if (!jump || !wstates || !h_buf) { free(jump); rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
            for (uint32_t g = 0; g < shards; ++g) jump[g] = (job->first_read + first[g]) * C / 4;   /* generator at the shard's first draw, n = 0 */
            const int jrc = qvz_hip_well_jump_many(job->well_state, jump, shards, wstates);
            free(jump);
            if (jrc) { rc = qvz_job_fail(jrc, "%s", qvz_hip_last_error()); goto done; }
            HIP(qvz_hip_copy_d2h(h_buf, d_packed, total, NULL));
            HIP(qvz_hip_stream_sync(NULL));
            uint64_t at = 0;
            for (uint32_t g = 0; g < shards; at += sizes[g], ++g) {
                char name[4096];
                const uint64_t reads = first[g + 1] - first[g];
                FILE *f;
                size_t need;
                if ((size_t)snprintf(name, sizeof(name), "%s.%04u", prefix, g) >= sizeof(name)) { rc = qvz_job_fail(QVZ_E_ARG, "output name too long"); goto done; }
                need = qvz_shard_file_image(NULL, K, C, reads, (const qvz_codebook_t *const *)job->cbs, wstates + (size_t)g * 32, NULL, sizes[g]);
                if (need > img_cap) {
                    free(image);
                    image = (uint8_t *)malloc(need);
                    img_cap = need;
                    if (!image) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
                }
                qvz_shard_file_image(image, K, C, reads, (const qvz_codebook_t *const *)job->cbs, wstates + (size_t)g * 32, h_buf + at, sizes[g]);
                f = fopen(name, "wb");
                if (!f) { rc = qvz_job_fail(QVZ_E_ARG, "Unable to open output file"); goto done; }
                if (fwrite(image, 1, need, f) != need) { fclose(f); rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
                fclose(f);
                ftotal += need;
            }
        }
        if (fu) {                                                               /* the -u text: wait for it (or write it now) */
            if (text_running) { pthread_join(th_text, NULL); text_running = 0; } else side_text(&task_text);
            if (task_text.rc) { rc = qvz_job_fail(task_text.rc, "%s", task_text.err); goto done; }
        }
        if (stream_bytes) *stream_bytes = total;
        if (file_bytes) *file_bytes = ftotal;
        job->t_egress = now() - t2;
    }
    job->t_encode = now() - t0;
done:
    if (prep_running) pthread_join(th_prep, NULL);
    if (err_running) pthread_join(th_err, NULL);                                /* they read device buffers freed below */
    if (text_running) pthread_join(th_text, NULL);
    free(h_ct); free(h_qs); free(sizes); free(first); free(h_buf); free(image); free(wstates);
    qvz_hip_free(d_ct); qvz_hip_free(d_qs); qvz_hip_free(d_dist);
    if (!planes_early) { job_free(job, d_err); job_free(job, d_qv); job_free(job, d_qi); job_free(job, d_st); }   /* (the early planes go with the job) */
    job_free(job, d_out_own); job_free(job, d_packed_own); qvz_hip_free(d_status);
    if (!job->keep_resident || rc) {
        if (job->d_draws) { job_free(job, job->d_draws); job->d_draws = NULL; }
        if (job->well_stream && !(job->pool && job->pool->well_stream == job->well_stream)) qvz_hip_stream_destroy(job->well_stream);
        job->well_stream = NULL;
    }
    return rc;
}

int qvz_job_compress_sharded(qvz_job_t *job, const char *prefix, FILE *fu, uint32_t shards, uint64_t *stream_bytes,
                             double *distortion_out) {
    return compress_sharded_impl(job, prefix, NULL, fu, shards, stream_bytes, NULL, distortion_out);
}

int qvz_job_compress_shardset(qvz_job_t *job, FILE *fout, FILE *fu, uint32_t shards, uint64_t *stream_bytes,
                              uint64_t *file_bytes, double *distortion_out) {
    return compress_sharded_impl(job, NULL, fout, fu, shards, stream_bytes, file_bytes, distortion_out);
}

void qvz_job_release(qvz_job_t *job) {
    if (job->cbs) { for (uint32_t j = 0; j < job->K; ++j) qvz_codebook_free(job->cbs[j]); free(job->cbs); }
    free(job->h_labels); free(job->h_means); free(job->h_counts); free(job->h_acc);
    if (!job->sym_borrowed) qvz_hip_free(job->d_sym);
    job_free(job, job->d_labels); qvz_hip_free(job->d_means); qvz_hip_free(job->d_cnt);
    qvz_hip_free(job->d_counts); qvz_hip_free(job->d_acc); qvz_hip_free(job->d_moved); qvz_hip_free(job->d_kstat);
    qvz_hip_free(job->d_cnt64); qvz_hip_free(job->d_counts64);
    job_free(job, job->d_draws);
    if (job->well_stream && !(job->pool && job->pool->well_stream == job->well_stream)) qvz_hip_stream_destroy(job->well_stream);
    workspace_join(job);
    qvz_hip_free(job->d_draws_early);                                           /* never handed over: no draws were started */
    job_free(job, job->d_work);
    qvz_hip_free(job->d_errsum); free(job->r_sizes);
    qvz_hip_free(job->d_planes);
    memset(job, 0, sizeof(*job));
}

/* ------------------------------------------------------------------------------------------------------- */
/* a rate-distortion sweep's side of encode_buffer: the points, where their outputs go, where their results go */
typedef struct {
    const qvz_sweep_point_t *points;
    uint32_t n;
    const char *out_path, *u_path;
    int keep_all;
    qvz_result_t *res;            /* [n] or NULL */
} sweep_t;

static int encode_buffer(const uint8_t *text, uint64_t bytes, FILE *fout, FILE *fu, const qvz_options_t *opts,
                         qvz_result_t *res, const char *shard_prefix, uint32_t shards, const sweep_t *sw);

int qvz_encode_buffer(const uint8_t *text, uint64_t bytes, FILE *fout, FILE *fu, const qvz_options_t *opts,
                      qvz_result_t *res) {
    return encode_buffer(text, bytes, fout, fu, opts, res, NULL, 0, NULL);
}

static void job_drop_codebooks(qvz_job_t *job) {
    if (job->cbs) { for (uint32_t j = 0; j < job->K; ++j) qvz_codebook_free(job->cbs[j]); free(job->cbs); job->cbs = NULL; }
    job->deferred_decisions = 0;
}

/* the operating point of a sweep: what -f / -r / -d / -D set (main.c:240-297); nothing the clustering or the counters read */
static int job_set_point(qvz_job_t *job, const qvz_sweep_point_t *pt) {
    job->opts.mode = pt->mode;
    job->opts.ratio = pt->ratio;
    job->opts.distortion = pt->distortion;
    job->opts.dist_file = pt->dist_file;
    if (pt->distortion == QVZ_DIST_CUSTOM) {
        if (qvz_distortion_matrix_file(pt->dist_file, job->h_dist)) return qvz_job_fail(QVZ_E_ARG, "%s", qvz_hip_last_error());
    } else if (qvz_distortion_matrix(pt->distortion, job->h_dist)) {
        return qvz_job_fail(QVZ_E_ARG, "%s", qvz_hip_last_error());
    }
    return 0;
}

static FILE *sweep_open(const char *path, uint32_t idx, int keep_all, const char *how) {
    char name[4096];
    if (!keep_all) return fopen(path, how);
    if ((size_t)snprintf(name, sizeof(name), "%s.%02u", path, idx) >= sizeof(name)) return NULL;
    return fopen(name, how);
}

static int encode_buffer(const uint8_t *text, uint64_t bytes, FILE *fout, FILE *fu, const qvz_options_t *opts,
                         qvz_result_t *res, const char *shard_prefix, uint32_t shards, const sweep_t *sw) {
    qvz_job_t job;
    qvz_result_t R;
    memset(&R, 0, sizeof(R));
    double t_start = now();
    uint8_t *fq_rows = NULL;                                      /* FASTQ input: the quality lines, in load_file's layout */
    const uint32_t npoints = sw ? sw->n : 1;
    if (opts->input_format == QVZ_INPUT_FASTQ || (opts->input_format == QVZ_INPUT_AUTO && qvz_fastq_detect(text, bytes))) {
        uint64_t reads = 0;
        uint32_t columns = 0;
        int frc = qvz_fastq_qualities(text, bytes, &fq_rows, &reads, &columns);
        if (frc) return frc;
        if (opts->verbose) printf("FASTQ input: %llu reads of %u quality values\n", (unsigned long long)reads, columns);
        text = fq_rows;
        bytes = reads * ((uint64_t)columns + 1);
    }
    int rc = qvz_job_init(&job, text, bytes, 0, 0, opts);
    if (rc) { free(fq_rows); return rc; }
    /* pages of the big outputs, made by a helper thread beside everything up to their egress (host/qvz_egress.c): the shard
     * set first - its size is not known yet: 2 bits per symbol are assumed, what is written beyond goes through pwrite and
     * what is not needed is given back - then the -u text, N (C + 1) bytes (a sweep: per point, when its files are opened) */
    if (!sw && shards && fout && !shard_prefix) qvz_prealloc_start(fout, job.N * (uint64_t)job.C / 4);
    if (!sw && shards && fu) qvz_prealloc_start(fu, job.N * ((uint64_t)job.C + 1));
    job.want_text = sw ? sw->u_path != NULL : fu != NULL;
    if (shards) qvz_job_start_workspace(&job, shards);                         /* the coder's record workspace and the quantizer's planes, beside the upload */
    if ((rc = qvz_job_upload(&job))) goto done;
    if ((rc = qvz_job_kmeans(&job, 0))) goto done;
    if (opts->verbose) printf("Clustering took %.4f seconds\n", job.t_cluster);                      /* main.c:56-58 */
    if ((rc = qvz_job_stats(&job))) goto done;
    for (uint32_t p = 0; p < npoints; ++p) {
        uint64_t head_bytes = 0;
        if (sw) {                                                               /* the point's flags and files; the shared phases count for the first */
            if (p) { t_start = now(); job.t_upload = job.t_cluster = job.t_stats = 0.0; }
            job_drop_codebooks(&job);
            if ((rc = job_set_point(&job, &sw->points[p]))) goto done;
            if (sw->out_path && !shard_prefix) {
                fout = sweep_open(sw->out_path, p, sw->keep_all, "wb");        /* main.c:74 */
                if (!fout) { rc = qvz_job_fail(QVZ_E_ARG, "Unable to open output file"); goto done; }
            }
            if (sw->u_path) {
                fu = sweep_open(sw->u_path, p, sw->keep_all, "w");              /* main.c:81 */
                if (!fu) { rc = qvz_job_fail(QVZ_E_ARG, "Unable to open uncompressed file"); goto done; }
            }
            job.keep_resident = p + 1 < npoints;                                /* the matrix and the draws serve the next point too */
        }
        if (sw && shards && fout && !shard_prefix) qvz_prealloc_start(fout, job.N * (uint64_t)job.C / 4);
        if (sw && shards && fu) qvz_prealloc_start(fu, job.N * ((uint64_t)job.C + 1));
        if (shards && (rc = qvz_job_start_draws(&job))) goto done;              /* the WELL generator runs beside the codebooks */
        if ((rc = qvz_job_codebooks(&job))) goto done;
        if (opts->verbose) printf("Stats and codebook generation took %.4f seconds\n", job.t_stats + job.t_codebook);   /* main.c:66-67 */
        if (shards && !shard_prefix) {                                          /* one shard-set file */
            if ((rc = qvz_job_compress_shardset(&job, fout, fu, shards, &R.stream_bytes, &R.file_bytes, &R.distortion))) goto done;
        } else if (shards) {
            if ((rc = qvz_job_compress_sharded(&job, shard_prefix, fu, shards, &R.stream_bytes, &R.distortion))) goto done;
            head_bytes = qvz_shard_file_image(NULL, job.K, job.C, 0, (const qvz_codebook_t *const *)job.cbs, job.well_state, NULL, 0) - 128;
            R.file_bytes = (uint64_t)shards * (head_bytes + 128) + R.stream_bytes;
        } else {
            if ((rc = qvz_job_write_codebooks(&job, fout, &head_bytes))) goto done;
            if ((rc = qvz_job_compress(&job, fout, fu, &R.stream_bytes, &R.distortion))) goto done;
            R.file_bytes = head_bytes + 128 + R.stream_bytes;
        }
        R.lines = job.N;
        R.columns = job.C;
        R.kmeans_iterations = job.kmeans_iterations;
        R.deferred_decisions = job.deferred_decisions;
        R.t_upload = job.t_upload; R.t_cluster = job.t_cluster; R.t_stats = job.t_stats; R.t_codebook = job.t_codebook;
        R.t_encode = job.t_encode;
        R.t_quantize = job.t_quantize; R.t_coder = job.t_coder; R.t_egress = job.t_egress;
        R.t_total = now() - t_start;
        if (opts->verbose) {                                                       /* main.c:96-121 */
            switch (job.opts.distortion) {
                case QVZ_DIST_MANHATTAN: printf("L1 distortion: %f\n", R.distortion); break;
                case QVZ_DIST_MSE: printf("MSE distortion: %f\n", R.distortion); break;
                case QVZ_DIST_LORENTZ: printf("log(1+L1) distortion: %f\n", R.distortion); break;
                case QVZ_DIST_CUSTOM: printf("Custom distortion: %f\n", R.distortion); break;          /* main.c:110-112 */
                default: break;
            }
            printf("Lines: %llu\n", (unsigned long long)R.lines);
            printf("Columns: %u\n", R.columns);
            printf("Total bytes used: %llu\n", (unsigned long long)R.stream_bytes);
            printf("Encoding took %.4f seconds.\n", R.t_total);                    /* prints the TOTAL timer, main.c:119 */
            printf("Total time elapsed: %.4f seconds.\n", R.t_total);
        }
        if (opts->stats)                                                           /* main.c:124-126 */
            printf("rate, %.4f, distortion, %.4f, time, %.4f, size, %llu \n", (R.stream_bytes * 8.) / ((double)R.lines * R.columns),
                   R.distortion, R.t_total, (unsigned long long)R.stream_bytes);
        if (sw) {
            if (sw->res) sw->res[p] = R;
            qvz_prealloc_finish(fu);
            if (shards && !shard_prefix) qvz_prealloc_finish(fout);
            if (fu) { fclose(fu); fu = NULL; }
            if (fout) { fclose(fout); fout = NULL; }
        }
    }
done:
    qvz_prealloc_finish(fu);
    if (shards && !shard_prefix) qvz_prealloc_finish(fout);
    if (sw) { if (fu) fclose(fu); if (fout) fclose(fout); }                    /* a point that failed half-way */
    qvz_job_release(&job);
    free(fq_rows);
    if (res) *res = R;
    return rc;
}

uint32_t qvz_sweep_parse(const char *text, const qvz_options_t *opts, qvz_sweep_point_t *points, uint32_t max) {
    uint32_t n = 0;
    if (!text || !opts || !points) return 0;
    if (strcmp(text, "rd") == 0) {                                              /* generate_rd.sh:5-8: comp = idx * 0.05 through bc -l, -c 1 -f $comp */
        if (max < 20) return 0;
        for (uint32_t idx = 0; idx < 20; ++idx) {
            char comp[16];
            snprintf(comp, sizeof(comp), ".%02u", idx * 5);                    /* "0", ".05", ".10" ... as bc prints them; atof as main.c:241 */
            points[idx].mode = QVZ_MODE_RATIO;
            points[idx].ratio = idx ? atof(comp) : atof("0");
            points[idx].distortion = opts->distortion;
            points[idx].dist_file = opts->dist_file;
        }
        return 20;
    }
    if ((text[0] != 'f' && text[0] != 'r') || text[1] != ':') return 0;
    const char *q = text + 2;
    while (*q) {
        char *end;
        const double v = strtod(q, &end);                                       /* atof's conversion (main.c:241,247) */
        if (end == q || n == max) return 0;
        points[n].mode = text[0] == 'f' ? QVZ_MODE_RATIO : QVZ_MODE_FIXED;
        points[n].ratio = v;
        points[n].distortion = opts->distortion;
        points[n].dist_file = opts->dist_file;
        ++n;
        q = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') return 0;
    }
    return n;
}

int qvz_encode_file_sweep(const char *in_path, const char *out_path, const qvz_options_t *opts, const qvz_sweep_point_t *points,
                          uint32_t npoints, uint32_t shards, int keep_all, qvz_result_t *res) {
    if (!in_path || !out_path || !opts || !points || !npoints || npoints > 100) return qvz_job_fail(QVZ_E_ARG, "a sweep needs 1..100 points");
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return qvz_job_fail(QVZ_E_ARG, "load_file returned error: 1");       /* LF_ERROR_NOT_FOUND, lines.h:18 */
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return qvz_job_fail(QVZ_E_DATA, "empty or unreadable input"); }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);   /* lines.c:64 */
    if (map == MAP_FAILED) { close(fd); return qvz_job_fail(QVZ_E_NOMEM, "mmap failed"); }
    const sweep_t sw = {points, npoints, out_path, opts->uncompressed_name, keep_all, res};
    const int rc = encode_buffer((const uint8_t *)map, (uint64_t)st.st_size, NULL, NULL, opts, NULL, NULL, shards, &sw);
    munmap(map, (size_t)st.st_size);
    close(fd);
    return rc;
}

int qvz_encode_file(const char *in_path, const char *out_path, const qvz_options_t *opts, qvz_result_t *res) {
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return qvz_job_fail(QVZ_E_ARG, "load_file returned error: 1");       /* LF_ERROR_NOT_FOUND, lines.h:18 */
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return qvz_job_fail(QVZ_E_DATA, "empty or unreadable input"); }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);   /* lines.c:64 */
    if (map == MAP_FAILED) { close(fd); return qvz_job_fail(QVZ_E_NOMEM, "mmap failed"); }
    FILE *fout = fopen(out_path, "wb");                                         /* main.c:74 */
    if (!fout) { munmap(map, (size_t)st.st_size); close(fd); return qvz_job_fail(QVZ_E_ARG, "Unable to open output file"); }
    FILE *fu = NULL;
    if (opts->uncompressed_name) {
        fu = fopen(opts->uncompressed_name, "w");                               /* main.c:81 */
        if (!fu) { fclose(fout); munmap(map, (size_t)st.st_size); close(fd); return qvz_job_fail(QVZ_E_ARG, "Unable to open uncompressed file"); }
    }
    const int rc = qvz_encode_buffer((const uint8_t *)map, (uint64_t)st.st_size, fout, fu, opts, res);
    if (fu) fclose(fu);
    fclose(fout);
    munmap(map, (size_t)st.st_size);
    close(fd);
    return rc;
}

int qvz_encode_file_sharded(const char *in_path, const char *out_prefix, const qvz_options_t *opts, uint32_t shards,
                            qvz_result_t *res) {
    if (!shards) return qvz_job_fail(QVZ_E_ARG, "shards must be positive");
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return qvz_job_fail(QVZ_E_ARG, "load_file returned error: 1");
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return qvz_job_fail(QVZ_E_DATA, "empty or unreadable input"); }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { close(fd); return qvz_job_fail(QVZ_E_NOMEM, "mmap failed"); }
    FILE *fu = NULL;
    if (opts->uncompressed_name) {
        fu = fopen(opts->uncompressed_name, "w");
        if (!fu) { munmap(map, (size_t)st.st_size); close(fd); return qvz_job_fail(QVZ_E_ARG, "Unable to open uncompressed file"); }
    }
    const int rc = encode_buffer((const uint8_t *)map, (uint64_t)st.st_size, NULL, fu, opts, res, out_prefix, shards, NULL);
    if (fu) fclose(fu);
    munmap(map, (size_t)st.st_size);
    close(fd);
    return rc;
}

int qvz_encode_file_shardset(const char *in_path, const char *out_path, const qvz_options_t *opts, uint32_t shards,
                             qvz_result_t *res) {
    if (!shards) return qvz_job_fail(QVZ_E_ARG, "shards must be positive");
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return qvz_job_fail(QVZ_E_ARG, "load_file returned error: 1");
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return qvz_job_fail(QVZ_E_DATA, "empty or unreadable input"); }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { close(fd); return qvz_job_fail(QVZ_E_NOMEM, "mmap failed"); }
    FILE *fout = fopen(out_path, "wb");
    if (!fout) { munmap(map, (size_t)st.st_size); close(fd); return qvz_job_fail(QVZ_E_ARG, "Unable to open output file"); }
    FILE *fu = NULL;
    if (opts->uncompressed_name) {
        fu = fopen(opts->uncompressed_name, "w");
        if (!fu) { fclose(fout); munmap(map, (size_t)st.st_size); close(fd); return qvz_job_fail(QVZ_E_ARG, "Unable to open uncompressed file"); }
    }
    const int rc = encode_buffer((const uint8_t *)map, (uint64_t)st.st_size, fout, fu, opts, res, NULL, shards, NULL);
    if (fu) fclose(fu);
    fclose(fout);
    munmap(map, (size_t)st.st_size);
    close(fd);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------- */
/* The hot path over a matrix that is in HBM already (include/qvz_encode.h): the stages of encode() (main.c:42-91) without the
 * text on either side - no ingest, no files. What bench.py times; what a caller that holds the reads on the device calls. */
struct qvz_resident {
    int device;
    qvz_pool_t pool;
    qvz_job_t job;               /* the last encode's job: its codebooks, labels and products are the caller's to read until the next */
    int have_job;
    uint32_t shards;
    double model_ms, coder_ms;
};

int qvz_resident_open(int device, qvz_resident_t **out) {
    if (!out) return qvz_job_fail(QVZ_E_ARG, "null argument");
    qvz_resident_t *r = (qvz_resident_t *)calloc(1, sizeof(*r));
    if (!r) return qvz_job_fail(QVZ_E_NOMEM, "host alloc");
    r->device = device;
    if (qvz_hip_set_device(device)) { free(r); return qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); }
    (void)qvz_hip_codebook_warmup();                              /* once per process: the device's log2 against this host's libm */
    (void)qvz_hip_encode_planes_warmup();
    *out = r;
    return 0;
}

static void resident_drop_job(qvz_resident_t *r) {
    if (r->have_job) { qvz_job_release(&r->job); r->have_job = 0; }
}

void qvz_resident_close(qvz_resident_t *r) {
    if (!r) return;
    (void)qvz_hip_set_device(r->device);
    resident_drop_job(r);
    for (int i = 1; i < QVZ_POOL_ROLES; ++i) qvz_hip_free(r->pool.ptr[i]);
    if (r->pool.well_stream) qvz_hip_stream_destroy(r->pool.well_stream);
    for (int i = 0; i < 8; ++i) if (r->pool.cb_stream[i]) qvz_hip_stream_destroy(r->pool.cb_stream[i]);
    free(r);
}

int qvz_resident_encode(qvz_resident_t *r, const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint64_t first_read,
                        uint64_t n_total, const qvz_options_t *opts, uint32_t shards, int want_qv, const uint64_t *centroid_lines,
                        qvz_allreduce_fn allreduce, void *allreduce_ctx, qvz_result_t *res) {
    int rc = 0;
    qvz_result_t R;
    memset(&R, 0, sizeof(R));
    if (!r || !d_sym || !opts || !shards) return qvz_job_fail(QVZ_E_ARG, "null argument");
    if (ld != qvz_hip_ld(n)) return qvz_job_fail(QVZ_E_ARG, "the matrix's leading dimension must be qvz_hip_ld(n) = %llu", (unsigned long long)qvz_hip_ld(n));
    if (n_total < first_read + n) return qvz_job_fail(QVZ_E_ARG, "reads %llu + %llu beyond the file's %llu", (unsigned long long)first_read, (unsigned long long)n, (unsigned long long)n_total);
    if (!allreduce && (first_read || n_total != n)) return qvz_job_fail(QVZ_E_ARG, "a part of a file needs the all-reduce of its statistics");
    const double t_start = now();
    qvz_job_t *job = &r->job;
    uint8_t *rows = NULL;
    uint64_t *d_idx = NULL;
    uint32_t *d_cent = NULL, *h_cent = NULL;
    if (qvz_hip_set_device(r->device)) return qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error());
    resident_drop_job(r);                                         /* the previous encode's products end here */
    qvz_options_t o = *opts;
    o.device = r->device;
    o.uncompressed_name = NULL;
    if (o.seed_mode == QVZ_SEED_TIME && allreduce) return qvz_job_fail(QVZ_E_ARG, "the parts of one file need ONE WELL state: seed_mode debug or explicit");
    const uint32_t K = (uint32_t)o.clusters;
    const uint64_t stride = (uint64_t)c + 1;
    if (o.clusters < 1 || o.clusters > 255) return qvz_job_fail(QVZ_E_ARG, "clusters must be 1..255");
    /* the centroid reads (cluster.c:198-204): their numbers from libc rand() in a fresh process's state unless the caller names
     * them, their bytes from the part that holds each (a sum over the parts: everybody else contributes zeros) */
    uint64_t lines[256];
    if (centroid_lines) memcpy(lines, centroid_lines, K * sizeof(uint64_t));
    else {
        uint32_t block_count = (uint32_t)(n_total / QVZ_LINES_PER_BLOCK);
        if ((uint64_t)block_count * QVZ_LINES_PER_BLOCK != n_total) block_count += 1;
        srand(1);                                                 /* a process that has just started: where the reference binary draws them */
        for (uint32_t j = 0; j < K; ++j) {
            const uint32_t block_id = (uint32_t)rand() % block_count;
            const uint64_t left = n_total - (uint64_t)block_id * QVZ_LINES_PER_BLOCK;
            lines[j] = (uint64_t)block_id * QVZ_LINES_PER_BLOCK + (uint32_t)rand() % (left > QVZ_LINES_PER_BLOCK ? (uint32_t)QVZ_LINES_PER_BLOCK : (uint32_t)left);
        }
    }
    rows = (uint8_t *)malloc((size_t)K * stride);
    h_cent = (uint32_t *)malloc((size_t)K * c * sizeof(uint32_t));
    if (!rows || !h_cent) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto fail; }
    {
        uint64_t idx[256];
        for (uint32_t j = 0; j < K; ++j) idx[j] = lines[j] >= first_read && lines[j] < first_read + n ? lines[j] - first_read : ~0ull;
        if (qvz_hip_malloc((void **)&d_idx, K * sizeof(uint64_t)) || qvz_hip_malloc((void **)&d_cent, (size_t)K * c * sizeof(uint32_t)) ||
            qvz_hip_copy_h2d(d_idx, idx, K * sizeof(uint64_t), NULL) || qvz_hip_gather_reads(d_sym, ld, c, d_idx, K, d_cent, NULL) ||
            (allreduce && allreduce(allreduce_ctx, d_cent, (uint64_t)K * c, 4, NULL)) ||
            qvz_hip_copy_d2h(h_cent, d_cent, (size_t)K * c * sizeof(uint32_t), NULL) || qvz_hip_stream_sync(NULL)) {
            rc = qvz_job_fail(QVZ_E_HIP, "centroid reads: %s", qvz_hip_last_error());
            goto fail;
        }
        for (uint32_t j = 0; j < K; ++j) {
            for (uint32_t s = 0; s < c; ++s) rows[(size_t)j * stride + s] = (uint8_t)(h_cent[(size_t)j * c + s] + 33);
            rows[(size_t)j * stride + c] = '\n';
        }
    }
    const double t_setup = now();
    if ((rc = qvz_job_init_part(job, NULL, n, c, &o, first_read, rows))) goto fail;
    r->have_job = 1;
    job->pool = &r->pool;
    job->d_sym = (uint8_t *)d_sym;                                /* (read only) */
    job->sym_borrowed = 1;
    job->uploaded = 1;
    job->keep_device = 1;
    job->want_text = want_qv != 0;
    job->allreduce = allreduce;
    job->allreduce_ctx = allreduce_ctx;
    if (qvz_hip_malloc((void **)&job->d_errsum, 1025 * sizeof(double))) { rc = qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); goto fail; }
    if (job_malloc(job, (void **)&job->d_labels, ld, QVZ_POOL_LABELS) || qvz_hip_memset(job->d_labels, 0, ld, NULL)) { rc = qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); goto fail; }
    if ((rc = qvz_job_kmeans(job, 0))) goto fail;
    if ((rc = qvz_job_stats(job))) goto fail;
    if ((rc = qvz_job_start_draws(job))) goto fail;               /* the WELL generator runs beside the codebooks */
    if ((rc = qvz_job_codebooks(job))) goto fail;
    {   /* the coder's record workspace: everything at once if it fits, 96 GiB at most, 24 GiB of what is free left alone */
        uint64_t free_b = 0, want = qvz_hip_encode_planes_work_bytes((n + shards - 1) / shards + 4, c, K, shards);
        if (want > (96ull << 30)) want = 96ull << 30;
        if (r->pool.bytes[QVZ_POOL_WORK] < want && qvz_hip_mem_info(&free_b, NULL) == QVZ_OK) {
            const uint64_t planes = (r->pool.ptr[QVZ_POOL_QI] ? 0 : (uint64_t)c * ld * (want_qv ? 3 : 2) + n * 8) + (r->pool.ptr[QVZ_POOL_DRAWS] ? 0 : (uint64_t)c * ld);
            const uint64_t have = r->pool.bytes[QVZ_POOL_WORK];
            const uint64_t room = free_b + have > planes + (24ull << 30) ? free_b + have - planes - (24ull << 30) : 0;
            if (want > room) want = room;
        }
        if (want >= (1ull << 30) && job_malloc(job, &job->d_work, (size_t)want, QVZ_POOL_WORK) == QVZ_OK) { job->work_bytes = r->pool.bytes[QVZ_POOL_WORK]; job->work_shards = shards; }
        else job->d_work = NULL;
    }
    if ((rc = compress_sharded_impl(job, NULL, NULL, NULL, shards, &R.stream_bytes, NULL, &R.distortion))) goto fail;
    qvz_hip_encode_planes_last_ms(&r->model_ms, &r->coder_ms);
    r->shards = shards;
    if (allreduce) {                                              /* the file's distortion: the parts' sums, then / N (qv_compressor.c:139-140) */
        if (qvz_hip_copy_h2d(job->d_errsum, &job->r_errsum, sizeof(double), NULL) || allreduce(allreduce_ctx, job->d_errsum, 1, -8, NULL) ||
            qvz_hip_copy_d2h(&job->r_errsum, job->d_errsum, sizeof(double), NULL) || qvz_hip_stream_sync(NULL)) {
            rc = qvz_job_fail(QVZ_E_HIP, "distortion sum: %s", qvz_hip_last_error());
            goto fail;
        }
        R.distortion = job->r_errsum / ((double)n_total);
    }
    R.lines = n; R.columns = c;
    R.file_bytes = 0;
    R.kmeans_iterations = job->kmeans_iterations;
    R.deferred_decisions = job->deferred_decisions;
    R.t_cluster = job->t_cluster; R.t_stats = job->t_stats; R.t_codebook = job->t_codebook; R.t_encode = job->t_encode;
    R.t_quantize = job->t_quantize; R.t_coder = job->t_coder; R.t_egress = job->t_egress;
    R.t_total = now() - t_start;
    if (getenv("QVZ_TIMING"))
        fprintf(stderr, "qvz_timing: resident encode: before the stages %.4f s (previous job's release, centroid reads), stages %.4f (cluster %.4f stats %.4f "
                        "codebook %.4f quantize %.4f coder %.4f after the coder %.4f), total %.4f\n", t_setup - t_start, R.t_total - (t_setup - t_start),
                R.t_cluster, R.t_stats, R.t_codebook, R.t_quantize, R.t_coder, R.t_egress, R.t_total);
fail:
    job->centroid_rows = NULL;
    free(rows); free(h_cent);
    qvz_hip_free(d_idx); qvz_hip_free(d_cent);
    if (rc) resident_drop_job(r);
    if (res) *res = R;
    return rc;
}

int qvz_resident_products(const qvz_resident_t *r, qvz_resident_products_t *out) {
    if (!r || !out) return qvz_job_fail(QVZ_E_ARG, "null argument");
    if (!r->have_job || !r->job.r_sizes) return qvz_job_fail(QVZ_E_ARG, "no encode has completed on this handle");
    memset(out, 0, sizeof(*out));
    out->d_coded = r->job.r_out;
    out->cap_per_stream = r->job.r_cap;
    out->d_streams = r->job.r_packed;
    out->h_sizes = r->job.r_sizes;
    out->shards = r->shards;
    out->stream_bytes = r->job.r_total;
    out->d_qv = r->job.r_qv;
    out->d_labels = r->job.d_labels;
    out->codebooks = (const qvz_codebook_t *const *)r->job.cbs;
    out->clusters = r->job.K;
    memcpy(out->well_state, r->job.well_state, sizeof(out->well_state));
    out->model_kernels_ms = r->model_ms;
    out->coder_kernels_ms = r->coder_ms;
    out->kmeans_kernels_ms = r->job.k_kmeans_ms;
    out->pmf_kernel_ms = r->job.k_pmf_ms;
    out->quantize_kernel_ms = r->job.k_quantize_ms;
    return 0;
}

int qvz_resident_pack(qvz_resident_t *r, const uint8_t **d_packed) {
    int rc = 0;
    if (!r || !r->have_job || !r->job.r_sizes) return qvz_job_fail(QVZ_E_ARG, "no encode has completed on this handle");
    qvz_job_t *job = &r->job;
    if (!job->r_packed) {
        uint8_t *dst = NULL;
        if (qvz_hip_set_device(r->device)) return qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error());
        /* into the front of the record workspace when the stream area lies behind it there (the coder has finished with it) */
        if (job->r_out_in_work && job->d_work && job->r_total <= (uint64_t)(job->r_out - (uint8_t *)job->d_work)) dst = (uint8_t *)job->d_work;
        else HIP(job_malloc(job, (void **)&dst, job->r_total ? job->r_total : 1, QVZ_POOL_PACKED));
        HIP(qvz_hip_gather_streams(job->r_out, job->r_cap, job->r_sizes, r->shards, dst, NULL));
        HIP(qvz_hip_stream_sync(NULL));
        job->r_packed = dst;
    }
    if (d_packed) *d_packed = job->r_packed;
done:
    return rc;
}
