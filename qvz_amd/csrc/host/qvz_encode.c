/* qvz_encode.c - host C driver of the encode path (see include/qvz_encode.h). Counterpart of encode() in the
 * reference's src/main.c:18-127; prints the same -v / -s lines. All heavy phases are HIP kernel calls through
 * the C-ABI; this file owns file I/O, the k-means loop condition, libc rand() semantics and the serial coder. */
#define _GNU_SOURCE
#include "qvz_encode.h"

#include <fcntl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "qvz_codebook.h"
#include "qvz_coder.h"

#define A QVZ_A
#define LINES_PER_BLOCK 1000000ull   /* MAX_LINES_PER_BLOCK, reference include/lines.h:12 */
#define MAX_COLUMNS 1022u            /* MAX_READS_PER_LINE, reference include/lines.h:13 */

static __thread char g_err[512] = "";
const char *qvz_encode_last_error(void) { return g_err; }

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP(call)                                                                                   \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_) { rc = fail(rc_, "%s: %s", #call, qvz_hip_last_error()); goto done; }              \
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

/* One set of per-chunk buffers; two sets alternate so the GPU prepares chunk i+1 while the host codes chunk i */
typedef struct {
    void *stream;
    uint8_t *d_qv, *d_qi, *d_st, *d_text;
    uint16_t *d_feed;
    uint32_t *d_words;
    double *d_err;
    int32_t *d_status;
    uint16_t *h_feed;
    uint8_t *h_labels, *h_text;
    double *h_err;
    int32_t *h_status;
    uint64_t r0, n;               /* reads [r0, r0+n) currently in flight in this set */
} chunk_t;

static void chunk_free(chunk_t *ck) {
    qvz_hip_free(ck->d_qv); qvz_hip_free(ck->d_qi); qvz_hip_free(ck->d_st); qvz_hip_free(ck->d_text);
    qvz_hip_free(ck->d_feed); qvz_hip_free(ck->d_words); qvz_hip_free(ck->d_err); qvz_hip_free(ck->d_status);
    qvz_hip_free_host(ck->h_feed); qvz_hip_free_host(ck->h_labels); qvz_hip_free_host(ck->h_text);
    qvz_hip_free_host(ck->h_err); qvz_hip_free_host(ck->h_status);
    if (ck->stream) qvz_hip_stream_destroy(ck->stream);
    memset(ck, 0, sizeof(*ck));
}

int qvz_encode_buffer(const uint8_t *text, uint64_t bytes, FILE *fout, FILE *fu, const qvz_options_t *opts,
                      qvz_result_t *res) {
    int rc = 0;
    qvz_result_t R;
    memset(&R, 0, sizeof(R));
    const double t_start = now();

    /* load_file (lines.c:44-54): columns from the first line, lines = size / (columns+1) */
    const uint8_t *nl = (const uint8_t *)memchr(text, '\n', bytes < MAX_COLUMNS + 2 ? bytes : MAX_COLUMNS + 2);
    if (!nl) return fail(QVZ_E_DATA, "first line longer than %u symbols or missing newline (LF_ERROR_TOO_LONG)", MAX_COLUMNS);
    const uint32_t C = (uint32_t)(nl - text);
    if (C == 0) return fail(QVZ_E_DATA, "empty first line");
    const uint64_t N = bytes / ((uint64_t)C + 1);
    if (N == 0) return fail(QVZ_E_DATA, "no complete line in the input");
    if (N >= (1ull << 32)) return fail(QVZ_E_ARG, "more than 2^32-1 lines: the .q header and the u32 counters cannot hold it (codebook.c:482)");
    const uint32_t K = (uint32_t)opts->clusters;
    if (K < 1 || K > 255) return fail(QVZ_E_ARG, "clusters must be 1..255");
    const uint64_t stride = (uint64_t)C + 1;
    const uint64_t ld = qvz_hip_ld(N);

    uint8_t *d_sym = NULL, *d_labels = NULL, *d_means = NULL, *d_up = NULL;
    uint32_t *d_cnt = NULL, *d_counts = NULL, *d_bad = NULL;
    uint64_t *d_acc = NULL;
    double *d_moved = NULL, *d_dist = NULL;
    int32_t *d_kstat = NULL;
    uint16_t *d_ct = NULL, *d_qs = NULL, *h_ct = NULL, *h_qs = NULL;
    qvz_codebook_t **cbs = NULL;
    qvz_coder_t *coder = NULL;
    chunk_t ck[2];
    memset(ck, 0, sizeof(ck));
    double h_dist[A * A];

    /* Centroid picks (cluster.c:198-204) use libc rand() in whatever state the process is in. They are drawn
     * BEFORE the first HIP call: initialising the HIP runtime consumes rand() values of its own, which would
     * shift the sequence away from what the reference binary sees in a fresh process. */
    uint64_t cent_lines[256];
    {
        uint32_t block_count = (uint32_t)(N / LINES_PER_BLOCK);               /* lines.c:93-96 */
        if ((uint64_t)block_count * LINES_PER_BLOCK != N) block_count += 1;
        for (uint32_t j = 0; j < K; ++j) {
            const uint32_t block_id = (uint32_t)rand() % block_count;
            const uint64_t left = N - (uint64_t)block_id * LINES_PER_BLOCK;
            const uint32_t count = left > LINES_PER_BLOCK ? (uint32_t)LINES_PER_BLOCK : (uint32_t)left;
            const uint32_t line_id = (uint32_t)rand() % count;
            cent_lines[j] = (uint64_t)block_id * LINES_PER_BLOCK + line_id;
            if (opts->verbose) printf("Chose block %d, line %d.\n", block_id, line_id);
        }
    }

    HIP(qvz_hip_set_device(opts->device));
    HIP(qvz_distortion_matrix(opts->distortion, h_dist));

    /* ---- upload + transpose: replaces the row-major pointer view of load_file ---------------------------- */
    const uint64_t up_reads = 1u << 21;   /* ~2M reads per staging copy */
    HIP(qvz_hip_malloc((void **)&d_sym, (size_t)C * ld));
    HIP(qvz_hip_malloc((void **)&d_labels, ld));
    HIP(qvz_hip_malloc((void **)&d_up, (size_t)(N < up_reads ? N : up_reads) * stride));
    HIP(qvz_hip_malloc((void **)&d_bad, sizeof(uint32_t)));
    HIP(qvz_hip_memset(d_bad, 0, sizeof(uint32_t), NULL));
    HIP(qvz_hip_memset(d_labels, 0, ld, NULL));                  /* calloc'd line_t.cluster, lines.c:116 */
    for (uint64_t r0 = 0; r0 < N; r0 += up_reads) {
        const uint64_t n = N - r0 < up_reads ? N - r0 : up_reads;
        HIP(qvz_hip_copy_h2d(d_up, text + r0 * stride, n * stride, NULL));
        HIP(qvz_hip_transpose_text(d_up, n, C, d_sym + r0, ld, d_bad, NULL));   /* r0 is a multiple of 256 */
        HIP(qvz_hip_stream_sync(NULL));
    }
    {
        uint32_t bad = 0;
        HIP(qvz_hip_copy_d2h(&bad, d_bad, sizeof(bad), NULL));
        HIP(qvz_hip_stream_sync(NULL));
        if (bad) { rc = fail(QVZ_E_DATA, "%u input bytes outside '!'..'h' or lines without '\\n' (undefined behaviour in the reference, pmf.c:372-381)", bad); goto done; }
    }
    qvz_hip_free(d_up); d_up = NULL;
    R.t_upload = now() - t_start;

    /* ---- k-means (cluster.c:192-244) ------------------------------------------------------------------ */
    const double t_cluster0 = now();
    {
        uint8_t *h_means = (uint8_t *)malloc((size_t)K * C);
        double *h_moved = (double *)malloc((K + 1) * sizeof(double));
        if (!h_means || !h_moved) { free(h_means); free(h_moved); rc = fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        for (uint32_t j = 0; j < K; ++j) {
            const uint8_t *row = text + cent_lines[j] * stride;
            for (uint32_t s = 0; s < C; ++s) h_means[(size_t)j * C + s] = (uint8_t)(row[s] - 33);
        }
        int krc = 0;
        if ((krc = qvz_hip_malloc((void **)&d_means, (size_t)K * C)) || (krc = qvz_hip_malloc((void **)&d_counts, K * sizeof(uint32_t))) ||
            (krc = qvz_hip_malloc((void **)&d_acc, (size_t)K * C * sizeof(uint64_t))) ||
            (krc = qvz_hip_malloc((void **)&d_moved, (K + 1) * sizeof(double))) ||
            (krc = qvz_hip_malloc((void **)&d_kstat, sizeof(int32_t))) ||
            (krc = qvz_hip_copy_h2d(d_means, h_means, (size_t)K * C, NULL)) || (krc = qvz_hip_memset(d_kstat, 0, sizeof(int32_t), NULL))) {
            free(h_means); free(h_moved);
            rc = fail(krc, "k-means setup: %s", qvz_hip_last_error());
            goto done;
        }
        uint32_t iter = 0;
        int loop = 1;
        while (iter < 1000 && loop) {                                          /* cluster.c:221, cluster.h:9 */
            int32_t kstat = 0;
            if ((krc = qvz_hip_memset(d_counts, 0, K * sizeof(uint32_t), NULL)) ||
                (krc = qvz_hip_memset(d_acc, 0, (size_t)K * C * sizeof(uint64_t), NULL)) ||
                (krc = qvz_hip_kmeans_assign(d_sym, N, C, ld, K, d_means, d_labels, d_counts, NULL)) ||
                (krc = qvz_hip_kmeans_accumulate(d_sym, d_labels, N, C, ld, K, d_acc, NULL)) ||
                (krc = qvz_hip_kmeans_update(d_acc, d_counts, K, C, d_means, d_moved, d_kstat, NULL)) ||
                (krc = qvz_hip_copy_d2h(h_moved, d_moved, (K + 1) * sizeof(double), NULL)) ||
                (krc = qvz_hip_copy_d2h(&kstat, d_kstat, sizeof(kstat), NULL)) || (krc = qvz_hip_stream_sync(NULL)))
                break;
            if (kstat) { krc = QVZ_E_EMPTY; break; }
            if (opts->verbose) {
                for (uint32_t j = 0; j < K; ++j) printf("Cluster %d moved %f.\n", j, h_moved[1 + j]);   /* cluster.c:126-127 */
                printf("\n");                                                  /* cluster.c:236-238 */
            }
            loop = h_moved[0] > opts->cluster_threshold;                       /* cluster.c:230-233 */
            iter += 1;
        }
        free(h_means); free(h_moved);
        if (krc == QVZ_E_EMPTY) { rc = fail(krc, "a cluster went empty during k-means (the reference divides by zero here, cluster.c:113)"); goto done; }
        if (krc) { rc = fail(krc, "k-means: %s", qvz_hip_last_error()); goto done; }
        if (opts->verbose) printf("\nTotal number of iterations: %d.\n", iter);   /* cluster.c:241-243 */
        R.kmeans_iterations = iter;
    }
    R.t_cluster = now() - t_cluster0;
    if (opts->verbose) printf("Clustering took %.4f seconds\n", R.t_cluster);     /* main.c:56-58 */

    /* ---- statistics + codebooks (main.c:61-67) -------------------------------------------------------- */
    const double t_stats0 = now();
    HIP(qvz_hip_malloc((void **)&d_cnt, (size_t)K * C * A * A * sizeof(uint32_t)));
    HIP(qvz_hip_memset(d_cnt, 0, (size_t)K * C * A * A * sizeof(uint32_t), NULL));
    HIP(qvz_hip_pmf_hist(d_sym, K > 1 ? d_labels : NULL, N, C, ld, K, d_cnt, NULL));
    HIP(qvz_hip_stream_sync(NULL));
    R.t_stats = now() - t_stats0;
    cbs = (qvz_codebook_t **)calloc(K, sizeof(*cbs));
    if (!cbs) { rc = fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    for (uint32_t j = 0; j < K; ++j) {
        cbs[j] = qvz_codebook_alloc(C);
        if (!cbs[j]) { rc = fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        HIP(qvz_hip_generate_codebook(d_cnt + (size_t)j * C * A * A, C, h_dist, opts->mode, opts->ratio, 0.0, cbs[j], NULL));
        R.deferred_decisions += qvz_hip_codebook_deferred_decisions();
    }
    R.t_codebook = now() - t_stats0 - R.t_stats;
    if (opts->verbose) printf("Stats and codebook generation took %.4f seconds\n", now() - t_stats0);   /* main.c:66-67 */

    /* ---- write_codebooks (codebook.c:474-498) + WELL state (qv_stream.c:70-89) ------------------------ */
    const double t_encode0 = now();
    uint64_t file_bytes = 0;
    {
        uint8_t hdr[9];
        qvz_codebook_write_header(hdr, K, C, N);
        if (fwrite(hdr, 1, 9, fout) != 9) { rc = fail(QVZ_E_ARG, "write failed"); goto done; }
        file_bytes += 9;
        for (uint32_t j = 0; j < K; ++j) {
            const size_t sz = qvz_codebook_write(NULL, cbs[j]);
            uint8_t *buf = (uint8_t *)malloc(sz);
            if (!buf) { rc = fail(QVZ_E_NOMEM, "host alloc"); goto done; }
            qvz_codebook_write(buf, cbs[j]);
            const size_t w = fwrite(buf, 1, sz, fout);
            free(buf);
            if (w != sz) { rc = fail(QVZ_E_ARG, "write failed"); goto done; }
            file_bytes += sz;
        }
    }
    uint32_t well_state[32];
    if (opts->seed_mode == QVZ_SEED_EXPLICIT) {
        memcpy(well_state, opts->well_state, sizeof(well_state));
    } else {
        srand((uint32_t)time(0));                                              /* qv_stream.c:77 (also in -DDEBUG builds) */
        for (int i = 0; i < 32; ++i) well_state[i] = opts->seed_mode == QVZ_SEED_DEBUG ? 0x55555555u : (uint32_t)rand();
    }
    if (fwrite(well_state, sizeof(uint32_t), 32, fout) != 32) { rc = fail(QVZ_E_ARG, "write failed"); goto done; }
    file_bytes += 128;

    /* ---- quantizer tables ----------------------------------------------------------------------------- */
    {
        const size_t ctn = (size_t)K * C * QVZ_CT_STRIDE, qsn = (size_t)K * C * QVZ_QS_STRIDE;
        h_ct = (uint16_t *)malloc(ctn * sizeof(uint16_t));
        h_qs = (uint16_t *)malloc(qsn * sizeof(uint16_t));
        if (!h_ct || !h_qs) { rc = fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        for (uint32_t j = 0; j < K; ++j)
            qvz_codebook_tables(cbs[j], h_ct + (size_t)j * C * QVZ_CT_STRIDE, h_qs + (size_t)j * C * QVZ_QS_STRIDE);
        HIP(qvz_hip_malloc((void **)&d_ct, ctn * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&d_qs, qsn * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&d_dist, sizeof(h_dist)));
        HIP(qvz_hip_copy_h2d(d_ct, h_ct, ctn * sizeof(uint16_t), NULL));
        HIP(qvz_hip_copy_h2d(d_qs, h_qs, qsn * sizeof(uint16_t), NULL));
        HIP(qvz_hip_copy_h2d(d_dist, h_dist, sizeof(h_dist), NULL));
        HIP(qvz_hip_stream_sync(NULL));
    }

    /* ---- quantize (GPU) + arithmetic coding (host), chunked and overlapped (qv_compressor.c:48-143) ---- */
    coder = qvz_coder_create(K, C, (const qvz_codebook_t *const *)cbs);
    if (!coder) { rc = fail(QVZ_E_NOMEM, "coder alloc"); goto done; }
    uint64_t chunk = opts->chunk_reads ? opts->chunk_reads : (1u << 20);
    chunk = (chunk + 255) & ~255ull;
    if (chunk > ((N + 255) & ~255ull)) chunk = (N + 255) & ~255ull;
    const uint64_t ldc = qvz_hip_ld(chunk);
    for (int b = 0; b < 2; ++b) {
        chunk_t *k_ = &ck[b];
        HIP(qvz_hip_stream_create(&k_->stream));
        HIP(qvz_hip_malloc((void **)&k_->d_qv, (size_t)C * ldc));
        HIP(qvz_hip_malloc((void **)&k_->d_qi, (size_t)C * ldc));
        HIP(qvz_hip_malloc((void **)&k_->d_st, (size_t)C * ldc));
        HIP(qvz_hip_malloc((void **)&k_->d_feed, chunk * C * sizeof(uint16_t)));
        HIP(qvz_hip_malloc((void **)&k_->d_words, (chunk * C / 4 + 8) * sizeof(uint32_t)));
        HIP(qvz_hip_malloc((void **)&k_->d_err, chunk * sizeof(double)));
        HIP(qvz_hip_malloc((void **)&k_->d_status, sizeof(int32_t)));
        HIP(qvz_hip_malloc_host((void **)&k_->h_feed, chunk * C * sizeof(uint16_t)));
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
        /* issue(): GPU side of one chunk on its own stream */
#define ISSUE(set)                                                                                               \
    do {                                                                                                         \
        chunk_t *q_ = &ck[set];                                                                                  \
        q_->r0 = next;                                                                                           \
        q_->n = N - next < chunk ? N - next : chunk;                                                             \
        const uint64_t fd_ = q_->r0 * C, w0_ = fd_ >> 2, w1_ = ((q_->r0 + q_->n) * C + 3) >> 2;                  \
        HIP(qvz_hip_memset(q_->d_status, 0, sizeof(int32_t), q_->stream));                                       \
        HIP(qvz_hip_well_words(well_state, w0_, w1_ - w0_, q_->d_words, q_->stream));                            \
        HIP(qvz_hip_quantize(d_sym + q_->r0, K > 1 ? d_labels + q_->r0 : NULL, q_->n, C, ld, K, d_ct, d_qs,      \
                             q_->d_words, fd_, d_dist, q_->d_qv, q_->d_qi, q_->d_st, qvz_hip_ld(q_->n), q_->d_err, \
                             q_->d_status, q_->stream));                                                         \
        HIP(qvz_hip_pack_feed(q_->d_qi, q_->d_st, q_->n, C, qvz_hip_ld(q_->n), q_->d_feed, q_->stream));          \
        HIP(qvz_hip_copy_d2h(q_->h_feed, q_->d_feed, q_->n * C * sizeof(uint16_t), q_->stream));                 \
        HIP(qvz_hip_copy_d2h(q_->h_err, q_->d_err, q_->n * sizeof(double), q_->stream));                         \
        HIP(qvz_hip_copy_d2h(q_->h_status, q_->d_status, sizeof(int32_t), q_->stream));                          \
        if (K > 1) HIP(qvz_hip_copy_d2h(q_->h_labels, d_labels + q_->r0, q_->n, q_->stream));                    \
        if (fu) {                                                                                                \
            HIP(qvz_hip_untranspose_text(q_->d_qv, q_->n, C, qvz_hip_ld(q_->n), q_->d_text, q_->stream));         \
            HIP(qvz_hip_copy_d2h(q_->h_text, q_->d_text, q_->n * stride, q_->stream));                           \
        }                                                                                                        \
        next += q_->n;                                                                                           \
    } while (0)

        HIP(qvz_hip_stream_sync(NULL));
        ISSUE(cur);
        while (ck[cur].n) {
            chunk_t *q = &ck[cur];
            HIP(qvz_hip_stream_sync(q->stream));
            ck[cur ^ 1].n = 0;
            if (next < N) ISSUE(cur ^ 1);      /* the GPU prepares the next chunk while this one is coded */
            if (*q->h_status) { rc = fail(QVZ_E_DEAD_CTX, "quantizer reached a context the codebook does not define (assert in the reference, codebook.c:164)"); goto done; }
            if (opts->verbose) {
                /* "Line: %dM" at the first line of every 1M-line block (qv_compressor.c:79-81) */
                for (uint64_t blk = (q->r0 + LINES_PER_BLOCK - 1) / LINES_PER_BLOCK; blk * LINES_PER_BLOCK < q->r0 + q->n; ++blk)
                    printf("Line: %dM\n", (int)blk);
            }
            if (qvz_coder_encode(coder, K > 1 ? q->h_labels : NULL, q->h_feed, q->n)) { rc = fail(QVZ_E_DATA, "coder feed out of range"); goto done; }
            for (uint64_t i = 0; i < q->n; ++i) distortion += q->h_err[i] / ((double)C);   /* qv_compressor.c:127 */
            if (fu && fwrite(q->h_text, 1, q->n * stride, fu) != q->n * stride) { rc = fail(QVZ_E_ARG, "write failed"); goto done; }
            {
                uint64_t sz;
                const uint8_t *data = qvz_coder_data(coder, &sz);
                if (sz && fwrite(data, 1, sz, fout) != sz) { rc = fail(QVZ_E_ARG, "write failed"); goto done; }
                qvz_coder_drop(coder, sz);
            }
            cur ^= 1;
        }
#undef ISSUE
        qvz_coder_finish(coder);                                               /* encoder_last_step, arith.c:99-116 */
        {
            uint64_t sz;
            const uint8_t *data = qvz_coder_data(coder, &sz);
            if (sz && fwrite(data, 1, sz, fout) != sz) { rc = fail(QVZ_E_ARG, "write failed"); goto done; }
            qvz_coder_drop(coder, sz);
        }
        R.distortion = distortion / ((double)N);                               /* qv_compressor.c:139-140 */
    }
    R.stream_bytes = qvz_coder_bytes(coder);
    R.file_bytes = file_bytes + R.stream_bytes;
    R.lines = N;
    R.columns = C;
    R.t_encode = now() - t_encode0;
    R.t_total = now() - t_start;

    if (opts->verbose) {                                                       /* main.c:96-121 */
        switch (opts->distortion) {
            case QVZ_DIST_MANHATTAN: printf("L1 distortion: %f\n", R.distortion); break;
            case QVZ_DIST_MSE: printf("MSE distortion: %f\n", R.distortion); break;
            case QVZ_DIST_LORENTZ: printf("log(1+L1) distortion: %f\n", R.distortion); break;
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

done:
    for (int b = 0; b < 2; ++b) chunk_free(&ck[b]);
    qvz_coder_destroy(coder);
    if (cbs) { for (uint32_t j = 0; j < K; ++j) qvz_codebook_free(cbs[j]); free(cbs); }
    free(h_ct); free(h_qs);
    qvz_hip_free(d_sym); qvz_hip_free(d_labels); qvz_hip_free(d_means); qvz_hip_free(d_up); qvz_hip_free(d_cnt);
    qvz_hip_free(d_counts); qvz_hip_free(d_bad); qvz_hip_free(d_acc); qvz_hip_free(d_moved); qvz_hip_free(d_dist);
    qvz_hip_free(d_kstat); qvz_hip_free(d_ct); qvz_hip_free(d_qs);
    if (res) *res = R;
    return rc;
}

int qvz_encode_file(const char *in_path, const char *out_path, const qvz_options_t *opts, qvz_result_t *res) {
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return fail(QVZ_E_ARG, "load_file returned error: 1");       /* LF_ERROR_NOT_FOUND, lines.h:18 */
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return fail(QVZ_E_DATA, "empty or unreadable input"); }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);   /* lines.c:64 */
    if (map == MAP_FAILED) { close(fd); return fail(QVZ_E_NOMEM, "mmap failed"); }
    FILE *fout = fopen(out_path, "wb");                                         /* main.c:74 */
    if (!fout) { munmap(map, (size_t)st.st_size); close(fd); return fail(QVZ_E_ARG, "Unable to open output file"); }
    FILE *fu = NULL;
    if (opts->uncompressed_name) {
        fu = fopen(opts->uncompressed_name, "w");                               /* main.c:81 */
        if (!fu) { fclose(fout); munmap(map, (size_t)st.st_size); close(fd); return fail(QVZ_E_ARG, "Unable to open uncompressed file"); }
    }
    const int rc = qvz_encode_buffer((const uint8_t *)map, (uint64_t)st.st_size, fout, fu, opts, res);
    if (fu) fclose(fu);
    fclose(fout);
    munmap(map, (size_t)st.st_size);
    close(fd);
    return rc;
}
