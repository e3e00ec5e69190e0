/* qvz_compat.c - the reference's entry points (include/qvz_compat.h) on top of the staged GPU job (qvz_job.h).
 * Each function does what its namesake does in the reference, as seen through the structs it is handed, and
 * reports errors the reference's way: a message on stdout and exit(1) (src/main.c:43-46,75-78). */
#define _GNU_SOURCE
#include "qvz_compat.h"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "qvz_codebook.h"
#include "qvz_job.h"

#define A QVZ_A
#define NOT_FOUND_U32 0xFFFFFFFFu          /* ALPHABET_SYMBOL_NOT_FOUND, pmf.h:9 */
#define INDEX_HINT 72                      /* ALPHABET_INDEX_SIZE_HINT, pmf.h:11 */
#define MAX_COLUMNS 1022u                  /* MAX_READS_PER_LINE, lines.h:13 */

static qvz_job_t g_job;                    /* device-side state of the one file being encoded */
static const struct quality_file_t *g_info;

static void die(const char *what) {
    printf("%s: %s\n", what, qvz_encode_last_error());
    exit(1);
}

static void index_alphabet(struct alphabet_t *a) {           /* alphabet_compute_index, pmf.c:365-382 */
    free(a->indexes);
    a->indexes = (uint32_t *)calloc(INDEX_HINT, sizeof(uint32_t));
    for (uint32_t i = 0; i < INDEX_HINT; ++i) a->indexes[i] = NOT_FOUND_U32;
    for (uint32_t i = 0; i < a->size; ++i) a->indexes[a->symbols[i]] = i;
}

static struct alphabet_t *alphabet_from(const uint8_t *symbols, uint32_t size) {
    struct alphabet_t *a = (struct alphabet_t *)calloc(1, sizeof(*a));
    a->size = size;
    a->symbols = (symbol_t *)calloc(size ? size : 1, sizeof(symbol_t));
    memcpy(a->symbols, symbols, size);
    index_alphabet(a);
    return a;
}

struct alphabet_t *alloc_alphabet(uint32_t size) {           /* pmf.c:11-24: symbols 0..size-1 */
    uint8_t syms[256];
    for (uint32_t i = 0; i < size && i < 256; ++i) syms[i] = (uint8_t)i;
    return alphabet_from(syms, size);
}

static struct pmf_t *new_pmf(const struct alphabet_t *alphabet) {   /* alloc_pmf, pmf.c:43-49 */
    struct pmf_t *p = (struct pmf_t *)calloc(1, sizeof(*p));
    p->alphabet = alphabet;
    p->pmf = (double *)calloc(alphabet->size, sizeof(double));
    p->counts = (uint32_t *)calloc(alphabet->size, sizeof(uint32_t));
    return p;
}

struct distortion_t *generate_distortion_matrix(uint8_t symbols, int type) {   /* distortion.c:30-45 */
    if (symbols != A) { printf("Only the 72-symbol alphabet is supported.\n"); exit(1); }
    struct distortion_t *d = (struct distortion_t *)calloc(1, sizeof(*d));
    d->symbols = symbols;
    d->distortion = (double *)calloc((size_t)A * A, sizeof(double));
    if (qvz_distortion_matrix(type, d->distortion)) {
        printf("Invalid distortion type %d specified.\n", type);                /* distortion.c:42-43 */
        exit(1);
    }
    return d;
}

struct distortion_t *gen_custom_distortion(uint8_t symbols, const char *filename) {     /* distortion.c:100-145 */
    (void)symbols;
    struct distortion_t *d = (struct distortion_t *)calloc(1, sizeof(*d));
    d->symbols = A;
    d->distortion = (double *)calloc((size_t)A * A, sizeof(double));
    if (qvz_distortion_matrix_file(filename, d->distortion)) {
        perror("Unable to open distortion definition file");
        exit(1);
    }
    return d;
}

uint32_t load_file(const char *path, struct quality_file_t *info, uint64_t max_lines) {   /* lines.c:27-82 */
    info->path = strdup(path);
    FILE *fp = fopen(path, "rt");
    const int fd = open(path, O_RDONLY);
    if (!fp || fd == -1) return LF_ERROR_NOT_FOUND;
    char line[MAX_COLUMNS + 2];
    if (!fgets(line, sizeof(line), fp)) line[0] = 0;                              /* lines.c:44-45 */
    info->columns = (uint32_t)strlen(line) - 1;
    fclose(fp);
    if (info->columns > MAX_COLUMNS) return LF_ERROR_TOO_LONG;
    struct stat st;
    fstat(fd, &st);
    info->lines = (uint64_t)st.st_size / ((uint64_t)info->columns + 1);           /* lines.c:54 */
    if (max_lines > 0 && info->lines > max_lines) info->lines = max_lines;
    /* alloc_blocks, lines.c:88-125 */
    info->block_count = (uint32_t)(info->lines / QVZ_LINES_PER_BLOCK);
    if ((uint64_t)info->block_count * QVZ_LINES_PER_BLOCK != info->lines) info->block_count += 1;
    info->blocks = (struct line_block_t *)calloc(info->block_count ? info->block_count : 1, sizeof(struct line_block_t));
    if (!info->blocks) return LF_ERROR_NO_MEMORY;
    uint64_t left = info->lines;
    for (uint32_t b = 0; b < info->block_count; ++b) {
        info->blocks[b].count = left > QVZ_LINES_PER_BLOCK ? (uint32_t)QVZ_LINES_PER_BLOCK : (uint32_t)left;
        left -= info->blocks[b].count;
        info->blocks[b].lines = (struct line_t *)calloc(info->blocks[b].count, sizeof(struct line_t));
        if (!info->blocks[b].lines) return LF_ERROR_NO_MEMORY;
    }
    const uint8_t *map = (const uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);   /* lines.c:64 */
    if (map == MAP_FAILED) return LF_ERROR_NO_MEMORY;
    uint64_t i = 0;
    for (uint32_t b = 0; b < info->block_count; ++b)
        for (uint32_t l = 0; l < info->blocks[b].count; ++l, ++i)
            info->blocks[b].lines[l].m_data = map + i * ((uint64_t)info->columns + 1);   /* lines.c:71 */
    return LF_ERROR_NONE;
}

struct cluster_list_t *alloc_cluster_list(struct quality_file_t *info) {         /* cluster.c:24-43 */
    struct cluster_list_t *rtn = (struct cluster_list_t *)calloc(1, sizeof(*rtn));
    rtn->count = info->cluster_count;
    rtn->clusters = (struct cluster_t *)calloc(info->cluster_count, sizeof(struct cluster_t));
    rtn->distances = (double *)calloc(info->cluster_count, sizeof(double));
    const uint32_t npmf = 1 + info->alphabet->size * (info->columns - 1);        /* codebook.c:19 */
    for (uint32_t j = 0; j < info->cluster_count; ++j) {
        struct cluster_t *c = &rtn->clusters[j];
        c->id = (uint8_t)j;
        c->mean = (symbol_t *)calloc(info->columns, sizeof(symbol_t));
        c->accumulator = (uint64_t *)calloc(info->columns, sizeof(uint64_t));
        struct cond_pmf_list_t *l = (struct cond_pmf_list_t *)calloc(1, sizeof(*l));
        l->columns = info->columns;
        l->alphabet = info->alphabet;
        l->pmfs = (struct pmf_t **)calloc(npmf, sizeof(struct pmf_t *));
        for (uint32_t i = 0; i < npmf; ++i) l->pmfs[i] = new_pmf(info->alphabet);
        c->training_stats = l;
    }
    return rtn;
}

static void options_from(const struct quality_file_t *info, qvz_options_t *o) {
    const struct qv_options_t *r = info->opts;
    qvz_options_default(o);
    o->verbose = r->verbose;
    o->stats = r->stats;
    o->mode = r->mode == QVZ_MODE_FIXED ? QVZ_MODE_FIXED : QVZ_MODE_RATIO;        /* uninitialised in main.c without -f/-r */
    o->clusters = info->cluster_count;
    o->distortion = r->distortion;
    o->dist_file = r->dist_file;                                                   /* -D, main.c:293-297 */
    o->ratio = r->ratio;
    o->cluster_threshold = r->cluster_threshold;
    const char *seed = getenv("QVZ_WELL_SEED");
    if (seed && strcmp(seed, "debug") == 0) o->seed_mode = QVZ_SEED_DEBUG;
    if (getenv("QVZ_DEVICE")) o->device = atoi(getenv("QVZ_DEVICE"));
    if (getenv("QVZ_CHUNK_READS")) o->chunk_reads = strtoull(getenv("QVZ_CHUNK_READS"), 0, 10);
}

static qvz_job_t *job_for(struct quality_file_t *info) {
    if (g_info != info) { printf("qvz_compat: do_kmeans_clustering must be called first for this quality_file_t\n"); exit(1); }
    return &g_job;
}

void do_kmeans_clustering(struct quality_file_t *info) {                         /* cluster.c:212-244 */
    qvz_options_t o;
    options_from(info, &o);
    if (info->lines == 0 || info->block_count == 0) { printf("do_kmeans_clustering: empty input\n"); exit(1); }
    const uint64_t stride = (uint64_t)info->columns + 1;
    const uint8_t *base = info->blocks[0].lines[0].m_data;
    const struct line_block_t *lastb = &info->blocks[info->block_count - 1];
    if (lastb->lines[lastb->count - 1].m_data != base + (info->lines - 1) * stride) {
        printf("qvz_compat: lines are not one contiguous mapping (load_file builds them that way, lines.c:69-79)\n");
        exit(1);
    }
    if (g_info) qvz_job_release(&g_job);
    g_info = info;
    /* the matrix the caller built (main.c:30-38) is the one that counts; qvz_job_init makes its own from the options */
    if (info->dist && info->dist->distortion && o.distortion == QVZ_DIST_CUSTOM) o.distortion = QVZ_DIST_MSE;
    if (qvz_job_init(&g_job, base, 0, info->lines, info->columns, &o)) die("do_kmeans_clustering");
    if (info->dist && info->dist->distortion) memcpy(g_job.h_dist, info->dist->distortion, sizeof(g_job.h_dist));
    if (qvz_job_kmeans(&g_job, 1)) die("do_kmeans_clustering");
    uint64_t i = 0;
    for (uint32_t b = 0; b < info->block_count; ++b)
        for (uint32_t l = 0; l < info->blocks[b].count; ++l, ++i) info->blocks[b].lines[l].cluster = g_job.h_labels[i];
    for (uint32_t j = 0; j < g_job.K; ++j) {
        struct cluster_t *c = &info->clusters->clusters[j];
        c->count = g_job.h_counts[j];
        for (uint32_t s = 0; s < g_job.C; ++s) {
            c->mean[s] = (symbol_t)(g_job.h_means[(size_t)j * g_job.C + s] + 33);            /* raw bytes, cluster.c:201 */
            c->accumulator[s] = g_job.h_acc[(size_t)j * g_job.C + s] + 33ull * c->count;     /* sums of raw bytes, :98 */
        }
    }
}

void calculate_statistics(struct quality_file_t *info) {                         /* codebook.c:185-220 */
    qvz_job_t *job = job_for(info);
    if (qvz_job_stats(job)) die("calculate_statistics");
    const uint32_t C = job->C, K = job->K;
    const size_t words = (size_t)C * A * A;
    uint32_t *cnt = (uint32_t *)malloc(words * sizeof(uint32_t));
    double *marg = (double *)malloc((size_t)C * A * sizeof(double));
    for (uint32_t j = 0; j < K; ++j) {
        if (qvz_hip_copy_d2h(cnt, job->d_cnt + (size_t)j * words, words * sizeof(uint32_t), NULL) || qvz_hip_stream_sync(NULL) ||
            qvz_hip_marginals(job->d_cnt + (size_t)j * words, C, marg, NULL)) {
            printf("calculate_statistics: %s\n", qvz_hip_last_error());
            exit(1);
        }
        struct cond_pmf_list_t *l = info->clusters->clusters[j].training_stats;
        for (uint32_t s = 0; s < C; ++s)
            for (uint32_t p = 0; p < (s ? (uint32_t)A : 1u); ++p) {
                struct pmf_t *pmf = l->pmfs[s ? 1 + (s - 1) * A + p : 0];          /* get_cond_pmf, codebook.c:116-120 */
                const uint32_t *row = cnt + ((size_t)s * A + p) * A;
                uint32_t total = 0;
                for (int x = 0; x < A; ++x) { pmf->counts[x] = row[x]; total += row[x]; }
                pmf->total = total;
                pmf->pmf_ready = 1;                                               /* recalculate_pmf, pmf.c:219-230 */
                for (int x = 0; x < A; ++x) pmf->pmf[x] = total ? ((double)row[x]) / (double)total : 0.0;
            }
        l->marginal_pmfs = (struct pmf_list_t *)calloc(1, sizeof(struct pmf_list_t));     /* codebook.c:212 */
        l->marginal_pmfs->size = C;
        l->marginal_pmfs->pmfs = (struct pmf_t **)calloc(C, sizeof(struct pmf_t *));
        for (uint32_t s = 0; s < C; ++s) {
            struct pmf_t *m = new_pmf(l->alphabet);
            memcpy(m->pmf, marg + (size_t)s * A, A * sizeof(double));
            m->pmf_ready = 1;
            l->marginal_pmfs->pmfs[s] = m;
        }
    }
    free(cnt);
    free(marg);
}

/* cond_quantizer_list_t of one cluster from the flat codebook (alloc_conditional_quantizer_list / cond_quantizer_init_column /
 * store_cond_quantizers_indexed, codebook.c:56-64,101-110,152-157) */
static struct cond_quantizer_list_t *qlist_from_codebook(const qvz_codebook_t *cb, struct quality_file_t *info) {
    const uint32_t C = cb->C;
    struct cond_quantizer_list_t *ql = (struct cond_quantizer_list_t *)calloc(1, sizeof(*ql));
    ql->columns = C;
    ql->input_alphabets = (struct alphabet_t **)calloc(C, sizeof(struct alphabet_t *));
    ql->q = (struct quantizer_t ***)calloc(C, sizeof(struct quantizer_t **));
    ql->ratio = (double **)calloc(C, sizeof(double *));
    ql->qratio = (uint8_t **)calloc(C, sizeof(uint8_t *));
    ql->options = info->opts;
    for (uint32_t s = 0; s < C; ++s) {
        const uint32_t n = cb->nctx[s];
        ql->input_alphabets[s] = alphabet_from(cb->ctx_sym + (size_t)s * A, n);
        ql->q[s] = (struct quantizer_t **)calloc(2 * n, sizeof(struct quantizer_t *));
        ql->ratio[s] = (double *)calloc(n, sizeof(double));
        ql->qratio[s] = (uint8_t *)calloc(n, sizeof(uint8_t));
        for (uint32_t ctx = 0; ctx < n; ++ctx) {
            ql->ratio[s][ctx] = cb->ratio[(size_t)s * A + ctx];
            ql->qratio[s][ctx] = cb->qratio[(size_t)s * A + ctx];
            for (uint32_t b = 0; b < 2; ++b) {
                const size_t t = ((size_t)s * 2 * A + 2 * ctx + b);
                struct quantizer_t *q = (struct quantizer_t *)calloc(1, sizeof(*q));
                q->alphabet = info->alphabet;
                q->q = (symbol_t *)calloc(A, sizeof(symbol_t));
                memcpy(q->q, cb->q + t * A, A);
                q->output_alphabet = alphabet_from(cb->recon + t * A, cb->nstates[t]);   /* quantizer.c:119-121 */
                q->ratio = b ? 1 - ql->ratio[s][ctx] : ql->ratio[s][ctx];                /* codebook.c:448-449 */
                ql->q[s][2 * ctx + b] = q;
            }
        }
    }
    return ql;
}

/* the inverse: the flat codebook of a cond_quantizer_list_t, whoever built it */
static qvz_codebook_t *codebook_from_qlist(const struct cond_quantizer_list_t *ql) {
    const uint32_t C = ql->columns;
    qvz_codebook_t *cb = qvz_codebook_alloc(C);
    if (!cb) return NULL;
    memset(cb->ctx_idx, QVZ_NOT_FOUND, (size_t)C * A);
    memset(cb->out_idx, QVZ_NOT_FOUND, (size_t)C * 2 * A * A);
    memset(cb->live, 1, (size_t)C * A);
    for (uint32_t s = 0; s < C; ++s) {
        const struct alphabet_t *ia = ql->input_alphabets[s];
        const uint32_t n = ia->size > (uint32_t)A ? (uint32_t)A : ia->size;
        cb->nctx[s] = (uint8_t)n;
        for (uint32_t ctx = 0; ctx < n; ++ctx) {
            const uint32_t sym = ia->symbols[ctx] < A ? ia->symbols[ctx] : 0;
            cb->ctx_sym[(size_t)s * A + ctx] = (uint8_t)sym;
            cb->ctx_idx[(size_t)s * A + sym] = (uint8_t)ctx;
            cb->qratio[(size_t)s * A + ctx] = ql->qratio[s][ctx];
            cb->ratio[(size_t)s * A + ctx] = ql->ratio[s] ? ql->ratio[s][ctx] : ql->qratio[s][ctx] / 128.0;
            for (uint32_t b = 0; b < 2; ++b) {
                const struct quantizer_t *q = ql->q[s][2 * ctx + b];
                const size_t t = (size_t)s * 2 * A + 2 * ctx + b;
                memcpy(cb->q + t * A, q->q, A);
                const uint32_t ns = q->output_alphabet->size > (uint32_t)A ? (uint32_t)A : q->output_alphabet->size;
                cb->nstates[t] = (uint8_t)ns;
                for (uint32_t z = 0; z < ns; ++z) {
                    const uint32_t v = q->output_alphabet->symbols[z] < A ? q->output_alphabet->symbols[z] : 0;
                    cb->recon[t * A + z] = (uint8_t)v;
                    cb->out_idx[t * A + v] = (uint8_t)z;
                }
            }
        }
    }
    return cb;
}

void generate_codebooks(struct quality_file_t *info) {                           /* codebook.c:355-468 */
    qvz_job_t *job = job_for(info);
    if (qvz_job_codebooks(job)) die("generate_codebooks");
    for (uint32_t j = 0; j < job->K; ++j) info->clusters->clusters[j].qlist = qlist_from_codebook(job->cbs[j], info);
}

void write_codebooks(FILE *fp, struct quality_file_t *info) {                    /* codebook.c:474-498 */
    if (qvz_job_write_codebooks(job_for(info), fp, NULL)) die("write_codebooks");
}

uint32_t start_qv_compression(struct quality_file_t *info, FILE *fout, double *dis, FILE *funcompressed) {   /* qv_compressor.c:48-143 */
    qvz_job_t *job = job_for(info);
    uint64_t stream_bytes = 0;
    double distortion = 0.0;
    if (qvz_job_compress(job, fout, funcompressed, &stream_bytes, &distortion)) die("start_qv_compression");
    memset(&info->well, 0, sizeof(info->well));                                  /* qv_stream.c:70 */
    memcpy(info->well.state, job->well_state, sizeof(job->well_state));
    if (dis) *dis = distortion;
    return (uint32_t)stream_bytes;                                               /* int -> uint32_t, arith.c:115 */
}

/* ---- decode direction (main.c:132-160) ------------------------------------------------------------------------------ */
static uint8_t *slurp_rest(FILE *fp, size_t *size) {            /* everything from the current position to the end */
    size_t cap = 1u << 20, len = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    while (buf) {
        const size_t got = fread(buf + len, 1, cap - len, fp);
        len += got;
        if (len < cap) break;
        uint8_t *nb = (uint8_t *)realloc(buf, cap * 2);
        if (!nb) { free(buf); buf = NULL; break; }
        buf = nb;
        cap *= 2;
    }
    *size = len;
    return buf;
}

void read_codebooks(FILE *fp, struct quality_file_t *info) {                     /* codebook.c:560-582 */
    const long start = ftell(fp);
    size_t size = 0, pos = 9;
    uint8_t *f = slurp_rest(fp, &size);
    qvz_codebook_t **cbs = NULL;
    if (!f || size < 9) { printf("read_codebooks: not a qvz file\n"); exit(1); }
    info->cluster_count = f[0];
    info->columns = ((uint32_t)f[1] << 24) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 8) | f[4];     /* ntohl, :569-572 */
    info->lines = ((uint32_t)f[5] << 24) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 8) | f[8];
    if (info->cluster_count < 1 || info->columns < 1 || info->columns > MAX_COLUMNS) { printf("read_codebooks: implausible header\n"); exit(1); }
    info->clusters = alloc_cluster_list(info);                                   /* :575 */
    if (qvz_codebooks_parse(f, size, &pos, info->cluster_count, info->columns, &cbs)) die("read_codebooks");
    for (uint32_t j = 0; j < info->cluster_count; ++j) {
        info->clusters->clusters[j].qlist = qlist_from_codebook(cbs[j], info);
        qvz_codebook_free(cbs[j]);
    }
    free(cbs);
    free(f);
    fseek(fp, start + (long)pos, SEEK_SET);                                      /* the WELL state comes next (qv_stream.c:72-74) */
}

void start_qv_decompression(FILE *fout, FILE *fin, struct quality_file_t *info) {   /* qv_compressor.c:145-231 */
    size_t size = 0;
    uint8_t *f = slurp_rest(fin, &size);
    const uint32_t K = info->cluster_count;
    qvz_codebook_t **cbs = (qvz_codebook_t **)calloc(K ? K : 1, sizeof(*cbs));
    if (!f || size < 128 || !cbs) { printf("start_qv_decompression: file ends before the WELL state\n"); exit(1); }
    for (uint32_t j = 0; j < K; ++j)
        if (!(cbs[j] = codebook_from_qlist(info->clusters->clusters[j].qlist))) { printf("start_qv_decompression: out of memory\n"); exit(1); }
    memset(&info->well, 0, sizeof(info->well));                                  /* qv_stream.c:70 */
    memcpy(info->well.state, f, 128);                                            /* qv_stream.c:72-74 */
    if (qvz_decode_stream_host(K, info->columns, info->lines, (const qvz_codebook_t *const *)cbs, info->well.state, f + 128, size - 128,
                               fout, info->opts && info->opts->verbose))
        die("start_qv_decompression");
    for (uint32_t j = 0; j < K; ++j) qvz_codebook_free(cbs[j]);
    free(cbs);
    free(f);
}

/* ---- symbol-level interface (qv_compressor.h:63-93): host C, one serial chain by definition ---------------------------- */
struct os_stream_t *alloc_os_stream(FILE *fp, uint8_t in) {                      /* os_stream.c:7-22 */
    struct os_stream_t *os = (struct os_stream_t *)calloc(1, sizeof(*os));
    os->fp = fp;
    os->buf = (uint8_t *)calloc(OS_STREAM_BUF_LEN, 1);
    if (in && fread(os->buf, 1, OS_STREAM_BUF_LEN, fp) == 0) { /* an empty stream reads as zeros */ }
    return os;
}

void free_os_stream(struct os_stream_t *os) { free(os->buf); free(os); }         /* the FILE is the caller's, os_stream.c:27-31 */

void stream_write_buffer(struct os_stream_t *os) {                               /* os_stream.c:115-120 */
    fwrite(os->buf, 1, os->bufPos, os->fp);
    memset(os->buf, 0, os->bufPos);
    os->written += os->bufPos;
    os->bufPos = 0;
}

/* The byte under construction holds its bits so far shifted left by one (the reference ORs the new bit into bit 0 and
 * shifts after every bit but the eighth, os_stream.c:72-92): kept that way so that the struct means the same to both. */
void stream_write_bit(struct os_stream_t *os, uint8_t bit) {
    uint8_t *b = &os->buf[os->bufPos];
    *b |= bit & 1;
    if (++os->bitPos == 8) {
        os->bitPos = 0;
        if (++os->bufPos == OS_STREAM_BUF_LEN) stream_write_buffer(os);
    } else {
        *b = (uint8_t)(*b << 1);
    }
}

void stream_write_bits(struct os_stream_t *os, uint32_t dw, uint8_t len) {       /* msb first, os_stream.c:94-100 */
    for (int bit = (int)len - 1; bit >= 0; --bit) stream_write_bit(os, (uint8_t)(dw >> bit));
}

void stream_finish_byte(struct os_stream_t *os) {                                /* os_stream.c:105-110 */
    os->buf[os->bufPos] = (uint8_t)(os->buf[os->bufPos] << (7 - os->bitPos));
    os->bitPos = 0;
    os->bufPos += 1;
    stream_write_buffer(os);
}

uint8_t stream_read_bit(struct os_stream_t *os) {                                /* consumes the byte from its top, os_stream.c:36-52 */
    const uint8_t bit = os->buf[os->bufPos] >> 7;
    os->buf[os->bufPos] = (uint8_t)(os->buf[os->bufPos] << 1);
    if (++os->bitPos == 8) {
        os->bitPos = 0;
        if (++os->bufPos == OS_STREAM_BUF_LEN) {
            if (fread(os->buf, 1, OS_STREAM_BUF_LEN, os->fp) == 0) memset(os->buf, 0, OS_STREAM_BUF_LEN);
            os->bufPos = 0;
        }
    }
    return bit;
}

uint32_t stream_read_bits(struct os_stream_t *os, uint8_t len) {
    uint32_t v = 0;
    for (int bit = (int)len - 1; bit >= 0; --bit) v |= (uint32_t)stream_read_bit(os) << bit;
    return v;
}

Arithmetic_code initialize_arithmetic_encoder(uint32_t m) {                      /* arith.c:5-16 */
    Arithmetic_code a = (Arithmetic_code)calloc(1, sizeof(*a));
    a->m = m;
    a->r = 1u << (m - 3);
    a->u = (1u << m) - 1;
    return a;
}

void update_stats(stream_stats_ptr_t stats, uint32_t x, uint32_t r) {            /* qv_stream.c:9-25 */
    stats->counts[x] += stats->step;
    stats->n += stats->step;
    if (stats->n > r) {
        uint32_t n = 0;
        for (uint32_t i = 0; i < stats->alphabetCard; ++i)
            if (stats->counts[i]) { stats->counts[i] = (stats->counts[i] >> 1) + 1; n += stats->counts[i]; }
        stats->n = n;
    }
}

static stream_stats_ptr_t new_stats(uint32_t card) {                             /* qv_stream.c:44-57, :96-104 */
    stream_stats_ptr_t s = (stream_stats_ptr_t)calloc(1, sizeof(*s));
    s->counts = (uint32_t *)calloc(card ? card : 1, sizeof(uint32_t));
    for (uint32_t k = 0; k < card; ++k) s->counts[k] = 1;
    s->n = s->alphabetCard = card;
    s->step = 8;
    return s;
}

stream_stats_ptr_t **initialize_stream_stats(struct cond_quantizer_list_t *ql) {   /* qv_stream.c:32-61 */
    stream_stats_ptr_t **s = (stream_stats_ptr_t **)calloc(ql->columns, sizeof(*s));
    for (uint32_t i = 0; i < ql->columns; ++i) {
        const uint32_t nq = 2 * ql->input_alphabets[i]->size;
        s[i] = (stream_stats_ptr_t *)calloc(nq, sizeof(**s));
        for (uint32_t j = 0; j < nq; ++j) s[i][j] = new_stats(ql->q[i][j]->output_alphabet->size);
    }
    return s;
}

/* new bounds of the interval for symbol x (arith.c:35-48); returns 0 when x is outside the alphabet */
static void narrow(Arithmetic_code a, const struct stream_stats_t *st, uint32_t x) {
    const uint64_t range = (uint64_t)a->u - a->l + 1;
    uint32_t lo = 0;
    for (uint32_t i = 0; i < x; ++i) lo += st->counts[i];
    const uint32_t hi = lo + st->counts[x];
    a->u = a->l + (uint32_t)((range * hi) / st->n) - 1;
    a->l = a->l + (uint32_t)((range * lo) / st->n);
}

static inline uint32_t lead_zeros(uint32_t v, uint32_t width) { return v ? (uint32_t)__builtin_clz(v) - (32 - width) : width; }

void arithmetic_encoder_step(Arithmetic_code a, stream_stats_ptr_t stats, int32_t x, osStream os) {   /* arith.c:24-97 */
    const uint32_t m = a->m, msb = 1u << (m - 1), mask = (1u << m) - 1, mask1 = msb - 1;
    if (x < 0 || (uint32_t)x >= stats->alphabetCard) { printf("arithmetic_encoder_step: symbol outside the alphabet\n"); exit(1); }   /* assert, :37 */
    narrow(a, stats, (uint32_t)x);
    /* E1/E2 then E3, each as one batch: after the k shared leading bits are out, msb(l) = 0 and msb(u) = 1, and E3 shifts
     * keep it so (same integers as the loop of :66-96) */
    const uint32_t diff = a->l ^ a->u;
    if (!(diff & msb)) {
        const uint32_t k = lead_zeros(diff, m), top = a->l >> (m - k), first = top >> (k - 1);
        stream_write_bit(os, (uint8_t)first);
        for (; a->scale3 > 0; a->scale3 -= 1) stream_write_bit(os, (uint8_t)!first);
        if (k > 1) stream_write_bits(os, top & ((1u << (k - 1)) - 1), (uint8_t)(k - 1));
        a->l = (a->l << k) & mask;
        a->u = ((a->u << k) & mask) | ((1u << k) - 1);
    }
    const uint32_t ones = lead_zeros(~a->l & mask1, m - 1), zeros = lead_zeros(a->u & mask1, m - 1), j = ones < zeros ? ones : zeros;
    if (j) {
        a->scale3 += (int32_t)j;
        a->l = (a->l << j) & mask1;
        a->u = ((a->u << j) & mask1) | msb | ((1u << j) - 1);
    }
}

int encoder_last_step(Arithmetic_code a, osStream os) {                          /* arith.c:99-116 */
    const uint8_t first = (uint8_t)(a->l >> (a->m - 1));
    stream_write_bit(os, first);
    for (; a->scale3 > 0; a->scale3 -= 1) stream_write_bit(os, !first);
    stream_write_bits(os, a->l, (uint8_t)(a->m - 1));
    stream_finish_byte(os);
    return (int)os->written;
}

static uint32_t symbol_under_tag(const struct Arithmetic_code_t *a, const struct stream_stats_t *st) {   /* arith.c:130-136, :196-203 */
    const uint64_t range = (uint64_t)a->u - a->l + 1, gap = (uint64_t)a->t - a->l + 1;
    const uint64_t sub = (gap * st->n - 1) / range;
    uint64_t cum = 0;
    uint32_t k = 0;
    while (k < st->alphabetCard && sub >= cum) cum += st->counts[k++];
    return k ? k - 1 : 0;
}

uint32_t decoder_last_step(Arithmetic_code a, stream_stats_ptr_t stats) { return symbol_under_tag(a, stats); }

uint32_t arithmetic_decoder_step(Arithmetic_code a, stream_stats_ptr_t stats, osStream is) {            /* arith.c:118-188 */
    const uint32_t m = a->m, msb = 1u << (m - 1), mask1 = msb - 1;
    const uint32_t x = symbol_under_tag(a, stats);
    narrow(a, stats, x);
    for (;;) {
        if ((a->l >> (m - 1)) == (a->u >> (m - 1))) {                            /* E1 / E2 */
            a->l = (a->l & mask1) << 1;
            a->u = ((a->u & mask1) << 1) + 1;
            a->t = ((a->t & mask1) << 1) + stream_read_bit(is);
        } else if ((a->l >> (m - 2)) == 1 && (a->u >> (m - 2)) == 2) {           /* E3 */
            a->l = (a->l << 1) & mask1;
            a->u = (((a->u << 1) & mask1) | msb) + 1;
            a->t = (((a->t & mask1) << 1) ^ msb) + stream_read_bit(is);
        } else break;
    }
    return x;
}

void compress_qv(arithStream as, uint32_t x, uint8_t cluster, uint32_t column, uint32_t idx) {          /* qv_compressor.c:8-11 */
    arithmetic_encoder_step(as->a, as->stats[cluster][column][idx], (int32_t)x, as->os);
    update_stats(as->stats[cluster][column][idx], x, as->a->r);
}

void qv_write_cluster(arithStream as, uint8_t cluster) {                                                /* qv_compressor.c:19-22 */
    arithmetic_encoder_step(as->a, as->cluster_stats, cluster, as->os);
    update_stats(as->cluster_stats, cluster, as->a->r);
}

uint32_t decompress_qv(arithStream as, uint8_t cluster, uint32_t column, uint32_t idx) {                /* qv_compressor.c:27-34 */
    const uint32_t x = arithmetic_decoder_step(as->a, as->stats[cluster][column][idx], as->os);
    update_stats(as->stats[cluster][column][idx], x, as->a->r);
    return x;
}

uint8_t qv_read_cluster(arithStream as) {                                                               /* qv_compressor.c:36-43 */
    const uint32_t x = arithmetic_decoder_step(as->a, as->cluster_stats, as->os);
    update_stats(as->cluster_stats, x, as->a->r);
    return (uint8_t)x;
}

qv_compressor initialize_qv_compressor(FILE *fout, uint8_t streamDirection, struct quality_file_t *info) {   /* qv_stream.c:66-125 */
    memset(&info->well, 0, sizeof(info->well));
    if (streamDirection) {
        if (fread(info->well.state, sizeof(uint32_t), 32, fout) != 32) { printf("initialize_qv_compressor: file ends before the WELL state\n"); exit(1); }
    } else {
        const char *seed = getenv("QVZ_WELL_SEED");
        const int debug = seed && strcmp(seed, "debug") == 0;                    /* the reference's -DDEBUG build, qv_stream.c:79-83 */
        srand((uint32_t)time(0));                                                /* qv_stream.c:77 */
        for (int i = 0; i < 32; ++i) info->well.state[i] = debug ? 0x55555555u : (uint32_t)rand();
        fwrite(info->well.state, sizeof(uint32_t), 32, fout);
    }
    arithStream as = (arithStream)calloc(1, sizeof(*as));
    as->cluster_stats = new_stats(info->cluster_count);
    as->stats = (stream_stats_ptr_t ***)calloc(info->cluster_count, sizeof(*as->stats));
    for (uint32_t j = 0; j < info->cluster_count; ++j) as->stats[j] = initialize_stream_stats(info->clusters->clusters[j].qlist);
    as->a = initialize_arithmetic_encoder(m_arith);
    as->os = alloc_os_stream(fout, streamDirection);
    as->a->t = streamDirection ? stream_read_bits(as->os, (uint8_t)as->a->m) : 0;
    qv_compressor qvc = (qv_compressor)calloc(1, sizeof(*qvc));
    qvc->Quals = as;
    return qvc;
}

uint32_t well_1024a(struct well_state_t *w) {                                    /* well.c:8-24 */
    uint32_t *s = w->state;
    const uint32_t n = w->n & 31;
    const uint32_t z0 = s[(n + 31) & 31], v1 = s[(n + 3) & 31], v2 = s[(n + 24) & 31], v3 = s[(n + 10) & 31];
    const uint32_t z1 = s[n] ^ v1 ^ (v1 >> 8);
    const uint32_t z2 = v2 ^ (v2 << 19) ^ v3 ^ (v3 << 14);
    s[n] = z1 ^ z2;
    w->n = (n + 31) & 31;
    s[w->n] = z0 ^ (z0 << 11) ^ z1 ^ (z1 << 7) ^ z2 ^ (z2 << 13);
    return s[w->n];
}

uint32_t well_1024a_bits(struct well_state_t *w, uint8_t bits) {                 /* well.c:33-46 */
    if (w->bits_left < bits) { w->bit_output = well_1024a(w); w->bits_left = 32; }
    const uint32_t v = w->bit_output & ((1u << bits) - 1);
    w->bit_output >>= bits;
    w->bits_left -= bits;
    return v;
}

struct quantizer_t *choose_quantizer(struct cond_quantizer_list_t *list, struct well_state_t *well, uint32_t column,
                                     symbol_t prev, uint32_t *q_idx) {           /* codebook.c:162-171 */
    const uint32_t idx = list->input_alphabets[column]->indexes[prev];
    if (idx == NOT_FOUND_U32) { printf("choose_quantizer: no context for value %u in column %u\n", prev, column); exit(1); }   /* assert, :164 */
    *q_idx = 2 * idx + (well_1024a_bits(well, 7) >= list->qratio[column][idx] ? 1u : 0u);
    return list->q[column][*q_idx];
}
