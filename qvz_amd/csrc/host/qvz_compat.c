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

void generate_codebooks(struct quality_file_t *info) {                           /* codebook.c:355-468 */
    qvz_job_t *job = job_for(info);
    if (qvz_job_codebooks(job)) die("generate_codebooks");
    const uint32_t C = job->C;
    for (uint32_t j = 0; j < job->K; ++j) {
        const qvz_codebook_t *cb = job->cbs[j];
        struct cond_quantizer_list_t *ql = (struct cond_quantizer_list_t *)calloc(1, sizeof(*ql));   /* codebook.c:56-64 */
        ql->columns = C;
        ql->input_alphabets = (struct alphabet_t **)calloc(C, sizeof(struct alphabet_t *));
        ql->q = (struct quantizer_t ***)calloc(C, sizeof(struct quantizer_t **));
        ql->ratio = (double **)calloc(C, sizeof(double *));
        ql->qratio = (uint8_t **)calloc(C, sizeof(uint8_t *));
        ql->options = info->opts;
        for (uint32_t s = 0; s < C; ++s) {
            const uint32_t n = cb->nctx[s];
            ql->input_alphabets[s] = alphabet_from(cb->ctx_sym + (size_t)s * A, n);          /* cond_quantizer_init_column :101-110 */
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
        info->clusters->clusters[j].qlist = ql;
    }
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

void read_codebooks(FILE *fp, struct quality_file_t *info) {
    (void)fp; (void)info;
    printf("Decoding (-x) is not part of this build: use the reference decoder on the .q file (SURVEY.md section 8f).\n");
    exit(1);
}

void start_qv_decompression(FILE *fout, FILE *fin, struct quality_file_t *info) {
    (void)fout; (void)fin;
    read_codebooks(NULL, info);
}
