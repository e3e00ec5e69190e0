/* qvz_decode.c - the decode direction of the drop-in surface (`qvz -x`), host C.
 *
 *   qvz_codebook_read      read_codebooks (reference src/codebook.c:560-669): codebook text -> qvz_codebook_t, output
 *                          alphabets by run de-duplication of the quantizer tables (find_output_alphabet,
 *                          src/quantizer.c:167-191), context sets by merging them (alphabet_union, src/pmf.c:312-357)
 *   qvz_decode_file        start_qv_decompression (src/qv_compressor.c:145-231) for the reference's one-stream file:
 *                          the serial chain (arithmetic_decoder_step / decoder_last_step, src/arith.c:118-205) and the
 *                          WELL draws (src/well.c:8-46) run on one host core as in the reference: a single stream offers
 *                          the GPU one lane. A shard set (qvz_shardset.h) given as input goes to the GPU decoder instead.
 *   qvz_decode_shard_files the files written by qvz_encode_file_sharded ("<prefix>.0000" ...): every shard is one lane of
 *                          the GPU decoder (qvz_hip_decode_shards); output = the decodes in order
 */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/statvfs.h>
#include <time.h>

#include "qvz_codebook.h"
#include "qvz_coder.h"
#include "qvz_encode.h"
#include "qvz_hip.h"
#include "qvz_job.h"
#include "qvz_shardset.h"

#define A QVZ_A
#define M_BITS 22u
#define MSB (1u << (M_BITS - 1))
#define MASK_M ((1u << M_BITS) - 1)
#define MASK_M1 ((1u << (M_BITS - 1)) - 1)
#define RESCALE (1u << (M_BITS - 3))
#define STEP 8u
#define NO_TABLE 0xFFFFFFFFu

/* find_output_alphabet, quantizer.c:167-191: a new symbol whenever the table value changes */
static uint32_t dedup_alphabet(const uint8_t *q, uint8_t *out) {
    uint32_t size = 1;
    uint8_t p = q[0];
    out[0] = p;
    for (int x = 1; x < A; ++x)
        if (q[x] != p) { p = q[x]; out[size++] = p; }
    return size;
}

/* alphabet_union, pmf.c:312-357: merge of two ascending symbol lists */
static uint32_t union_lists(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb, uint8_t *out) {
    uint8_t tmp[4 * A];
    uint32_t i = 0, j = 0, k = 0;
    while (i < na && j < nb && k < 4 * A) {
        if (a[i] < b[j]) tmp[k] = a[i++];
        else if (a[i] == b[j]) { tmp[k] = a[i]; ++i; ++j; }
        else tmp[k] = b[j++];
        ++k;
    }
    while (i < na && k < 4 * A) tmp[k++] = a[i++];
    while (j < nb && k < 4 * A) tmp[k++] = b[j++];
    memcpy(out, tmp, k);
    return k;
}

/* A table byte is a quality value + 33 ('!'..'h'), a ratio byte (uint8)(ratio * 128) + 33 with ratio <= 1: anything else
 * would index past the 72-entry rows below (the reference does so, codebook.c:586-663: undefined behaviour there). */
static int set_quantizer(qvz_codebook_t *cb, size_t t, const uint8_t *text) {
    for (int i = 0; i < A; ++i) {
        if (text[i] < 33 || text[i] >= 33 + A) return qvz_job_fail(QVZ_E_DATA, "codebook byte 0x%02x is not a quality value", text[i]);
        cb->q[t * A + i] = (uint8_t)(text[i] - 33);
    }
    const uint32_t ns = dedup_alphabet(cb->q + t * A, cb->recon + t * A);
    cb->nstates[t] = (uint8_t)ns;
    memset(cb->out_idx + t * A, QVZ_NOT_FOUND, A);
    for (uint32_t z = 0; z < ns; ++z) cb->out_idx[t * A + cb->recon[t * A + z]] = (uint8_t)z;
    return QVZ_OK;
}

static int ratio_byte(uint8_t b, uint8_t *out) {
    if (b < 33 || b > 33 + 128) return qvz_job_fail(QVZ_E_DATA, "codebook ratio byte 0x%02x out of range", b);
    *out = (uint8_t)(b - 33);
    return QVZ_OK;
}

int qvz_codebook_read(const uint8_t *f, size_t size, size_t *ppos, qvz_codebook_t *cb) {
    size_t pos = *ppos;
    const uint32_t C = cb->C;
    uint8_t uni[4 * A];
    uint32_t nu;
#define NEED(nb) do { if (pos + (nb) > size) return qvz_job_fail(QVZ_E_DATA, "codebook ends early"); } while (0)
    memset(cb->ctx_idx, QVZ_NOT_FOUND, (size_t)C * A);
    memset(cb->live, 1, (size_t)C * A);
    NEED(2 + 2 * (A + 1));                                                     /* codebook.c:596-618 */
    cb->nctx[0] = 1; cb->ctx_sym[0] = 0; cb->ctx_idx[0] = 0;
    if (ratio_byte(f[pos], &cb->qratio[0])) return QVZ_E_DATA;
    cb->ratio[0] = cb->qratio[0] / 128.0;
    pos += 2;
    for (int b = 0; b < 2; ++b) { if (set_quantizer(cb, (size_t)b, f + pos)) return QVZ_E_DATA; pos += A + 1; }
    nu = union_lists(cb->recon, cb->nstates[0], cb->recon + A, cb->nstates[1], uni);
    for (uint32_t s = 1; s < C; ++s) {                                         /* :623-663 */
        const uint32_t n = nu;
        if (n == 0 || n > A) return qvz_job_fail(QVZ_E_DATA, "column %u: %u contexts", s, n);
        cb->nctx[s] = (uint8_t)n;
        memcpy(cb->ctx_sym + (size_t)s * A, uni, n);
        for (uint32_t i = 0; i < n; ++i) cb->ctx_idx[(size_t)s * A + uni[i]] = (uint8_t)i;
        NEED(n + 1 + 2 * ((size_t)n * A + 1));
        for (uint32_t j = 0; j < n; ++j) {
            if (ratio_byte(f[pos + j], &cb->qratio[(size_t)s * A + j])) return QVZ_E_DATA;
            cb->ratio[(size_t)s * A + j] = cb->qratio[(size_t)s * A + j] / 128.0;
        }
        pos += n + 1;
        nu = 0;
        for (int b = 0; b < 2; ++b) {
            for (uint32_t j = 0; j < n; ++j) {
                const size_t t = (size_t)s * 2 * A + 2 * j + b;
                if (set_quantizer(cb, t, f + pos)) return QVZ_E_DATA;
                pos += A;
                nu = union_lists(uni, nu, cb->recon + t * A, cb->nstates[t], uni);
            }
            pos += 1;
        }
    }
#undef NEED
    *ppos = pos;
    return QVZ_OK;
}

/* ---- the serial chain on the host ---------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *buf;
    uint64_t size, pos;
    uint32_t bitpos;
    uint32_t l, u, t;
} hdec_t;

static inline uint32_t rd_bit(hdec_t *d) {                                      /* stream_read_bit, os_stream.c:36-52 */
    uint32_t b = 0;
    if (d->pos < d->size) b = (d->buf[d->pos] >> (7 - d->bitpos)) & 1;
    if (++d->bitpos == 8) { d->bitpos = 0; d->pos++; }
    return b;
}

/* arithmetic_decoder_step (arith.c:118-188) / decoder_last_step (:190-205) + update_stats (qv_stream.c:9-25) */
static uint32_t hdec_symbol(hdec_t *d, uint32_t *t, int last) {
    const uint32_t n = t[0], card = t[1];
    const uint64_t range = (uint64_t)d->u - d->l + 1, gap = (uint64_t)d->t - d->l + 1;
    const uint64_t sub = (gap * n - 1) / range;
    uint64_t cum = 0;
    uint32_t k = 0;
    while (k < card && sub >= cum) cum += t[2 + k++];
    if (sub >= cum) return card;
    const uint32_t x = k - 1;
    if (last) return x;
    const uint32_t fx = t[2 + x];
    d->u = d->l + (uint32_t)((range * cum) / n) - 1;
    d->l = d->l + (uint32_t)((range * (cum - fx)) / n);
    for (;;) {
        if ((d->l >> (M_BITS - 1)) == (d->u >> (M_BITS - 1))) {                 /* E1 / E2 */
            d->l = (d->l & MASK_M1) << 1;
            d->u = ((d->u & MASK_M1) << 1) + 1;
            d->t = ((d->t & MASK_M1) << 1) + rd_bit(d);
        } else if ((d->l >> (M_BITS - 2)) == 1 && (d->u >> (M_BITS - 2)) == 2) {   /* E3 */
            d->l = (d->l << 1) & MASK_M1;
            d->u = (((d->u << 1) & MASK_M1) | MSB) + 1;
            d->t = (((d->t & MASK_M1) << 1) ^ MSB) + rd_bit(d);
        } else break;
    }
    t[2 + x] = fx + STEP;
    uint32_t nn = n + STEP;
    if (nn > RESCALE) {
        nn = 0;
        for (uint32_t i = 0; i < card; ++i)
            if (t[2 + i]) { t[2 + i] = (t[2 + i] >> 1) + 1; nn += t[2 + i]; }
    }
    t[0] = nn;
    return x;
}

/* WELL-1024a on the host (reference src/well.c:8-46) for the serial decoder: one stream means one draw at a time */
typedef struct { uint32_t s[32]; uint32_t n, word, left; } hwell_t;

static inline uint32_t hwell_draw(hwell_t *w) {                                  /* well_1024a_bits(w, 7) */
    if (w->left < 7) {
        uint32_t *s = w->s;
        const uint32_t n = w->n;
        const uint32_t z0 = s[(n + 31) & 31], v1 = s[(n + 3) & 31], v2 = s[(n + 24) & 31], v3 = s[(n + 10) & 31];
        const uint32_t z1 = s[n] ^ v1 ^ (v1 >> 8);
        const uint32_t z2 = v2 ^ (v2 << 19) ^ v3 ^ (v3 << 14);
        s[n] = z1 ^ z2;
        w->n = (n + 31) & 31;
        s[w->n] = z0 ^ (z0 << 11) ^ z1 ^ (z1 << 7) ^ z2 ^ (z2 << 13);
        w->word = s[w->n];
        w->left = 32;
    }
    const uint32_t d = w->word & 127;
    w->word >>= 7;
    w->left -= 7;
    return d;
}

void qvz_qfile_release(qvz_qfile_t *q) {
    if (q->cbs) { for (uint32_t j = 0; j < q->K; ++j) qvz_codebook_free(q->cbs[j]); free(q->cbs); }
    memset(q, 0, sizeof(*q));
}

/* `clusters` codebooks (read_codebook, codebook.c:586-669) starting at f[*pos] */
int qvz_codebooks_parse(const uint8_t *f, size_t size, size_t *pos, uint32_t K, uint32_t C, qvz_codebook_t ***out) {
    qvz_codebook_t **cbs = (qvz_codebook_t **)calloc(K, sizeof(*cbs));
    if (!cbs) return qvz_job_fail(QVZ_E_NOMEM, "host alloc");
    for (uint32_t j = 0; j < K; ++j) {
        cbs[j] = qvz_codebook_alloc(C);
        const int rc = cbs[j] ? qvz_codebook_read(f, size, pos, cbs[j]) : qvz_job_fail(QVZ_E_NOMEM, "host alloc");
        if (rc) { for (uint32_t i = 0; i <= j; ++i) qvz_codebook_free(cbs[i]); free(cbs); return rc; }
    }
    *out = cbs;
    return QVZ_OK;
}

/* header (codebook.c:565-573), codebooks, WELL state (qv_stream.c:72-74) */
int qvz_qfile_parse(const uint8_t *f, size_t size, qvz_qfile_t *q) {
    memset(q, 0, sizeof(*q));
    if (size < 9) return qvz_job_fail(QVZ_E_DATA, "not a qvz file");
    q->K = f[0];
    q->C = ((uint32_t)f[1] << 24) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 8) | f[4];
    q->lines = ((uint32_t)f[5] << 24) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 8) | f[8];
    if (q->K < 1 || q->C < 1 || q->C > (1u << 20)) return qvz_job_fail(QVZ_E_DATA, "implausible header (%u clusters, %u columns)", q->K, q->C);
    size_t pos = 9;
    const int rc = qvz_codebooks_parse(f, size, &pos, q->K, q->C, &q->cbs);
    if (rc) { memset(q, 0, sizeof(*q)); return rc; }
    if (pos + 128 > size) { qvz_qfile_release(q); return qvz_job_fail(QVZ_E_DATA, "file ends before the WELL state"); }
    memcpy(q->well, f + pos, 128);
    q->stream_at = pos + 128;
    return QVZ_OK;
}

/* A file somebody wrote, mapped read-only (no copy: a shard set of 100M x 150 reads is 2.6 GB, and its streams go from the page
 * cache through pinned slabs to the device). qvz_unmap_file releases it. */
int qvz_map_file(const char *path, const uint8_t **buf, size_t *size) {
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0) return qvz_job_fail(QVZ_E_ARG, "Unable to open input file %s", path);
    if (fstat(fd, &st) || st.st_size <= 0) { close(fd); return qvz_job_fail(QVZ_E_ARG, "cannot read %s", path); }
    void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return qvz_job_fail(QVZ_E_NOMEM, "cannot map %s", path);
    *buf = (const uint8_t *)m;
    *size = (size_t)st.st_size;
    return QVZ_OK;
}
void qvz_unmap_file(const uint8_t *buf, size_t size) { if (buf) munmap((void *)buf, size); }

int qvz_read_whole_file(const char *path, uint8_t **buf, size_t *size) {
    FILE *f = fopen(path, "rb");
    if (!f) return qvz_job_fail(QVZ_E_ARG, "Unable to open input file %s", path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    *buf = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
    if (!*buf || sz < 0 || fread(*buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(*buf); *buf = NULL; return qvz_job_fail(QVZ_E_ARG, "cannot read %s", path); }
    fclose(f);
    *size = (size_t)sz;
    return QVZ_OK;
}

/* start_qv_decompression (qv_compressor.c:145-231) over memory: one stream, one host core, as in the reference */
int qvz_decode_stream_host(uint32_t K, uint32_t C, uint64_t lines, const qvz_codebook_t *const *cbs, const uint32_t *well,
                           const uint8_t *stream, size_t size, FILE *fo, int verbose) {
    uint32_t *arena = NULL, *off = NULL, words = 0;
    uint8_t *row = NULL;
    int rc = QVZ_OK;
    if (!lines) return QVZ_OK;
    if (qvz_model_build(K, C, cbs, 0, &arena, &off, &words)) return qvz_job_fail(QVZ_E_NOMEM, "model build failed");
    row = (uint8_t *)malloc((size_t)C + 1);
    if (!row) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    {
        hwell_t w;
        hdec_t d = { stream, size, 0, 0, 0, MASK_M, 0 };
        memcpy(w.s, well, 128);
        w.n = 0; w.word = 0; w.left = 0;                                        /* qv_stream.c:70,93 */
        for (int bit = (int)M_BITS - 1; bit >= 0; --bit) d.t |= rd_bit(&d) << bit;   /* qv_stream.c:112-113 */
        for (uint64_t i = 0; i < lines; ++i) {
            if (verbose && i % QVZ_LINES_PER_BLOCK == 0) printf("Line: %dM\n", (int)(i / QVZ_LINES_PER_BLOCK));   /* qv_compressor.c:167-169 */
            /* qv_read_cluster, qv_compressor.c:36-43: coded for every read (a no-op on the interval when K = 1) */
            const uint32_t cl = hdec_symbol(&d, arena, 0);
            if (cl >= K) { rc = qvz_job_fail(QVZ_E_DATA, "stream does not fit the codebooks (cluster id)"); goto done; }
            const qvz_codebook_t *cb = cbs[cl];
            uint32_t prev = 0;
            for (uint32_t s = 0; s < C; ++s) {
                const uint32_t ctx = cb->ctx_idx[(size_t)s * A + prev];
                if (ctx == QVZ_NOT_FOUND) { rc = qvz_job_fail(QVZ_E_DATA, "stream does not fit the codebooks (context)"); goto done; }
                const uint32_t draw = hwell_draw(&w);                           /* choose_quantizer, codebook.c:162-171 */
                const uint32_t qi = 2 * ctx + (draw >= cb->qratio[(size_t)s * A + ctx] ? 1u : 0u);
                const uint32_t o = off[((size_t)cl * C + s) * 2 * A + qi];
                const int last = i + 1 == lines && s + 1 == C;                  /* qv_compressor.c:222-225 */
                const uint32_t x = hdec_symbol(&d, arena + o, last);
                if (x >= arena[o + 1]) { rc = qvz_job_fail(QVZ_E_DATA, "stream does not fit the codebooks (state)"); goto done; }
                prev = cb->recon[((size_t)s * 2 * A + qi) * A + x];
                row[s] = (uint8_t)(prev + 33);
            }
            row[C] = '\n';
            if (fwrite(row, 1, (size_t)C + 1, fo) != (size_t)C + 1) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
            /* The reference reads zeros past the end of a short file and goes on for as many lines as the header claims
             * (stream_read_bit, os_stream.c:36-52). A whole stream is never read more than the 22 bits of look-ahead past
             * its last byte, so further than that means the file was cut or its line count is wrong. */
            if (d.pos > (uint64_t)size + 8 && i + 1 < lines) { rc = qvz_job_fail(QVZ_E_DATA, "the stream ends before line %llu of %llu", (unsigned long long)(i + 2), (unsigned long long)lines); goto done; }
        }
    }
done:
    free(row); free(arena); free(off);
    return rc;
}

int qvz_decode_file(const char *in_path, const char *out_path, int device, qvz_result_t *res) {
    const uint8_t *f = NULL;
    size_t size = 0;
    qvz_qfile_t q;
    int rc;
    memset(&q, 0, sizeof(q));
    if ((rc = qvz_map_file(in_path, &f, &size))) return rc;
    if (qvz_shardset_is(f, size)) {                                             /* a shard set (include/qvz_shardset.h): all streams at once on the GPU */
        qvz_unmap_file(f, size);
        return qvz_shardset_decode(in_path, out_path, device, res);
    }
    if ((rc = qvz_qfile_parse(f, size, &q))) { qvz_unmap_file(f, size); return rc; }
    FILE *fo = fopen(out_path, "w");
    if (!fo) rc = qvz_job_fail(QVZ_E_ARG, "Unable to open output file");
    else {
        rc = qvz_decode_stream_host(q.K, q.C, q.lines, (const qvz_codebook_t *const *)q.cbs, q.well, f + q.stream_at, size - q.stream_at, fo, 0);
        fclose(fo);
    }
    if (!rc && res) { memset(res, 0, sizeof(*res)); res->lines = q.lines; res->columns = q.C; res->stream_bytes = size - q.stream_at; res->file_bytes = size; }
    qvz_unmap_file(f, size);
    qvz_qfile_release(&q);
    return rc;
}

/* S streams of one file (stream g = reads [first[g], first[g+1]) with qvz_shard_bounds(N, S)) decoded at once by the GPU
 * decoder and written as text. streams[g] points at sizes[g] bytes. */
static double dec_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* A file is decoded in SLABS of whole streams when its matrices - the draws and the quantized values, 2 C bytes per read - do not fit
 * the device at once (a file of 2^32 reads and more always: 1.3 TB at 150 columns): the streams of a slab are decoded together,
 * their text is appended, the next slab follows with the WELL generator advanced to its first read. 0 = by the device's free memory;
 * the setter is a test hook (slabs of a few thousand reads on small files must give the same text). */
static uint64_t g_slab_reads = 0;
static uint32_t g_slab_streams = 0;
void qvz_decode_set_slab_reads(uint64_t reads) { g_slab_reads = reads; g_slab_streams = 0; }
void qvz_decode_set_slab(uint64_t reads, uint32_t streams) { g_slab_reads = reads; g_slab_streams = streams; }

int qvz_decode_streams_gpu(uint32_t K, uint32_t C, uint64_t N, const qvz_codebook_t *const *cbs, const uint32_t *well,
                           uint32_t shards, const uint64_t *bounds, const uint8_t *const *streams, const uint64_t *sizes,
                           const char *out_path, int device) {
    int rc = QVZ_OK;
    uint8_t *h_packed = NULL, *d_packed = NULL, *d_in = NULL, *d_draws = NULL, *d_qv = NULL, *d_labels = NULL;
    uint64_t *first = NULL, *rel = NULL;
    FILE *fo = NULL;
    /* N and C come from a file somebody else wrote: every product below must be the number it looks like */
    if (!qvz_plausible_matrix(N, C) || shards == 0)
        return qvz_job_fail(QVZ_E_DATA, "implausible size: %llu lines of %u columns in %u streams", (unsigned long long)N, C, shards);
    first = (uint64_t *)malloc(((size_t)shards + 1) * sizeof(uint64_t));
    rel = (uint64_t *)malloc(((size_t)shards + 1) * sizeof(uint64_t));
    if (!first || !rel) { free(first); free(rel); return qvz_job_fail(QVZ_E_NOMEM, "host alloc"); }
    if (bounds) memcpy(first, bounds, ((size_t)shards + 1) * sizeof(uint64_t));
    else qvz_shard_bounds(N, shards, first);
    for (uint32_t g = 0; g < shards; ++g)
        if (first[g + 1] < first[g] || first[g + 1] > N) { free(first); free(rel); return qvz_job_fail(QVZ_E_DATA, "the streams' read ranges are not in order"); }
    if (qvz_hip_set_device(device)) { free(first); free(rel); return qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); }
    fo = fopen(out_path, "w");
    if (!fo) { free(first); free(rel); return qvz_job_fail(QVZ_E_ARG, "Unable to open output file"); }
    {
        /* The text's pages are made while the GPU decodes - but N and C are a header's claim until the streams have decoded, and
         * a small crafted file may claim 1e9 x 150: pages are made ahead only for a text the output's file system reports room
         * for. A file system that reports less (or nothing: some FUSE, network and quota-limited mounts) gets no pages made
         * ahead and the writes decide - a valid decode is not refused on a mount's word. (N * (C + 1) itself cannot wrap:
         * qvz_plausible_matrix above.) */
        const uint64_t text = N * ((uint64_t)C + 1);
        struct statvfs vfs;
        struct stat ost;
        if (fstat(fileno(fo), &ost) == 0 && S_ISREG(ost.st_mode) && fstatvfs(fileno(fo), &vfs) == 0 && vfs.f_blocks != 0 &&
            text <= (uint64_t)vfs.f_bavail * (uint64_t)vfs.f_frsize)
            qvz_prealloc_start(fo, text);
    }
    /* reads per slab: what two matrices of C bytes per read may take of the free memory (40 % of it; the streams and the labels
     * are small beside them), never 2^32 or more (the kernels' per-call read counts), at least one stream */
    uint64_t slab_reads = g_slab_reads;
    if (!slab_reads) {
        uint64_t free_b = 0;
        slab_reads = 1ull << 31;
        if (qvz_hip_mem_info(&free_b, NULL) == QVZ_OK && free_b / 5 * 2 / (2ull * C + 1) < slab_reads) slab_reads = free_b / 5 * 2 / (2ull * C + 1);
    }
    if (slab_reads >= (1ull << 32)) slab_reads = (1ull << 32) - 256;
    /* ... and streams per slab: every stream of a slab has its adaptive tables in device memory (0.9 MB at 150 columns): a third of
     * the free memory for them at most (a million short streams would otherwise ask for a terabyte) */
    uint32_t slab_streams = shards;
    {
        uint64_t free_b = 0;
        const uint64_t per = qvz_hip_decode_tables_bytes(C, K, cbs);
        if (per && qvz_hip_mem_info(&free_b, NULL) == QVZ_OK && free_b / 3 / per < slab_streams) slab_streams = (uint32_t)(free_b / 3 / per);
        if (g_slab_streams && g_slab_streams < slab_streams) slab_streams = g_slab_streams;
        if (slab_streams < 1) slab_streams = 1;
    }
#define HIPC(call) do { if ((call) != QVZ_OK) { rc = qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); goto done; } } while (0)
    double t_upload = 0.0, t_decode = 0.0, t_egress = 0.0;
    uint32_t nslabs = 0;
    for (uint32_t g0 = 0; g0 < shards;) {
        /* the slab: streams g0 .. g1 - 1 */
        uint32_t g1 = g0 + 1;
        while (g1 < shards && g1 - g0 < slab_streams && first[g1 + 1] - first[g0] <= slab_reads) ++g1;
        const uint32_t ns = g1 - g0;
        const uint64_t r0 = first[g0], n = first[g1] - r0, ld = qvz_hip_ld(n);
        uint64_t cap = 0, total = 0;
        int contiguous = 1;
        if (n >= (1ull << 32)) { rc = qvz_job_fail(QVZ_E_DATA, "stream %u: %llu reads from read %llu cannot be decoded in one piece", g0, (unsigned long long)n, (unsigned long long)r0); goto done; }
        for (uint32_t g = g0; g < g1; ++g) {
            if (sizes[g] > cap) cap = sizes[g];
            if (g + 1 < g1 && streams[g + 1] != streams[g] + sizes[g]) contiguous = 0;
            total += sizes[g];
            rel[g - g0] = first[g] - r0;
        }
        rel[ns] = n;
        cap = (cap + 7) & ~3ull;
        if (cap > (UINT64_MAX >> 1) / ns) { rc = qvz_job_fail(QVZ_E_DATA, "implausible stream sizes"); goto done; }
        const double t_0 = dec_now();
        /* the streams go up packed (they lie back to back in a shard set; the pieces of shard FILES are packed first) through
         * pinned slabs and are spread to the decoder's fixed-stride layout on the device */
        if (!contiguous) {
            h_packed = (uint8_t *)malloc(total ? total : 1);
            if (!h_packed) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
            for (uint64_t g = g0, at = 0; g < g1; at += sizes[g], ++g) memcpy(h_packed + at, streams[g], sizes[g]);
        }
        HIPC(qvz_hip_malloc((void **)&d_packed, total ? total : 1));
        HIPC(qvz_hip_malloc((void **)&d_in, (size_t)ns * cap));
        HIPC(qvz_hip_malloc((void **)&d_draws, (size_t)C * ld));
        HIPC(qvz_hip_well_draws(well, r0, n, C, d_draws, ld, NULL));            /* beside the upload; the generator advanced to the slab's first read */
        if ((rc = qvz_ingress_bytes(d_packed, contiguous ? streams[g0] : h_packed, total))) goto done;
        HIPC(qvz_hip_scatter_streams(d_packed, sizes + g0, ns, d_in, cap, NULL));
        qvz_hip_free(d_packed); d_packed = NULL;
        free(h_packed); h_packed = NULL;
        const double t_up = dec_now();
        HIPC(qvz_hip_malloc((void **)&d_qv, (size_t)C * ld));
        HIPC(qvz_hip_malloc((void **)&d_labels, ld));
        if (qvz_hip_decode_shards_bounds(d_in, cap, sizes + g0, n, C, K, cbs, ns, rel, d_draws, ld, d_qv, d_labels, NULL)) {
            rc = qvz_job_fail(QVZ_E_DATA, "%s", qvz_hip_last_error());
            goto done;
        }
        HIPC(qvz_hip_stream_sync(NULL));
        const double t_dec = dec_now();
        qvz_hip_free(d_draws); d_draws = NULL;
        qvz_hip_free(d_in); d_in = NULL;
        if ((rc = qvz_egress_text(fo, d_qv, n, C, ld, NULL))) goto done;        /* the text leaves in slabs: no N-sized host buffer */
        qvz_hip_free(d_qv); d_qv = NULL;
        qvz_hip_free(d_labels); d_labels = NULL;
        t_upload += t_up - t_0; t_decode += t_dec - t_up; t_egress += dec_now() - t_dec;
        nslabs += 1;
        g0 = g1;
    }
#undef HIPC
    if (getenv("QVZ_TIMING"))                                                  /* where the wall time went, one line on stderr */
        fprintf(stderr, "qvz_timing: decode upload=%.4f decode=%.4f egress=%.4f lines=%llu columns=%u streams=%u slabs=%u\n", t_upload, t_decode, t_egress,
                (unsigned long long)N, C, shards, nslabs);
done:
    qvz_prealloc_finish(fo);
    if (fo) fclose(fo);
    qvz_hip_free(d_packed); qvz_hip_free(d_in); qvz_hip_free(d_draws); qvz_hip_free(d_qv); qvz_hip_free(d_labels);
    free(h_packed); free(first); free(rel);
    return rc;
}

int qvz_decode_shard_files(const char *prefix, uint32_t shards, const char *out_path, int device, qvz_result_t *res) {
    int rc = QVZ_OK;
    uint8_t **files = NULL;
    const uint8_t **streams = NULL;
    size_t *fsize = NULL;
    uint64_t *sizes = NULL, *first = NULL;
    qvz_qfile_t q0;
    memset(&q0, 0, sizeof(q0));
    if (!shards) return qvz_job_fail(QVZ_E_ARG, "shards must be positive");
    files = (uint8_t **)calloc(shards, sizeof(*files));
    streams = (const uint8_t **)calloc(shards, sizeof(*streams));
    fsize = (size_t *)calloc(shards, sizeof(*fsize));
    sizes = (uint64_t *)calloc(shards, sizeof(*sizes));
    first = (uint64_t *)calloc((size_t)shards + 1, sizeof(*first));
    if (!files || !streams || !fsize || !sizes || !first) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    uint64_t N = 0;
    size_t head = 0;
    for (uint32_t g = 0; g < shards; ++g) {
        char name[4096];
        if ((size_t)snprintf(name, sizeof(name), "%s.%04u", prefix, g) >= sizeof(name)) { rc = qvz_job_fail(QVZ_E_ARG, "name too long"); goto done; }
        if ((rc = qvz_read_whole_file(name, &files[g], &fsize[g]))) goto done;
        if (g == 0) {
            if ((rc = qvz_qfile_parse(files[0], fsize[0], &q0))) goto done;
            head = q0.stream_at;
        } else {                                                                /* same codebooks in every shard: compare the bytes */
            if (fsize[g] < head || memcmp(files[g] + 9, files[0] + 9, head - 128 - 9) != 0 || files[g][0] != files[0][0] ||
                memcmp(files[g] + 1, files[0] + 1, 4) != 0) { rc = qvz_job_fail(QVZ_E_DATA, "%s does not carry the codebooks of shard 0", name); goto done; }
        }
        const uint64_t lines = ((uint32_t)files[g][5] << 24) | ((uint32_t)files[g][6] << 16) | ((uint32_t)files[g][7] << 8) | files[g][8];
        first[g] = N;
        N += lines;
        sizes[g] = fsize[g] - head;
        streams[g] = files[g] + head;
    }
    first[shards] = N;
    /* The cut is the one the files' headers state (whatever formula made it: an earlier revision dealt ceil(N / S) reads
     * rounded up to 4, qvz_shard_bounds deals groups of 4); what the decoder needs is that every shard but the last holds
     * whole WELL words, i.e. starts on a multiple of 4 reads - qvz_hip_decode_shards_bounds checks that. */
    /* shard 0 carries the file's WELL state */
    rc = qvz_decode_streams_gpu(q0.K, q0.C, N, (const qvz_codebook_t *const *)q0.cbs, q0.well, shards, first, streams, sizes, out_path, device);
    if (!rc && res) { memset(res, 0, sizeof(*res)); res->lines = N; res->columns = q0.C; for (uint32_t g = 0; g < shards; ++g) { res->stream_bytes += sizes[g]; res->file_bytes += fsize[g]; } }
done:
    if (files) for (uint32_t g = 0; g < shards; ++g) free(files[g]);
    free(files); free(streams); free(fsize); free(sizes); free(first);
    qvz_qfile_release(&q0);
    return rc;
}
