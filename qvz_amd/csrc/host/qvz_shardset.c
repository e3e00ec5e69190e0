/* qvz_shardset.c - the one-file form of a sharded encode (include/qvz_shardset.h): writer, reader, the splitter that
 * materialises reference-format shard files, and the decode entry `qvz -x` takes for such a file. Host C; the GPU is used
 * for the WELL jump-ahead of the splitter and for decoding all streams at once. */
#include "qvz_shardset.h"

#include <stdlib.h>
#include <string.h>

#include "qvz_coder.h"
#include "qvz_job.h"

static void put_u32(uint8_t *p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
static void put_u64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
static uint32_t get_u32(const uint8_t *p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[i] << (8 * i); return v; }
static uint64_t get_u64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

int qvz_shardset_is(const uint8_t *buf, size_t size) {
    return buf && size >= QVZ_SHARDSET_FIXED && memcmp(buf, QVZ_SHARDSET_MAGIC, 8) == 0;
}

int qvz_shardset_write_head(FILE *f, uint32_t clusters, uint32_t columns, uint64_t lines, uint32_t shards,
                            const uint32_t *well_state, const qvz_codebook_t *const *cbs, const uint64_t *sizes,
                            qvz_shardset_info_t *info) {
    if (!f || !clusters || !columns || !shards || !well_state || !cbs || !sizes) return qvz_job_fail(QVZ_E_ARG, "bad shard-set arguments");
    const size_t hbytes = QVZ_SHARDSET_FIXED + (size_t)shards * 8;
    uint8_t *head = (uint8_t *)calloc(1, hbytes);
    uint8_t *cbuf = NULL;
    int rc = QVZ_OK;
    uint64_t cb_total = 0, st_total = 0;
    if (!head) return qvz_job_fail(QVZ_E_NOMEM, "host alloc");
    memcpy(head, QVZ_SHARDSET_MAGIC, 8);
    put_u32(head + 8, clusters);
    put_u32(head + 12, columns);
    put_u64(head + 16, lines);
    put_u32(head + 24, shards);
    memcpy(head + 32, well_state, 128);                                         /* native little-endian words, as qv_stream.c:89 */
    for (uint32_t g = 0; g < shards; ++g) put_u64(head + QVZ_SHARDSET_FIXED + (size_t)g * 8, sizes[g]);
    if (fwrite(head, 1, hbytes, f) != hbytes) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
    for (uint32_t j = 0; j < clusters; ++j) {                                   /* write_codebook, codebook.c:505-555 */
        const size_t sz = qvz_codebook_write(NULL, cbs[j]);
        cbuf = (uint8_t *)malloc(sz);
        if (!cbuf) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        qvz_codebook_write(cbuf, cbs[j]);
        const size_t w = fwrite(cbuf, 1, sz, f);
        free(cbuf); cbuf = NULL;
        if (w != sz) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
        cb_total += sz;
    }
    for (uint32_t g = 0; g < shards; ++g) st_total += sizes[g];
    if (info) {
        memset(info, 0, sizeof(*info));
        info->clusters = clusters; info->columns = columns; info->shards = shards; info->lines = lines;
        info->header_bytes = hbytes; info->codebook_bytes = cb_total; info->stream_bytes = st_total;
        info->file_bytes = hbytes + cb_total + st_total;
    }
done:
    free(head);
    return rc;
}

int qvz_shardset_write(FILE *f, uint32_t clusters, uint32_t columns, uint64_t lines, uint32_t shards,
                       const uint32_t *well_state, const qvz_codebook_t *const *cbs, const uint8_t *streams,
                       uint64_t stride, const uint64_t *sizes, qvz_shardset_info_t *info) {
    qvz_shardset_info_t si;
    if (!streams) return qvz_job_fail(QVZ_E_ARG, "bad shard-set arguments");
    const int rc = qvz_shardset_write_head(f, clusters, columns, lines, shards, well_state, cbs, sizes, &si);
    if (rc) return rc;
    if (stride == 0) {                                                          /* packed: one write */
        if (si.stream_bytes && fwrite(streams, 1, si.stream_bytes, f) != si.stream_bytes) return qvz_job_fail(QVZ_E_ARG, "write failed");
    } else {
        for (uint32_t g = 0; g < shards; ++g)
            if (sizes[g] && fwrite(streams + (size_t)g * stride, 1, sizes[g], f) != sizes[g]) return qvz_job_fail(QVZ_E_ARG, "write failed");
    }
    if (info) *info = si;
    return QVZ_OK;
}

/* a shard set in memory: header fields, parsed codebooks, where the streams lie */
typedef struct {
    uint32_t K, C, S;
    uint64_t N;
    uint32_t well[32];
    uint64_t *sizes, *at;          /* [S] length and file offset of every stream */
    qvz_codebook_t **cbs;
    size_t cb_at, cb_bytes;
} sset_t;

static void sset_release(sset_t *t) {
    if (t->cbs) { for (uint32_t j = 0; j < t->K; ++j) qvz_codebook_free(t->cbs[j]); free(t->cbs); }
    free(t->sizes); free(t->at);
    memset(t, 0, sizeof(*t));
}

static int sset_parse(const uint8_t *f, size_t size, sset_t *t) {
    memset(t, 0, sizeof(*t));
    if (!qvz_shardset_is(f, size)) return qvz_job_fail(QVZ_E_DATA, "not a qvz shard set");
    t->K = get_u32(f + 8); t->C = get_u32(f + 12); t->N = get_u64(f + 16); t->S = get_u32(f + 24);
    if (t->K < 1 || t->K > 255 || t->C < 1 || t->C > (1u << 20) || t->S < 1 || t->S > qvz_shard_limit(t->N) || t->N == 0)
        return qvz_job_fail(QVZ_E_DATA, "implausible shard-set header (%u clusters, %u columns, %llu lines, %u shards)", t->K, t->C, (unsigned long long)t->N, t->S);
    /* the line count is 64 bits of somebody else's file: N * (C + 1), C * ld and first_read * C are computed from it later */
    if (!qvz_plausible_matrix(t->N, t->C))
        return qvz_job_fail(QVZ_E_DATA, "implausible shard-set header: %llu lines of %u columns", (unsigned long long)t->N, t->C);
    if ((uint64_t)size < QVZ_SHARDSET_FIXED + (uint64_t)t->S * 8) return qvz_job_fail(QVZ_E_DATA, "shard set ends inside its size table");
    memcpy(t->well, f + 32, 128);
    t->sizes = (uint64_t *)calloc(t->S, sizeof(uint64_t));
    t->at = (uint64_t *)calloc(t->S, sizeof(uint64_t));
    if (!t->sizes || !t->at) { sset_release(t); return qvz_job_fail(QVZ_E_NOMEM, "host alloc"); }
    size_t pos = QVZ_SHARDSET_FIXED + (size_t)t->S * 8;
    t->cb_at = pos;
    const int rc = qvz_codebooks_parse(f, size, &pos, t->K, t->C, &t->cbs);
    if (rc) { sset_release(t); return rc; }
    t->cb_bytes = pos - t->cb_at;
    uint64_t at = pos;
    for (uint32_t g = 0; g < t->S; ++g) {
        t->sizes[g] = get_u64(f + QVZ_SHARDSET_FIXED + (size_t)g * 8);
        t->at[g] = at;
        if (t->sizes[g] > (uint64_t)size - at) { sset_release(t); return qvz_job_fail(QVZ_E_DATA, "shard set ends inside stream %u", g); }
        at += t->sizes[g];
    }
    return QVZ_OK;
}

int qvz_shardset_stat(const char *path, qvz_shardset_info_t *info) {
    uint8_t *f = NULL;
    size_t size = 0;
    sset_t t;
    int rc = qvz_read_whole_file(path, &f, &size);
    if (rc) return rc;
    if ((rc = sset_parse(f, size, &t))) { free(f); return rc; }
    memset(info, 0, sizeof(*info));
    info->clusters = t.K; info->columns = t.C; info->shards = t.S; info->lines = t.N;
    info->header_bytes = QVZ_SHARDSET_FIXED + (uint64_t)t.S * 8;
    info->codebook_bytes = t.cb_bytes;
    for (uint32_t g = 0; g < t.S; ++g) info->stream_bytes += t.sizes[g];
    info->file_bytes = size;
    sset_release(&t);
    free(f);
    return QVZ_OK;
}

int qvz_shardset_split(const char *path, const char *prefix, uint32_t first_shard, uint32_t count, int device) {
    uint8_t *f = NULL, *image = NULL;
    size_t size = 0;
    sset_t t;
    uint64_t *first = NULL, *jump = NULL;
    uint32_t *wstates = NULL;
    int rc = qvz_read_whole_file(path, &f, &size);
    if (rc) return rc;
    if ((rc = sset_parse(f, size, &t))) { free(f); return rc; }
    if (first_shard >= t.S) { rc = qvz_job_fail(QVZ_E_ARG, "shard %u of %u", first_shard, t.S); goto done; }
    if (count == 0 || count > t.S - first_shard) count = t.S - first_shard;
    first = (uint64_t *)malloc(((size_t)t.S + 1) * sizeof(uint64_t));
    jump = (uint64_t *)malloc((size_t)count * sizeof(uint64_t));
    wstates = (uint32_t *)malloc((size_t)count * 128);
    if (!first || !jump || !wstates) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    qvz_shard_bounds(t.N, t.S, first);
    for (uint32_t i = 0; i < count; ++i) jump[i] = first[first_shard + i] * t.C / 4;   /* words before the shard's first draw */
    if (qvz_hip_set_device(device) || qvz_hip_well_jump_many(t.well, jump, count, wstates)) { rc = qvz_job_fail(QVZ_E_HIP, "%s", qvz_hip_last_error()); goto done; }
    {
        size_t img_cap = 0;
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t g = first_shard + i;
            const uint64_t reads = first[g + 1] - first[g];
            char name[4096];
            if (reads >= (1ull << 32)) { rc = qvz_job_fail(QVZ_E_ARG, "shard %u has %llu lines: more than a reference header can name (codebook.c:482)", g, (unsigned long long)reads); goto done; }
            if ((size_t)snprintf(name, sizeof(name), "%s.%04u", prefix, g) >= sizeof(name)) { rc = qvz_job_fail(QVZ_E_ARG, "output name too long"); goto done; }
            const size_t need = qvz_shard_file_image(NULL, t.K, t.C, reads, (const qvz_codebook_t *const *)t.cbs, wstates + (size_t)i * 32, NULL, t.sizes[g]);
            if (need > img_cap) {
                free(image);
                image = (uint8_t *)malloc(need);
                img_cap = need;
                if (!image) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
            }
            qvz_shard_file_image(image, t.K, t.C, reads, (const qvz_codebook_t *const *)t.cbs, wstates + (size_t)i * 32, f + t.at[g], t.sizes[g]);
            FILE *fo = fopen(name, "wb");
            if (!fo) { rc = qvz_job_fail(QVZ_E_ARG, "Unable to open output file %s", name); goto done; }
            const size_t w = fwrite(image, 1, need, fo);
            fclose(fo);
            if (w != need) { rc = qvz_job_fail(QVZ_E_ARG, "write failed"); goto done; }
        }
    }
done:
    free(first); free(jump); free(wstates); free(image);
    sset_release(&t);
    free(f);
    return rc;
}

int qvz_shardset_decode(const char *path, const char *out_path, int device, qvz_result_t *res) {
    const uint8_t *f = NULL;
    size_t size = 0;
    sset_t t;
    const uint8_t **streams = NULL;
    int rc = qvz_map_file(path, &f, &size);                                     /* mapped, not copied: the streams go page cache -> pinned slabs -> device */
    if (rc) return rc;
    if ((rc = sset_parse(f, size, &t))) { qvz_unmap_file(f, size); return rc; }
    streams = (const uint8_t **)calloc(t.S, sizeof(*streams));
    if (!streams) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    for (uint32_t g = 0; g < t.S; ++g) streams[g] = f + t.at[g];
    rc = qvz_decode_streams_gpu(t.K, t.C, t.N, (const qvz_codebook_t *const *)t.cbs, t.well, t.S, NULL, streams, t.sizes, out_path, device);
    if (!rc && res) {
        memset(res, 0, sizeof(*res));
        res->lines = t.N; res->columns = t.C; res->file_bytes = size;
        for (uint32_t g = 0; g < t.S; ++g) res->stream_bytes += t.sizes[g];
    }
done:
    free(streams);
    sset_release(&t);
    qvz_unmap_file(f, size);
    return rc;
}
