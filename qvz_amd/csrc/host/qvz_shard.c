/* qvz_shard.c - sharded coding: independent reference-format streams over contiguous read ranges, coded by a
 * pool of host threads that stream the (quantizer index, state) feed out of HBM (include/qvz_coder.h). */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_coder.h"

void qvz_shard_bounds(uint64_t n, uint32_t shards, uint64_t *first) {
    /* Shares are dealt in units of 4 reads (4 reads x C draws = C whole WELL words), shard g starting at group
     * floor(groups * g / shards): with shards <= groups no shard is empty (a reference-format file of 0 lines makes the
     * reference decoder loop over 2^32 lines, qv_compressor.c:166); the reads past the last whole group go to the last shard. */
    if (shards == 0) shards = 1;
    const uint64_t groups = n / 4;
    for (uint32_t g = 0; g < shards; ++g)
        first[g] = 4 * (uint64_t)(((unsigned __int128)groups * g) / shards);
    first[shards] = n;
}

uint32_t qvz_shard_limit(uint64_t n) {                       /* most shards n reads can be cut into without an empty one */
    const uint64_t groups = n / 4;
    return groups < 1 ? 1u : groups > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)groups;
}

typedef struct {
    const uint16_t *d_feed;
    const uint8_t *d_labels;
    uint32_t columns, clusters, shards, threads, tid;
    int device;
    const qvz_codebook_t *const *cbs;
    const uint64_t *first;
    uint8_t **streams;
    uint64_t *sizes;
    int rc;
} worker_t;

#define SLICE_READS (1u << 16)

static void *worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    void *stream = NULL;
    uint16_t *h_feed[2] = {NULL, NULL};
    uint8_t *h_lab[2] = {NULL, NULL};
    const uint32_t C = w->columns;
    w->rc = 0;
    if (qvz_hip_set_device(w->device) || qvz_hip_stream_create(&stream)) { w->rc = QVZ_E_HIP; return NULL; }
    for (int b = 0; b < 2; ++b)
        if (qvz_hip_malloc_host((void **)&h_feed[b], (size_t)SLICE_READS * C * sizeof(uint16_t)) ||
            qvz_hip_malloc_host((void **)&h_lab[b], SLICE_READS)) { w->rc = QVZ_E_NOMEM; goto out; }

    for (uint32_t g = w->tid; g < w->shards; g += w->threads) {
        const uint64_t r0 = w->first[g], r1 = w->first[g + 1];
        qvz_coder_t *cd = qvz_coder_create(w->clusters, C, w->cbs);
        if (!cd) { w->rc = QVZ_E_NOMEM; goto out; }
        /* double-buffered: slice i+1 is copied while slice i is coded */
        uint64_t next = r0;
        int cur = 0;
        uint64_t n_cur = 0, n_next = 0;
#define FETCH(buf, count)                                                                                      \
    do {                                                                                                       \
        (count) = r1 - next < SLICE_READS ? r1 - next : SLICE_READS;                                           \
        if (count) {                                                                                           \
            if (qvz_hip_copy_d2h(h_feed[buf], w->d_feed + next * C, (count) * C * sizeof(uint16_t), stream) ||  \
                (w->d_labels && qvz_hip_copy_d2h(h_lab[buf], w->d_labels + next, (count), stream))) {           \
                w->rc = QVZ_E_HIP;                                                                             \
            }                                                                                                  \
            next += (count);                                                                                   \
        }                                                                                                      \
    } while (0)
        FETCH(cur, n_cur);
        while (n_cur && !w->rc) {
            if (qvz_hip_stream_sync(stream)) { w->rc = QVZ_E_HIP; break; }
            FETCH(cur ^ 1, n_next);
            if (qvz_coder_encode(cd, w->d_labels ? h_lab[cur] : NULL, h_feed[cur], n_cur)) { w->rc = QVZ_E_DATA; break; }
            cur ^= 1;
            n_cur = n_next;
        }
#undef FETCH
        qvz_hip_stream_sync(stream);
        if (!w->rc) {
            qvz_coder_finish(cd);
            uint64_t sz;
            const uint8_t *data = qvz_coder_data(cd, &sz);
            w->streams[g] = (uint8_t *)malloc(sz ? sz : 1);
            if (!w->streams[g]) w->rc = QVZ_E_NOMEM;
            else { memcpy(w->streams[g], data, sz); w->sizes[g] = sz; }
        }
        qvz_coder_destroy(cd);
        if (w->rc) break;
    }
out:
    for (int b = 0; b < 2; ++b) { qvz_hip_free_host(h_feed[b]); qvz_hip_free_host(h_lab[b]); }
    if (stream) qvz_hip_stream_destroy(stream);
    return NULL;
}

int qvz_encode_shards_device(const uint16_t *d_feed, const uint8_t *d_labels, uint64_t n, uint32_t columns,
                             uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards, uint32_t threads,
                             int device, uint8_t **streams, uint64_t *sizes) {
    if (!d_feed || !n || !columns || !clusters || !shards || !streams || !sizes) return QVZ_E_ARG;
    if (threads == 0) threads = 1;
    if (threads > shards) threads = shards;
    uint64_t *first = (uint64_t *)malloc((shards + 1) * sizeof(uint64_t));
    worker_t *ws = (worker_t *)calloc(threads, sizeof(worker_t));
    pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
    if (!first || !ws || !th) { free(first); free(ws); free(th); return QVZ_E_NOMEM; }
    qvz_shard_bounds(n, shards, first);
    memset(streams, 0, shards * sizeof(uint8_t *));
    memset(sizes, 0, shards * sizeof(uint64_t));
    for (uint32_t t = 0; t < threads; ++t) {
        ws[t] = (worker_t){d_feed, d_labels, columns, clusters, shards, threads, t, device, cbs, first, streams, sizes, 0};
        if (pthread_create(&th[t], NULL, worker, &ws[t])) { ws[t].rc = QVZ_E_NOMEM; th[t] = 0; }
    }
    int rc = 0;
    for (uint32_t t = 0; t < threads; ++t) {
        if (th[t]) pthread_join(th[t], NULL);
        if (ws[t].rc && !rc) rc = ws[t].rc;
    }
    if (rc)
        for (uint32_t g = 0; g < shards; ++g) { free(streams[g]); streams[g] = NULL; }
    free(first); free(ws); free(th);
    return rc;
}

size_t qvz_shard_file_image(uint8_t *buf, uint32_t clusters, uint32_t columns, uint64_t shard_reads,
                            const qvz_codebook_t *const *cbs, const uint32_t *well_state_at_shard, const uint8_t *stream,
                            uint64_t stream_bytes) {
    size_t pos = qvz_codebook_write_header(buf, clusters, columns, shard_reads);
    for (uint32_t j = 0; j < clusters; ++j) pos += qvz_codebook_write(buf ? buf + pos : NULL, cbs[j]);
    if (buf) memcpy(buf + pos, well_state_at_shard, 128);
    pos += 128;
    if (buf) memcpy(buf + pos, stream, stream_bytes);
    return pos + stream_bytes;
}
