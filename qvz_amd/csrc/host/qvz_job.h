/* qvz_job.h - the encode path as resumable stages (internal to the host C layer).
 * qvz_encode_buffer (include/qvz_encode.h) runs them back to back; the compat layer (include/qvz_compat.h)
 * calls one stage per reference entry point (do_kmeans_clustering, calculate_statistics, ...). */
#ifndef QVZ_JOB_H
#define QVZ_JOB_H

#include <stdint.h>
#include <stdio.h>

#include "qvz_codebook.h"
#include "qvz_encode.h"

#define QVZ_LINES_PER_BLOCK 1000000ull   /* MAX_LINES_PER_BLOCK, reference include/lines.h:12 */

/* Device blocks that outlive a job (include/qvz_encode.h, qvz_resident_*): a process that encodes one resident matrix after another
 * keeps its big buffers - labels, draws, the quantizer's planes, the stream area, the record workspace - instead of returning 100+ GB
 * to the driver and asking for them again every time. A block is found by its role; one that is too small is replaced. */
enum { QVZ_POOL_LABELS = 1, QVZ_POOL_DRAWS, QVZ_POOL_QI, QVZ_POOL_ST, QVZ_POOL_QV, QVZ_POOL_ERR, QVZ_POOL_OUT, QVZ_POOL_PACKED, QVZ_POOL_WORK, QVZ_POOL_ROLES };
typedef struct qvz_pool {
    void *ptr[QVZ_POOL_ROLES];
    uint64_t bytes[QVZ_POOL_ROLES];
    /* side streams, made once and probed for running beside the default stream (and one another): the WELL generator's, and one per
     * cluster beyond the first for the codebook generations. The runtime spreads streams over a few hardware queues; a fresh stream
     * per encode lands on the default stream's queue every few encodes and its work then runs in turn instead of beside */
    void *well_stream, *cb_stream[8];
} qvz_pool_t;

typedef struct qvz_job {
    /* input view: rows of C+1 bytes (Q+33 ... '\n') in file order */
    const uint8_t *text;
    uint64_t N;
    uint32_t C, K;
    qvz_options_t opts;
    /* host-visible results of the stages */
    uint64_t centroid_lines[256];
    uint32_t kmeans_iterations;
    uint8_t *h_labels;          /* [N]      after qvz_job_kmeans (when requested)          */
    uint8_t *h_means;           /* [K][C]   symbols (byte-33)                              */
    uint32_t *h_counts;         /* [K]                                                       */
    uint64_t *h_acc;            /* [K][C]   symbol sums of the last iteration                */
    qvz_codebook_t **cbs;       /* [K]      after qvz_job_codebooks                          */
    uint64_t deferred_decisions;
    uint32_t well_state[32];
    double h_dist[QVZ_A * QVZ_A];
    /* device state */
    uint64_t ld;
    uint8_t *d_sym, *d_labels, *d_means;
    uint32_t *d_cnt, *d_counts;
    uint64_t *d_acc;
    uint64_t *d_cnt64, *d_counts64;   /* a part of a file: this part's counters widened and summed over all parts (u32 sums could wrap) */
    double *d_moved;
    int32_t *d_kstat;
    int uploaded, clustered, counted;
    /* ---- a job over a matrix that is in HBM already (qvz_resident_encode); all zero otherwise ---- */
    qvz_pool_t *pool;           /* the big device buffers come from here and go back there                                       */
    int sym_borrowed;           /* d_sym is the caller's                                                                          */
    int keep_device;            /* the sharded compress leaves its products on the device: no file, no host copy                  */
    uint8_t *r_packed, *r_qv;   /* the streams packed back to back (made on request) / the quantized values (pool blocks)         */
    uint8_t *r_out;             /* the streams where the coder wrote them: stream g at r_out + g * r_cap                           */
    uint64_t r_cap;
    int r_out_in_work;          /* ... inside the record workspace (behind the records)                                           */
    uint64_t *r_sizes, r_total; /* malloc'd [shards]                                                                              */
    double k_kmeans_ms, k_pmf_ms, k_quantize_ms;   /* busy time of the kernels (HIP events on their stream; k-means: all iterations) */
    double *d_errsum;           /* [1025] this part's distortion sum (qvz_hip_sum_errors)                                          */
    double r_errsum;
    int keep_resident;          /* a sweep: the sharded compress leaves the matrix and the draws for the next operating point */
    double t_upload, t_cluster, t_stats, t_codebook, t_encode;
    double t_quantize, t_coder, t_egress;   /* inside t_encode of a sharded encode: WELL wait + quantizer, coder kernels, what leaves the device afterwards */
    /* WELL draws of the whole part, started on a side stream before the codebooks are generated (they depend on the seed
     * only; qvz_job_start_draws), waited for by the quantizer */
    uint8_t *d_draws;
    void *well_stream;
    int well_chosen;
    /* record workspace of the GPU coder, allocated ahead of time by a host thread of its own while the text is still
     * being uploaded (qvz_job_start_workspace): a one-shot process cannot keep it between encodes as bench.py does, and a
     * 96 GiB hipMalloc in front of the coder kernels is time on the critical path */
    void *d_work;
    uint64_t work_bytes;
    uint32_t work_shards;
    /* ... and, by the same thread, the quantizer's planes (coder index and state, the quantized values when the caller wants the -u
     * text, the per-read errors) and the draw matrix: 45-60 GB at 100M x 150 whose hipMalloc calls otherwise sit in front of the
     * quantizer and the WELL generator - where a process that starts soon after another has freed its HBM was seen to stall for up
     * to a second, whichever large allocation came first (want_text: set by the caller before qvz_job_start_workspace) */
    void *d_planes;
    uint64_t planes_bytes;
    int want_text;
    uint8_t *d_draws_early;           /* the draw matrix: allocated first, an allocation of its own (given back after the quantizer) */
    int draws_early_state;            /* 0 none, 1 being allocated, 2 decided: qvz_job_start_draws waits for this, not for the thread */
    int work_running, work_rc;
    unsigned long work_thread;        /* pthread_t */
    /* ---- a job that holds PART of a file (one device of a multi-device encode, qvz_multi.c); all zero otherwise ---- */
    uint64_t first_read;        /* global number of this job's first read: WELL offset of its draws (a multiple of 4)         */
    const uint8_t *centroid_rows;  /* [K][C+1] text rows of the k centroid reads, wherever in the file they lie (cluster.c:198-201) */
    /* sum of a device buffer over all parts (count elements of width 4 or 8 bytes, unsigned), in place, on `stream`:
     * the k-means accumulators and counts per iteration, the PMF counters once */
    int (*allreduce)(void *ctx, void *d_buf, uint64_t count, int width, void *stream);
    void *allreduce_ctx;
    const uint64_t *shard_bounds;  /* [shards + 1] this part's cut into shards, relative to its first read (NULL: qvz_shard_bounds) */
    /* sharded compress hands its products back instead of writing files when `collect` is set */
    int collect;
    uint8_t *out_streams;       /* malloc'd: the part's streams packed back to back                                            */
    uint64_t *out_sizes;        /* malloc'd: [shards]                                                                          */
    double *out_err;            /* malloc'd: [N] per-read error sums, for the file-order distortion sum (qv_compressor.c:127)  */
} qvz_job_t;

/* Draws the centroid reads with libc rand() (cluster.c:198-204). Must run before the first HIP call of the process:
 * HIP runtime start-up consumes rand() values. No device work. */
int qvz_job_init(qvz_job_t *job, const uint8_t *text, uint64_t bytes_or_zero, uint64_t lines, uint32_t columns,
                 const qvz_options_t *opts);
/* A job over PART of a file: reads [first_read, first_read + lines) at `text`; no rand() draws (the whole file's centroid
 * reads come as centroid_rows, K rows of columns + 1 bytes). The caller sets allreduce / shard_bounds / collect. */
int qvz_job_init_part(qvz_job_t *job, const uint8_t *text, uint64_t lines, uint32_t columns, const qvz_options_t *opts,
                      uint64_t first_read, const uint8_t *centroid_rows);
int qvz_job_upload(qvz_job_t *job);                       /* text -> column-major symbols in HBM            */
int qvz_job_kmeans(qvz_job_t *job, int want_host_labels); /* do_kmeans_clustering, cluster.c:212-244        */
int qvz_job_stats(qvz_job_t *job);                        /* counting loop of calculate_statistics           */
int qvz_job_codebooks(qvz_job_t *job);                    /* generate_codebooks                              */
int qvz_job_write_codebooks(qvz_job_t *job, FILE *fout, uint64_t *bytes);   /* write_codebooks             */
/* start_qv_compression: writes the WELL state and the stream to fout, the -u text to fu (may be NULL) */
int qvz_job_compress(qvz_job_t *job, FILE *fout, FILE *fu, uint64_t *stream_bytes, double *distortion);
/* the same for sharded output: `shards` reference-format files "<prefix>.<g>" coded by the GPU coder */
int qvz_job_compress_sharded(qvz_job_t *job, const char *prefix, FILE *fu, uint32_t shards, uint64_t *stream_bytes,
                             double *distortion);
/* the same streams as ONE shard-set file (include/qvz_shardset.h): codebooks stored once */
int qvz_job_compress_shardset(qvz_job_t *job, FILE *fout, FILE *fu, uint32_t shards, uint64_t *stream_bytes,
                              uint64_t *file_bytes, double *distortion);
/* Chooses the WELL state (qv_stream.c:70-89) and starts generating the part's draw matrix on a side stream; called before
 * qvz_job_codebooks so that the generator runs beside it (the sharded compress calls it itself otherwise). */
int qvz_job_start_draws(qvz_job_t *job);
/* starts allocating the coder's record workspace for `shards` streams on a thread of its own; the sharded compress waits for it */
int qvz_job_start_workspace(qvz_job_t *job, uint32_t shards);
void qvz_job_release(qvz_job_t *job);

/* ---- qvz_egress.c: products leaving the device in slabs (pinned double buffers, threaded file writes) ---- */
/* `after`: a stream whose work so far must be finished before the device data is read - waited for on the device,
 * not by the host; NULL: the data is complete already */
int qvz_egress_text(FILE *fu, const uint8_t *d_qv, uint64_t N, uint32_t C, uint64_t ld, void *after);   /* -u rows at fu's position */
int qvz_prealloc_start(FILE *f, uint64_t bytes);    /* a thread makes the pages of the next `bytes` of f ahead of qvz_egress_*; 1 if started */
int qvz_host_write_slabs(FILE *f, const uint8_t *buf, uint64_t bytes, uint64_t estimate, uint64_t slab);   /* a host buffer the way the slab pipelines write */
void qvz_prealloc_finish(FILE *f);                  /* ... and must be ended before f is closed (no-op if none was started) */
int qvz_egress_bytes(FILE *f, const uint8_t *d_src, uint64_t bytes, void *after);
int qvz_sum_errors(const double *d_err, uint64_t N, uint32_t C, double *sum_out, double *keep, void *after);
int qvz_parallel_pwrite(int fd, const uint8_t *buf, size_t n, int64_t off);
void qvz_parallel_memcpy(uint8_t *dst, const uint8_t *src, size_t n);
int qvz_ingress_bytes(uint8_t *d_dst, const uint8_t *h_src, uint64_t bytes);

/* ---- decode side, shared by qvz_decode.c, qvz_shardset.c and qvz_compat.c ---- */
typedef struct qvz_qfile {
    uint32_t K, C;
    uint64_t lines;
    qvz_codebook_t **cbs;
    uint32_t well[32];
    size_t stream_at;
} qvz_qfile_t;
int qvz_qfile_parse(const uint8_t *f, size_t size, qvz_qfile_t *q);       /* header + codebooks + WELL state of a .q image */
void qvz_qfile_release(qvz_qfile_t *q);
int qvz_codebooks_parse(const uint8_t *f, size_t size, size_t *pos, uint32_t K, uint32_t C, qvz_codebook_t ***out);
int qvz_read_whole_file(const char *path, uint8_t **buf, size_t *size);
int qvz_map_file(const char *path, const uint8_t **buf, size_t *size);      /* read-only mapping of a whole file */
void qvz_unmap_file(const uint8_t *buf, size_t size);
/* start_qv_decompression (qv_compressor.c:145-231) over memory: the serial chain and the WELL draws on one host core */
int qvz_decode_stream_host(uint32_t K, uint32_t C, uint64_t lines, const qvz_codebook_t *const *cbs, const uint32_t *well,
                           const uint8_t *stream, size_t size, FILE *fo, int verbose);
/* S streams of one file decoded at once on the GPU, written as text to out_path; stream g holds the reads
 * [bounds[g], bounds[g+1]) (NULL: qvz_shard_bounds(N, S)) */
int qvz_decode_streams_gpu(uint32_t K, uint32_t C, uint64_t N, const qvz_codebook_t *const *cbs, const uint32_t *well,
                           uint32_t shards, const uint64_t *bounds, const uint8_t *const *streams, const uint64_t *sizes,
                           const char *out_path, int device);
/* Sizes taken from a file somebody else wrote: 1 when N lines of C columns give products - N * (C + 1), C * ld(N),
 * first_read * C - that fit 2^48 (256 TB of text: beyond any device or host this runs on, far below a 64-bit wrap) */
static inline int qvz_plausible_matrix(uint64_t N, uint32_t C) {
    return N != 0 && C != 0 && C <= (1u << 20) && N <= (1ull << 48) / ((uint64_t)C + 1);
}

int qvz_job_fail(int code, const char *fmt, ...);

#endif
