/* qvz_encode.h - host C driver of the MI355X encode path: the counterpart of encode() in the reference's
 * src/main.c:18-127, calling the HIP kernels through include/qvz_hip.h / qvz_codebook.h and running the serial
 * arithmetic coder (include/qvz_coder.h) on the host, overlapped with the GPU.
 *
 * Output files are byte-identical to the reference's for the same 128-byte WELL state:
 *   .q  = 9-byte header + codebooks (src/codebook.c:474-555) + WELL state (src/qv_stream.c:89) + stream
 *   -u  = quantized symbols as text (src/qv_compressor.c:100-125)
 */
#ifndef QVZ_ENCODE_H
#define QVZ_ENCODE_H

#include <stdint.h>
#include <stdio.h>

#include "qvz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define QVZ_SEED_TIME 0      /* srand(time(0)); 32 x rand()          (reference default, qv_stream.c:77-81) */
#define QVZ_SEED_DEBUG 1     /* 32 x 0x55555555                      (reference -DDEBUG, qv_stream.c:83)   */
#define QVZ_SEED_EXPLICIT 2  /* options.well_state                                                          */

#define QVZ_INPUT_AUTO 0      /* FASTQ if the text starts with a four-line record whose second line is bases, else qualities */
#define QVZ_INPUT_QUAL 1      /* one quality string per line: what the reference reads (src/lines.c:27-82) */
#define QVZ_INPUT_FASTQ 2     /* four-line FASTQ records: the fourth lines are encoded (reference README.md:30-34 leaves this to the user) */

typedef struct qvz_options {
    /* the fields of qv_options_t the encoder reads (reference include/codebook.h:28-40) */
    int verbose;
    int stats;
    int mode;                 /* QVZ_MODE_RATIO (-f) or QVZ_MODE_FIXED (-r) */
    int clusters;             /* -c */
    int distortion;           /* QVZ_DIST_* (-d) */
    double ratio;             /* argument of -f / -r */
    double cluster_threshold; /* -T */
    const char *uncompressed_name; /* -u, or NULL */
    /* additions of this implementation */
    int device;               /* HIP device ordinal */
    int seed_mode;            /* QVZ_SEED_* */
    uint32_t well_state[32];  /* used when seed_mode == QVZ_SEED_EXPLICIT */
    uint64_t chunk_reads;     /* reads per quantize/encode chunk (0 = default) */
    const char *dist_file;    /* -D FILE when distortion == QVZ_DIST_CUSTOM (reference codebook.h:35, main.c:30-32) */
    int input_format;         /* QVZ_INPUT_*: what the input text is (default: look at it) */
} qvz_options_t;

typedef struct qvz_result {
    uint64_t lines;
    uint32_t columns;
    uint64_t stream_bytes;    /* os->written: arithmetic stream only, what "-s" calls size (main.c:125) */
    uint64_t file_bytes;      /* whole .q file */
    double distortion;        /* bit-identical to the reference's value (sequential sum in read order) */
    uint32_t kmeans_iterations;
    uint64_t deferred_decisions; /* codebook search decisions taken by host libm inside the guard band */
    double t_upload, t_cluster, t_stats, t_codebook, t_encode, t_total;   /* seconds */
    /* sharded encodes: what t_encode consists of - waiting for the WELL draws + the quantizer, the coder kernels, and
     * everything that leaves the device afterwards (-u text, per-read errors, streams: PCIe + file writes) */
    double t_quantize, t_coder, t_egress;
} qvz_result_t;

void qvz_options_default(qvz_options_t *o);   /* reference defaults, main.c:198-204 (+ mode = RATIO) */

/* FASTQ front end (host/qvz_fastq.c). detect: does the text start with a four-line record (line 1 '@', line 2 bases,
 * line 3 '+', lines 2 and 4 equally long)? qualities: the fourth line of every record as rows of columns+1 bytes
 * ('\n'-terminated, the layout of src/lines.c:44-71) in a malloc'd buffer the caller frees; all reads must have the
 * length of the first (QVZ_E_DATA otherwise, with the byte offset of the offending record in the message). */
int qvz_fastq_detect(const uint8_t *text, uint64_t bytes);
int qvz_fastq_qualities(const uint8_t *text, uint64_t bytes, uint8_t **rows_out, uint64_t *reads_out, uint32_t *columns_out);

/* Text rows in host memory (n rows of c+1 bytes, Q+33 ... '\n': the layout load_file maps, src/lines.c:44-71) -> the
 * column-major symbol matrix d_sym (c x ld, include/qvz_hip.h) through two pinned staging buffers: host copy, PCIe transfer
 * and the transposing kernel of consecutive chunks overlap. *bad_out: bytes outside '!'..'h' or rows without '\n'. */
int qvz_ingest_rows(const uint8_t *text, uint64_t n, uint32_t c, uint8_t *d_sym, uint64_t ld, uint32_t *bad_out);
/* The inverse: the reads of a column-major matrix in HBM (c x ld symbols, include/qvz_hip.h) written to `path` as text
 * rows - how a -u text or a decoded file leaves the device: slab by slab through two pinned buffers, the file written by
 * a few threads; no n * (c + 1)-byte buffer on either side. */
int qvz_egress_rows(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, const char *path);

/* text: the whole input file image (rows of c+1 bytes, or FASTQ: see input_format). fout receives the .q file, fu (may be
 * NULL) the -u text. */
int qvz_encode_buffer(const uint8_t *text, uint64_t bytes, FILE *fout, FILE *fu, const qvz_options_t *opts,
                      qvz_result_t *res);
/* mmaps `in_path` like load_file (lines.c:27-82) and encodes it to `out_path`. */
int qvz_encode_file(const char *in_path, const char *out_path, const qvz_options_t *opts, qvz_result_t *res);

/* Sharded output (option B): the same pipeline, but the reads are cut into `shards` contiguous
 * ranges (qvz_shard_bounds) coded as independent streams by the GPU coder, and every range is written as a complete
 * reference-format file "<out_prefix>.0000", "<out_prefix>.0001", ...: header with the shard's line count, the whole
 * file's codebooks, the WELL state advanced to the shard's first draw, the stream. The unmodified reference decoder
 * decodes each; the decodes concatenate to the -u text of the single-stream encode. res->stream_bytes is the sum of
 * the streams (what -s reports), res->file_bytes the sum of the files. */
int qvz_encode_file_sharded(const char *in_path, const char *out_prefix, const qvz_options_t *opts, uint32_t shards,
                            qvz_result_t *res);

/* The same encode on SEVERAL devices of this process (host/qvz_multi.c; QVZ_DEVICES=0,1,... in the qvz front end): the reads
 * are dealt to the devices in whole shards of qvz_shard_bounds(N, shards), every device runs the single-device stages on
 * its reads, the k-means accumulators (per iteration) and the PMF counters (once) are summed over the devices - RCCL
 * (ncclAllReduce over xGMI) when the ordinals are distinct, a host-staged sum when an ordinal is listed more than once -
 * and the output is the one shard set a single device writes for the same `shards`, byte for byte; distortion and -u too.
 * shards >= ndev. */
int qvz_encode_file_devices(const char *in_path, const char *out_path, const qvz_options_t *opts, const int *devices,
                            uint32_t ndev, uint32_t shards, qvz_result_t *res);

/* Rate-distortion sweep (reference generate_rd.sh:1-17: twenty `qvz -c 1 -f 0 ... 0.95 -s` encodes of one file; SURVEY.md
 * section 8f row 2). One file, `npoints` operating points: the upload, the cluster labels (cluster.c:212-244) and the conditional
 * counters (codebook.c:185-205) do not depend on -f / -r / -d / -D and are made ONCE; per point what is left is the codebooks,
 * the quantizer and the coding. Everything else in `opts` (clusters, threshold, seed, device, -u, -s, -v) holds for all points.
 * shards == 0: every point's output is the reference's single-stream file, byte for byte what `qvz <flags of the point>`
 * writes for the same WELL state; shards > 0: a shard set per point (GPU coder), what qvz_encode_file_shardset writes.
 * Outputs: "<out_path>.NN" (and "<-u name>.NN") with NN = the point's two-digit index when keep_all, else every point
 * overwrites out_path as the reference's script does. With opts->stats every point prints the reference's -s line
 * (main.c:125) - the lines generate_rd.sh collects; a point's `time` is its own seconds, the first point's includes the shared
 * phases. res (may be NULL): npoints results. */
typedef struct qvz_sweep_point {
    int mode;                 /* QVZ_MODE_RATIO (-f) or QVZ_MODE_FIXED (-r) */
    double ratio;
    int distortion;           /* QVZ_DIST_* */
    const char *dist_file;    /* -D FILE when distortion == QVZ_DIST_CUSTOM */
} qvz_sweep_point_t;
int qvz_encode_file_sweep(const char *in_path, const char *out_path, const qvz_options_t *opts, const qvz_sweep_point_t *points,
                          uint32_t npoints, uint32_t shards, int keep_all, qvz_result_t *res);
/* The points of a sweep from text (the qvz front end's QVZ_SWEEP): "rd" = the reference script's twenty, -f 0, .05 ... .95, spelled
 * as `bc -l` prints them; else "<f|r>:<v>,<v>,..." with `opts`' distortion. Returns the number of points (0: not understood,
 * or more than `max`). */
uint32_t qvz_sweep_parse(const char *text, const qvz_options_t *opts, qvz_sweep_point_t *points, uint32_t max);

/* The hot path over a matrix that is ALREADY in HBM: the stages of the reference's encode() (main.c:42-91: do_kmeans_clustering,
 * calculate_statistics, generate_codebooks, start_qv_compression) with neither text nor files on either side - a caller that
 * holds the reads on the device (column-major symbols, include/qvz_hip.h: d_sym[s * ld + i] = byte - 33, ld = qvz_hip_ld(n)),
 * and the step bench.py times. `shards` independent reference-format streams coded by the GPU coder stay on the device packed
 * each in its slot of the stream area (qvz_resident_pack: back to back), with the quantized values (want_qv) and the labels; qvz_resident_products says where. A handle keeps its big
 * device buffers between encodes; nothing computed is carried from one encode to the next.
 *   A PART of a file (one GPU of several, one process each): reads [first_read, first_read + n) of n_total, first_read a multiple of
 * 4, and `allreduce` - the sum, in place on `stream`, of `count` elements of a device buffer over all parts: width 4 / 8 = unsigned
 * integers of that many bytes, -8 = doubles. RCCL's ncclAllReduce is that function (bench.py hands torch.distributed's in);
 * it is called for the centroid reads (k x c, once), the k-means sums (per iteration), the counters (once) and the distortion.
 * centroid_lines: the k read numbers cluster.c:198-201 draws (NULL: drawn here with libc rand() after srand(1) - the state a
 * process starts in, which is where the reference binary draws them).
 * res: lines / columns of the part, stream_bytes of its streams, the FILE's distortion (a fixed-order device sum, not the
 * reference's sequential one: within 1e-6 relative), iterations, phase seconds. */
typedef struct qvz_resident qvz_resident_t;
typedef int (*qvz_allreduce_fn)(void *ctx, void *d_buf, uint64_t count, int width, void *stream);
typedef struct qvz_resident_products {
    const uint8_t *d_coded;            /* device: stream g at d_coded + g * cap_per_stream, h_sizes[g] bytes                */
    uint64_t cap_per_stream;
    const uint8_t *d_streams;          /* device: the streams packed back to back - NULL until qvz_resident_pack has made them */
    const uint64_t *h_sizes;           /* host: [shards]                                                                    */
    uint32_t shards, clusters;
    uint64_t stream_bytes;
    const uint8_t *d_qv;               /* device: quantized values, the layout of d_sym (NULL unless want_qv)               */
    const uint8_t *d_labels;           /* device: [ld] cluster of every read                                                */
    const struct qvz_codebook *const *codebooks;   /* [clusters] (include/qvz_codebook.h)                                   */
    uint32_t well_state[32];           /* the file's WELL state (shard g starts first_read + bounds[g] reads into its draws) */
    double model_kernels_ms, coder_kernels_ms;     /* busy time of the coder phase's two kernel families                    */
    double kmeans_kernels_ms, pmf_kernel_ms, quantize_kernel_ms;   /* HIP events around the launches, on their stream (k-means: all iterations) */
} qvz_resident_products_t;
int qvz_resident_open(int device, qvz_resident_t **out);
int qvz_resident_encode(qvz_resident_t *r, const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint64_t first_read,
                        uint64_t n_total, const qvz_options_t *opts, uint32_t shards, int want_qv, const uint64_t *centroid_lines,
                        qvz_allreduce_fn allreduce, void *allreduce_ctx, qvz_result_t *res);
/* what the last encode left (valid until the next encode on the handle, or its close) */
int qvz_resident_products(const qvz_resident_t *r, qvz_resident_products_t *out);
/* the last encode's streams packed back to back on the device (what travels to the host and into a shard set: 1-2 bits per symbol
 * instead of the area's 6): made on the first call, *d_packed (may be NULL) receives the address; products then reports it too */
int qvz_resident_pack(qvz_resident_t *r, const uint8_t **d_packed);
void qvz_resident_close(qvz_resident_t *r);

/* Decode direction (`qvz -x`, reference src/main.c:132-183, start_qv_decompression src/qv_compressor.c:145-231).
 * qvz_decode_file: a reference-format .q file (one stream) -> text; the serial arithmetic decoder runs on one host
 * core as in the reference, the WELL draws come from the GPU word stream. qvz_decode_shard_files: the files
 * "<prefix>.0000" ... written by qvz_encode_file_sharded, all shards decoded at once by the GPU decoder
 * (qvz_hip_decode_shards); the output is the concatenation of the decodes. res (may be NULL): lines, columns,
 * stream_bytes, file_bytes. */
int qvz_decode_file(const char *in_path, const char *out_path, int device, qvz_result_t *res);
int qvz_decode_shard_files(const char *prefix, uint32_t shards, const char *out_path, int device, qvz_result_t *res);
/* The GPU decoder works in slabs of whole streams - draws, decode, text, next slab - when a file's matrices (2 C bytes per read)
 * do not fit the device at once, as a file of 2^32 reads and more never does. reads: most reads per slab (a stream longer than
 * that is a slab by itself); 0 = by the device's free memory, the default. Process-wide; the output does not depend on it. */
void qvz_decode_set_slab_reads(uint64_t reads);
/* the same with a limit on the streams of a slab too (every stream of a slab has its adaptive tables in device memory; 0 = by the
 * device's free memory) */
void qvz_decode_set_slab(uint64_t reads, uint32_t streams);

const char *qvz_encode_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
