/* qvz_compat.h - the reference's OWN entry points, implemented on the MI355X.
 *
 * qvz has no plugin/FFI interface: the seam a maintainer can cut at is the set of functions encode() calls
 * (reference src/main.c:18-127). libqvzcompat.so exports exactly those names with the reference's signatures
 * and struct layouts (x86-64 ABI), so the reference's unmodified src/main.c + src/util.c link against it in
 * place of codebook.o, cluster.o, lines.o, pmf.o, distortion.o, quantizer.o, qv_compressor.o, qv_stream.o,
 * arith.o, os_stream.o and well.o (see INTEGRATION.md, and oracle/build_ref.sh which builds that binary as
 * oracle/_ref/qvz_dropin for the tests).
 *
 * The structs below restate the layouts of include/lines.h:25-77, include/codebook.h:28-69, include/pmf.h:22-45,
 * include/quantizer.h:16-22, include/distortion.h:15-18 and include/well.h:9-14 of the reference: field order
 * and types are interface facts that must match. Everything the reference leaves in those structs after each
 * call is filled in here too (labels per line, cluster means/counts/accumulators, conditional and marginal pmfs,
 * the conditional quantizer lists), so a caller that swaps only some of the functions still finds what it
 * expects. The library keeps the device-side state of the one file being encoded (the reference is single-file,
 * single-thread, never frees: src/main.c:18-127); it is not re-entrant.
 */
#ifndef QVZ_COMPAT_H
#define QVZ_COMPAT_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t symbol_t;                                   /* pmf.h:16 */

struct alphabet_t { uint32_t size; symbol_t *symbols; uint32_t *indexes; };                      /* pmf.h:22-26 */
struct pmf_t { uint8_t pmf_ready; const struct alphabet_t *alphabet; double *pmf; uint32_t *counts; uint32_t total; };
struct pmf_list_t { uint32_t size; struct pmf_t **pmfs; };                                       /* pmf.h:42-45 */
struct distortion_t { double *distortion; uint8_t symbols; };                                    /* distortion.h:15-18 */
struct well_state_t { uint32_t state[32]; uint32_t n; uint32_t bit_output; uint32_t bits_left; }; /* well.h:9-14 */
struct quantizer_t {                                                                             /* quantizer.h:16-22 */
    const struct alphabet_t *alphabet;
    struct alphabet_t *output_alphabet;
    symbol_t *q;
    double ratio;
    double mse;            /* not computed by this implementation (nothing reads it: codebook.c:410,453 are dead) */
};
struct qv_options_t {                                                                            /* codebook.h:28-40 */
    uint8_t verbose, stats, mode, clusters, uncompressed, distortion;
    char *dist_file;
    char *uncompressed_name;
    double ratio, e_dist, cluster_threshold;
};
struct cond_pmf_list_t { uint32_t columns; const struct alphabet_t *alphabet; struct pmf_t **pmfs; struct pmf_list_t *marginal_pmfs; };
struct cond_quantizer_list_t {                                                                   /* codebook.h:61-69 */
    uint32_t columns, lines;
    struct alphabet_t **input_alphabets;
    struct quantizer_t ***q;
    double **ratio;
    uint8_t **qratio;
    struct qv_options_t *options;
};
struct line_t { uint8_t cluster; const symbol_t *m_data; };                                      /* lines.h:25-28 */
struct line_block_t { uint32_t count; struct line_t *lines; };
struct cluster_t {                                                                               /* lines.h:39-49 */
    uint8_t id;
    uint32_t count;
    symbol_t *mean;
    uint64_t *accumulator;
    struct cond_pmf_list_t *training_stats;
    struct cond_quantizer_list_t *qlist;
};
struct cluster_list_t { uint8_t count; struct cluster_t *clusters; double *distances; };
struct quality_file_t {                                                                          /* lines.h:65-77 */
    struct alphabet_t *alphabet;
    char *path;
    uint64_t lines;
    uint32_t columns;
    uint32_t block_count;
    struct line_block_t *blocks;
    uint8_t cluster_count;
    struct cluster_list_t *clusters;
    struct distortion_t *dist;
    struct qv_options_t *opts;
    struct well_state_t well;
};

#define LF_ERROR_NONE 0                                      /* lines.h:17-20 */
#define LF_ERROR_NOT_FOUND 1
#define LF_ERROR_NO_MEMORY 2
#define LF_ERROR_TOO_LONG 4

/* the calls of encode(), in order (main.c:20-91) */
struct alphabet_t *alloc_alphabet(uint32_t size);                                    /* pmf.h:48,  pmf.c:11    */
struct distortion_t *generate_distortion_matrix(uint8_t symbols, int type);          /* distortion.h:25, distortion.c:30 */
struct distortion_t *gen_custom_distortion(uint8_t symbols, const char *filename);   /* distortion.h:26, distortion.c:100 */
uint32_t load_file(const char *path, struct quality_file_t *info, uint64_t max_lines);  /* lines.h:80, lines.c:27 */
struct cluster_list_t *alloc_cluster_list(struct quality_file_t *info);              /* cluster.h:12, cluster.c:24 */
void do_kmeans_clustering(struct quality_file_t *info);                              /* cluster.h:24, cluster.c:212 */
void calculate_statistics(struct quality_file_t *info);                              /* codebook.h:90, codebook.c:185 */
void generate_codebooks(struct quality_file_t *info);                                /* codebook.h:92, codebook.c:355 */
void write_codebooks(FILE *fp, struct quality_file_t *info);                         /* codebook.h:95, codebook.c:474 */
uint32_t start_qv_compression(struct quality_file_t *info, FILE *fout, double *dis, FILE *funcompressed);  /* qv_compressor.h:95 */

/* decode() (main.c:132-160): the codebooks are parsed by the host C layer (qvz_codebook_read), the single serial stream
 * is decoded on one host core exactly as the reference does it (include/qvz_encode.h, qvz_decode_file); info->columns,
 * lines, cluster_count, clusters[].qlist and well are filled in as the reference leaves them. */
void read_codebooks(FILE *fp, struct quality_file_t *info);                          /* codebook.h:96, codebook.c:560 */
void start_qv_decompression(FILE *fout, FILE *fin, struct quality_file_t *info);     /* qv_compressor.h:96, qv_compressor.c:145 */

/* ---- the symbol-level interface of include/qv_compressor.h:26-93 ----------------------------------------------------
 * start_qv_compression / start_qv_decompression above replace the reference's loops wholesale (GPU quantizer, staged
 * coder). A caller that drives the coder symbol by symbol instead (the loops of qv_compressor.c:76-135 and :166-229) finds
 * the same structs and functions here, computed on the host: this interface is one serial chain by definition. */
#define m_arith 22                                           /* qv_compressor.h:19 */
#define OS_STREAM_BUF_LEN (4096 * 4096)                      /* qv_compressor.h:21 */
#define COMPRESSION 0
#define DECOMPRESSION 1

typedef struct Arithmetic_code_t { int32_t scale3; uint32_t l, u, t, m, r; } *Arithmetic_code;        /* qv_compressor.h:26-35 */
typedef struct os_stream_t { FILE *fp; uint8_t *buf; uint32_t bufPos; uint8_t bitPos; uint64_t written; } *osStream;   /* :37-43 */
typedef struct stream_stats_t { uint32_t *counts; uint32_t alphabetCard, step, n; } *stream_stats_ptr_t;   /* :45-50 */
typedef struct arithStream_t {                                                                         /* :52-57 */
    stream_stats_ptr_t cluster_stats;
    stream_stats_ptr_t ***stats;
    Arithmetic_code a;
    osStream os;
} *arithStream;
typedef struct qv_compressor_t { arithStream Quals; } *qv_compressor;                                   /* :59-61 */

struct os_stream_t *alloc_os_stream(FILE *fp, uint8_t in);                           /* os_stream.c:7 */
void free_os_stream(struct os_stream_t *os);
uint8_t stream_read_bit(struct os_stream_t *os);                                     /* os_stream.c:36 */
uint32_t stream_read_bits(struct os_stream_t *os, uint8_t len);
void stream_write_bit(struct os_stream_t *os, uint8_t bit);                          /* os_stream.c:72 */
void stream_write_bits(struct os_stream_t *os, uint32_t dw, uint8_t len);
void stream_finish_byte(struct os_stream_t *os);                                     /* os_stream.c:105 */
void stream_write_buffer(struct os_stream_t *os);                                    /* os_stream.c:115 */

Arithmetic_code initialize_arithmetic_encoder(uint32_t m);                           /* arith.c:5 */
void arithmetic_encoder_step(Arithmetic_code a, stream_stats_ptr_t stats, int32_t x, osStream os);   /* arith.c:24 */
int encoder_last_step(Arithmetic_code a, osStream os);                               /* arith.c:99 */
uint32_t arithmetic_decoder_step(Arithmetic_code a, stream_stats_ptr_t stats, osStream is);          /* arith.c:118 */
uint32_t decoder_last_step(Arithmetic_code a, stream_stats_ptr_t stats);             /* arith.c:190 */

stream_stats_ptr_t **initialize_stream_stats(struct cond_quantizer_list_t *q_list);  /* qv_stream.c:32 */
void update_stats(stream_stats_ptr_t stats, uint32_t x, uint32_t r);                 /* qv_stream.c:9 */

void compress_qv(arithStream as, uint32_t x, uint8_t cluster, uint32_t column, uint32_t idx);   /* qv_compressor.c:8 */
void qv_write_cluster(arithStream as, uint8_t cluster);                              /* qv_compressor.c:19 */
uint32_t decompress_qv(arithStream as, uint8_t cluster, uint32_t column, uint32_t idx);         /* qv_compressor.c:27 */
uint8_t qv_read_cluster(arithStream as);                                             /* qv_compressor.c:36 */
qv_compressor initialize_qv_compressor(FILE *fout, uint8_t streamDirection, struct quality_file_t *info);   /* qv_stream.c:120 */

/* what those loops call besides: the lo/hi choice by one 7-bit WELL draw (codebook.c:162-171, well.c:8-46) */
uint32_t well_1024a(struct well_state_t *state);
uint32_t well_1024a_bits(struct well_state_t *state, uint8_t bits);
struct quantizer_t *choose_quantizer(struct cond_quantizer_list_t *list, struct well_state_t *well, uint32_t column,
                                     symbol_t prev, uint32_t *q_idx);

#ifdef __cplusplus
}
#endif
#endif
