/* qvz_hip.h - C-ABI of the MI355X (gfx950) implementation of qvz's encode hot path.
 *
 * Plain C: pointers, sizes, int return codes. No C++ or torch types cross this boundary.
 * The reference (mikelhernaez/qvz) has no FFI: its seam is the seven calls encode() makes
 * (reference src/main.c:42-91). Each entry point below names the reference function(s) it replaces.
 * The host-side mirror that keeps the reference's own names and struct-based signatures
 * (do_kmeans_clustering, calculate_statistics, generate_codebooks, start_qv_compression ...) is
 * declared in qvz_compat.h and is a thin C layer over this file.
 *
 * Conventions
 *  - "d_" pointers are device (HBM) pointers, "h_" pointers are host pointers.
 *  - stream: a hipStream_t passed as void* (NULL = default stream). Calls are asynchronous on that
 *    stream unless they return host-visible results.
 *  - Symbol matrix layout (the one layout every kernel is written for): column-major, offset removed:
 *        d_sym[s * ld + i] = byte(read i, column s) - 33   in [0,72)
 *    ld = qvz_hip_ld(n): n rounded up to a multiple of 256, so every column starts 256-byte aligned and
 *    16-byte vector loads never straddle a column. Entries i >= n inside the padding are ignored.
 *  - Alphabet is fixed at 72 (reference src/main.c:13, include/pmf.h:11).
 *  - Return value 0 = success; negative = QVZ_E_*; qvz_hip_last_error() gives the message.
 */
#ifndef QVZ_HIP_H
#define QVZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QVZ_A 72
#define QVZ_NOT_FOUND 0xFF

#define QVZ_OK 0
#define QVZ_E_ARG (-1)       /* bad argument (shape, alignment, range)                          */
#define QVZ_E_HIP (-2)       /* HIP runtime error                                               */
#define QVZ_E_DATA (-3)      /* input bytes outside [33,105) or missing '\n' (reference: UB)    */
#define QVZ_E_EMPTY (-4)     /* a k-means cluster went empty (reference: SIGFPE, cluster.c:113) */
#define QVZ_E_DEAD_CTX (-5)  /* quantizer met a context the codebook does not define            */
#define QVZ_E_NOMEM (-6)

#define QVZ_MODE_RATIO 0     /* -f, reference include/codebook.h:21 */
#define QVZ_MODE_FIXED 1     /* -r, reference include/codebook.h:22 */
#define QVZ_DIST_MANHATTAN 1 /* reference include/distortion.h:7-9 */
#define QVZ_DIST_MSE 2
#define QVZ_DIST_LORENTZ 3
#define QVZ_DIST_CUSTOM 4    /* -D FILE, reference include/distortion.h:10 */

const char *qvz_hip_last_error(void);
/* Records `msg` as the text qvz_hip_last_error() returns and hands `code` back: how the host C layer of the library
 * (qvz_amd/csrc/host, no HIP headers) reports through the same channel as the kernels' launchers. */
int qvz_hip_set_error(int code, const char *msg);
uint64_t qvz_hip_ld(uint64_t n);
int qvz_hip_device_count(void);

/* Device memory / stream plumbing so that host C code (qvz_amd/csrc/host) needs no HIP headers.
 * The reference has no counterpart: it mmaps the input and callocs everything (src/lines.c:64,116). */
int qvz_hip_set_device(int device);
int qvz_hip_mem_info(uint64_t *free_bytes, uint64_t *total_bytes);   /* HBM of the current device */
/* events on a stream, for timing a launch from C: elapsed_ms waits for the second event */
int qvz_hip_event_create(void **event);
int qvz_hip_event_record(void *event, void *stream);
int qvz_hip_event_elapsed_ms(void *first, void *second, double *ms);
int qvz_hip_event_destroy(void *event);
int qvz_hip_malloc(void **d_ptr, size_t bytes);
int qvz_hip_free(void *d_ptr);
int qvz_hip_malloc_host(void **h_ptr, size_t bytes);   /* pinned host memory for async copies */
int qvz_hip_free_host(void *h_ptr);
int qvz_hip_memset(void *d_ptr, int value, size_t bytes, void *stream);
int qvz_hip_copy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int qvz_hip_copy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int qvz_hip_stream_create(void **stream);
/* *overlap = 1 if kernels launched on the two streams (NULL = the default stream) run side by side, 0 if the runtime has
 * put them on one hardware queue (a queue runs its kernels in order): a ~1 ms probe, for streams that are made once and
 * kept. Both streams are synchronised. */
int qvz_hip_streams_overlap(void *stream_a, void *stream_b, int *overlap);
int qvz_hip_stream_destroy(void *stream);
int qvz_hip_stream_sync(void *stream);
/* work submitted to `stream` from now on starts after everything submitted to `other` so far has finished (no host wait) */
int qvz_hip_stream_wait(void *stream, void *other);

/* ---------------------------------------------------------------------------------------------
 * Input staging. Replaces the row-major pointer view built by load_file (reference
 * src/lines.c:27-82: C from the first line, N = size/(C+1), one line_t per read).
 * ------------------------------------------------------------------------------------------- */
/* d_text: n rows of c+1 bytes (Q+33 ... '\n') already in HBM. Writes d_sym (c columns x ld).
 * d_bad (1 x u32, zeroed by the caller): number of bytes outside [33,105) or rows not ending '\n'. */
int qvz_hip_transpose_text(const uint8_t *d_text, uint64_t n, uint32_t c, uint8_t *d_sym, uint64_t ld,
                           uint32_t *d_bad, void *stream);
/* Inverse, for the "-u" dump (reference src/qv_compressor.c:100-125): quantized symbols -> text rows. */
int qvz_hip_untranspose_text(const uint8_t *d_qv, uint64_t n, uint32_t c, uint64_t ld, uint8_t *d_text,
                             void *stream);
/* The same for the reads [first, first + n) of the matrix only (first a multiple of 4), written as rows 0 .. n-1 of
 * d_text: how a whole-file -u text leaves the device slab by slab instead of as one N * (c + 1)-byte buffer. */
int qvz_hip_untranspose_rows(const uint8_t *d_qv, uint64_t first, uint64_t n, uint32_t c, uint64_t ld, uint8_t *d_text,
                             void *stream);
/* QGEN-1 synthetic reads [first, first+n) generated directly in d_sym layout. */
int qvz_hip_qgen1(uint64_t seed, uint64_t first, uint64_t n, uint32_t c, uint8_t *d_sym, uint64_t ld,
                  void *stream);

/* ---------------------------------------------------------------------------------------------
 * k-means. Replaces do_kmeans_clustering and helpers (reference src/cluster.c:65-244).
 * Means are SYMBOLS (byte-33); the reference keeps raw bytes, distances and floor-means are the same.
 * ------------------------------------------------------------------------------------------- */
/* cluster_lines/do_cluster_assignment/find_distance/assign_cluster (cluster.c:65,136,176,149):
 * d_labels[i] = lowest j minimising sum_s (x_s - mean_j,s)^2; d_counts[j] += reads labelled j. */
int qvz_hip_kmeans_assign(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint32_t k,
                          const uint8_t *d_means /*[k][c]*/, uint8_t *d_labels /*[ld]*/,
                          uint32_t *d_counts /*[k], +=*/, void *stream);
/* first pass of recalculate_means (cluster.c:92-101): d_acc[j][s] += sum of symbols of reads labelled j */
int qvz_hip_kmeans_accumulate(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c,
                              uint64_t ld, uint32_t k, uint64_t *d_acc /*[k][c], +=*/, void *stream);
/* One iteration's assignment AND sums in a single pass over the symbols: exactly qvz_hip_kmeans_assign followed by
 * qvz_hip_kmeans_accumulate (same labels, counts and sums), which it calls for shapes its one-pass kernel does not take
 * (k outside 2..8, more than 160 columns). This is what the drivers call; symbols must be below 128. */
int qvz_hip_kmeans_step(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint32_t k,
                        const uint8_t *d_means /*[k][c]*/, uint8_t *d_labels /*[ld]*/, uint32_t *d_counts /*[k], +=*/,
                        uint64_t *d_acc /*[k][c], +=*/, void *stream);
/* second pass of recalculate_means (cluster.c:104-128): mean = acc/count (floor), moved_j = sum (dmean)^2.
 * d_moved: k+1 doubles (exact integers): [0] = max_j moved_j, [1+j] = moved_j.
 * d_status[0] |= 1 if a cluster is empty. */
int qvz_hip_kmeans_update(const uint64_t *d_acc, const uint32_t *d_counts, uint32_t k, uint32_t c,
                          uint8_t *d_means, double *d_moved, int32_t *d_status, void *stream);

/* the same with 64-bit read counts (a file of 2^32 reads or more, counted in parts: host/qvz_multi.c) */
int qvz_hip_kmeans_update64(const uint64_t *d_acc, const uint64_t *d_counts, uint32_t k, uint32_t c,
                            uint8_t *d_means, double *d_moved, int32_t *d_status, void *stream);
/* d_dst[i] = d_src[i]: a part's u32 counters widened so that their sum over the parts cannot wrap */
int qvz_hip_widen_u32(const uint32_t *d_src, uint64_t *d_dst, uint64_t count, void *stream);
/* The k reads the reference copies as initial centroids (initialize_kmeans_clustering, cluster.c:198-201) out of a matrix that is in
 * HBM: d_out[j][s] = symbol s of read d_idx[j] (a read number of THIS matrix), all zero for d_idx[j] = ~0 (a read another part
 * of the file owns: the SUM over the parts then holds every centroid; SURVEY.md section 8e, collective 1). */
int qvz_hip_gather_reads(const uint8_t *d_sym, uint64_t ld, uint32_t c, const uint64_t *d_idx /*[k]*/, uint32_t k,
                         uint32_t *d_out /*[k][c]*/, void *stream);
/* d_sum[0] = sum over the reads of d_err[i] / columns (the term start_qv_compression adds per line, qv_compressor.c:127) in a
 * fixed order - reproducible, but not the reference's one sequential sum: within 1e-6 relative of it (north_star's tolerance for
 * the distortion), what a part of a sharded file contributes to the all-reduced total. d_sum: 1025 doubles (the sum + scratch). */
int qvz_hip_sum_errors(const double *d_err, uint64_t n, uint32_t c, double *d_sum /*[1025]*/, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Conditional-PMF counters. Replaces the counting loop of calculate_statistics
 * (reference src/codebook.c:193-205, pmf_increment src/pmf.c:211-214).
 * d_cnt[((j*c + s)*72 + prev)*72 + cur] += 1 for every read labelled j (column 0: prev = 0).
 * d_labels may be NULL when k == 1. Counters are u32 (reference pmf.h:35-36) => n < 2^32.
 * ------------------------------------------------------------------------------------------- */
int qvz_hip_pmf_hist(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c, uint64_t ld,
                     uint32_t k, uint32_t *d_cnt, void *stream);

/* ---------------------------------------------------------------------------------------------
 * WELL-1024a draws. Replaces well_1024a / well_1024a_bits (reference src/well.c:8-46) for the
 * quantizer: draw g (global index i*C+s) = (word[g>>2] >> 7*(g&3)) & 127.
 * Writes words [first_word, first_word+nwords) of the stream seeded with h_state (32 words as stored
 * in the .q file, n = 0) to d_words. Uses GF(2) jump-ahead so any range can be produced in parallel.
 * ------------------------------------------------------------------------------------------- */
int qvz_hip_well_words(const uint32_t *h_state, uint64_t first_word, uint64_t nwords, uint32_t *d_words,
                       void *stream);
/* The same draws as a matrix in the symbols' layout: d_draws[s*ld + i] = draw of (read first_read+i, column s), one
 * byte each (0..127). This is what the quantizer consumes; internally the word stream is generated in chunks into a
 * reusable workspace and transposed through LDS. */
int qvz_hip_well_draws(const uint32_t *h_state, uint64_t first_read, uint64_t n, uint32_t c, uint8_t *d_draws,
                       uint64_t ld, void *stream);
/* State of the generator after `words` outputs, rotated to n = 0 (what a shard's .q header stores). */
int qvz_hip_well_jump(const uint32_t *h_state, uint64_t words, uint32_t *h_state_out);
/* The same for `count` different jumps at once: h_states_out[i*32 ..] = state after h_words[i] outputs (the WELL states
 * of the headers of `count` shard files: one masked batched GF(2) mat-vec per bit of the jumps). */
int qvz_hip_well_jump_many(const uint32_t *h_state, const uint64_t *h_words, uint64_t count, uint32_t *h_states_out);

/* ---------------------------------------------------------------------------------------------
 * Quantizer tables in device form (one per cluster), built from a codebook (see qvz_codebook.h).
 *   ct[s][72]      u16 : low byte = context index of previous quantized value (QVZ_NOT_FOUND if none),
 *                        high byte = qratio of that context           (choose_quantizer codebook.c:162-171)
 *   qs[s][144][72] u16 : low byte = quantized value q[2ctx+b][x], high byte = its state index in that
 *                        quantizer's output alphabet                  (qv_compressor.c:93-95,109-110)
 * ------------------------------------------------------------------------------------------- */
#define QVZ_CT_STRIDE 72
#define QVZ_QS_STRIDE (144 * 72)

/* Replaces the quantize half of start_qv_compression (reference src/qv_compressor.c:76-135).
 * d_ct / d_qs: k tables back to back ([k][c][72] and [k][c][144][72] u16).
 * d_draws: draw matrix of these n reads (qvz_hip_well_draws), pitch ld_draws.
 * h_dist: 72x72 doubles d[x + 72*y] (reference distortion.c:151).
 * Outputs (each may be NULL): d_qv / d_qi / d_st column-major with pitch ld_out (a multiple of 256, >= n; lets a
 * chunk of a large matrix be quantized into chunk-sized buffers): quantized value, quantizer
 * index 2*ctx+b and state index per symbol; d_err[i] = per-read sum of d(x,qv) accumulated in column
 * order exactly as the reference does (bit-identical double); d_status |= QVZ_E_DEAD_CTX flags. */
int qvz_hip_quantize(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c, uint64_t ld,
                     uint32_t k, const uint16_t *d_ct, const uint16_t *d_qs, const uint8_t *d_draws,
                     uint64_t ld_draws, const double *d_dist, uint8_t *d_qv, uint8_t *d_qi,
                     uint8_t *d_st, uint64_t ld_out, double *d_err, int32_t *d_status, void *stream);

/* Coder feed: column-major quantizer indices / state indices (pitch ld) -> read-major pairs
 * d_feed[i*c + s] = qi | st << 8, the order compress_qv consumes them in (qv_compressor.c:89-119). */
int qvz_hip_pack_feed(const uint8_t *d_qi, const uint8_t *d_st, uint64_t n, uint32_t c, uint64_t ld,
                      uint16_t *d_feed, void *stream);

#ifdef __cplusplus
}
#endif
#endif
