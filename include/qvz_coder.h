/* qvz_coder.h - host-side adaptive arithmetic coder of the qvz bitstream (plain C).
 *
 * Replaces compress_qv / qv_write_cluster (reference src/qv_compressor.c:8-22), arithmetic_encoder_step and
 * encoder_last_step (src/arith.c:24-116), update_stats and the stats initialisation (src/qv_stream.c:9-61,96-108)
 * and the MSB-first bit writer (src/os_stream.c:72-120). The reference's stream is one serial chain of
 * interval updates, so this part stays on a host core; the GPU hands it the (quantizer index, state) feed.
 *
 * Exactness: output bytes are identical to the reference's. Two restructurings are used, both exact:
 *   - the E1/E2 and E3 renormalisation loops run as one count-leading-zeros batch each per symbol (after the
 *     E1/E2 batch msb(l)=0, msb(u)=1 and E3 shifts preserve that, so one batch of each suffices);
 *   - floor(range*cum/n) is taken as (uint64)((double)(range*cum) / (double)n): with range*cum < 2^42 and
 *     n < 2^21 the true quotient is either an integer or at least 2^-21 away from one, far more than the
 *     2^-53 relative rounding of one IEEE division, so the truncation is exact.
 */
#ifndef QVZ_CODER_H
#define QVZ_CODER_H

#include <stddef.h>
#include <stdint.h>

#include "qvz_codebook.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qvz_coder qvz_coder_t;

/* cbs: `clusters` codebooks (their nctx / nstates define the model, qv_stream.c:46-53). The coder appends to
 * an internal growable byte buffer. */
qvz_coder_t *qvz_coder_create(uint32_t clusters, uint32_t columns, const qvz_codebook_t *const *cbs);
void qvz_coder_destroy(qvz_coder_t *cd);

/* Encodes `reads` reads in file order. labels[i] (NULL when clusters == 1), feed[i*columns + s] = quantizer
 * index (low byte) | state index (high byte) as produced by qvz_hip_pack_feed. Returns 0, or -1 when a feed
 * entry addresses a quantizer the codebook does not define or a state outside its alphabet. */
int qvz_coder_encode(qvz_coder_t *cd, const uint8_t *labels, const uint16_t *feed, uint64_t reads);

/* encoder_last_step (arith.c:99-116): flush the tag and pad. After this the buffer is complete. */
void qvz_coder_finish(qvz_coder_t *cd);

/* Bytes produced so far (complete bytes only until qvz_coder_finish) and access to them. take() hands out the
 * first `*size` bytes and drops them from the coder, so a long stream can be written to disk incrementally. */
uint64_t qvz_coder_bytes(const qvz_coder_t *cd);
const uint8_t *qvz_coder_data(const qvz_coder_t *cd, uint64_t *size);
void qvz_coder_drop(qvz_coder_t *cd, uint64_t size);

/* ---------------------------------------------------------------------------------------------------------
 * Sharded coding (option B). The reference format has exactly one serial stream per
 * file, so one stream caps the whole encoder at one host core. Here reads [0,n) are cut into `shards`
 * contiguous ranges, every range starting at a multiple of 4 reads (= a WELL word boundary, 4*C draws = C
 * words); each range is coded as an independent stream with a fresh adaptive model - byte-for-byte what the
 * reference writes for a file holding only those reads, given the codebooks of the whole file and the WELL
 * state advanced to the range's first draw (stored rotated to n = 0). Wrapped with qvz_shard_file_image every
 * shard is a file the unmodified reference decoder accepts; the decodes concatenate to the single-stream -u
 * output. Cost: each shard re-learns its model (slightly more bits per symbol) and repeats the codebooks.
 * ------------------------------------------------------------------------------------------------------- */
/* first[g] for g = 0..shards (first[shards] = n): the n/4 groups of 4 reads dealt evenly, shard g starting at group
 * floor(groups * g / shards); no shard is empty as long as shards <= qvz_shard_limit(n) = max(1, n/4) */
void qvz_shard_bounds(uint64_t n, uint32_t shards, uint64_t *first);
uint32_t qvz_shard_limit(uint64_t n);

/* Codes all shards with `threads` host threads pulling the feed straight from HBM (d_feed[i*columns+s] u16,
 * d_labels[i] or NULL) in slices over their own streams. streams[g] is malloc'd (free with free()),
 * sizes[g] its length. device: HIP ordinal the buffers live on. */
int qvz_encode_shards_device(const uint16_t *d_feed, const uint8_t *d_labels, uint64_t n, uint32_t columns,
                             uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                             uint32_t threads, int device, uint8_t **streams, uint64_t *sizes);

/* The sharded coding on the GPU, fed straight from the quantizer's output planes (d_qi[s*ld+i], d_st[s*ld+i], as written by
 * qvz_hip_quantize): stream g is byte-identical to what qvz_encode_shards_device - and the reference, for a file holding only
 * that shard's reads (src/qv_compressor.c:76-135, src/arith.c:24-116, src/qv_stream.c:9-25) - produces. d_out: shards *
 * cap_per_shard bytes in HBM, stream g at d_out + g * cap_per_shard; h_sizes[g] receives its length. QVZ_E_NOMEM if a stream
 * outgrows cap_per_shard (worst case 22 bits per symbol), QVZ_E_DEAD_CTX if a symbol reaches a dead context (the models hold
 * live contexts only, qvz_codebook_t.live).
 *
 * The adaptive model of a column depends on that column's symbols only (update_stats touches one table and a table
 * belongs to one column, qv_stream.c:9-25), so the model does not have to run inside the serial chain:
 *   model  one wave per (stream, column), one lane per read: the counts a symbol sees are the table's counts plus
 *          8 x (earlier reads of the chunk with the same table / the same state / a smaller state), taken from lane
 *          bit-masks in LDS; chunks in which a table passes the rescale threshold (n > 2^19) are replayed serially.
 *          It leaves an 8-byte record per symbol: the integers n, cum_lo and n - cum_hi (21 bits each).
 *   coder  one lane per stream walks its records: two loader waves turn a record into the two quotients (reciprocal + one
 *          Newton step), two fused multiply-adds give the new bounds exactly (arith.c:47-48), then E1/E2/E3 renormalisation
 *          (arith.c:50-97) and the bit output, split over three more waves (interval chain / pending-bit count / byte stream)
 *          that hand blocks of 16 symbols over through LDS.
 * Records pass through d_work (work_bytes, 16-byte aligned) in segments of as many reads per stream as fit:
 * 8 bytes x (shards rounded up to 64) x (columns, + 8 reserved positions when clusters > 1, rounded up to 8) per read;
 * d_work = NULL: a workspace of 4/5 of the free device memory, at most 96 GiB and what the whole input needs, is
 * allocated for the call. With room for two segments the model kernel of one runs beside the coder kernel of the other.
 * Codebooks whose largest column needs more table space than LDS offers fall back to a model kernel with one lane per
 * (stream, column) and the tables in HBM.
 * d_labels must be readable up to n rounded up to a multiple of 4. The emitter checks its room once per 16 symbols:
 * a stream that comes within 256 bytes of cap_per_shard is reported as overflowed (QVZ_E_NOMEM). */
int qvz_hip_encode_planes(const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                          uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                          uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes, void *d_work, uint64_t work_bytes,
                          void *stream);
/* The same with the caller's cut of the n reads into `shards` ranges: h_bounds[0] = 0 < ... < h_bounds[shards] = n, every
 * bound but the last a multiple of 4 reads (NULL: qvz_shard_bounds(n, shards)). A device that holds reads
 * [first[a], first[b]) of a file cut by qvz_shard_bounds(N, S) codes its shards a .. b-1 with h_bounds[g] = first[a+g] - first[a]. */
/* What a process's FIRST qvz_hip_encode_planes(_bounds) call does before its first kernel (code object load, the kept
 * buffers sized and allocated, the model / coder stream pair made and probed: 40-50 ms), without coding anything, so that
 * a one-shot process can do it on a host thread beside the quantizer. Same arguments as the call that follows. */
int qvz_hip_encode_planes_prepare(uint64_t n, uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs,
                                  uint32_t shards, const uint64_t *h_bounds, uint64_t work_bytes);
/* Record workspace (bytes) with which streams of up to longest_reads reads are coded in one segment; less means more
 * segments (each costs the model kernel a table save and load). For callers that allocate d_work themselves. */
uint64_t qvz_hip_encode_planes_work_bytes(uint64_t longest_reads, uint32_t columns, uint32_t clusters, uint32_t shards);

/* How qvz_hip_encode_planes cuts a stream of `longest_reads` reads into segments of at most `seg_reads` (what half the
 * workspace holds, or an eighth / a quarter of a stream): `overlap` != 0 - the model kernel of a segment beside the coder kernel
 * of the one before - makes the first and the last segment a quarter of the others. No device needed. Returns the number of
 * segments; first[i], reads[i] of the first min(count, cap) are written. (No counterpart in the reference: its coder is one
 * loop over the lines, src/qv_compressor.c:76-135.) */
uint32_t qvz_coder_segment_list(uint64_t longest_reads, uint64_t seg_reads, int overlap, uint64_t *first, uint64_t *reads, uint32_t cap);
/* ... with the first segment seg_reads / first_div reads long (4: the rule above, used with one cluster, where the coder kernels
 * bound the phase and idle until the first model kernel has ended; 1: with several clusters, where the model kernels bound it and
 * a short first segment would only be one more table save and load per workgroup) */
uint32_t qvz_coder_segment_list2(uint64_t longest_reads, uint64_t seg_reads, int overlap, uint32_t first_div, uint64_t *first, uint64_t *reads, uint32_t cap);
int qvz_hip_encode_planes_bounds(const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                                 uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                 const uint64_t *h_bounds, uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes, void *d_work,
                                 uint64_t work_bytes, void *stream);
/* Decoder of sharded streams on the GPU, one lane per stream: the reference's
 * start_qv_decompression (src/qv_compressor.c:145-231) with arithmetic_decoder_step / decoder_last_step
 * (src/arith.c:118-205) and update_stats (src/qv_stream.c:9-25), per shard. d_in holds stream g at
 * d_in + g * cap_per_shard (h_sizes[g] bytes, the layout the GPU coders write); d_draws is the draw matrix of the
 * same reads (qvz_hip_well_draws with the file's WELL state). Writes the quality values qv[s*ld + i] (the layout of
 * the quantizer's `qv` plane: qvz_hip_untranspose_text turns it into the reference's decoded text) and, for
 * clusters > 1, the decoded cluster id of every read. With one shard this decodes the reference's own stream.
 * QVZ_E_DATA if a stream does not fit the codebooks. */
int qvz_hip_decode_shards(const uint8_t *d_in, uint64_t cap_per_shard, const uint64_t *h_sizes, uint64_t n,
                          uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                          const uint8_t *d_draws, uint64_t ld, uint8_t *d_qv, uint8_t *d_labels, void *stream);
/* The same with the cut given (as qvz_hip_encode_planes_bounds on the encode side): stream g holds the reads
 * [h_bounds[g], h_bounds[g+1]), h_bounds[0] = 0, h_bounds[shards] = n, every start a multiple of 4 reads (a WELL word
 * boundary for any column count). NULL = qvz_shard_bounds(n, shards). This is what lets shard FILES, whose headers carry
 * their own line counts, decode whatever formula cut them. */
int qvz_hip_decode_shards_bounds(const uint8_t *d_in, uint64_t cap_per_shard, const uint64_t *h_sizes, uint64_t n,
                                 uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                 const uint64_t *h_bounds, const uint8_t *d_draws, uint64_t ld, uint8_t *d_qv, uint8_t *d_labels,
                                 void *stream);

/* Bytes of adaptive tables the decoder keeps per stream in device memory (a call needs that for every stream it is given, and
 * once more): what a caller that decodes a file in slabs of streams budgets with (host/qvz_decode.c). */
uint64_t qvz_hip_decode_tables_bytes(uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs);

/* The streams of a GPU coder call (stream g at d_coded + g * cap_per_shard, h_sizes[g] bytes) packed back to back into
 * d_packed (sum of the sizes): what travels to the host and into a shard set, instead of the whole fixed-stride area. */
int qvz_hip_gather_streams(const uint8_t *d_coded, uint64_t cap_per_shard, const uint64_t *h_sizes, uint32_t shards,
                           uint8_t *d_packed, void *stream);
/* The inverse, for a decoder fed from a shard set: packed streams -> stream g at d_coded + g * cap_per_shard, the rest of
 * the area zeroed. */
int qvz_hip_scatter_streams(const uint8_t *d_packed, const uint64_t *h_sizes, uint32_t shards, uint8_t *d_coded,
                            uint64_t cap_per_shard, void *stream);

/* GPU time of the last qvz_hip_encode_planes call, split into its model kernels and its coder kernels (HIP events on
 * the call's stream, summed over the segments). Either pointer may be NULL. */
void qvz_hip_encode_planes_last_ms(double *model_ms, double *coder_ms);
/* What the first qvz_hip_encode_planes call of a process needs and no codebook (its stream pair made and tested for running side
 * by side, this file's code object loaded, the loaders' reciprocal check): ~20-25 ms that a one-shot process spends beside its
 * upload, on a thread of its own, instead of in front of its first coder kernel. Optional; the current device is the caller's. */
int qvz_hip_encode_planes_warmup(void);
/* The loaders of the coder kernel turn a packed record's integers into quotients with v_rcp_f64 and ONE Newton step when that is
 * within 2^-47 of 1 / (n 2^10) for every n a record can hold (n = 1 .. 2^19 + 1024) on the current device - measured once per process
 * and device over all of them -, with two steps otherwise (and with the test hook QVZ_CODER=newton=2). Returns 1 for one step; *worst (may be
 * NULL) receives the worst relative error measured for one step. */
int qvz_coder_rcp_one_step_ok_ex(double *worst);
int qvz_coder_rcp_one_step_ok(void);
/* qvz_hip_encode_planes keeps its internal device buffers (table images, per-stream table saves, read lists: a few GB at
 * 100M x 150) and its two streams per device between calls; this gives the buffers of the current device back. */
void qvz_hip_encode_planes_release(void);

/* ---------------------------------------------------------------------------------------------------------
 * Strict single-stream mode with the model on the GPU. The reference's file has ONE serial arithmetic stream
 * (src/qv_stream.c:109-110); its interval chain (src/arith.c:35-97) stays on one host core, but update_stats and the
 * cumulative counts (src/qv_stream.c:9-25, src/arith.c:39-43) do not: a table belongs to one column, so a model session
 * runs the adaptive model of the whole file on the GPU (a wave per column, tables in LDS, state carried from chunk to
 * chunk) and leaves, per symbol, a 16-byte record of two 64-bit integers: floor(2^62 * c / n) + 1 for c = cum_lo and
 * c = cum_hi. The host coder (qvz_coder_encode_records) turns a record into the new bounds with two 64 x 64 -> 128 bit
 * multiplies - exactly floor(range * c / n) of arith.c:47-48, proof in qvz_coder.c - renormalises and writes bits.
 * Byte-identical to qvz_coder_encode on the feed.
 * open fails with QVZ_E_NOMEM when a column's tables do not fit LDS (the caller then keeps the model on the host).
 * Record (row, pos) of a chunk lies at ((row / 2) * (pitch / 4) + pos / 4) * 8 + (row % 2) * 4 + pos % 4 (in records);
 * positions per read: one cluster: pos = column; several: pos 0 = the cluster id, 1..7 unused, 8 + column.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct qvz_model_session qvz_model_session_t;
int qvz_hip_model_session_open(uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, qvz_model_session_t **out);
uint32_t qvz_hip_model_session_pitch(const qvz_model_session_t *s);     /* record positions per read, a multiple of 8 */
/* records of the next n reads of the stream (planes d_qi / d_st / d_labels start at this chunk's first read) into d_rec
 * ((n rounded up to 2) * pitch * 16 bytes); asynchronous on `stream`, chunks in file order */
int qvz_hip_model_session_records(qvz_model_session_t *s, const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld,
                                  const uint8_t *d_labels, uint64_t n, void *d_rec, void *stream);
int qvz_hip_model_session_status(qvz_model_session_t *s, void *stream); /* QVZ_E_DEAD_CTX if a symbol had no table; synchronises */
void qvz_hip_model_session_close(qvz_model_session_t *s);
/* the host half: `reads` reads of records (layout above; h_rec in host memory) through the interval chain of `cd` */
int qvz_coder_encode_records(qvz_coder_t *cd, const void *h_rec, uint64_t reads, uint32_t pitch);

/* Initial adaptive model as a flat arena of tables [n][card][counts...] (cluster-id table at index 0) plus the
 * (cluster, column, quantizer) -> arena index map (0xFFFFFFFF = none). Shared by the host and GPU coders.
 * Caller frees *arena_out and *offset_out with free(). Returns 0 on success. */
#define QVZ_MODEL_LIVE_ONLY 1   /* leave out tables of dead contexts (qvz_codebook_t.live == 0) */
#define QVZ_MODEL_ALIGN16 2     /* start every table on a 16-byte boundary */
int qvz_model_build(uint32_t clusters, uint32_t columns, const qvz_codebook_t *const *cbs, int flags,
                    uint32_t **arena_out, uint32_t **offset_out, uint32_t *words_out);

/* Reference-format file image of one shard: header(clusters, columns, shard_reads) + codebooks + 128-byte WELL
 * state + stream. buf == NULL returns the size. */
size_t qvz_shard_file_image(uint8_t *buf, uint32_t clusters, uint32_t columns, uint64_t shard_reads,
                            const qvz_codebook_t *const *cbs, const uint32_t *well_state_at_shard,
                            const uint8_t *stream, uint64_t stream_bytes);

#ifdef __cplusplus
}
#endif
#endif
