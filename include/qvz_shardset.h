/* qvz_shardset.h - ONE file for a sharded encode.
 *
 * The reference's .q format (src/codebook.c:474-555, src/qv_stream.c:89) holds exactly one serial arithmetic stream, so a
 * sharded encode (include/qvz_coder.h, SURVEY.md section 8e option B) as reference-format files repeats the codebooks in
 * every shard file: at 100M x 150 reads and 8192 streams that is ~8 GB of codebook copies beside 2.6 GB of streams.
 * A shard set stores what the shards share once:
 *
 *   offset 0    u8   0x00              a reference file starts with its cluster count (>= 1, codebook.c:483): never 0
 *          1    7 B  "QVZSET1"
 *          8    u32  clusters K        (little-endian from here on)
 *         12    u32  columns C
 *         16    u64  lines N           64 bits: more than the 2^32 - 1 lines a reference header can name
 *         24    u32  shards S
 *         28    u32  0
 *         32    128 B WELL-1024a state of read 0 (as qv_stream.c:89 stores it)
 *        160    u64  stream_bytes[S]
 *        ...    the K codebooks, byte for byte as write_codebook writes them (codebook.c:505-555)
 *        ...    the S streams back to back; stream g codes reads [first[g], first[g+1]) of qvz_shard_bounds(N, S)
 *
 * Stream g together with the codebooks, a 9-byte header naming first[g+1] - first[g] lines and the WELL state advanced by
 * first[g] * C / 4 words (stored with n = 0) IS the reference-format file of those reads: qvz_shardset_split writes these
 * files for anyone who wants the unmodified reference decoder to read them; qvz_shardset_decode / `qvz -x` decodes the
 * set directly (all streams at once on the GPU). */
#ifndef QVZ_SHARDSET_H
#define QVZ_SHARDSET_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "qvz_codebook.h"
#include "qvz_encode.h"

#ifdef __cplusplus
extern "C" {
#endif

#define QVZ_SHARDSET_MAGIC "\0QVZSET1"
#define QVZ_SHARDSET_FIXED 160u          /* bytes before the stream-size table */

typedef struct qvz_shardset_info {
    uint32_t clusters, columns, shards;
    uint64_t lines;
    uint64_t header_bytes;     /* fixed part + size table */
    uint64_t codebook_bytes;   /* the K codebooks, once */
    uint64_t stream_bytes;     /* sum of the streams: what the reference's -s calls "size" (main.c:125) */
    uint64_t file_bytes;       /* everything */
} qvz_shardset_info_t;

/* does this buffer start like a shard set? */
int qvz_shardset_is(const uint8_t *buf, size_t size);

/* Writes a shard set to f. streams: stream g at streams + g * stride, sizes[g] bytes of it valid (the GPU coder's
 * fixed-stride layout), or with stride = 0 the streams packed back to back (qvz_hip_gather_streams). info (may be NULL)
 * receives the byte counts. */
int qvz_shardset_write(FILE *f, uint32_t clusters, uint32_t columns, uint64_t lines, uint32_t shards,
                       const uint32_t *well_state, const qvz_codebook_t *const *cbs, const uint8_t *streams,
                       uint64_t stride, const uint64_t *sizes, qvz_shardset_info_t *info);

/* Everything of a shard set that comes before the streams - fixed part, size table, the codebooks - for a writer that
 * sends the streams after it piece by piece (the whole-file driver streams them out of the device through pinned slabs).
 * info->stream_bytes / file_bytes count the streams the size table promises. */
int qvz_shardset_write_head(FILE *f, uint32_t clusters, uint32_t columns, uint64_t lines, uint32_t shards,
                            const uint32_t *well_state, const qvz_codebook_t *const *cbs, const uint64_t *sizes,
                            qvz_shardset_info_t *info);

/* header fields and byte counts of the shard set at `path` */
int qvz_shardset_stat(const char *path, qvz_shardset_info_t *info);

/* Materialises the reference-format files "<prefix>.0000" ... "<prefix>.<S-1>" of shards [first_shard, first_shard +
 * count) (count = 0: all). Each is accepted by the unmodified reference decoder; their decodes concatenate to the -u text.
 * Needs the GPU for the WELL jump-ahead (qvz_hip_well_jump_many). */
int qvz_shardset_split(const char *path, const char *prefix, uint32_t first_shard, uint32_t count, int device);

/* `qvz -x` for a shard set: every stream is one lane of the GPU decoder (qvz_hip_decode_shards). */
int qvz_shardset_decode(const char *path, const char *out_path, int device, qvz_result_t *res);

/* The encode driver with a shard set as its output (the default of a sharded encode: QVZ_SHARDS=N qvz in out). */
int qvz_encode_file_shardset(const char *in_path, const char *out_path, const qvz_options_t *opts, uint32_t shards,
                             qvz_result_t *res);

#ifdef __cplusplus
}
#endif
#endif
