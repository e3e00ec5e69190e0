/* fuzz_stubs.c - the device half of libqvzhip.so as seen by the parser fuzz harness (tests/fuzz_host.c): every
 * qvz_hip_* entry point the host C sources call, each failing with QVZ_E_HIP ("no device"). TEST INFRASTRUCTURE. The
 * harness is a CPU program by construction: host C sources + this file, one plain clang invocation, no device code, no
 * HIP runtime, nothing that could start GPU work - whichever box it is built on. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "qvz_codebook.h"
#include "qvz_coder.h"
#include "qvz_encode.h"
#include "qvz_hip.h"
#include "qvz_shardset.h"

static char g_err[512] = "fuzz build: no device";
const char *qvz_hip_last_error(void) { return g_err; }
int qvz_hip_set_error(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); return code; }
uint64_t qvz_hip_ld(uint64_t n) { return (n + 255) & ~(uint64_t)255; }
static int nodev(void) { return qvz_hip_set_error(QVZ_E_HIP, "fuzz build: no device"); }
#define U(x) (void)(x)

uint64_t qvz_hip_codebook_deferred_decisions(void) { return 0; }
int qvz_hip_copy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) { U(h_dst); U(d_src); U(bytes); U(stream); return nodev(); }
int qvz_hip_copy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) { U(d_dst); U(h_src); U(bytes); U(stream); return nodev(); }
int qvz_hip_decode_shards_bounds(const uint8_t *d_in, uint64_t cap_per_shard, const uint64_t *h_sizes, uint64_t n, uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards, const uint64_t *h_bounds, const uint8_t *d_draws, uint64_t ld, uint8_t *d_qv, uint8_t *d_labels, void *stream) { U(d_in); U(cap_per_shard); U(h_sizes); U(n); U(columns); U(clusters); U(cbs); U(shards); U(h_bounds); U(d_draws); U(ld); U(d_qv); U(d_labels); U(stream); return nodev(); }
int qvz_hip_encode_planes_bounds(const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n, uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards, const uint64_t *h_bounds, uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes, void *d_work, uint64_t work_bytes, void *stream) { U(d_qi); U(d_st); U(ld); U(d_labels); U(n); U(columns); U(clusters); U(cbs); U(shards); U(h_bounds); U(d_out); U(cap_per_shard); U(h_sizes); U(d_work); U(work_bytes); U(stream); return nodev(); }
int qvz_hip_free(void *d_ptr) { U(d_ptr); return nodev(); }
int qvz_hip_free_host(void *h_ptr) { U(h_ptr); return nodev(); }
int qvz_hip_gather_streams(const uint8_t *d_coded, uint64_t cap_per_shard, const uint64_t *h_sizes, uint32_t shards, uint8_t *d_packed, void *stream) { U(d_coded); U(cap_per_shard); U(h_sizes); U(shards); U(d_packed); U(stream); return nodev(); }
int qvz_hip_generate_codebook(const uint32_t *d_cnt, uint32_t C, const double *h_dist, int mode, double ratio, double guard, qvz_codebook_t *out, void *stream) { U(d_cnt); U(C); U(h_dist); U(mode); U(ratio); U(guard); U(out); U(stream); return nodev(); }
int qvz_hip_generate_codebook64(const uint64_t *d_cnt, uint32_t C, const double *h_dist, int mode, double ratio, double guard, qvz_codebook_t *out, void *stream) { U(d_cnt); U(C); U(h_dist); U(mode); U(ratio); U(guard); U(out); U(stream); return nodev(); }
int qvz_hip_kmeans_step(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint32_t k, const uint8_t *d_means , uint8_t *d_labels , uint32_t *d_counts , uint64_t *d_acc , void *stream) { U(d_sym); U(n); U(c); U(ld); U(k); U(d_means); U(d_labels); U(d_counts); U(d_acc); U(stream); return nodev(); }
int qvz_hip_kmeans_update(const uint64_t *d_acc, const uint32_t *d_counts, uint32_t k, uint32_t c, uint8_t *d_means, double *d_moved, int32_t *d_status, void *stream) { U(d_acc); U(d_counts); U(k); U(c); U(d_means); U(d_moved); U(d_status); U(stream); return nodev(); }
int qvz_hip_kmeans_update64(const uint64_t *d_acc, const uint64_t *d_counts, uint32_t k, uint32_t c, uint8_t *d_means, double *d_moved, int32_t *d_status, void *stream) { U(d_acc); U(d_counts); U(k); U(c); U(d_means); U(d_moved); U(d_status); U(stream); return nodev(); }
int qvz_hip_malloc(void **d_ptr, size_t bytes) { U(d_ptr); U(bytes); return nodev(); }
int qvz_hip_malloc_host(void **h_ptr, size_t bytes) { U(h_ptr); U(bytes); return nodev(); }
int qvz_hip_memset(void *d_ptr, int value, size_t bytes, void *stream) { U(d_ptr); U(value); U(bytes); U(stream); return nodev(); }
void qvz_hip_model_session_close(qvz_model_session_t *s) { U(s); }
int qvz_hip_model_session_open(uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, qvz_model_session_t **out) { U(columns); U(clusters); U(cbs); U(out); return nodev(); }
uint32_t qvz_hip_model_session_pitch(const qvz_model_session_t *s) { U(s); return 0; }
int qvz_hip_model_session_records(qvz_model_session_t *s, const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n, void *d_rec, void *stream) { U(s); U(d_qi); U(d_st); U(ld); U(d_labels); U(n); U(d_rec); U(stream); return nodev(); }
int qvz_hip_model_session_status(qvz_model_session_t *s, void *stream) { U(s); U(stream); return nodev(); }
int qvz_hip_pack_feed(const uint8_t *d_qi, const uint8_t *d_st, uint64_t n, uint32_t c, uint64_t ld, uint16_t *d_feed, void *stream) { U(d_qi); U(d_st); U(n); U(c); U(ld); U(d_feed); U(stream); return nodev(); }
int qvz_hip_pmf_hist(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c, uint64_t ld, uint32_t k, uint32_t *d_cnt, void *stream) { U(d_sym); U(d_labels); U(n); U(c); U(ld); U(k); U(d_cnt); U(stream); return nodev(); }
int qvz_hip_quantize(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c, uint64_t ld, uint32_t k, const uint16_t *d_ct, const uint16_t *d_qs, const uint8_t *d_draws, uint64_t ld_draws, const double *d_dist, uint8_t *d_qv, uint8_t *d_qi, uint8_t *d_st, uint64_t ld_out, double *d_err, int32_t *d_status, void *stream) { U(d_sym); U(d_labels); U(n); U(c); U(ld); U(k); U(d_ct); U(d_qs); U(d_draws); U(ld_draws); U(d_dist); U(d_qv); U(d_qi); U(d_st); U(ld_out); U(d_err); U(d_status); U(stream); return nodev(); }
int qvz_hip_set_device(int device) { U(device); return nodev(); }
int qvz_hip_stream_create(void **stream) { U(stream); return nodev(); }
int qvz_hip_stream_destroy(void *stream) { U(stream); return nodev(); }
int qvz_hip_stream_sync(void *stream) { U(stream); return nodev(); }
int qvz_hip_transpose_text(const uint8_t *d_text, uint64_t n, uint32_t c, uint8_t *d_sym, uint64_t ld, uint32_t *d_bad, void *stream) { U(d_text); U(n); U(c); U(d_sym); U(ld); U(d_bad); U(stream); return nodev(); }
int qvz_hip_untranspose_text(const uint8_t *d_qv, uint64_t n, uint32_t c, uint64_t ld, uint8_t *d_text, void *stream) { U(d_qv); U(n); U(c); U(ld); U(d_text); U(stream); return nodev(); }
int qvz_hip_well_draws(const uint32_t *h_state, uint64_t first_read, uint64_t n, uint32_t c, uint8_t *d_draws, uint64_t ld, void *stream) { U(h_state); U(first_read); U(n); U(c); U(d_draws); U(ld); U(stream); return nodev(); }
int qvz_hip_well_jump_many(const uint32_t *h_state, const uint64_t *h_words, uint64_t count, uint32_t *h_states_out) { U(h_state); U(h_words); U(count); U(h_states_out); return nodev(); }
int qvz_hip_widen_u32(const uint32_t *d_src, uint64_t *d_dst, uint64_t count, void *stream) { U(d_src); U(d_dst); U(count); U(stream); return nodev(); }
int qvz_hip_stream_wait(void *stream, void *other) { U(stream); U(other); return nodev(); }
int qvz_hip_streams_overlap(void *stream_a, void *stream_b, int *overlap) { U(stream_a); U(stream_b); U(overlap); return nodev(); }
int qvz_hip_untranspose_rows(const uint8_t *d_qv, uint64_t first, uint64_t n, uint32_t c, uint64_t ld, uint8_t *d_text, void *stream) { U(d_qv); U(first); U(n); U(c); U(ld); U(d_text); U(stream); return nodev(); }
int qvz_hip_scatter_streams(const uint8_t *d_packed, const uint64_t *h_sizes, uint32_t shards, uint8_t *d_coded, uint64_t cap_per_shard, void *stream) { U(d_packed); U(h_sizes); U(shards); U(d_coded); U(cap_per_shard); U(stream); return nodev(); }
uint64_t qvz_hip_decode_tables_bytes(uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs) { U(columns); U(clusters); U(cbs); return 0; }
int qvz_hip_mem_info(uint64_t *free_bytes, uint64_t *total_bytes) { U(free_bytes); U(total_bytes); return nodev(); }
uint64_t qvz_hip_encode_planes_work_bytes(uint64_t longest_reads, uint32_t columns, uint32_t clusters, uint32_t shards) { U(longest_reads); U(columns); U(clusters); U(shards); return 0; }
int qvz_hip_encode_planes_prepare(uint64_t n, uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards, const uint64_t *h_bounds, uint64_t work_bytes) { U(n); U(columns); U(clusters); U(cbs); U(shards); U(h_bounds); U(work_bytes); return nodev(); }
int qvz_hip_codebook_warmup(void) { return nodev(); }
int qvz_hip_codebook_exact_entropies(void) { return 0; }
int qvz_hip_encode_planes_warmup(void) { return QVZ_E_HIP; }
int qvz_hip_event_create(void **event) { U(event); return nodev(); }
int qvz_hip_event_record(void *event, void *stream) { U(event); U(stream); return nodev(); }
int qvz_hip_event_elapsed_ms(void *first, void *second, double *ms) { U(first); U(second); U(ms); return nodev(); }
int qvz_hip_event_destroy(void *event) { U(event); return nodev(); }
int qvz_hip_sum_errors(const double *d_err, uint64_t n, uint32_t c, double *d_sum, void *stream) { U(d_err); U(n); U(c); U(d_sum); U(stream); return nodev(); }
int qvz_hip_gather_reads(const uint8_t *d_sym, uint64_t ld, uint32_t c, const uint64_t *d_idx, uint32_t k, uint32_t *d_out, void *stream) { U(d_sym); U(ld); U(c); U(d_idx); U(k); U(d_out); U(stream); return nodev(); }
void qvz_hip_encode_planes_last_ms(double *model_ms, double *coder_ms) { if (model_ms) *model_ms = 0.0; if (coder_ms) *coder_ms = 0.0; }
