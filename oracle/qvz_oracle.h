/* qvz_oracle: CPU restatement of the reference qvz ENCODE hot path, flat arrays, single thread.
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; nothing under qvz_amd/ (the product) links, imports or executes it.
 *
 * Parity status: PINNED. Checked byte-for-byte against the unmodified reference compiled from
 * /root/reference by oracle/build_ref.sh (oracle/_ref/qvz_ref_fixedseed, libqvzref.so) and against
 * the committed fixtures those binaries produced (tests/golden/, generator: tools/make_golden.py).
 *
 * Every function cites the reference file:line it restates. Symbols are byte-33 in [0,72).
 */
#ifndef QVZ_ORACLE_H
#define QVZ_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_A 72            /* alphabet, reference src/main.c:13 */
#define ORC_NOT_FOUND 0xFF  /* flat stand-in for ALPHABET_SYMBOL_NOT_FOUND, include/pmf.h:9 */

enum { ORC_MODE_RATIO = 0, ORC_MODE_FIXED = 1 };              /* include/codebook.h:21-22 */
enum { ORC_DIST_MANHATTAN = 1, ORC_DIST_MSE = 2, ORC_DIST_LORENTZ = 3 }; /* include/distortion.h:7-9 */

/* One cluster's conditional quantizers as flat tables (reference: cond_quantizer_list_t,
 * include/codebook.h:61-69 + quantizer_t include/quantizer.h:16-22). All arrays caller-visible. */
typedef struct orc_codebook {
    uint32_t C;
    uint8_t *nctx;     /* [C]          contexts in column s (1 for s=0)                         */
    uint8_t *ctx_sym;  /* [C][72]      context symbols, alphabet order (input_alphabets[s])      */
    uint8_t *ctx_idx;  /* [C][72]      symbol -> context index, ORC_NOT_FOUND if absent          */
    uint8_t *qratio;   /* [C][72]      (uint8)(ratio*128), codebook.c:156                        */
    double  *ratio;    /* [C][72]                                                                */
    uint8_t *q;        /* [C][144][72] quantizer tables, index 2*ctx+{0 lo,1 hi}                 */
    uint8_t *nstates;  /* [C][144]     output alphabet size                                      */
    uint8_t *recon;    /* [C][144][72] output alphabet symbols (first nstates valid)             */
    uint8_t *out_idx;  /* [C][144][72] value -> state index (last writer wins), ORC_NOT_FOUND    */
    double  *mse;      /* [C][144]                                                               */
} orc_codebook_t;

orc_codebook_t *orc_codebook_alloc(uint32_t C);
void orc_codebook_free(orc_codebook_t *cb);

/* distortion.c:50-95 : d[x + 72*y] */
int orc_distortion(int type, double *d /*[72*72]*/);

/* cluster.c:192-206 : centroid read indices chosen with libc rand() exactly as the reference does
 * (block = rand()%block_count, line = rand()%count). reseed!=0 calls srand(1) first, which is the
 * state a fresh process starts in. */
void orc_pick_centroids(uint64_t n, uint32_t k, int reseed, uint64_t *lines_out);

/* cluster.c:65-244. rows: text layout, stride c+1, raw bytes (Q+33). labels must be zeroed by
 * the caller (calloc in lines.c:116). means_out: k*c raw-byte means. Returns iterations, or -1 if a
 * cluster went empty (reference divides by zero there, cluster.c:113). */
int orc_kmeans(const uint8_t *rows, uint64_t n, uint32_t c, uint32_t k, double threshold,
               const uint64_t *centroid_lines, uint8_t *labels, uint8_t *means_out,
               uint32_t *counts_out);

/* One assignment pass of the loop above (cluster.c:65-75, :136-187) against given means (raw bytes): labels and counts. The
 * reference's final labels are those of its LAST assignment pass, i.e. against the means before the last update. */
void orc_assign(const uint8_t *rows, uint64_t n, uint32_t c, uint32_t k, const uint8_t *means, uint8_t *labels, uint32_t *counts);

/* codebook.c:193-205 : cnt[k][C][72][72] u32 (column 0 uses prev row 0). Increments: cnt zeroed by the
 * caller for a file's counters, or carried over to count a file in pieces. */
void orc_stats(const uint8_t *rows, const uint8_t *labels, uint64_t n, uint32_t c, uint32_t k,
               uint32_t *cnt);

/* codebook.c:208-219 + pmf.c:189-230 for ONE cluster: cnt[C][72][72] -> marg[C][72] */
void orc_marginals(const uint32_t *cnt, uint32_t C, double *marg);

/* quantizer.c:34-132 (+ apply_quantizer :139-161, get_entropy pmf.c:141-155).
 * Returns entropy of the output pmf; fills q[72], recon[S], out_idx[72], *mse. */
double orc_lloyd_max(const double *p, const double *d, uint32_t S, uint8_t *q, uint8_t *recon,
                     uint8_t *out_idx, double *mse);

/* codebook.c:355-468 for ONE cluster. */
void orc_generate_codebook(const uint32_t *cnt, uint32_t C, const double *d, int mode, double ratio,
                           orc_codebook_t *cb);

/* codebook.c:474-555 : header (9 bytes) is written by orc_write_header, one codebook per cluster by
 * orc_write_codebook. Both return bytes written (or needed, if buf==NULL). */
size_t orc_write_header(uint8_t *buf, uint32_t k, uint32_t C, uint64_t N);
size_t orc_write_codebook(uint8_t *buf, const orc_codebook_t *cb);

/* well.c:8-46 */
typedef struct { uint32_t state[32]; uint32_t n, bit_output, bits_left; } orc_well_t;
uint32_t orc_well_word(orc_well_t *w);
uint32_t orc_well_bits(orc_well_t *w, uint32_t bits);
/* states after words[i] (ascending) calls from state0 (n = 0), each rotated to n = 0; by stepping only */
void orc_well_states_at(const uint32_t *state0, const uint64_t *words, uint64_t count, uint32_t *out /*[count][32]*/);

/* qv_compressor.c:48-143 + qv_stream.c + arith.c + os_stream.c.
 * cbs: k codebooks. well_state: the 32 words written to the file before the stream.
 * u_out (optional): n*(c+1) bytes of "-u" text. stream_out: capacity stream_cap; returns stream
 * bytes (os->written) or (uint64_t)-1 on overflow. qi_out/st_out (optional): n*c coder feed. */
uint64_t orc_encode_stream(const uint8_t *rows, const uint8_t *labels, uint64_t n, uint32_t c,
                           uint32_t k, orc_codebook_t *const *cbs, const double *d,
                           const uint32_t *well_state, uint8_t *u_out, uint8_t *stream_out,
                           uint64_t stream_cap, double *distortion_out, uint8_t *qi_out,
                           uint8_t *st_out);

/* Whole encode() of main.c:18-127 on an in-memory text buffer: returns .q size written into q_out
 * (capacity q_cap), fills u_out (n*(c+1)) if non-NULL. stats_out: [0]=rate [1]=distortion
 * [2]=stream bytes [3]=kmeans iterations. */
uint64_t orc_encode(const uint8_t *rows, uint64_t n, uint32_t c, uint32_t k, int mode, double ratio,
                    int dist_type, double threshold, const uint32_t *well_state, uint8_t *q_out,
                    uint64_t q_cap, uint8_t *u_out, double *stats_out);

/* Decoder side (qv_compressor.c:145-231, codebook.c:560-669, arith.c:118-205): parses a whole .q
 * image and reproduces the text the reference's `qvz -x` writes. Returns bytes written to out. */
uint64_t orc_decode(const uint8_t *qfile, uint64_t qsize, uint8_t *out, uint64_t out_cap);

/* QGEN-1 rows [first, first+n) in reference text layout (stride c+1, Q+33, '\n'); qgen1.h */
void orc_qgen1_rows(uint64_t seed, uint64_t first, uint64_t n, uint32_t c, uint8_t *rows);

#ifdef __cplusplus
}
#endif
#endif
