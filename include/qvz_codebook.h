/* qvz_codebook.h - C-ABI for codebook generation on the MI355X and the flat host-side codebook it fills.
 *
 * Replaces, for one cluster, the marginal pass of calculate_statistics (reference src/codebook.c:208-219) and
 * generate_codebooks (src/codebook.c:355-468) with its helpers optimize_for_entropy (:230-269),
 * compute_qpmf_quan_list (:274-289), compute_qpmf_list (:291-330), compute_xpmf_list (:332-349) and
 * generate_quantizer / apply_quantizer (src/quantizer.c:34-161).
 *
 * Where the arithmetic runs: every probability table, every Lloyd-Max design and every output pmf is computed
 * by HIP kernels in the reference's summation order (FP64, no FMA contraction). The entropies that decide codebook
 * bytes (src/pmf.c:141-155, src/codebook.c:244-268) are libm log2 values in the reference; since round 3 the device
 * evaluates glibc's own log2 (qvz_amd/csrc/qvz_log2_glibc.h: the routine restated operation by operation, its
 * constants read out of libm.so.6) once a run-time check has found it bit-identical to this host's log2
 * (qvz_hip_codebook_warmup / qvz_hip_codebook_exact_entropies). Until then - or with QVZ_CB_LOG2=host - the device's
 * own log2 only steers the search, inside a guard band, and the host's libm takes every decision that falls inside
 * the band and computes the entropies that become `ratio`.
 */
#ifndef QVZ_CODEBOOK_H
#define QVZ_CODEBOOK_H

#include <stddef.h>
#include <stdint.h>

#include "qvz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Flat form of cond_quantizer_list_t + quantizer_t (reference include/codebook.h:61-69, quantizer.h:16-22).
 * Index 2*ctx+0 is the "lo" quantizer of a context, 2*ctx+1 the "hi" one. */
typedef struct qvz_codebook {
    uint32_t C;
    uint8_t *nctx;     /* [C]            number of contexts of column s (1 for s = 0)                       */
    uint8_t *ctx_sym;  /* [C][72]        context symbols in alphabet (merge) order                            */
    uint8_t *ctx_idx;  /* [C][72]        previous quantized value -> context index, QVZ_NOT_FOUND if none     */
    uint8_t *qratio;   /* [C][72]        (uint8)(ratio*128)              store_cond_quantizers_indexed :156   */
    double  *ratio;    /* [C][72]        probability of using the lo quantizer                                */
    uint8_t *q;        /* [C][144][72]   quantizer tables                                                     */
    uint8_t *nstates;  /* [C][144]       size of the output alphabet                                          */
    uint8_t *recon;    /* [C][144][72]   output alphabet (reconstruction points, first nstates valid)         */
    uint8_t *out_idx;  /* [C][144][72]   value -> state index in the output alphabet (last writer wins)       */
    uint8_t *live;     /* [C][72]        1 if the context's pmf P(X_s | Q_{s-1}) has mass, 0 for a "dead" context:
                          a previous value no read can produce (the reference still stores quantizers for it,
                          SURVEY.md Appendix E.5). Not in the file format; lets per-shard coder models skip them. */
} qvz_codebook_t;

qvz_codebook_t *qvz_codebook_alloc(uint32_t C);
void qvz_codebook_free(qvz_codebook_t *cb);

/* distortion matrices d[x + 72*y] (reference src/distortion.c:50-95); host, 41 KB */
int qvz_distortion_matrix(int type, double *h_dist /*[72*72]*/);
/* -D FILE (reference src/distortion.c:100-145): up to 72 rows of comma-separated doubles (atof), row x column y =
 * the cost of reconstructing true value x as y = d[x + 72*y]; lines starting with '#' are comments. Entries or rows
 * the file does not hold are 0.0, with the reference's warning on stdout (there the short-row case never returns:
 * its fill loop does not advance). Returns QVZ_OK, or QVZ_E_ARG when the file cannot be opened. */
int qvz_distortion_matrix_file(const char *path, double *h_dist /*[72*72]*/);

/* d_cnt: this cluster's counters [C][72][72] (device). mode/ratio: -f / -r and its argument.
 * guard: width of the band (absolute, on entropies in bits) inside which device-side search decisions are
 * deferred to the host; pass 0 for the default (1e-9). A huge guard makes the host take every decision
 * (used by the tests to prove the two routes agree). Returns QVZ_OK and fills `out`. */
int qvz_hip_generate_codebook(const uint32_t *d_cnt, uint32_t C, const double *h_dist, int mode, double ratio,
                              double guard, qvz_codebook_t *out, void *stream);

/* The same from 64-bit counters: the counters of a file of 2^32 reads or more, summed over the parts that counted it
 * (host/qvz_multi.c). The reference's u32 counts and totals wrap there (include/pmf.h:35-36), so this is an extension,
 * not a restatement; whenever the u32 counters do not wrap it gives the same codebook, bit for bit. */
int qvz_hip_generate_codebook64(const uint64_t *d_cnt, uint32_t C, const double *h_dist, int mode, double ratio,
                                double guard, qvz_codebook_t *out, void *stream);

/* Marginal pmfs of one cluster, h_marg[s*72 + x]: the second half of calculate_statistics
 * (reference src/codebook.c:208-219, combine_pmfs src/pmf.c:189-205), bit-identical to the reference's values. */
int qvz_hip_marginals(const uint32_t *d_cnt, uint32_t C, double *h_marg, void *stream);

/* Starts, on a host thread of its own, the one-time check that lets the device's entropies decide: glibc's log2 as restated
 * for the device (qvz_amd/csrc/qvz_log2_glibc.h) against this host's log2 on 2^20 arguments, bit for bit. Optional: the
 * first qvz_hip_generate_codebook of a process runs (and waits for) the check itself; a caller that knows a generation
 * is coming calls this early. QVZ_CB_LOG2=host: no check, the host's libm takes every decision as before round 3. */
int qvz_hip_codebook_warmup(void);
/* 1 if the check has passed on the current device (runs / waits for it): the device's entropies take the decisions;
 * 0: they only steer and the host's libm decides (the check failed, or QVZ_CB_LOG2=host) */
int qvz_hip_codebook_exact_entropies(void);
/* The run-time check is a sample (2^20 arguments): it catches another libm, it cannot prove equality on the arguments a given
 * pmf produces - "the reference's bytes on this host" is a statistical claim as far as the ratios go. The DECISIONS are
 * protected further: an `entropy < target` comparison whose two sides lie within 1e-12 of each other (a one-bit miss of the
 * log2 restatement moves an entropy by ~1e-15) is taken again by the host's libm from the device-made pmfs, target included.
 * Process-wide counts of such re-decisions and of those where the libm's numbers differed from the device's by a bit. */
void qvz_hip_codebook_exact_recheck_counts(unsigned long long *rechecked, unsigned long long *mismatched);

/* Number of host-side decisions the last qvz_hip_generate_codebook call OF THE CALLING THREAD had to take inside the
 * guard band. (qvz_hip_generate_codebook may run concurrently on several host threads, one cluster and one stream each:
 * the clusters' codebooks are independent, and a generation is a chain of small kernels and host round trips.) */
uint64_t qvz_hip_codebook_deferred_decisions(void);

/* Device tables for the quantizer (layout in qvz_hip.h) from a host codebook: h_ct [C][72], h_qs [C][144][72] */
void qvz_codebook_tables(const qvz_codebook_t *cb, uint16_t *h_ct, uint16_t *h_qs);

/* Serialisation, byte-identical to write_codebooks / write_codebook (reference src/codebook.c:474-555).
 * buf == NULL returns the size only. */
size_t qvz_codebook_write_header(uint8_t *buf, uint32_t clusters, uint32_t columns, uint64_t lines);
size_t qvz_codebook_write(uint8_t *buf, const qvz_codebook_t *cb);
/* The inverse, read_codebooks (reference src/codebook.c:560-669) for one cluster: parses the codebook text at
 * buf[*pos ...] into cb (allocated for the file's column count), advancing *pos. Output alphabets are derived the
 * way the reference's decoder derives them (run de-duplication of each table, src/quantizer.c:167-191); `ratio` is
 * qratio / 128 and every context counts as live. */
int qvz_codebook_read(const uint8_t *buf, size_t size, size_t *pos, qvz_codebook_t *cb);

#ifdef __cplusplus
}
#endif
#endif
