/* qvz_coder.c - see include/qvz_coder.h. Plain C, one stream per coder object, no global state. */
#include "qvz_coder.h"

#include <stdlib.h>
#include <string.h>

#define A QVZ_A
#define M_BITS 22u                           /* m_arith, reference include/qvz_compressor.h:19 */
#define MSB (1u << (M_BITS - 1))
#define MASK_M ((1u << M_BITS) - 1)
#define MASK_M1 ((1u << (M_BITS - 1)) - 1)
#define RESCALE (1u << (M_BITS - 3))         /* a->r, arith.c:11 */
#define STEP 8u                              /* qv_stream.c:56 */
#define NO_TABLE 0xFFFFFFFFu

struct qvz_coder {
    uint32_t clusters, columns;
    uint32_t *arena;          /* tables back to back: [n][card][counts...] */
    uint32_t *offset;         /* [clusters][columns][144] -> arena index or NO_TABLE */
    uint32_t cluster_table;   /* arena index of the cluster-id model */
    uint32_t l, u;
    uint64_t scale3;
    uint64_t acc;             /* bit accumulator, MSB-first */
    uint32_t nbits;
    uint8_t *buf;
    uint64_t len, cap, dropped;
};

static int buf_reserve(qvz_coder_t *cd, uint64_t extra) {
    if (cd->len + extra <= cd->cap) return 0;
    uint64_t ncap = cd->cap ? cd->cap * 2 : (1u << 20);
    while (ncap < cd->len + extra) ncap *= 2;
    uint8_t *nb = (uint8_t *)realloc(cd->buf, ncap);
    if (!nb) return -1;
    cd->buf = nb;
    cd->cap = ncap;
    return 0;
}

/* append `count` (<= 32) bits, most significant first (stream_write_bit order, os_stream.c:72-92) */
static inline void put_bits(qvz_coder_t *cd, uint32_t value, uint32_t count) {
    cd->acc = (cd->acc << count) | (value & (count == 32 ? 0xFFFFFFFFu : ((1u << count) - 1)));
    cd->nbits += count;
    while (cd->nbits >= 8) {
        cd->nbits -= 8;
        cd->buf[cd->len++] = (uint8_t)(cd->acc >> cd->nbits);
    }
}

static inline void put_run(qvz_coder_t *cd, uint32_t bit, uint64_t count) {
    const uint32_t pattern = bit ? 0xFFFFFFFFu : 0u;
    while (count) {
        uint32_t c = count > 32 ? 32 : (uint32_t)count;
        if (cd->len + 8 > cd->cap) buf_reserve(cd, 4096);
        put_bits(cd, pattern, c);
        count -= c;
    }
}

/* Initial adaptive model (initialize_stream_stats qv_stream.c:32-61 and :96-104) as one flat arena of tables
 * [n][card][counts...]: the cluster-id table first (index 0), then one table per (cluster, column, quantizer).
 * flags: QVZ_MODEL_LIVE_ONLY leaves out the quantizers of dead contexts (qvz_codebook_t.live == 0; they map to
 * 0xFFFFFFFF), QVZ_MODEL_ALIGN16 starts every table on a 16-byte boundary (vector loads on the GPU). */
int qvz_model_build(uint32_t clusters, uint32_t columns, const qvz_codebook_t *const *cbs, int flags,
                    uint32_t **arena_out, uint32_t **offset_out, uint32_t *words_out) {
    size_t words;
    const int aligned = flags & QVZ_MODEL_ALIGN16;
    const int live_only = flags & QVZ_MODEL_LIVE_ONLY;
#define PAD4(w) (aligned ? (((w) + 3u) & ~3u) : (w))
    words = PAD4(2 + clusters);
    for (uint32_t c = 0; c < clusters; ++c)
        for (uint32_t s = 0; s < columns; ++s)
            for (uint32_t j = 0; j < 2u * cbs[c]->nctx[s]; ++j)
                if (!live_only || cbs[c]->live[(size_t)s * A + j / 2]) words += PAD4(2u + cbs[c]->nstates[(size_t)s * 2 * A + j]);
    if (words >= NO_TABLE) return -1;
    uint32_t *arena = (uint32_t *)calloc(words, sizeof(uint32_t));
    uint32_t *offset = (uint32_t *)malloc((size_t)clusters * columns * 2 * A * sizeof(uint32_t));
    if (!arena || !offset) { free(arena); free(offset); return -1; }
    uint32_t pos = 0;
    arena[pos++] = clusters;                                    /* cluster_stats: n = card = k, counts 1 */
    arena[pos++] = clusters;
    for (uint32_t c = 0; c < clusters; ++c) arena[pos++] = 1;
    pos = PAD4(pos);
    for (uint32_t c = 0; c < clusters; ++c)
        for (uint32_t s = 0; s < columns; ++s)
            for (uint32_t j = 0; j < 2 * A; ++j) {
                uint32_t *off = &offset[((size_t)c * columns + s) * 2 * A + j];
                if (j >= 2u * cbs[c]->nctx[s] || (live_only && !cbs[c]->live[(size_t)s * A + j / 2])) { *off = NO_TABLE; continue; }
                const uint32_t card = cbs[c]->nstates[(size_t)s * 2 * A + j];   /* qv_stream.c:46-53 */
                *off = pos;
                arena[pos++] = card;       /* n */
                arena[pos++] = card;       /* alphabetCard */
                for (uint32_t i = 0; i < card; ++i) arena[pos++] = 1;
                pos = PAD4(pos);
            }
#undef PAD4
    *arena_out = arena;
    *offset_out = offset;
    *words_out = pos;
    return 0;
}

qvz_coder_t *qvz_coder_create(uint32_t clusters, uint32_t columns, const qvz_codebook_t *const *cbs) {
    qvz_coder_t *cd = (qvz_coder_t *)calloc(1, sizeof(*cd));
    if (!cd) return NULL;
    cd->clusters = clusters;
    cd->columns = columns;
    uint32_t words;
    if (qvz_model_build(clusters, columns, cbs, 0, &cd->arena, &cd->offset, &words) || buf_reserve(cd, 1u << 20)) {
        qvz_coder_destroy(cd);
        return NULL;
    }
    cd->cluster_table = 0;
    cd->l = 0;                                                  /* arith.c:12-13 */
    cd->u = MASK_M;
    return cd;
}

void qvz_coder_destroy(qvz_coder_t *cd) {
    if (!cd) return;
    free(cd->arena); free(cd->offset); free(cd->buf);
    free(cd);
}

static inline uint32_t clz_field(uint32_t v, uint32_t width) {   /* leading zeros inside a `width`-bit field */
    return v ? (uint32_t)__builtin_clz(v) - (32 - width) : width;
}

/* arithmetic_encoder_step (arith.c:24-97) + update_stats (qv_stream.c:9-25) on table t, symbol x */
static inline void encode_symbol(qvz_coder_t *cd, uint32_t *t, uint32_t x) {
    const uint32_t n = t[0], card = t[1];
    uint32_t *counts = t + 2;
    uint32_t cum_lo = 0;
    for (uint32_t i = 0; i < x; ++i) cum_lo += counts[i];
    const uint32_t cum_hi = cum_lo + counts[x];
    const uint64_t range = (uint64_t)cd->u - cd->l + 1;
    const double dn = (double)n;
    uint32_t u = cd->l + (uint32_t)(uint64_t)((double)(range * cum_hi) / dn) - 1;   /* arith.c:47 */
    uint32_t l = cd->l + (uint32_t)(uint64_t)((double)(range * cum_lo) / dn);       /* arith.c:48 */

    const uint32_t diff = l ^ u;
    if (!(diff & MSB)) {                                        /* E1/E2 batch, arith.c:66-79 */
        const uint32_t k = clz_field(diff, M_BITS);
        const uint32_t top = l >> (M_BITS - k);                 /* the k leading bits shared by l and u */
        const uint32_t b = top >> (k - 1);
        put_bits(cd, b, 1);
        if (cd->scale3) { put_run(cd, !b, cd->scale3); cd->scale3 = 0; }
        if (k > 1) put_bits(cd, top & ((1u << (k - 1)) - 1), k - 1);
        l = (l << k) & MASK_M;
        u = ((u << k) & MASK_M) | ((1u << k) - 1);
    }
    {                                                           /* E3 batch, arith.c:80-84 */
        const uint32_t ones_l = clz_field((~l) & MASK_M1, M_BITS - 1);
        const uint32_t zeros_u = clz_field(u & MASK_M1, M_BITS - 1);
        const uint32_t j = ones_l < zeros_u ? ones_l : zeros_u;
        if (j) {
            cd->scale3 += j;
            l = (l << j) & MASK_M1;
            u = ((u << j) & MASK_M1) | MSB | ((1u << j) - 1);
        }
    }
    cd->l = l;
    cd->u = u;

    counts[x] += STEP;                                          /* qv_stream.c:12-13 */
    uint32_t nn = n + STEP;
    if (nn > RESCALE) {                                         /* qv_stream.c:15-24 */
        nn = 0;
        for (uint32_t i = 0; i < card; ++i)
            if (counts[i]) { counts[i] = (counts[i] >> 1) + 1; nn += counts[i]; }
    }
    t[0] = nn;
}

int qvz_coder_encode(qvz_coder_t *cd, const uint8_t *labels, const uint16_t *feed, uint64_t reads) {
    const uint32_t C = cd->columns;
    for (uint64_t i = 0; i < reads; ++i) {
        /* worst case per symbol is M_BITS bits plus a pending run that put_run grows for on its own */
        if (buf_reserve(cd, (uint64_t)(C + 1) * 4 + 64)) return -1;
        const uint32_t cl = labels ? labels[i] : 0;
        if (cl >= cd->clusters) return -1;
        encode_symbol(cd, cd->arena + cd->cluster_table, cl);   /* qv_write_cluster, qv_compressor.c:19-22 */
        const uint32_t *off = cd->offset + (size_t)cl * C * 2 * A;
        const uint16_t *f = feed + i * C;
        for (uint32_t s = 0; s < C; ++s) {
            const uint32_t qi = f[s] & 0xFF, st = f[s] >> 8;
            if (qi >= 2 * A) return -1;
            const uint32_t o = off[(size_t)s * 2 * A + qi];
            if (o == NO_TABLE || st >= cd->arena[o + 1]) return -1;
            encode_symbol(cd, cd->arena + o, st);               /* compress_qv, qv_compressor.c:8-11 */
        }
    }
    return 0;
}

/* The interval chain over records (include/qvz_coder.h, "strict single-stream mode with the model on the GPU"): per
 * symbol the model left two 64-bit integers, X(c) = floor(2^62 * c / n) + 1 for c = cum_lo and c = cum_hi. With l and ~u
 * kept left-aligned in 32 bits (l32 = l << 10, nu32 = (~u & mask) << 10), rd = 2^32 - (l32 + nu32) = (u - l + 1) * 1024 and
 *     floor((u - l + 1) * c / n) = (rd * X(c)) >> 72                                               (arith.c:47-48)
 * exactly: rd * X / 2^72 = R * X / 2^62 lies in (R c / n, R c / n + 2^-40] (R = u - l + 1 <= 2^22), and R c / n is an
 * integer or at least 1 / n >= 2^-19.1 below the next one. So a symbol costs this core two independent 64 x 64 -> 128 bit
 * multiplies, the renormalisation and the bit output - no table, no cumulative sum, no division, no floating point.
 * The new bounds: l += q_lo; u = l_old + q_hi - 1, i.e. ~u loses q_hi and gains R: nu += R - q_hi.
 * E1/E2/E3 as in encode_symbol above, on the aligned words: k = leading bits on which l and u agree, j = leading
 * positions below the top where l has 1 and u has 0. Same integers, same bits as arith.c:50-97. */
/* The chain below is one dependent sequence per symbol (about 24 instructions deep): it is compiled a second time for
 * BMI2 / LZCNT hosts (shlx, lzcnt instead of shl %cl and bsr + xor; chosen when the library is loaded): 7.9 -> 7.5 ns per
 * symbol on the GPU boxes' host cores. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("arch=haswell", "default")))
#endif
int qvz_coder_encode_records(qvz_coder_t *cd, const void *records, uint64_t reads, uint32_t pitch) {
    const uint64_t *rec = (const uint64_t *)records;
    const uint32_t C = cd->columns, pos0 = cd->clusters > 1 ? 8u : 0u;             /* several clusters: 0 = cluster id, 8 + column */
    if (!rec || (pitch & 7) || pitch < pos0 + C) return -1;
    uint32_t l32 = cd->l << 10, nu32 = (~cd->u & MASK_M) << 10;
    uint64_t scale3 = cd->scale3, acc = cd->acc;
    uint32_t nbits = cd->nbits;
    const uint32_t quads = pitch >> 2;
    for (uint64_t i = 0; i < reads; ++i) {
        if (cd->len + (uint64_t)(C + 1) * 4 + 64 > cd->cap) {
            cd->acc = acc; cd->nbits = nbits;
            if (buf_reserve(cd, (uint64_t)(C + 1) * 4 + 64)) return -1;
        }
        const uint64_t *row = rec + ((i >> 1) * quads * 8 + (i & 1) * 4) * 2;      /* unit (i / 2, 0), half i % 2 */
        for (uint32_t t = 0; t < C + (pos0 ? 1u : 0u); ++t) {
            const uint32_t p = !pos0 ? t : t == 0 ? 0u : pos0 + t - 1;              /* qv_write_cluster first (qv_compressor.c:86), then the columns */
            const uint64_t *r = row + ((size_t)(p >> 2) * 8 + (p & 3)) * 2;
            const uint64_t rd = 0x100000000ull - (uint64_t)(l32 + nu32);           /* (u - l + 1) * 1024; l32 + nu32 < 2^32 */
            const uint32_t q_lo = (uint32_t)(((unsigned __int128)rd * r[0]) >> 72);
            const uint32_t q_hi = (uint32_t)(((unsigned __int128)rd * r[1]) >> 72);
            l32 += q_lo << 10;
            nu32 += (uint32_t)rd - (q_hi << 10);                                   /* (R - q_hi) << 10; rd = 2^32 wraps to 0 with q_hi = R = 2^22 */
            const uint32_t k = (uint32_t)__builtin_clz(~(l32 ^ nu32));             /* the zero low bits stop it at 22 */
            /* E1/E2 batch, written without a branch on k (it is zero for about every other symbol, unpredictably): the k
             * agreed bits with the pending run behind the first of them are top + ((2^s3 - 1) << (k - 1)) whatever the first
             * bit is, k + s3 bits long; k = 0 makes that zero bits of value 0 and leaves the run pending */
            const uint64_t s3 = k ? scale3 : 0;
            if (__builtin_expect(s3 > 9, 0)) {                                     /* a long run: piecewise, any length */
                const uint32_t top = l32 >> (32 - k), b = top >> (k - 1);
                cd->acc = acc; cd->nbits = nbits;
                put_bits(cd, b, 1);
                put_run(cd, !b, s3);
                if (k > 1) put_bits(cd, top & ((1u << (k - 1)) - 1), k - 1);
                acc = cd->acc; nbits = cd->nbits;
            } else {                                                               /* at most 22 + 9 bits */
                const uint64_t top = (uint64_t)l32 >> (32 - k);
                acc = (acc << (k + s3)) | (top + ((((uint64_t)1 << s3) - 1) << ((k - 1) & 63)));
                nbits += k + (uint32_t)s3;
            }
            scale3 -= s3;
            l32 <<= k;
            nu32 <<= k;
            if (nbits >= 32) {                                                     /* four whole bytes, most significant first */
                nbits -= 32;
                const uint32_t w = __builtin_bswap32((uint32_t)(acc >> nbits));
                memcpy(cd->buf + cd->len, &w, 4);
                cd->len += 4;
            }
            const uint32_t j = (uint32_t)__builtin_clz(~(l32 & nu32) << 1);        /* E3 batch (j = 0: nothing) */
            scale3 += j;
            l32 = (l32 << j) & 0x7FFFFFFFu;
            nu32 = (nu32 << j) & 0x7FFFFFFFu;
        }
    }
    cd->l = l32 >> 10;
    cd->u = ~(nu32 >> 10) & MASK_M;
    cd->scale3 = scale3;
    cd->acc = acc;
    cd->nbits = nbits;
    while (cd->nbits >= 8) {                                                      /* back to the byte-wise invariant of put_bits */
        cd->nbits -= 8;
        cd->buf[cd->len++] = (uint8_t)(cd->acc >> cd->nbits);
    }
    return 0;
}

void qvz_coder_finish(qvz_coder_t *cd) {                        /* encoder_last_step, arith.c:99-116 */
    buf_reserve(cd, 64);
    const uint32_t b = cd->l >> (M_BITS - 1);
    put_bits(cd, b, 1);
    if (cd->scale3) { put_run(cd, !b, cd->scale3); cd->scale3 = 0; }
    buf_reserve(cd, 64);
    put_bits(cd, cd->l & MASK_M1, M_BITS - 1);                  /* stream_write_bits(os, l, m-1) */
    /* stream_finish_byte (os_stream.c:105-110): pad with zeros; an aligned stream still gets one 0x00 byte */
    if (cd->nbits) put_bits(cd, 0, 8 - cd->nbits);
    else cd->buf[cd->len++] = 0;
}

uint64_t qvz_coder_bytes(const qvz_coder_t *cd) { return cd->dropped + cd->len; }

const uint8_t *qvz_coder_data(const qvz_coder_t *cd, uint64_t *size) {
    *size = cd->len;
    return cd->buf;
}

void qvz_coder_drop(qvz_coder_t *cd, uint64_t size) {
    if (size > cd->len) size = cd->len;
    memmove(cd->buf, cd->buf + size, cd->len - size);
    cd->len -= size;
    cd->dropped += size;
}
