/* qvz_oracle.c - see qvz_oracle.h. TEST INFRASTRUCTURE: a plain, literal CPU restatement of the
 * reference encode path (and the decoder needed to close round trips on the GPU box).
 * Build: gcc -O2 -ffp-contract=off (the reference is built without FMA, src/Makefile:10; a fused
 * multiply-add changes decision-bearing sums, SURVEY.md section 7 hard part 2).
 *
 * Exact restructurings used (each re-verified against oracle/_ref):
 *  (i)  p_temp of compute_qpmf_list hoisted out of the idx loop (same per-term order);
 *  (ii) Lloyd-Max designs for an all-zero pmf memoised per (distortion matrix, states).
 */
#include "qvz_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define A ORC_A

/* ------------------------------------------------------------------------------------------ */
/* codebook container                                                                          */
/* ------------------------------------------------------------------------------------------ */
orc_codebook_t *orc_codebook_alloc(uint32_t C) {
    orc_codebook_t *cb = (orc_codebook_t *)calloc(1, sizeof(*cb));
    cb->C = C;
    cb->nctx = (uint8_t *)calloc(C, 1);
    cb->ctx_sym = (uint8_t *)calloc((size_t)C * A, 1);
    cb->ctx_idx = (uint8_t *)malloc((size_t)C * A);
    memset(cb->ctx_idx, ORC_NOT_FOUND, (size_t)C * A);
    cb->qratio = (uint8_t *)calloc((size_t)C * A, 1);
    cb->ratio = (double *)calloc((size_t)C * A, sizeof(double));
    cb->q = (uint8_t *)calloc((size_t)C * 2 * A * A, 1);
    cb->nstates = (uint8_t *)calloc((size_t)C * 2 * A, 1);
    cb->recon = (uint8_t *)calloc((size_t)C * 2 * A * A, 1);
    cb->out_idx = (uint8_t *)malloc((size_t)C * 2 * A * A);
    memset(cb->out_idx, ORC_NOT_FOUND, (size_t)C * 2 * A * A);
    cb->mse = (double *)calloc((size_t)C * 2 * A, sizeof(double));
    return cb;
}

void orc_codebook_free(orc_codebook_t *cb) {
    if (!cb) return;
    free(cb->nctx); free(cb->ctx_sym); free(cb->ctx_idx); free(cb->qratio); free(cb->ratio);
    free(cb->q); free(cb->nstates); free(cb->recon); free(cb->out_idx); free(cb->mse);
    free(cb);
}

/* ------------------------------------------------------------------------------------------ */
/* distortion matrices : distortion.c:50-95, indexing d[x + 72*y] (distortion.c:151-153)        */
/* ------------------------------------------------------------------------------------------ */
int orc_distortion(int type, double *d) {
    for (int x = 0; x < A; ++x) {
        for (int y = 0; y < A; ++y) {
            int diff = abs(x - y);
            switch (type) {
                case ORC_DIST_MANHATTAN: d[x + y * A] = diff; break;                 /* :50-61 */
                case ORC_DIST_MSE:       d[x + y * A] = (x - y) * (x - y); break;    /* :66-77 */
                case ORC_DIST_LORENTZ:   d[x + y * A] = log2(1.0 + (double)diff); break; /* :82-95 */
                default: return -1;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* k-means : cluster.c                                                                         */
/* ------------------------------------------------------------------------------------------ */
#define ORC_LINES_PER_BLOCK 1000000ull /* include/lines.h:12 */

void orc_pick_centroids(uint64_t n, uint32_t k, int reseed, uint64_t *lines_out) {
    /* lines.c:88-125: block_count = ceil(n/1e6); all blocks full except the last */
    uint32_t block_count = (uint32_t)(n / ORC_LINES_PER_BLOCK);
    if ((uint64_t)block_count * ORC_LINES_PER_BLOCK != n) block_count += 1;
    if (reseed) srand(1);
    for (uint32_t j = 0; j < k; ++j) {            /* cluster.c:198-201 */
        uint32_t block_id = (uint32_t)rand() % block_count;
        uint64_t left = n - (uint64_t)block_id * ORC_LINES_PER_BLOCK;
        uint32_t count = left > ORC_LINES_PER_BLOCK ? (uint32_t)ORC_LINES_PER_BLOCK : (uint32_t)left;
        uint32_t line_id = (uint32_t)rand() % count;
        lines_out[j] = (uint64_t)block_id * ORC_LINES_PER_BLOCK + line_id;
    }
}

/* One assignment pass, cluster_lines -> do_cluster_assignment (cluster.c:65-75, :136-171): every read goes to the nearest of the k
 * means (raw bytes), the lowest id winning ties (strict <, cluster.c:160); counts[j] = reads assigned to j. */
void orc_assign(const uint8_t *rows, uint64_t n, uint32_t c, uint32_t k, const uint8_t *means, uint8_t *labels, uint32_t *counts) {
    const size_t stride = (size_t)c + 1;
    double *dist = (double *)malloc(k * sizeof(double));
    for (uint32_t j = 0; j < k; ++j) counts[j] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *x = rows + i * stride;
        for (uint32_t j = 0; j < k; ++j) {     /* find_distance cluster.c:176-187 */
            double dd = 0.0;
            const uint8_t *m = means + (size_t)j * c;
            for (uint32_t s = 0; s < c; ++s) {
                uint32_t data = x[s], mean = m[s];
                dd += (data - mean) * (data - mean);   /* uint32 wraparound square, exact */
            }
            dist[j] = dd;
        }
        uint8_t id = 0;                        /* assign_cluster cluster.c:149-171 */
        double best = dist[0];
        for (uint32_t j = 1; j < k; ++j)
            if (dist[j] < best) { id = (uint8_t)j; best = dist[j]; }
        labels[i] = id;
        counts[id] += 1;
    }
    free(dist);
}

int orc_kmeans(const uint8_t *rows, uint64_t n, uint32_t c, uint32_t k, double threshold,
               const uint64_t *centroid_lines, uint8_t *labels, uint8_t *means, uint32_t *counts) {
    const size_t stride = (size_t)c + 1;
    uint64_t *acc = (uint64_t *)malloc((size_t)k * c * sizeof(uint64_t));
    for (uint32_t j = 0; j < k; ++j)               /* cluster.c:201 memcpy of raw bytes */
        memcpy(means + (size_t)j * c, rows + centroid_lines[j] * stride, c);

    int iter = 0, loop = 1;
    while (iter < 1000 && loop) {                  /* cluster.c:221, cluster.h:9 */
        orc_assign(rows, n, c, k, means, labels, counts);
        /* recalculate_means cluster.c:80-131 */
        memset(acc, 0, (size_t)k * c * sizeof(uint64_t));
        for (uint64_t i = 0; i < n; ++i) {
            const uint8_t *x = rows + i * stride;
            uint64_t *a = acc + (size_t)labels[i] * c;
            for (uint32_t s = 0; s < c; ++s) a[s] += x[s];
        }
        double move_max = 0.0;
        for (uint32_t j = 0; j < k; ++j) {
            if (counts[j] == 0) { free(acc); return -1; }
            double moved = 0.0;
            for (uint32_t s = 0; s < c; ++s) {
                uint8_t new_mean = (uint8_t)(acc[(size_t)j * c + s] / counts[j]);
                double dd = new_mean - means[(size_t)j * c + s];
                moved += dd * dd;
                means[(size_t)j * c + s] = new_mean;
            }
            if (moved > move_max) move_max = moved;
        }
        loop = move_max > threshold;               /* cluster.c:230-233 */
        iter += 1;
    }
    free(acc);
    return iter;
}

/* ------------------------------------------------------------------------------------------ */
/* statistics : codebook.c:185-220                                                             */
/* ------------------------------------------------------------------------------------------ */
void orc_stats(const uint8_t *rows, const uint8_t *labels, uint64_t n, uint32_t c, uint32_t k,
               uint32_t *cnt) {
    (void)k;
    const size_t stride = (size_t)c + 1;
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *x = rows + i * stride;
        uint32_t *t = cnt + (size_t)labels[i] * c * A * A;
        t[x[0] - 33] += 1;                                        /* codebook.c:200 */
        for (uint32_t s = 1; s < c; ++s)                          /* codebook.c:201-203 */
            t[(size_t)s * A * A + (size_t)(x[s - 1] - 33) * A + (x[s] - 33)] += 1;
    }
}

/* pmf.c:219-230 : P[x] = cnt[x]/total by true division; all zero when total == 0 */
static void counts_to_pmf(const uint32_t *cnt, double *p) {
    uint32_t total = 0;
    for (int x = 0; x < A; ++x) total += cnt[x];       /* pmf_increment keeps total, pmf.c:211-214 */
    if (total == 0) { for (int x = 0; x < A; ++x) p[x] = 0.0; return; }
    double t = (double)total;
    for (int x = 0; x < A; ++x) p[x] = ((double)cnt[x]) / t;
}

/* all conditional pmfs of one cluster: P[s][prev][x] (column 0 only row 0 meaningful) */
static double *cond_pmfs(const uint32_t *cnt, uint32_t C) {
    double *P = (double *)malloc((size_t)C * A * A * sizeof(double));
    for (size_t r = 0; r < (size_t)C * A; ++r) counts_to_pmf(cnt + r * A, P + r * A);
    return P;
}

static void marginals_from_P(const double *P, uint32_t C, double *marg) {
    /* codebook.c:213 : combine_pmfs(cond(0,0), m0, 1.0, 0.0, m0), pmf.c:200-202 */
    for (int x = 0; x < A; ++x) marg[x] = 1.0 * P[x] + 0.0 * 0.0;
    for (uint32_t s = 1; s < C; ++s) {                 /* codebook.c:214-218 */
        double *m = marg + (size_t)s * A;
        const double *mp = marg + (size_t)(s - 1) * A;
        for (int x = 0; x < A; ++x) m[x] = 0.0;
        for (int j = 0; j < A; ++j) {
            const double *pj = P + ((size_t)s * A + j) * A;
            double w = mp[j];
            for (int x = 0; x < A; ++x) m[x] = 1.0 * m[x] + w * pj[x];
        }
    }
}

void orc_marginals(const uint32_t *cnt, uint32_t C, double *marg) {
    double *P = cond_pmfs(cnt, C);
    marginals_from_P(P, C, marg);
    free(P);
}

/* ------------------------------------------------------------------------------------------ */
/* Lloyd-Max : quantizer.c:34-132                                                              */
/* ------------------------------------------------------------------------------------------ */
static double entropy72(const double *p) {             /* pmf.c:141-155 */
    double e = 0.0;
    for (int i = 0; i < A; ++i)
        if (p[i] > 0.0) e -= p[i] * log2(p[i]);
    return e;
}

double orc_lloyd_max(const double *p, const double *d, uint32_t S, uint8_t *q, uint8_t *recon,
                     uint8_t *out_idx, double *mse_out) {
    uint8_t bounds[A + 1];
    uint32_t changed = 1, iter = 0, i, j, r, min_r;
    double mse, min_mse, next_mse;

    bounds[0] = 0;                                     /* quantizer.c:45-52 */
    bounds[S] = A;
    for (j = 1; j < S; ++j) bounds[j] = (uint8_t)((j * A) / S);
    for (j = 0; j < S; ++j) recon[j] = (uint8_t)((bounds[j] + bounds[j + 1] - 1) / 2);

    while (changed && iter < 100) {                    /* quantizer.c:57, quantizer.h:10 */
        changed = 0;
        iter += 1;
        for (j = 0; j < S; ++j) {                      /* :62-87 */
            min_mse = DBL_MAX;
            min_r = bounds[j];
            for (r = bounds[j]; r < bounds[j + 1]; ++r) {
                mse = 0.0;
                for (i = bounds[j]; i < bounds[j + 1]; ++i) mse += p[i] * d[i + A * r];
                if (mse < min_mse) { min_r = r; min_mse = mse; }
            }
            if (min_r != recon[j]) { changed = 1; recon[j] = (uint8_t)min_r; }
        }
        r = 0;                                         /* :93-107, stale bounds are kept */
        for (j = 1; j < A - 1 && r < S - 1; ++j) {
            mse = d[j + A * recon[r]];
            next_mse = d[j + A * recon[r + 1]];
            if (next_mse < mse) { r += 1; bounds[r] = (uint8_t)j; }
        }
    }

    memset(q, 0, A);                                   /* calloc in alloc_quantizer :11-16 */
    for (j = 0; j < S; ++j)                            /* :112-116 */
        for (i = bounds[j]; i < bounds[j + 1]; ++i) q[i] = recon[j];

    memset(out_idx, ORC_NOT_FOUND, A);                 /* :119-121, pmf.c:365-382 */
    for (j = 0; j < S; ++j) out_idx[recon[j]] = (uint8_t)j;

    double m = 0.0;                                    /* :124-129 */
    for (j = 0; j < S; ++j)
        for (i = bounds[j]; i < bounds[j + 1]; ++i) m += d[i + A * recon[j]] * p[i];
    *mse_out = m;

    double o[A];                                       /* apply_quantizer :139-161 */
    memset(o, 0, sizeof(o));
    for (i = 0; i < A; ++i) o[q[i]] += p[i];
    return entropy72(o);
}

/* memo of designs for the all-zero pmf, keyed by the distortion matrix contents */
typedef struct { uint8_t q[A], recon[A], out_idx[A]; double mse, H; int valid; } lm_design_t;
static lm_design_t g_zero_memo[A + 1];
static double g_zero_memo_d[A * A];
static int g_zero_memo_set = 0;

static void lm_design(const double *p, int p_is_zero, const double *d, uint32_t S, lm_design_t *out) {
    if (p_is_zero) {
        if (!g_zero_memo_set || memcmp(g_zero_memo_d, d, sizeof(g_zero_memo_d)) != 0) {
            memcpy(g_zero_memo_d, d, sizeof(g_zero_memo_d));
            for (int s = 0; s <= A; ++s) g_zero_memo[s].valid = 0;
            g_zero_memo_set = 1;
        }
        lm_design_t *z = &g_zero_memo[S];
        if (!z->valid) {
            memset(z->recon, 0, A);
            z->H = orc_lloyd_max(p, d, S, z->q, z->recon, z->out_idx, &z->mse);
            z->valid = 1;
        }
        *out = *z;
        return;
    }
    memset(out->recon, 0, A);
    out->H = orc_lloyd_max(p, d, S, out->q, out->recon, out->out_idx, &out->mse);
    out->valid = 1;
}

/* optimize_for_entropy codebook.c:230-269. Returns ratio; *Slo,*Shi = states of lo and hi. */
static double optimize_for_entropy(const double *p, const double *d, double target,
                                   lm_design_t *lo, lm_design_t *hi, uint32_t *Slo, uint32_t *Shi) {
    int zero = 1;
    for (int i = 0; i < A; ++i) if (p[i] != 0.0) { zero = 0; break; }
    uint32_t states = 1;
    if (target == 0.0) {                               /* :236-242 */
        lm_design(p, zero, d, 1, lo);
        lm_design(p, zero, d, 1, hi);
        *Slo = *Shi = 1;
        return 1.0;
    }
    lm_design(p, zero, d, states, hi);                 /* :244-246 */
    double hi_entropy = hi->H, lo_entropy;
    do {                                               /* :249-258 */
        *lo = *hi;
        lo_entropy = hi_entropy;
        states += 1;
        lm_design(p, zero, d, states, hi);
        hi_entropy = hi->H;
    } while (hi_entropy < target && states < A);
    *Slo = states - 1;
    *Shi = states;
    if (hi_entropy < target) return 0.0;               /* :263-268 */
    else if (lo_entropy >= target || hi_entropy == lo_entropy) return 1.0;
    else return (target - hi_entropy) / (lo_entropy - hi_entropy);
}

static void store_pair(orc_codebook_t *cb, uint32_t s, uint32_t ctx, const lm_design_t *lo,
                       const lm_design_t *hi, uint32_t Slo, uint32_t Shi, double ratio) {
    /* store_cond_quantizers_indexed codebook.c:152-157 */
    size_t base = ((size_t)s * 2 * A + 2 * ctx) * A;
    memcpy(cb->q + base, lo->q, A);             memcpy(cb->q + base + A, hi->q, A);
    memcpy(cb->recon + base, lo->recon, A);     memcpy(cb->recon + base + A, hi->recon, A);
    memcpy(cb->out_idx + base, lo->out_idx, A); memcpy(cb->out_idx + base + A, hi->out_idx, A);
    cb->nstates[(size_t)s * 2 * A + 2 * ctx] = (uint8_t)Slo;
    cb->nstates[(size_t)s * 2 * A + 2 * ctx + 1] = (uint8_t)Shi;
    cb->mse[(size_t)s * 2 * A + 2 * ctx] = lo->mse;
    cb->mse[(size_t)s * 2 * A + 2 * ctx + 1] = hi->mse;
    cb->ratio[(size_t)s * A + ctx] = ratio;
    cb->qratio[(size_t)s * A + ctx] = (uint8_t)(ratio * 128.);
}

/* alphabet_union pmf.c:312-357: a MERGE of two symbol lists (not a set union if unsorted) */
static uint32_t merge_alphabets(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb,
                                uint8_t *out) {
    uint8_t tmp[2 * A * 2];
    uint32_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) tmp[k] = a[i++];
        else if (a[i] == b[j]) { tmp[k] = a[i]; i++; j++; }
        else tmp[k] = b[j++];
        k++;
    }
    while (i < na) tmp[k++] = a[i++];
    while (j < nb) tmp[k++] = b[j++];
    memcpy(out, tmp, k);
    return k;
}

void orc_generate_codebook(const uint32_t *cnt, uint32_t C, const double *d, int mode, double opt_ratio,
                           orc_codebook_t *cb) {
    double *P = cond_pmfs(cnt, C);                     /* P[s][prev][x] */
    double *marg = (double *)malloc((size_t)C * A * sizeof(double));
    marginals_from_P(P, C, marg);

    double *qpmf = (double *)calloc((size_t)A * A, sizeof(double));      /* [k=x][idx] */
    double *prev_qpmf = (double *)calloc((size_t)A * A, sizeof(double));
    double *xpmf = (double *)calloc((size_t)A * A, sizeof(double));      /* [idx][k] */
    double *ptemp = (double *)calloc((size_t)A * A, sizeof(double));     /* [k][j] */
    lm_design_t lo, hi;
    uint32_t Slo, Shi;

    /* column 0 : codebook.c:392-411 */
    cb->nctx[0] = 1;
    cb->ctx_sym[0] = 0;
    cb->ctx_idx[0] = 0;
    const double *p0 = P;                              /* get_cond_pmf(list,0,0) */
    double target = (mode == ORC_MODE_RATIO) ? entropy72(p0) * opt_ratio : opt_ratio;
    double ratio = optimize_for_entropy(p0, d, target, &lo, &hi, &Slo, &Shi);
    store_pair(cb, 0, 0, &lo, &hi, Slo, Shi, ratio);

    for (uint32_t s = 1; s < C; ++s) {                 /* codebook.c:419-462 */
        /* union of the previous column's output alphabets, :421-424 */
        uint32_t nprev = cb->nctx[s - 1];
        uint8_t uni[2 * A];
        uint32_t nu;
        {
            size_t b0 = ((size_t)(s - 1) * 2 * A) * A;
            nu = cb->nstates[(size_t)(s - 1) * 2 * A];
            memcpy(uni, cb->recon + b0, nu);
            for (uint32_t j = 1; j < 2 * nprev; ++j)
                nu = merge_alphabets(uni, nu, cb->recon + b0 + (size_t)j * A,
                                     cb->nstates[(size_t)(s - 1) * 2 * A + j], uni);
        }
        cb->nctx[s] = (uint8_t)nu;
        memcpy(cb->ctx_sym + (size_t)s * A, uni, nu);
        for (uint32_t i = 0; i < nu; ++i) cb->ctx_idx[(size_t)s * A + uni[i]] = (uint8_t)i;

        memset(qpmf, 0, (size_t)A * A * sizeof(double));
        if (s == 1) {                                  /* compute_qpmf_quan_list :274-289 */
            const uint8_t *qlo = cb->q, *qhi = cb->q + A;   /* column 0, ctx 0 */
            double r0 = cb->ratio[0];
            for (int x = 0; x < A; ++x)
                for (uint32_t idx = 0; idx < nu; ++idx) {
                    uint32_t sym = uni[idx];
                    if (qlo[x] == sym) qpmf[x * A + idx] += r0;
                    if (qhi[x] == sym) qpmf[x * A + idx] += (1 - r0);
                }
        } else {                                       /* compute_qpmf_list :291-330 */
            const double *Pc = P + (size_t)(s - 1) * A * A;      /* P(X_{s-1}=k | X_{s-2}=x) */
            const double *mm = marg + (size_t)(s - 2) * A;
            for (int k = 0; k < A; ++k)
                for (uint32_t j = 0; j < nprev; ++j) {           /* hoisted p_temp, :318-321 */
                    double pt = 0;
                    for (int x = 0; x < A; ++x)
                        pt += prev_qpmf[x * A + j] * Pc[(size_t)x * A + k] * mm[x];
                    ptemp[k * A + j] = pt;
                }
            for (int k = 0; k < A; ++k) {
                for (uint32_t idx = 0; idx < nu; ++idx) {
                    uint32_t sym = uni[idx];
                    double acc = 0.0;
                    for (uint32_t j = 0; j < nprev; ++j) {
                        double p_q_xq = 0.0;
                        size_t qb = ((size_t)(s - 1) * 2 * A + 2 * j) * A;
                        double rj = cb->ratio[(size_t)(s - 1) * A + j];
                        if (cb->q[qb + k] == sym) p_q_xq += rj;            /* q_lo->ratio */
                        if (cb->q[qb + A + k] == sym) p_q_xq += (1 - rj);  /* q_hi->ratio */
                        acc += p_q_xq * ptemp[k * A + j];
                    }
                    qpmf[k * A + idx] = acc;
                }
                double total = 0;                      /* renormalize_pmf pmf.c:235-254 */
                for (uint32_t idx = 0; idx < nu; ++idx) total += qpmf[k * A + idx];
                if (total > 0)
                    for (uint32_t idx = 0; idx < nu; ++idx) qpmf[k * A + idx] = qpmf[k * A + idx] / total;
            }
        }

        /* compute_xpmf_list :332-349 */
        {
            const double *Pc = P + (size_t)s * A * A;            /* P(X_s=k | X_{s-1}=x) */
            const double *mm = marg + (size_t)(s - 1) * A;
            for (uint32_t idx = 0; idx < nu; ++idx) {
                for (int k = 0; k < A; ++k) {
                    double acc = 0.0;
                    for (int x = 0; x < A; ++x)
                        acc += qpmf[x * A + idx] * Pc[(size_t)x * A + k] * mm[x];
                    xpmf[idx * A + k] = acc;
                }
                double total = 0;
                for (int k = 0; k < A; ++k) total += xpmf[idx * A + k];
                if (total > 0)
                    for (int k = 0; k < A; ++k) xpmf[idx * A + k] = xpmf[idx * A + k] / total;
            }
        }

        for (uint32_t j = 0; j < nu; ++j) {            /* :441-454 */
            const double *pj = xpmf + (size_t)j * A;
            target = (mode == ORC_MODE_RATIO) ? entropy72(pj) * opt_ratio : opt_ratio;
            ratio = optimize_for_entropy(pj, d, target, &lo, &hi, &Slo, &Shi);
            store_pair(cb, s, j, &lo, &hi, Slo, Shi, ratio);
        }
        double *t = prev_qpmf; prev_qpmf = qpmf; qpmf = t;
    }
    free(P); free(marg); free(qpmf); free(prev_qpmf); free(xpmf); free(ptemp);
}

/* ------------------------------------------------------------------------------------------ */
/* codebook serialisation : codebook.c:474-555                                                 */
/* ------------------------------------------------------------------------------------------ */
size_t orc_write_header(uint8_t *buf, uint32_t k, uint32_t C, uint64_t N) {
    if (buf) {
        uint32_t lines = (uint32_t)N;                  /* truncation, codebook.c:482 */
        buf[0] = (uint8_t)k;
        buf[1] = (uint8_t)(C >> 24); buf[2] = (uint8_t)(C >> 16); buf[3] = (uint8_t)(C >> 8); buf[4] = (uint8_t)C;
        buf[5] = (uint8_t)(lines >> 24); buf[6] = (uint8_t)(lines >> 16); buf[7] = (uint8_t)(lines >> 8); buf[8] = (uint8_t)lines;
    }
    return 9;
}

size_t orc_write_codebook(uint8_t *buf, const orc_codebook_t *cb) {
    size_t pos = 0;
#define PUT(b) do { if (buf) buf[pos] = (uint8_t)(b); pos++; } while (0)
    PUT(cb->qratio[0] + 33); PUT('\n');                                   /* :515-517 */
    for (int i = 0; i < A; ++i) PUT(cb->q[i] + 33);                       /* :520-522 */
    PUT('\n');
    for (int i = 0; i < A; ++i) PUT(cb->q[A + i] + 33);                   /* :525-528 */
    PUT('\n');
    for (uint32_t s = 1; s < cb->C; ++s) {                                /* :531-554 */
        uint32_t n = cb->nctx[s];
        for (uint32_t j = 0; j < n; ++j) PUT(cb->qratio[(size_t)s * A + j] + 33);
        PUT('\n');
        for (int b = 0; b < 2; ++b) {
            for (uint32_t j = 0; j < n; ++j) {
                const uint8_t *q = cb->q + ((size_t)s * 2 * A + 2 * j + b) * A;
                for (int i = 0; i < A; ++i) PUT(q[i] + 33);
            }
            PUT('\n');
        }
    }
#undef PUT
    return pos;
}

/* ------------------------------------------------------------------------------------------ */
/* WELL-1024a : well.c:8-46                                                                    */
/* ------------------------------------------------------------------------------------------ */
uint32_t orc_well_word(orc_well_t *w) {
    uint32_t *s = w->state, n = w->n;
    uint32_t z0 = s[(n + 31) & 31];
    uint32_t v1 = s[(n + 3) & 31], v2 = s[(n + 24) & 31], v3 = s[(n + 10) & 31];
    uint32_t z1 = s[n] ^ (v1 ^ (v1 >> 8));
    uint32_t z2 = (v2 ^ (v2 << 19)) ^ (v3 ^ (v3 << 14));
    s[n] = z1 ^ z2;
    n = (n + 31) & 31;
    s[n] = (z0 ^ (z0 << 11)) ^ (z1 ^ (z1 << 7)) ^ (z2 ^ (z2 << 13));
    w->n = n;
    return s[n];
}

/* The generator's state after words[i] calls from `state0` (32 words, n = 0), for ascending words[]: found by stepping
 * orc_well_word, nothing else. States are handed back in the form a file stores them (qv_stream.c:89-93: the 32 words, n = 0):
 * out[i][j] = state[(n + j) & 31]. This is how the tests get the WELL state of a shard in the middle of a file WITHOUT the
 * product's jump-ahead. ~1.5 ns per word. */
void orc_well_states_at(const uint32_t *state0, const uint64_t *words, uint64_t count, uint32_t *out) {
    orc_well_t w;
    uint64_t done = 0;
    memcpy(w.state, state0, sizeof(w.state));
    w.n = 0; w.bit_output = 0; w.bits_left = 0;
    for (uint64_t i = 0; i < count; ++i) {
        for (; done < words[i]; ++done) (void)orc_well_word(&w);
        for (uint32_t j = 0; j < 32; ++j) out[i * 32 + j] = w.state[(w.n + j) & 31];
    }
}

uint32_t orc_well_bits(orc_well_t *w, uint32_t bits) {
    uint32_t mask = (1u << bits) - 1, rtn;
    if (w->bits_left < bits) { w->bit_output = orc_well_word(w); w->bits_left = 32; }
    rtn = w->bit_output & mask;
    w->bit_output >>= bits;
    w->bits_left -= bits;
    return rtn;
}

/* ------------------------------------------------------------------------------------------ */
/* adaptive stats + arithmetic coder + bit stream (literal)                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t *counts; uint32_t card, step, n; } stats_t;    /* qv_compressor.h:45-50 */

static void stats_init(stats_t *st, uint32_t card) {                      /* qv_stream.c:46-57 */
    st->counts = (uint32_t *)malloc(card * sizeof(uint32_t));
    for (uint32_t i = 0; i < card; ++i) st->counts[i] = 1;
    st->n = card; st->card = card; st->step = 8;
}

static void stats_update(stats_t *st, uint32_t x, uint32_t r) {           /* qv_stream.c:9-25 */
    st->counts[x] += st->step;
    st->n += st->step;
    if (st->n > r) {
        st->n = 0;
        for (uint32_t i = 0; i < st->card; ++i)
            if (st->counts[i]) { st->counts[i] >>= 1; st->counts[i] += 1; st->n += st->counts[i]; }
    }
}

typedef struct { uint8_t *buf; uint64_t cap, pos; uint8_t cur; uint32_t bitpos; int overflow; } bitw_t;

static void bw_bit(bitw_t *w, uint32_t bit) {          /* os_stream.c:72-89 (MSB first) */
    w->cur = (uint8_t)((w->cur << 1) | (bit & 1));
    if (++w->bitpos == 8) {
        if (w->pos < w->cap) w->buf[w->pos] = w->cur; else w->overflow = 1;
        w->pos++; w->cur = 0; w->bitpos = 0;
    }
}

static void bw_finish(bitw_t *w) {                     /* os_stream.c:105-110: always emits a byte */
    uint8_t b = (uint8_t)(w->cur << (8 - w->bitpos));
    if (w->bitpos == 0) b = 0;
    if (w->pos < w->cap) w->buf[w->pos] = b; else w->overflow = 1;
    w->pos++; w->cur = 0; w->bitpos = 0;
}

typedef struct { int32_t scale3; uint32_t l, u, t, m, r; } arith_t;      /* qv_compressor.h:26-35 */

static void arith_init(arith_t *a) {                   /* arith.c:5-16, m = 22 */
    memset(a, 0, sizeof(*a));
    a->m = 22; a->r = 1u << (a->m - 3); a->l = 0; a->u = (1u << a->m) - 1;
}

static void arith_encode(arith_t *a, const stats_t *st, uint32_t x, bitw_t *os) {  /* arith.c:24-97 */
    const uint32_t msb_shift = a->m - 1, smsb_shift = a->m - 2, mask = (1u << msb_shift) - 1;
    uint64_t range = (uint64_t)a->u - a->l + 1;
    uint32_t cum1 = 0;
    for (uint32_t i = 0; i < x; ++i) cum1 += st->counts[i];
    uint32_t cum = cum1 + st->counts[x];
    a->u = a->l + (uint32_t)((range * cum) / st->n) - 1;
    a->l = a->l + (uint32_t)((range * cum1) / st->n);
    for (;;) {
        uint32_t msbL = a->l >> msb_shift, msbU = a->u >> msb_shift;
        if (msbL == msbU) {                            /* E1/E2 */
            bw_bit(os, msbL);
            a->l = (a->l & mask) << 1;
            a->u = ((a->u & mask) << 1) + 1;
            while (a->scale3 > 0) { bw_bit(os, !msbL); a->scale3 -= 1; }
        } else if ((a->l >> smsb_shift) == 1 && (a->u >> smsb_shift) == 2) {   /* E3 */
            a->scale3 += 1;
            a->u = (((a->u << 1) & mask) | (1u << msb_shift)) + 1;
            a->l = (a->l << 1) & mask;
        } else break;
    }
}

static void arith_finish(arith_t *a, bitw_t *os) {     /* arith.c:99-116 */
    uint32_t msbL = a->l >> (a->m - 1);
    bw_bit(os, msbL);
    while (a->scale3 > 0) { bw_bit(os, !msbL); a->scale3 -= 1; }
    for (int bit = (int)a->m - 2; bit >= 0; --bit) bw_bit(os, (a->l >> bit) & 1);   /* os_stream.c:94-100 */
    bw_finish(os);
}

typedef struct { stats_t cluster; stats_t **tab; uint32_t k, C; } model_t;

static void model_init(model_t *m, uint32_t k, uint32_t C, orc_codebook_t *const *cbs) {
    m->k = k; m->C = C;                                /* qv_stream.c:96-108 */
    stats_init(&m->cluster, k);
    m->tab = (stats_t **)malloc(k * sizeof(stats_t *));
    for (uint32_t c = 0; c < k; ++c) {
        m->tab[c] = (stats_t *)calloc((size_t)C * 2 * A, sizeof(stats_t));
        for (uint32_t s = 0; s < C; ++s)
            for (uint32_t j = 0; j < 2u * cbs[c]->nctx[s]; ++j)
                stats_init(&m->tab[c][(size_t)s * 2 * A + j], cbs[c]->nstates[(size_t)s * 2 * A + j]);
    }
}

static void model_free(model_t *m) {
    free(m->cluster.counts);
    for (uint32_t c = 0; c < m->k; ++c) {
        for (size_t t = 0; t < (size_t)m->C * 2 * A; ++t) free(m->tab[c][t].counts);
        free(m->tab[c]);
    }
    free(m->tab);
}

uint64_t orc_encode_stream(const uint8_t *rows, const uint8_t *labels, uint64_t n, uint32_t c,
                           uint32_t k, orc_codebook_t *const *cbs, const double *d,
                           const uint32_t *well_state, uint8_t *u_out, uint8_t *stream_out,
                           uint64_t stream_cap, double *distortion_out, uint8_t *qi_out,
                           uint8_t *st_out) {
    const size_t stride = (size_t)c + 1;
    orc_well_t well;
    memset(&well, 0, sizeof(well));                    /* qv_stream.c:70,93 */
    memcpy(well.state, well_state, 32 * sizeof(uint32_t));
    model_t model;
    model_init(&model, k, c, cbs);
    arith_t a;
    arith_init(&a);
    bitw_t os = { stream_out, stream_cap, 0, 0, 0, 0 };
    double distortion = 0.0;

    for (uint64_t i = 0; i < n; ++i) {                 /* qv_compressor.c:76-135 */
        const uint8_t *x = rows + i * stride;
        uint32_t cl = labels[i];
        const orc_codebook_t *cb = cbs[cl];
        arith_encode(&a, &model.cluster, cl, &os);     /* qv_write_cluster :19-22 */
        stats_update(&model.cluster, cl, a.r);
        double error = 0.0;
        uint8_t prev = 0;
        for (uint32_t s = 0; s < c; ++s) {
            uint32_t ctx = cb->ctx_idx[(size_t)s * A + prev];            /* choose_quantizer codebook.c:162-171 */
            uint32_t draw = orc_well_bits(&well, 7);
            uint32_t qi = 2 * ctx + (draw >= cb->qratio[(size_t)s * A + ctx] ? 1 : 0);
            uint8_t data = (uint8_t)(x[s] - 33);
            uint8_t qv = cb->q[((size_t)s * 2 * A + qi) * A + data];
            uint32_t state = cb->out_idx[((size_t)s * 2 * A + qi) * A + qv];
            stats_t *st = &model.tab[cl][(size_t)s * 2 * A + qi];
            arith_encode(&a, st, state, &os);          /* compress_qv :8-11 */
            stats_update(st, state, a.r);
            if (s == 0) error = d[data + A * qv]; else error += d[data + A * qv];
            if (u_out) u_out[i * stride + s] = (uint8_t)(qv + 33);
            if (qi_out) qi_out[i * c + s] = (uint8_t)qi;
            if (st_out) st_out[i * c + s] = (uint8_t)state;
            prev = qv;
        }
        if (u_out) u_out[i * stride + c] = '\n';
        distortion += error / ((double)c);
    }
    arith_finish(&a, &os);
    if (distortion_out) *distortion_out = distortion / ((double)n);
    model_free(&model);
    return os.overflow ? (uint64_t)-1 : os.pos;
}

/* ------------------------------------------------------------------------------------------ */
/* whole encode() : main.c:18-127                                                              */
/* ------------------------------------------------------------------------------------------ */
uint64_t orc_encode(const uint8_t *rows, uint64_t n, uint32_t c, uint32_t k, int mode, double ratio,
                    int dist_type, double threshold, const uint32_t *well_state, uint8_t *q_out,
                    uint64_t q_cap, uint8_t *u_out, double *stats_out) {
    double d[A * A];
    if (orc_distortion(dist_type, d)) return (uint64_t)-1;
    uint8_t *labels = (uint8_t *)calloc(n ? n : 1, 1);
    uint8_t *means = (uint8_t *)malloc((size_t)k * c);
    uint32_t *counts = (uint32_t *)malloc(k * sizeof(uint32_t));
    uint64_t *cent = (uint64_t *)malloc(k * sizeof(uint64_t));
    orc_pick_centroids(n, k, 1, cent);
    int iters = orc_kmeans(rows, n, c, k, threshold, cent, labels, means, counts);

    uint32_t *cnt = (uint32_t *)calloc((size_t)k * c * A * A, sizeof(uint32_t));
    orc_stats(rows, labels, n, c, k, cnt);
    orc_codebook_t **cbs = (orc_codebook_t **)malloc(k * sizeof(*cbs));
    for (uint32_t j = 0; j < k; ++j) {
        cbs[j] = orc_codebook_alloc(c);
        orc_generate_codebook(cnt + (size_t)j * c * A * A, c, d, mode, ratio, cbs[j]);
    }

    uint64_t pos = 0, result = (uint64_t)-1;
    size_t need = orc_write_header(NULL, k, c, n);
    for (uint32_t j = 0; j < k; ++j) need += orc_write_codebook(NULL, cbs[j]);
    need += 128;
    if (need <= q_cap) {
        pos += orc_write_header(q_out, k, c, n);
        for (uint32_t j = 0; j < k; ++j) pos += orc_write_codebook(q_out + pos, cbs[j]);
        memcpy(q_out + pos, well_state, 128);          /* qv_stream.c:89 native little-endian */
        pos += 128;
        double dist = 0;
        uint64_t sb = orc_encode_stream(rows, labels, n, c, k, cbs, d, well_state, u_out, q_out + pos,
                                        q_cap - pos, &dist, NULL, NULL);
        if (sb != (uint64_t)-1) {
            result = pos + sb;
            if (stats_out) {
                stats_out[0] = (sb * 8.) / ((double)n * c);    /* main.c:125 */
                stats_out[1] = dist;
                stats_out[2] = (double)sb;
                stats_out[3] = (double)iters;
            }
        }
    }
    for (uint32_t j = 0; j < k; ++j) orc_codebook_free(cbs[j]);
    free(cbs); free(cnt); free(labels); free(means); free(counts); free(cent);
    return result;
}

/* ------------------------------------------------------------------------------------------ */
/* decoder : codebook.c:560-669, qv_compressor.c:145-231, arith.c:118-205                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const uint8_t *buf; uint64_t size, pos; uint32_t bitpos; } bitr_t;

static uint32_t br_bit(bitr_t *r) {                    /* os_stream.c:36-52; past the end reads 0 */
    uint32_t b = 0;
    if (r->pos < r->size) b = (r->buf[r->pos] >> (7 - r->bitpos)) & 1;
    if (++r->bitpos == 8) { r->bitpos = 0; r->pos++; }
    return b;
}

static uint32_t arith_decode(arith_t *a, const stats_t *st, bitr_t *is, int last) {
    const uint32_t msb_shift = a->m - 1, smsb_shift = a->m - 2, mask = (1u << msb_shift) - 1;
    uint64_t range = (uint64_t)a->u - a->l + 1;
    uint64_t tagGap = (uint64_t)a->t - a->l + 1;
    uint64_t sub = (tagGap * st->n - 1) / range;       /* arith.c:134 / :199 */
    uint32_t subRange = last ? (uint32_t)0 : (uint32_t)sub, k = 0, cumCount = 0;
    if (last) {                                        /* decoder_last_step compares in 64 bits */
        while (sub >= cumCount) cumCount += st->counts[k++];
        return k - 1;
    }
    while (subRange >= cumCount) cumCount += st->counts[k++];
    uint32_t x = k - 1, cum1 = 0;
    for (uint32_t i = 0; i < x; ++i) cum1 += st->counts[i];
    uint32_t cum = cum1 + st->counts[x];
    a->u = a->l + (uint32_t)((range * cum) / st->n) - 1;
    a->l = a->l + (uint32_t)((range * cum1) / st->n);
    for (;;) {
        uint32_t msbL = a->l >> msb_shift, msbU = a->u >> msb_shift;
        if (msbL == msbU) {
            a->l = (a->l & mask) << 1;
            a->u = ((a->u & mask) << 1) + 1;
            a->t = ((a->t & mask) << 1) + br_bit(is);
        } else if ((a->l >> smsb_shift) == 1 && (a->u >> smsb_shift) == 2) {
            a->l = (a->l << 1) & mask;
            a->u = (((a->u << 1) & mask) | (1u << msb_shift)) + 1;
            a->t = (((a->t & mask) << 1) ^ (1u << msb_shift)) + br_bit(is);
        } else break;
    }
    return x;
}

/* find_output_alphabet quantizer.c:167-191 : run de-duplication */
static uint32_t dedup_alphabet(const uint8_t *q, uint8_t *out) {
    uint32_t size = 1;
    uint8_t p = q[0];
    out[0] = p;
    for (int x = 1; x < A; ++x)
        if (q[x] != p) { p = q[x]; out[size++] = p; }
    return size;
}

static int read_codebook(const uint8_t *f, uint64_t size, uint64_t *ppos, orc_codebook_t *cb) {
    uint64_t pos = *ppos;
    uint32_t C = cb->C;
    uint8_t uni[2 * A]; uint32_t nu;
#define NEED(nb) do { if (pos + (nb) > size) return -1; } while (0)
    NEED(2 + 2 * (A + 1));
    cb->nctx[0] = 1; cb->ctx_sym[0] = 0; cb->ctx_idx[0] = 0;
    cb->qratio[0] = (uint8_t)(f[pos] - 33); pos += 2;                     /* :602-603 */
    for (int b = 0; b < 2; ++b) {
        for (int i = 0; i < A; ++i) cb->q[b * A + i] = (uint8_t)(f[pos + i] - 33);
        pos += A + 1;
    }
    for (int b = 0; b < 2; ++b) {
        uint32_t ns = dedup_alphabet(cb->q + b * A, cb->recon + b * A);
        cb->nstates[b] = (uint8_t)ns;
        for (uint32_t j = 0; j < ns; ++j) cb->out_idx[b * A + cb->recon[b * A + j]] = (uint8_t)j;
    }
    nu = merge_alphabets(cb->recon, cb->nstates[0], cb->recon + A, cb->nstates[1], uni);   /* :617-618 */
    for (uint32_t s = 1; s < C; ++s) {                                    /* :623-663 */
        uint32_t n = nu;
        cb->nctx[s] = (uint8_t)n;
        memcpy(cb->ctx_sym + (size_t)s * A, uni, n);
        for (uint32_t i = 0; i < n; ++i) cb->ctx_idx[(size_t)s * A + uni[i]] = (uint8_t)i;
        NEED(n + 1 + 2 * ((uint64_t)n * A + 1));
        for (uint32_t j = 0; j < n; ++j) cb->qratio[(size_t)s * A + j] = (uint8_t)(f[pos + j] - 33);
        pos += n + 1;
        nu = 0;
        for (int b = 0; b < 2; ++b) {
            for (uint32_t j = 0; j < n; ++j) {
                size_t t = ((size_t)s * 2 * A + 2 * j + b);
                for (int i = 0; i < A; ++i) cb->q[t * A + i] = (uint8_t)(f[pos + i] - 33);
                pos += A;
                uint32_t ns = dedup_alphabet(cb->q + t * A, cb->recon + t * A);
                cb->nstates[t] = (uint8_t)ns;
                for (uint32_t z = 0; z < ns; ++z) cb->out_idx[t * A + cb->recon[t * A + z]] = (uint8_t)z;
                nu = merge_alphabets(uni, nu, cb->recon + t * A, ns, uni);
            }
            pos += 1;
        }
    }
#undef NEED
    *ppos = pos;
    return 0;
}

uint64_t orc_decode(const uint8_t *f, uint64_t size, uint8_t *out, uint64_t out_cap) {
    if (size < 9) return (uint64_t)-1;
    uint32_t k = f[0];                                                    /* codebook.c:565-573 */
    uint32_t C = ((uint32_t)f[1] << 24) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 8) | f[4];
    uint32_t lines = ((uint32_t)f[5] << 24) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 8) | f[8];
    uint64_t pos = 9;
    if ((uint64_t)lines * (C + 1) > out_cap || C < 2 || lines < 1 || k < 1) return (uint64_t)-1;
    orc_codebook_t **cbs = (orc_codebook_t **)calloc(k, sizeof(*cbs));
    uint64_t result = (uint64_t)-1;
    for (uint32_t j = 0; j < k; ++j) {
        cbs[j] = orc_codebook_alloc(C);
        if (read_codebook(f, size, &pos, cbs[j])) goto done;
    }
    if (pos + 128 > size) goto done;
    {
        orc_well_t well;
        memset(&well, 0, sizeof(well));
        memcpy(well.state, f + pos, 128);                                 /* qv_stream.c:72-74 */
        pos += 128;
        model_t model;
        model_init(&model, k, C, cbs);
        arith_t a;
        arith_init(&a);
        bitr_t is = { f + pos, size - pos, 0, 0 };
        for (int bit = (int)a.m - 1; bit >= 0; --bit) a.t |= br_bit(&is) << bit;   /* qv_stream.c:112-113 */
        for (uint32_t i = 0; i < lines; ++i) {
            uint8_t *o = out + (uint64_t)i * (C + 1);
            uint32_t cl = arith_decode(&a, &model.cluster, &is, 0);       /* qv_read_cluster :36-43 */
            stats_update(&model.cluster, cl, a.r);
            if (cl >= k) { model_free(&model); goto done; }
            const orc_codebook_t *cb = cbs[cl];
            uint8_t prev = 0;
            for (uint32_t s = 0; s < C; ++s) {
                uint32_t ctx = cb->ctx_idx[(size_t)s * A + prev];
                if (ctx == ORC_NOT_FOUND) { model_free(&model); goto done; }
                uint32_t draw = orc_well_bits(&well, 7);
                uint32_t qi = 2 * ctx + (draw >= cb->qratio[(size_t)s * A + ctx] ? 1 : 0);
                stats_t *st = &model.tab[cl][(size_t)s * 2 * A + qi];
                int last = (i == lines - 1 && s == C - 1);                /* :222-225 */
                uint32_t state = arith_decode(&a, st, &is, last);
                if (!last) stats_update(st, state, a.r);
                prev = cb->recon[((size_t)s * 2 * A + qi) * A + state];
                o[s] = (uint8_t)(prev + 33);
            }
            o[C] = '\n';
        }
        model_free(&model);
        result = (uint64_t)lines * (C + 1);
    }
done:
    for (uint32_t j = 0; j < k; ++j) orc_codebook_free(cbs[j]);
    free(cbs);
    return result;
}

/* ------------------------------------------------------------------------------------------ */
/* QGEN-1 synthetic input (this repo's own generator) in reference text     */
/* layout: reads [first, first+n), stride c+1, bytes Q+33, '\n' terminated.                      */
/* ------------------------------------------------------------------------------------------ */
#include "qgen1.h"
void orc_qgen1_rows(uint64_t seed, uint64_t first, uint64_t n, uint32_t c, uint8_t *rows) {
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t *r = rows + i * ((size_t)c + 1);
        qgen1_read(seed, first + i, c, r);
        for (uint32_t j = 0; j < c; ++j) r[j] = (uint8_t)(r[j] + 33);
        r[c] = '\n';
    }
}
