/* qvz_codebook_host.c - the host-side half of include/qvz_codebook.h: the flat codebook container, the distortion
 * matrices and the codebook text of the reference's file format. Plain C, no device code and no HIP header: these are
 * the functions every reader and writer of a .q file needs whether or not a GPU is present (the parser fuzz harness of
 * tests/ links this file and the other host sources against stubs for the qvz_hip_* entry points).
 *
 * Reference: cond_quantizer_list_t (include/codebook.h:61-69), generate_distortion_matrix (src/distortion.c:29-95),
 * gen_custom_distortion (src/distortion.c:100-145), write_codebooks / write_codebook (src/codebook.c:474-555). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_codebook.h"

#define A QVZ_A

static int fail(int code, const char *what, const char *arg) {
    char msg[512];
    snprintf(msg, sizeof(msg), what, arg);
    return qvz_hip_set_error(code, msg);
}

qvz_codebook_t *qvz_codebook_alloc(uint32_t C) {
    qvz_codebook_t *cb = (qvz_codebook_t *)calloc(1, sizeof(*cb));
    if (!cb) return NULL;
    cb->C = C;
    cb->nctx = (uint8_t *)calloc(C, 1);
    cb->ctx_sym = (uint8_t *)calloc((size_t)C * A, 1);
    cb->ctx_idx = (uint8_t *)malloc((size_t)C * A);
    cb->qratio = (uint8_t *)calloc((size_t)C * A, 1);
    cb->ratio = (double *)calloc((size_t)C * A, sizeof(double));
    cb->q = (uint8_t *)calloc((size_t)C * 2 * A * A, 1);
    cb->nstates = (uint8_t *)calloc((size_t)C * 2 * A, 1);
    cb->recon = (uint8_t *)calloc((size_t)C * 2 * A * A, 1);
    cb->out_idx = (uint8_t *)malloc((size_t)C * 2 * A * A);
    cb->live = (uint8_t *)malloc((size_t)C * A);
    if (!cb->nctx || !cb->ctx_sym || !cb->ctx_idx || !cb->qratio || !cb->ratio || !cb->q || !cb->nstates ||
        !cb->recon || !cb->out_idx || !cb->live) {
        qvz_codebook_free(cb);
        return NULL;
    }
    memset(cb->live, 1, (size_t)C * A);      /* conservative default: every context may be visited */
    memset(cb->ctx_idx, QVZ_NOT_FOUND, (size_t)C * A);
    memset(cb->out_idx, QVZ_NOT_FOUND, (size_t)C * 2 * A * A);
    return cb;
}

void qvz_codebook_free(qvz_codebook_t *cb) {
    if (!cb) return;
    free(cb->nctx); free(cb->ctx_sym); free(cb->ctx_idx); free(cb->qratio); free(cb->ratio);
    free(cb->q); free(cb->nstates); free(cb->recon); free(cb->out_idx); free(cb->live);
    free(cb);
}

int qvz_distortion_matrix(int type, double *d) {
    for (int y = 0; y < A; ++y)
        for (int x = 0; x < A; ++x) {
            const int diff = x > y ? x - y : y - x;
            double v;
            switch (type) {
                case QVZ_DIST_MANHATTAN: v = diff; break;                          /* distortion.c:50-61 */
                case QVZ_DIST_MSE: v = diff * diff; break;                         /* distortion.c:66-77 */
                case QVZ_DIST_LORENTZ: v = log2(1.0 + (double)diff); break;        /* distortion.c:82-95 */
                default: {
                    char num[32];
                    snprintf(num, sizeof(num), "%d", type);
                    return fail(QVZ_E_ARG, "unknown distortion type %s", num);
                }
            }
            d[x + A * y] = v;
        }
    return QVZ_OK;
}

int qvz_distortion_matrix_file(const char *path, double *d) {                     /* distortion.c:100-145 */
    if (!path || !d) return fail(QVZ_E_ARG, "null distortion file name%s", "");
    FILE *fp = fopen(path, "rt");
    if (!fp) return fail(QVZ_E_ARG, "Unable to open distortion definition file %s", path);
    memset(d, 0, sizeof(double) * A * A);
    char *line = (char *)malloc(65536);
    if (!line) { fclose(fp); return fail(QVZ_E_NOMEM, "distortion file line buffer%s", ""); }
    int x = 0;
    while (x < A && fgets(line, 65536, fp) != NULL) {
        if (line[0] == '#') continue;
        const char *field = line;
        int y = 0;
        while (y < A && field != NULL) {
            d[x + A * y] = atof(field);
            field = strchr(field, ',');
            if (field) field += 1;
            y += 1;
        }
        if (y < A) {
            printf("Warning: one or more entries in the distortion matrix on line %d were missing", x);
            printf(" they have been filled with 0.0\n");
        }
        x += 1;
    }
    free(line);
    fclose(fp);
    return QVZ_OK;
}

size_t qvz_codebook_write_header(uint8_t *buf, uint32_t clusters, uint32_t columns, uint64_t lines) {
    if (buf) {                                                   /* codebook.c:481-486 (htonl, u32 truncation) */
        const uint32_t l32 = (uint32_t)lines;
        buf[0] = (uint8_t)clusters;
        for (int b = 0; b < 4; ++b) {
            buf[1 + b] = (uint8_t)(columns >> (24 - 8 * b));
            buf[5 + b] = (uint8_t)(l32 >> (24 - 8 * b));
        }
    }
    return 9;
}

/* one byte / one 72-entry table of the codebook text (every stored value is offset by 33 = '!') */
static inline size_t put(uint8_t *buf, size_t pos, uint32_t b) {
    if (buf) buf[pos] = (uint8_t)b;
    return pos + 1;
}
static inline size_t put_table(uint8_t *buf, size_t pos, const uint8_t *q) {
    for (int i = 0; i < A; ++i) pos = put(buf, pos, q[i] + 33u);
    return pos;
}

size_t qvz_codebook_write(uint8_t *buf, const qvz_codebook_t *cb) {
    size_t pos = 0;
    pos = put(buf, pos, cb->qratio[0] + 33u); pos = put(buf, pos, '\n');          /* codebook.c:515-517 */
    pos = put_table(buf, pos, cb->q); pos = put(buf, pos, '\n');                  /* :520-522 */
    pos = put_table(buf, pos, cb->q + A); pos = put(buf, pos, '\n');              /* :525-528 */
    for (uint32_t s = 1; s < cb->C; ++s) {                                        /* :531-554 */
        const uint32_t n = cb->nctx[s];
        for (uint32_t j = 0; j < n; ++j) pos = put(buf, pos, cb->qratio[(size_t)s * A + j] + 33u);
        pos = put(buf, pos, '\n');
        for (uint32_t b = 0; b < 2; ++b) {
            for (uint32_t j = 0; j < n; ++j) pos = put_table(buf, pos, cb->q + ((size_t)s * 2 * A + 2 * j + b) * A);
            pos = put(buf, pos, '\n');
        }
    }
    return pos;
}

void qvz_codebook_tables(const qvz_codebook_t *cb, uint16_t *ct, uint16_t *qs) {
    for (uint32_t s = 0; s < cb->C; ++s) {
        for (int p = 0; p < A; ++p) {
            const uint8_t ctx = cb->ctx_idx[(size_t)s * A + p];
            ct[(size_t)s * A + p] = ctx == QVZ_NOT_FOUND
                                        ? (uint16_t)QVZ_NOT_FOUND
                                        : (uint16_t)(ctx | (cb->qratio[(size_t)s * A + ctx] << 8));
        }
        for (uint32_t qi = 0; qi < 2 * A; ++qi) {
            const size_t t = ((size_t)s * 2 * A + qi) * A;
            /* a quantizer the column does not have (most of the 144: a -f codebook has a handful of contexts) is a row of one value
             * in `q` (zeros): all its table entries are the same, written without looking each up (this loop over all 1.5 million
             * entries of a 150-column codebook took 0.5 ms between the last codebook kernel and the quantizer's launch) */
            if (qi >= 2u * cb->nctx[s] && memcmp(cb->q + t, cb->q + t + 1, A - 1) == 0) {
                const uint8_t qv = cb->q[t];
                const uint16_t e = (uint16_t)(qv | (cb->out_idx[t + qv] << 8));
                for (int x = 0; x < A; ++x) qs[t + x] = e;
                continue;
            }
            for (int x = 0; x < A; ++x) {
                const uint8_t qv = cb->q[t + x];
                qs[t + x] = (uint16_t)(qv | (cb->out_idx[t + qv] << 8));
            }
        }
    }
}
