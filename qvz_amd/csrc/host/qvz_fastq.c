/* FASTQ front end (the reference leaves the extraction of the quality lines to the user,
 * README.md:30-34): the fourth line of every four-line record becomes one row of the (C+1)-byte text layout that
 * load_file maps (src/lines.c:44-71), which is what everything downstream ingests. Fixed read length, as in the
 * reference ("columns" comes from the first line, lines.c:45); '\r' before '\n' is dropped; wrapped (multi-line)
 * sequences are not FASTQ as sequencers write it and are refused.
 *
 * The scan is a memchr walk; up to 16 threads take byte ranges, each resynchronising on the first record start
 * at or after its offset: a line that begins with '@' whose next-but-one line begins with '+' and whose second and
 * fourth lines have the read length. Pass 1 counts records per range, pass 2 writes the rows. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "qvz_encode.h"
#include "qvz_hip.h"
#include "qvz_job.h"

typedef struct { const uint8_t *p; uint64_t len; } line_ref;      /* without the terminator */

/* the line starting at offset o; returns the offset of the next line (bytes when this was the last one) */
static uint64_t take_line(const uint8_t *t, uint64_t bytes, uint64_t o, line_ref *l) {
    const uint8_t *nl = (const uint8_t *)memchr(t + o, '\n', (size_t)(bytes - o));
    uint64_t end = nl ? (uint64_t)(nl - t) : bytes;
    l->p = t + o;
    l->len = end - o;
    if (l->len && l->p[l->len - 1] == '\r') l->len -= 1;
    return nl ? end + 1 : bytes;
}

static int bases_only(const line_ref *l) {
    for (uint64_t i = 0; i < l->len; ++i) {
        const uint8_t c = l->p[i] & 0xDF;                          /* upper case */
        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N' || c == 'U' || l->p[i] == '.' ||
              c == 'R' || c == 'Y' || c == 'K' || c == 'M' || c == 'S' || c == 'W' || c == 'B' || c == 'D' || c == 'H' || c == 'V'))
            return 0;
    }
    return 1;
}

/* is there a record at offset o (o = start of a line)? fills the four lines and the offset after it */
static int record_at(const uint8_t *t, uint64_t bytes, uint64_t o, uint32_t columns, line_ref L[4], uint64_t *next) {
    if (o >= bytes || t[o] != '@') return 0;
    uint64_t q = o;
    for (int i = 0; i < 4; ++i) {
        if (q >= bytes) return 0;                                  /* fewer than four lines left */
        q = take_line(t, bytes, q, &L[i]);
    }
    if (!L[2].len || L[2].p[0] != '+') return 0;
    if (L[1].len != L[3].len || L[1].len == 0) return 0;
    if (columns && L[3].len != columns) return 0;
    *next = q;
    return 1;
}

int qvz_fastq_detect(const uint8_t *text, uint64_t bytes) {
    line_ref L[4];
    uint64_t next;
    return text && bytes && record_at(text, bytes, 0, 0, L, &next) && bases_only(&L[1]);
}

typedef struct {
    const uint8_t *t;
    uint64_t bytes, lo, hi;       /* records that START in [lo, hi) belong to this range */
    uint32_t columns;
    uint8_t *out;                 /* NULL in the counting pass */
    uint64_t first_row, count;
    uint64_t start, end;          /* where this range's first record starts and its last one ends (count > 0) */
    int rc;
    uint64_t bad_at;
} fq_range;

static void *fq_worker(void *arg) {
    fq_range *r = (fq_range *)arg;
    const uint8_t *t = r->t;
    uint64_t o = r->lo;
    line_ref L[4];
    uint64_t next;
    if (o) {                      /* resynchronise: first line start at or after lo that begins a record */
        const uint8_t *nl = (const uint8_t *)memchr(t + o - 1, '\n', (size_t)(r->bytes - o + 1));
        o = nl ? (uint64_t)(nl - t) + 1 : r->bytes;
        while (o < r->hi && !record_at(t, r->bytes, o, r->columns, L, &next)) {
            nl = (const uint8_t *)memchr(t + o, '\n', (size_t)(r->bytes - o));
            o = nl ? (uint64_t)(nl - t) + 1 : r->bytes;
        }
    }
    const uint64_t stride = (uint64_t)r->columns + 1;
    uint64_t n = 0;
    while (o < r->hi && o < r->bytes) {
        if (!record_at(t, r->bytes, o, r->columns, L, &next)) {
            /* trailing blank lines are fine; anything else is a malformed or variable-length record */
            uint64_t z = o;
            while (z < r->bytes && (t[z] == '\n' || t[z] == '\r')) ++z;
            if (z == r->bytes) break;
            r->rc = 1; r->bad_at = o;
            break;
        }
        if (r->out) {
            uint8_t *row = r->out + (r->first_row + n) * stride;
            memcpy(row, L[3].p, r->columns);
            row[r->columns] = '\n';
        }
        if (n == 0) r->start = o;
        n += 1;
        o = next;
        r->end = o;
    }
    r->count = n;
    return NULL;
}

int qvz_fastq_qualities(const uint8_t *text, uint64_t bytes, uint8_t **rows_out, uint64_t *reads_out, uint32_t *columns_out) {
    line_ref L[4];
    uint64_t next;
    if (!text || !rows_out || !reads_out || !columns_out) return qvz_job_fail(QVZ_E_ARG, "null FASTQ argument");
    if (!record_at(text, bytes, 0, 0, L, &next)) return qvz_job_fail(QVZ_E_DATA, "the input does not start with a four-line FASTQ record");
    if (L[3].len > 0xFFFFFFF0ull) return qvz_job_fail(QVZ_E_DATA, "read too long");
    const uint32_t columns = (uint32_t)L[3].len;
    enum { T = 16 };
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = bytes < (1u << 20) ? 1 : bytes < (64u << 20) ? 4 : T;
    if (cores > 0 && nt > cores) nt = (int)cores;
    fq_range R[T];
    pthread_t th[T];
    uint8_t *rows = NULL;
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t row = 0;
        for (int i = 0; i < nt; ++i) {
            if (pass == 0) {
                memset(&R[i], 0, sizeof(R[i]));
                R[i].t = text; R[i].bytes = bytes; R[i].columns = columns;
                R[i].lo = bytes / (uint64_t)nt * (uint64_t)i;
                R[i].hi = i + 1 == nt ? bytes : bytes / (uint64_t)nt * (uint64_t)(i + 1);
            } else {
                R[i].out = rows; R[i].first_row = row; row += R[i].count;
            }
        }
        for (int i = 1; i < nt; ++i) pthread_create(&th[i], NULL, fq_worker, &R[i]);
        fq_worker(&R[0]);
        for (int i = 1; i < nt; ++i) pthread_join(th[i], NULL);
        uint64_t total = 0;
        for (int i = 0; i < nt; ++i) {
            if (R[i].rc) {
                free(rows);
                return qvz_job_fail(QVZ_E_DATA, "FASTQ record at byte %llu is malformed or its read length differs from %u",
                                    (unsigned long long)R[i].bad_at, columns);
            }
            total += R[i].count;
        }
        if (pass == 0) {
            /* A range other than the first finds its first record by skipping lines until one starts a record; a malformed
             * record (or one of another read length) right at the boundary would be skipped without a word. So the ranges
             * must chain: each one's first record starts where the previous one's last record ended. */
            uint64_t expect = 0;
            for (int i = 0; i < nt; ++i) {
                if (!R[i].count) continue;
                if (R[i].start != expect) {
                    free(rows);
                    return qvz_job_fail(QVZ_E_DATA, "FASTQ record at byte %llu is malformed or its read length differs from %u",
                                        (unsigned long long)expect, columns);
                }
                expect = R[i].end;
            }
        }
        if (pass == 0) {
            rows = (uint8_t *)malloc((size_t)(total ? total : 1) * ((size_t)columns + 1));
            if (!rows) return qvz_job_fail(QVZ_E_NOMEM, "no memory for %llu quality rows", (unsigned long long)total);
            *reads_out = total;
        }
    }
    *rows_out = rows;
    *columns_out = columns;
    return QVZ_OK;
}
