/* fuzz_host.c - mutation fuzzing of the host-side parsers that read files somebody else wrote: the .q container and its
 * codebooks (read_codebooks, codebook.c:560-669) with the serial decoder behind them (start_qv_decompression,
 * qv_compressor.c:145-231), the shard-set container (include/qvz_shardset.h), FASTQ text and -D distortion files.
 * TEST INFRASTRUCTURE: built by tests/test_fuzz_cpu.py with clang's AddressSanitizer + UBSan over the host C sources and
 * run on the CPU (no GPU call can succeed here; entry points that need the device must fail with an error, not crash).
 *
 *   fuzz_host <seed> <iterations> <work dir> <file.q>... [--fastq file] [--dist file] [--set file]
 *   fuzz_host --make-set <file.q> <out.set>        (a one-stream shard set to seed --set with)
 *
 * Every mutant must come back with a return code: a sanitizer report, a signal or a run-away loop fails the test. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_codebook.h"
#include "qvz_encode.h"
#include "qvz_hip.h"
#include "qvz_shardset.h"
#include "qvz_job.h"

static uint64_t rng_state;
static uint64_t rnd(void) {                                   /* splitmix64 */
    uint64_t x = (rng_state += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

typedef struct { uint8_t *p; size_t n; } blob_t;

static blob_t slurp(const char *path) {
    blob_t b = {NULL, 0};
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    b.n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    b.p = (uint8_t *)malloc(b.n + 1);
    if (fread(b.p, 1, b.n, f) != b.n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return b;
}

/* a mutant of `src` in a fresh exact-size allocation (so that reads past the end are seen by the sanitizer) */
static blob_t mutate(blob_t src, size_t head) {
    blob_t m;
    size_t n = src.n;
    const unsigned kind = (unsigned)(rnd() % 8);
    if (kind == 0 && n > 1) n = (size_t)(rnd() % n);                          /* truncate anywhere */
    else if (kind == 1 && head > 1) n = (size_t)(rnd() % head);               /* truncate inside the header / codebooks */
    m.n = n;
    m.p = (uint8_t *)malloc(n ? n : 1);
    memcpy(m.p, src.p, n);
    if (!n) return m;
    /* kinds 2, 3: the part behind the header (the arithmetic stream); 4, 5: the header part; 6, 7: anywhere */
    size_t lo = 0, span = n;
    if (head && head < n) {
        if (kind == 2 || kind == 3) { lo = head; span = n - head; }
        else if (kind == 4 || kind == 5) span = head;
    }
    const unsigned hits = 1 + (unsigned)(rnd() % 4);
    for (unsigned h = 0; h < hits && kind >= 2; ++h) {
        const size_t at = lo + (size_t)(rnd() % span);
        switch (rnd() % 5) {
        case 0: m.p[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 1: m.p[at] = (uint8_t)rnd(); break;
        case 2: m.p[at] = 0xFF; break;
        case 3: m.p[at] = 0; break;
        default: for (size_t i = at; i < n && i < at + 4; ++i) m.p[i] = (rnd() & 1) ? 0xFF : 0x00; break;
        }
    }
    return m;
}

static void spill(const char *path, blob_t b) {
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(b.p, 1, b.n, f) != b.n) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: fuzz_host seed iterations workdir file.q... [--fastq f] [--dist f] [--set f]\n"); return 2; }
    if (!strcmp(argv[1], "--make-set")) {                                     /* fuzz_host --make-set in.q out.set: a one-stream shard set as seed */
        blob_t b = slurp(argv[2]);
        qvz_qfile_t h;
        memset(&h, 0, sizeof(h));
        if (qvz_qfile_parse(b.p, b.n, &h)) { fprintf(stderr, "%s does not parse\n", argv[2]); return 2; }
        FILE *f = fopen(argv[3], "wb");
        const uint64_t size = b.n - h.stream_at;
        qvz_shardset_info_t info;
        const int rc = f ? qvz_shardset_write(f, h.K, h.C, h.lines, 1, h.well, (const qvz_codebook_t *const *)h.cbs, b.p + h.stream_at, 0, &size, &info) : -1;
        if (f) fclose(f);
        qvz_qfile_release(&h);
        return rc ? 2 : 0;
    }
    rng_state = strtoull(argv[1], NULL, 10);
    const long iters = strtol(argv[2], NULL, 10);
    const char *work = argv[3];
    blob_t q[32], fastq = {NULL, 0}, dist = {NULL, 0}, set = {NULL, 0};
    size_t qhead[32];
    int nq = 0;
    for (int i = 4; i < argc; ++i) {
        if (!strcmp(argv[i], "--fastq") && i + 1 < argc) fastq = slurp(argv[++i]);
        else if (!strcmp(argv[i], "--dist") && i + 1 < argc) dist = slurp(argv[++i]);
        else if (!strcmp(argv[i], "--set") && i + 1 < argc) set = slurp(argv[++i]);
        else if (nq < 32) {
            q[nq] = slurp(argv[i]);
            qvz_qfile_t h;
            memset(&h, 0, sizeof(h));
            if (qvz_qfile_parse(q[nq].p, q[nq].n, &h)) { fprintf(stderr, "seed file %s does not parse: %s\n", argv[i], qvz_encode_last_error()); return 2; }
            qhead[nq] = h.stream_at;
            qvz_qfile_release(&h);
            ++nq;
        }
    }
    char tmp[4096], out[4096];
    snprintf(tmp, sizeof(tmp), "%s/mutant", work);
    snprintf(out, sizeof(out), "%s/out", work);
    FILE *sink = fopen("/dev/null", "w");
    {   /* the output side once under the sanitizers: pages made ahead and the writer's mapping (host/qvz_egress.c), exact and wrong estimates */
        setenv("QVZ_PREALLOC", "min=1", 1);
        const size_t n = (3u << 20) + 77;
        uint8_t *buf = (uint8_t *)malloc(n), *back = (uint8_t *)malloc(n + 64);
        for (size_t i = 0; i < n; ++i) buf[i] = (uint8_t)(rnd() >> 7);
        const uint64_t est[4] = {n, n + 100000, 4096 + 5, 0};
        for (int e = 0; e < 4; ++e) {
            FILE *f = fopen(out, "w");
            if (!f || fwrite("head", 1, 4, f) != 4 || qvz_host_write_slabs(f, buf, n, est[e], 1u << 20)) { fprintf(stderr, "write through the mapping failed\n"); return 3; }
            fclose(f);
            f = fopen(out, "rb");
            const size_t got = fread(back, 1, n + 64, f);
            fclose(f);
            if (got != n + 4 || memcmp(back, "head", 4) || memcmp(back + 4, buf, n)) { fprintf(stderr, "estimate %d: the file differs\n", e); return 3; }
        }
        free(buf); free(back);
    }
    long parsed = 0, decoded = 0, rejected = 0, fq_ok = 0, set_ok = 0, dist_ok = 0;
    for (long it = 0; it < iters; ++it) {
        if (nq) {                                                             /* .q image: header + codebooks + stream */
            const int which = (int)(rnd() % (unsigned)nq);
            blob_t m = mutate(q[which], qhead[which]);
            qvz_qfile_t h;
            memset(&h, 0, sizeof(h));
            if (qvz_qfile_parse(m.p, m.n, &h) == QVZ_OK) {
                ++parsed;
                /* a mutated line count would only make the loop long; the decoder stops by itself once the stream is used up */
                if (qvz_decode_stream_host(h.K, h.C, h.lines, (const qvz_codebook_t *const *)h.cbs, h.well, m.p + h.stream_at,
                                           m.n - h.stream_at, sink, 0) == QVZ_OK) ++decoded;
                else ++rejected;
                qvz_qfile_release(&h);
            } else ++rejected;
            free(m.p);
        }
        if (set.p && it % 4 == 0) {                                           /* shard set: stat parses all of it; decode needs the device and must fail cleanly */
            blob_t m = mutate(set, QVZ_SHARDSET_FIXED + 64);
            spill(tmp, m);
            qvz_shardset_info_t info;
            if (qvz_shardset_stat(tmp, &info) == QVZ_OK) ++set_ok;
            (void)qvz_shardset_is(m.p, m.n);
            if (it % 16 == 0 && qvz_decode_file(tmp, out, 0, NULL) == QVZ_OK) { fprintf(stderr, "decoded a shard set without a GPU?\n"); return 1; }
            free(m.p);
        }
        if (fastq.p && it % 2 == 0) {
            blob_t m = mutate(fastq, 0);
            uint8_t *rows = NULL;
            uint64_t reads = 0;
            uint32_t cols = 0;
            (void)qvz_fastq_detect(m.p, m.n);
            if (qvz_fastq_qualities(m.p, m.n, &rows, &reads, &cols) == QVZ_OK) {
                ++fq_ok;
                uint64_t sum = 0;                                             /* touch every byte the contract promises */
                for (uint64_t i = 0; i < reads * ((uint64_t)cols + 1); ++i) sum += rows[i];
                if (sum == 1) putchar(' ');
            }
            free(rows);
            free(m.p);
        }
        if (dist.p && it % 8 == 0) {
            blob_t m = mutate(dist, 0);
            spill(tmp, m);
            static double d[72 * 72];
            FILE *keep = stdout;                                              /* the reference's warnings go to stdout: not into the test log */
            stdout = sink;
            if (qvz_distortion_matrix_file(tmp, d) == QVZ_OK) ++dist_ok;
            stdout = keep;
            free(m.p);
        }
    }
    printf("fuzz_host: %ld iterations: .q parsed %ld decoded %ld rejected %ld; shard sets accepted %ld; fastq accepted %ld; distortion files accepted %ld\n",
           iters, parsed, decoded, rejected, set_ok, fq_ok, dist_ok);
    return 0;
}
