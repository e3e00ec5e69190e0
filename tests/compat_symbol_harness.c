/* Test program for the symbol-level half of include/qvz_compat.h (TEST CODE, built by tests/test_compat_cpu.py with gcc
 * against qvz_amd/libqvzcompat.so). It drives the interface the way a caller of the reference's qv_compressor.h would:
 *
 *   harness decode IN.q OUT.txt OUT.labels    read_codebooks, initialize_qv_compressor(DECOMPRESSION), then one
 *                                             qv_read_cluster + C x (choose_quantizer, decompress_qv) per line, the last
 *                                             symbol of the file through decoder_last_step
 *   harness encode IN.q IN.txt IN.labels OUT.q  codebooks taken from IN.q; initialize_qv_compressor(COMPRESSION), then per line
 *                                             qv_write_cluster + C x (choose_quantizer, table look-up, compress_qv),
 *                                             encoder_last_step. With QVZ_WELL_SEED=debug OUT.q must equal IN.q.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_compat.h"

static void fail(const char *what) { fprintf(stderr, "harness: %s\n", what); exit(2); }

static long open_codebooks(const char *path, FILE **fp, struct quality_file_t *info, struct qv_options_t *opts) {
    memset(info, 0, sizeof(*info));
    memset(opts, 0, sizeof(*opts));
    info->alphabet = alloc_alphabet(72);
    info->opts = opts;
    *fp = fopen(path, "rb");
    if (!*fp) fail("cannot open the .q file");
    read_codebooks(*fp, info);
    return ftell(*fp);                       /* the WELL state starts here */
}

int main(int argc, char **argv) {
    struct quality_file_t info;
    struct qv_options_t opts;
    FILE *fq;
    if (argc == 5 && strcmp(argv[1], "decode") == 0) {
        open_codebooks(argv[2], &fq, &info, &opts);
        FILE *fo = fopen(argv[3], "wb"), *fl = fopen(argv[4], "wb");
        if (!fo || !fl) fail("cannot open outputs");
        qv_compressor qvc = initialize_qv_compressor(fq, DECOMPRESSION, &info);
        const uint32_t C = info.columns;
        char *line = (char *)malloc(C + 1);
        line[C] = '\n';
        for (uint64_t i = 0; i < info.lines; ++i) {
            const uint8_t cl = qv_read_cluster(qvc->Quals);
            if (cl >= info.cluster_count) fail("cluster id out of range");
            struct cond_quantizer_list_t *ql = info.clusters->clusters[cl].qlist;
            uint8_t prev = 0;
            for (uint32_t s = 0; s < C; ++s) {
                uint32_t idx;
                struct quantizer_t *q = choose_quantizer(ql, &info.well, s, prev, &idx);
                const int last = i + 1 == info.lines && s + 1 == C;
                const uint32_t st = last ? decoder_last_step(qvc->Quals->a, qvc->Quals->stats[cl][s][idx]) : decompress_qv(qvc->Quals, cl, s, idx);
                if (st >= q->output_alphabet->size) fail("state out of range");
                prev = q->output_alphabet->symbols[st];
                line[s] = (char)(prev + 33);
            }
            fwrite(line, 1, C + 1, fo);
            fputc(cl, fl);
        }
        fclose(fo); fclose(fl); fclose(fq);
        return 0;
    }
    if (argc == 6 && strcmp(argv[1], "encode") == 0) {
        const long head = open_codebooks(argv[2], &fq, &info, &opts);
        FILE *fi = fopen(argv[3], "rb"), *fl = fopen(argv[4], "rb"), *fo = fopen(argv[5], "wb");
        if (!fi || !fl || !fo) fail("cannot open inputs / output");
        char *hb = (char *)malloc((size_t)head);                                 /* header + codebooks as they are */
        fseek(fq, 0, SEEK_SET);
        if (fread(hb, 1, (size_t)head, fq) != (size_t)head) fail("short .q file");
        fwrite(hb, 1, (size_t)head, fo);
        qv_compressor qvc = initialize_qv_compressor(fo, COMPRESSION, &info);
        const uint32_t C = info.columns;
        char *line = (char *)malloc(C + 2);
        for (uint64_t i = 0; i < info.lines; ++i) {
            if (fread(line, 1, C + 1, fi) != C + 1) fail("short input text");
            const int cl = fgetc(fl);
            if (cl < 0 || cl >= info.cluster_count) fail("bad label");
            struct cond_quantizer_list_t *ql = info.clusters->clusters[cl].qlist;
            qv_write_cluster(qvc->Quals, (uint8_t)cl);
            uint8_t prev = 0;
            for (uint32_t s = 0; s < C; ++s) {
                uint32_t idx;
                struct quantizer_t *q = choose_quantizer(ql, &info.well, s, prev, &idx);
                const uint8_t qv = q->q[(uint8_t)line[s] - 33];
                compress_qv(qvc->Quals, q->output_alphabet->indexes[qv], (uint8_t)cl, s, idx);
                prev = qv;
            }
        }
        encoder_last_step(qvc->Quals->a, qvc->Quals->os);
        fclose(fo); fclose(fi); fclose(fl); fclose(fq);
        return 0;
    }
    fprintf(stderr, "usage: %s decode IN.q OUT.txt OUT.labels | encode IN.q IN.txt IN.labels OUT.q\n", argv[0]);
    return 2;
}
