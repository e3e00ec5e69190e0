/* ref_harness: drives the UNMODIFIED reference (linked from oracle/_ref/libqvzref.so, headers taken
 * from /root/reference/include at compile time, nothing copied) through the calls encode() makes
 * (reference src/main.c:42-67) and dumps the per-phase intermediates as a flat binary so that the
 * oracle's k-means labels, PMF counters and marginals can be pinned directly, not only via .q/.u.
 *
 *   ref_harness in.txt k mode(f|r) ratio dist(M|L|A) out.bin
 * out.bin: u64 N, u32 C, u32 k, labels[N] u8, means[k][C] u8, counts[k] u32,
 *          cnt[k][C][72][72] u32, marg[k][C][72] f64
 * TEST INFRASTRUCTURE (golden-vector generation in this container only). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "codebook.h"
#include "cluster.h"
#include "qv_compressor.h"

int main(int argc, char **argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s in k f|r ratio M|L|A out.bin\n", argv[0]); return 2; }
    struct qv_options_t opts;
    memset(&opts, 0, sizeof(opts));
    opts.clusters = (uint8_t)atoi(argv[2]);
    opts.mode = argv[3][0] == 'f' ? MODE_RATIO : MODE_FIXED;
    opts.ratio = atof(argv[4]);
    opts.distortion = argv[5][0] == 'L' ? DISTORTION_LORENTZ : (argv[5][0] == 'A' ? DISTORTION_MANHATTAN : DISTORTION_MSE);
    opts.cluster_threshold = 4;

    struct quality_file_t info;
    memset(&info, 0, sizeof(info));
    info.alphabet = alloc_alphabet(72);
    info.dist = generate_distortion_matrix(72, opts.distortion);
    info.cluster_count = opts.clusters;
    if (load_file(argv[1], &info, 0) != LF_ERROR_NONE) { fprintf(stderr, "load_file failed\n"); return 1; }
    info.clusters = alloc_cluster_list(&info);
    info.opts = &opts;
    do_kmeans_clustering(&info);
    calculate_statistics(&info);

    FILE *f = fopen(argv[6], "wb");
    uint64_t N = info.lines; uint32_t C = info.columns, k = info.cluster_count;
    fwrite(&N, 8, 1, f); fwrite(&C, 4, 1, f); fwrite(&k, 4, 1, f);
    for (uint32_t b = 0; b < info.block_count; ++b)
        for (uint32_t i = 0; i < info.blocks[b].count; ++i) fwrite(&info.blocks[b].lines[i].cluster, 1, 1, f);
    for (uint32_t j = 0; j < k; ++j) fwrite(info.clusters->clusters[j].mean, 1, C, f);
    for (uint32_t j = 0; j < k; ++j) fwrite(&info.clusters->clusters[j].count, 4, 1, f);
    uint32_t zeros[72]; memset(zeros, 0, sizeof(zeros));
    for (uint32_t j = 0; j < k; ++j) {
        struct cond_pmf_list_t *l = info.clusters->clusters[j].training_stats;
        for (uint32_t s = 0; s < C; ++s)
            for (uint32_t p = 0; p < 72; ++p) {
                if (s == 0 && p > 0) { fwrite(zeros, 4, 72, f); continue; }
                fwrite(get_cond_pmf(l, s, (symbol_t)p)->counts, 4, 72, f);
            }
    }
    for (uint32_t j = 0; j < k; ++j) {
        struct cond_pmf_list_t *l = info.clusters->clusters[j].training_stats;
        for (uint32_t s = 0; s < C; ++s) {
            double m[72];
            for (uint32_t x = 0; x < 72; ++x) m[x] = get_probability(l->marginal_pmfs->pmfs[s], x);
            fwrite(m, 8, 72, f);
        }
    }
    fclose(f);
    return 0;
}
