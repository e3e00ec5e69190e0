/* qvz - command line front end with the reference's grammar (mikelhernaez/qvz src/main.c:189-356):
 *     qvz (options) [input file] [output file]
 * one token per option, an option's value is the next argv, no combined short options. Encoding and decoding (`-x`) run
 * on the MI355X through libqvzhip.
 * Environment (additions): QVZ_WELL_SEED=debug|time (default time), QVZ_DEVICE=<ordinal>,
 * QVZ_CHUNK_READS=<n>, QVZ_SHARDS=<n> (sharded output: one shard-set file, include/qvz_shardset.h; with
 * QVZ_SHARD_FILES=1 one reference-format file per shard instead), QVZ_INPUT=fastq|qual, QVZ_TIMING=1 (phase seconds on stderr),
 * QVZ_SWEEP=rd | f:<v>,<v>,... | r:<v>,<v>,... (a rate-distortion sweep in one process, reference generate_rd.sh:1-17: the
 * cluster labels and the counters are made once, every point prints its -s line and writes <output>.NN; QVZ_SWEEP_OUT=last:
 * every point overwrites <output> as the script's encodes do). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qvz_encode.h"
#include "qvz_shardset.h"

static void usage(const char *name) {   /* main.c:166-184 */
    printf("Usage: %s (options) [input file] [output file]\n", name);
    printf("Options are:\n");
    printf("   -q           : Store quality values in compressed file (default)\n");
    printf("   -x           : Extract quality values from compressed file\n");
    printf("   -f [ratio]   : Compress using [ratio] bits per bit of input entropy per symbol\n");
    printf("   -r [rate]    : Compress using fixed [rate] bits per symbol\n");
    printf("   -d [M|L|A]   : Optimize for MSE, Log(1+L1), L1 distortions, respectively (default: MSE)\n");
    printf("   -D [FILE]    : Optimize using the custom distortion matrix specified in FILE\n");
    printf("   -c [#]       : Compress using [#] clusters (default: 1)\n");
    printf("   -T [#]       : Use [#] as a threshold for cluster center movement (L2 norm) to declare a stable solution (default: 4).\n");
    printf("   -u [FILE]    : Write the uncompressed lossy values to FILE (default: off)\n");
    printf("   -h           : Print this help\n");
    printf("   -s           : Print summary stats\n");
    printf("   -v           : Enable verbose output\n");
    printf("\nFor custom distortion matrices, a 72x72 matrix of values must be provided as the cost of reconstructing\n");
    printf("the x-th row as the y-th column, where x and y range from 0 to 71 (inclusive) corresponding to the possible\n");
    printf("Phred scores.\n");
}

int main(int argc, char **argv) {
    const char *input_name = 0, *output_name = 0;
    qvz_options_t opts;
    int extract = 0, file_idx = 0, i = 1;
    qvz_options_default(&opts);

    while (i < argc) {
        if (argv[i][0] != '-') {                                  /* main.c:210-229 */
            if (file_idx == 0) { input_name = argv[i]; file_idx = 1; }
            else if (file_idx == 1) { output_name = argv[i]; file_idx = 2; }
            else {
                printf("Garbage argument \"%s\" detected.\n", argv[i]);
                usage(argv[0]);
                exit(1);
            }
            i += 1;
            continue;
        }
        const int has_value = i + 1 < argc;
        switch (argv[i][1]) {                                     /* main.c:232-302 */
            case 'x': extract = 1; i += 1; break;
            case 'q': extract = 0; i += 1; break;
            case 'f': extract = 0; opts.ratio = has_value ? atof(argv[i + 1]) : 0; opts.mode = QVZ_MODE_RATIO; i += 2; break;
            case 'r': extract = 0; opts.ratio = has_value ? atof(argv[i + 1]) : 0; opts.mode = QVZ_MODE_FIXED; i += 2; break;
            case 'c': opts.clusters = has_value ? atoi(argv[i + 1]) : 1; i += 2; break;
            case 'v': opts.verbose = 1; i += 1; break;
            case 'h': usage(argv[0]); exit(0);
            case 's': opts.stats = 1; i += 1; break;
            case 'u': opts.uncompressed_name = has_value ? argv[i + 1] : NULL; i += 2; break;
            case 'T': opts.cluster_threshold = has_value ? atoi(argv[i + 1]) : 4; i += 2; break;
            case 'd':
                switch (has_value ? argv[i + 1][0] : 0) {
                    case 'M': opts.distortion = QVZ_DIST_MSE; break;
                    case 'L': opts.distortion = QVZ_DIST_LORENTZ; break;
                    case 'A': opts.distortion = QVZ_DIST_MANHATTAN; break;
                    default: printf("Distortion measure not supported, using MSE.\n"); break;
                }
                i += 2;
                break;
            case 'D':                                             /* main.c:293-297 */
                opts.distortion = QVZ_DIST_CUSTOM;
                opts.dist_file = has_value ? argv[i + 1] : NULL;
                i += 2;
                break;
            default:
                printf("Unrecognized option -%c.\n", argv[i][1]);
                usage(argv[0]);
                exit(1);
        }
    }
    if (getenv("QVZ_INPUT")) {                                    /* "fastq" / "qual": skip the look at the input */
        opts.input_format = strcmp(getenv("QVZ_INPUT"), "fastq") == 0 ? QVZ_INPUT_FASTQ : strcmp(getenv("QVZ_INPUT"), "qual") == 0 ? QVZ_INPUT_QUAL : QVZ_INPUT_AUTO;
    }
    if (file_idx != 2) {                                          /* main.c:305-309 */
        printf("Missing required filenames.\n");
        usage(argv[0]);
        exit(1);
    }
    if (opts.verbose) {                                           /* main.c:311-344 */
        if (extract) {
            printf("%s will be decoded to %s.\n", input_name, output_name);
        } else {
            printf("%s will be encoded as %s.\n", input_name, output_name);
            if (opts.mode == QVZ_MODE_RATIO) printf("Ratio mode selected, targeting %f compression ratio.\n", opts.ratio);
            else if (opts.mode == QVZ_MODE_FIXED) printf("Fixed-rate mode selected, targeting %f bits per symbol.\n", opts.ratio);
            switch (opts.distortion) {
                case QVZ_DIST_MSE: printf("MSE will be used as a distortion metric.\n"); break;
                case QVZ_DIST_LORENTZ: printf("log(1+L1) will be used as a distortion metric.\n"); break;
                case QVZ_DIST_MANHATTAN: printf("L1 will be used as a distortion metric.\n"); break;
                case QVZ_DIST_CUSTOM: printf("A custom distortion metric stored in %s will be used.\n", opts.dist_file); break;   /* main.c:334-336 */
            }
            printf("Compression will use %d clusters, with a movement threshold of %.0f.\n", opts.clusters, opts.cluster_threshold);
        }
    }
    if (extract) {                                                /* main.c:132-183 */
        /* a reference-format file or a shard set: qvz_decode_file looks at the first bytes. QVZ_SHARDS=N with
         * QVZ_SHARD_FILES=1: the input is the prefix of N reference-format shard files */
        const uint32_t nsh = getenv("QVZ_SHARDS") && getenv("QVZ_SHARD_FILES") && atoi(getenv("QVZ_SHARD_FILES")) ? (uint32_t)strtoul(getenv("QVZ_SHARDS"), 0, 10) : 0;
        const int dev = getenv("QVZ_DEVICE") ? atoi(getenv("QVZ_DEVICE")) : 0;
        qvz_result_t dres;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        const int drc = nsh ? qvz_decode_shard_files(input_name, nsh, output_name, dev, &dres)
                            : qvz_decode_file(input_name, output_name, dev, &dres);
        if (drc) {
            printf("%s\n", qvz_encode_last_error());
            exit(1);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (opts.verbose)                                         /* main.c:158-160 */
            printf("Decoded %llu lines in %f seconds.\n", (unsigned long long)dres.lines,
                   (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
        return 0;
    }
    const char *seed = getenv("QVZ_WELL_SEED");
    if (seed && strcmp(seed, "debug") == 0) opts.seed_mode = QVZ_SEED_DEBUG;
    if (getenv("QVZ_DEVICE")) opts.device = atoi(getenv("QVZ_DEVICE"));
    if (getenv("QVZ_CHUNK_READS")) opts.chunk_reads = strtoull(getenv("QVZ_CHUNK_READS"), 0, 10);

    qvz_result_t res;
    /* QVZ_SHARDS=N (not a reference option, so not on the command line): N independent streams coded on the GPU, written
     * as one shard set, or with QVZ_SHARD_FILES=1 as N reference-format files <output>.0000 ... */
    const uint32_t shards = getenv("QVZ_SHARDS") ? (uint32_t)strtoul(getenv("QVZ_SHARDS"), 0, 10) : 0;
    const int files = getenv("QVZ_SHARD_FILES") && atoi(getenv("QVZ_SHARD_FILES"));
    /* QVZ_DEVICES=0,1,...: the reads are dealt to several GPUs of this process (host/qvz_multi.c); the output is the shard
     * set a single GPU writes for the same QVZ_SHARDS (default: 8192 streams per device) */
    int devices[64];
    uint32_t ndev = 0;
    if (getenv("QVZ_DEVICES")) {
        const char *p = getenv("QVZ_DEVICES");
        while (*p && ndev < 64) {
            char *end;
            const long d = strtol(p, &end, 10);
            if (end == p) break;
            devices[ndev++] = (int)d;
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (getenv("QVZ_SWEEP")) {                                    /* generate_rd.sh in one process (include/qvz_encode.h) */
        qvz_sweep_point_t points[100];
        const uint32_t np = qvz_sweep_parse(getenv("QVZ_SWEEP"), &opts, points, 100);
        if (!np || ndev > 1 || files) {
            printf("QVZ_SWEEP: expected rd, f:<v>,<v>,... or r:<v>,<v>,... (at most 100 points; not with several QVZ_DEVICES or QVZ_SHARD_FILES)\n");
            exit(1);
        }
        if (ndev == 1) opts.device = devices[0];
        opts.stats = 1;                                           /* the -s lines are the sweep's product (generate_rd.sh:8,16) */
        const int keep_all = !(getenv("QVZ_SWEEP_OUT") && strcmp(getenv("QVZ_SWEEP_OUT"), "last") == 0);
        if (qvz_encode_file_sweep(input_name, output_name, &opts, points, np, shards, keep_all, NULL)) {
            printf("%s\n", qvz_encode_last_error());
            exit(1);
        }
        return 0;
    }
    if (ndev == 1) opts.device = devices[0];                      /* one ordinal named: that GPU, through the single-device driver */
    if (ndev > 1 && files) {                                      /* options that cannot be honoured together are refused, not ignored */
        printf("QVZ_SHARD_FILES=1 is not available with several QVZ_DEVICES: write the shard set and split it (qvz-shardset split)\n");
        exit(1);
    }
    const int rc = ndev > 1 ? qvz_encode_file_devices(input_name, output_name, &opts, devices, ndev, shards ? shards : 8192u * ndev, &res)
                   : !shards ? qvz_encode_file(input_name, output_name, &opts, &res)
                   : files ? qvz_encode_file_sharded(input_name, output_name, &opts, shards, &res)
                           : qvz_encode_file_shardset(input_name, output_name, &opts, shards, &res);
    if (rc) {
        printf("%s\n", qvz_encode_last_error());                  /* reference: printf + exit(1), main.c:43-46 */
        exit(1);
    }
    if (getenv("QVZ_TIMING"))                                     /* where the wall time went, one line on stderr (bench.py reads it) */
        fprintf(stderr, "qvz_timing: upload=%.4f cluster=%.4f stats=%.4f codebook=%.4f quantize=%.4f coder=%.4f egress=%.4f encode=%.4f total=%.4f "
                        "lines=%llu columns=%u stream_bytes=%llu file_bytes=%llu distortion=%.17g kmeans_iterations=%u\n",
                res.t_upload, res.t_cluster, res.t_stats, res.t_codebook, res.t_quantize, res.t_coder, res.t_egress, res.t_encode, res.t_total,
                (unsigned long long)res.lines, res.columns, (unsigned long long)res.stream_bytes, (unsigned long long)res.file_bytes,
                res.distortion, res.kmeans_iterations);
    return 0;
}
