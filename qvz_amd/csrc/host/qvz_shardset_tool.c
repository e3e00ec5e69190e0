/* qvz-shardset - inspects a shard set (include/qvz_shardset.h) and materialises reference-format shard files from it:
 *     qvz-shardset stat  SET
 *     qvz-shardset split SET PREFIX [FIRST [COUNT]]      -> PREFIX.0000 ..., each decodable by the unmodified reference
 * (the reference has no counterpart: its format holds one stream, src/codebook.c:474-555) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_shardset.h"

int main(int argc, char **argv) {
    if (argc >= 3 && strcmp(argv[1], "stat") == 0) {
        qvz_shardset_info_t si;
        if (qvz_shardset_stat(argv[2], &si)) { printf("%s\n", qvz_encode_last_error()); return 1; }
        const double sym = (double)si.lines * si.columns;
        printf("clusters, %u, columns, %u, lines, %llu, shards, %u\n", si.clusters, si.columns, (unsigned long long)si.lines, si.shards);
        printf("header bytes, %llu, codebook bytes, %llu, stream bytes, %llu, file bytes, %llu\n", (unsigned long long)si.header_bytes,
               (unsigned long long)si.codebook_bytes, (unsigned long long)si.stream_bytes, (unsigned long long)si.file_bytes);
        printf("rate, %.4f, rate on disk, %.4f\n", si.stream_bytes * 8. / sym, si.file_bytes * 8. / sym);
        return 0;
    }
    if (argc >= 4 && strcmp(argv[1], "split") == 0) {
        const uint32_t first = argc > 4 ? (uint32_t)strtoul(argv[4], 0, 10) : 0, count = argc > 5 ? (uint32_t)strtoul(argv[5], 0, 10) : 0;
        const int dev = getenv("QVZ_DEVICE") ? atoi(getenv("QVZ_DEVICE")) : 0;
        if (qvz_shardset_split(argv[2], argv[3], first, count, dev)) { printf("%s\n", qvz_encode_last_error()); return 1; }
        return 0;
    }
    printf("Usage: %s stat SET | split SET PREFIX [FIRST [COUNT]]\n", argv[0]);
    return 1;
}
