/* qgen1_tool seed N C out.txt : writes a QGEN-1 quality file in qvz's input format
 * (one read per line, bytes Q+33, '\n' line ends; reference loader: src/lines.c:27-82).
 * Test/bench infrastructure. */
#include <stdio.h>
#include <stdlib.h>
#include "qgen1.h"
int main(int argc, char **argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s seed N C out\n", argv[0]); return 2; }
    uint64_t seed = strtoull(argv[1], 0, 10), n = strtoull(argv[2], 0, 10);
    uint32_t c = (uint32_t)strtoul(argv[3], 0, 10);
    FILE *f = fopen(argv[4], "wb");
    if (!f) { perror("open"); return 1; }
    uint8_t *line = malloc(c + 1);
    for (uint64_t i = 0; i < n; ++i) {
        qgen1_read(seed, i, c, line);
        for (uint32_t j = 0; j < c; ++j) line[j] += 33;
        line[c] = '\n';
        fwrite(line, 1, c + 1, f);
    }
    fclose(f); free(line);
    return 0;
}
