/* well_far.c - golden WELL-1024a states far into the stream, made by STEPPING THE REFERENCE'S OWN well_1024a
 * (/root/reference/src/well.c:8-24, linked from oracle/_ref/libqvzref.so - nothing of it is copied here).
 * TEST INFRASTRUCTURE: run once in the build container by tools/make_well_golden.py, which commits the output as
 * tests/golden/well_far.json. A state is what the generator holds after `words` calls from 32 x 0x55555555, n = 0
 * (qv_stream.c:79-83 with -DDEBUG): the reference's raw state[32] and n, plus the next four words it returns.
 *
 *   well_far <words>...      (ascending word counts)                                                             */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct well_state_t {            /* the reference's layout, include/well.h:11-16 (an ABI fact of libqvzref.so) */
    uint32_t state[32];
    uint32_t n, bit_output, bits_left;
};
uint32_t well_1024a(struct well_state_t *state);

int main(int argc, char **argv) {
    struct well_state_t w;
    for (int i = 0; i < 32; ++i) w.state[i] = 0x55555555u;
    w.n = 0; w.bit_output = 0; w.bits_left = 0;
    uint64_t done = 0;
    printf("[\n");
    for (int a = 1; a < argc; ++a) {
        const uint64_t target = strtoull(argv[a], NULL, 10);
        if (target < done) { fprintf(stderr, "word counts must ascend\n"); return 2; }
        for (; done < target; ++done) (void)well_1024a(&w);
        struct well_state_t peek = w;
        printf(" {\"words\": %llu, \"n\": %u, \"state\": [", (unsigned long long)target, w.n);
        for (int i = 0; i < 32; ++i) printf("%u%s", w.state[i], i < 31 ? ", " : "");
        printf("], \"next\": [");
        for (int i = 0; i < 4; ++i) printf("%u%s", well_1024a(&peek), i < 3 ? ", " : "");
        printf("]}%s\n", a + 1 < argc ? "," : "");
        fflush(stdout);
    }
    printf("]\n");
    return 0;
}
