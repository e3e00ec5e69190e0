/* QGEN-1: this repository's own synthetic Q-score generator.
 * TEST / BENCH INFRASTRUCTURE. Counter-based (splitmix64), so any read can be produced
 * independently: the HIP generator in qvz_amd/csrc restates the same recurrence on device.
 * Check value: (seed=1, N=10000, C=100) -> text file md5 c304bc40ee06fefa23ab9594243295e2.
 */
#ifndef QGEN1_H
#define QGEN1_H
#include <stdint.h>

static inline uint64_t qgen1_sm64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline int qgen1_clamp(int v) { return v < 2 ? 2 : (v > 41 ? 41 : v); }

/* Writes the C symbols (Q values 2..41, NO +33 offset) of read i into out[0..C-1]. */
static inline void qgen1_read(uint64_t seed, uint64_t i, uint32_t C, uint8_t *out) {
    uint64_t r = qgen1_sm64(seed * 0xD1342543DE82EF95ull + i);
    unsigned cls = (unsigned)(r % 10);
    int base = cls < 6 ? 38 : (cls < 9 ? 30 : 20);
    int slope = (int)((r >> 8) % 13);
    int q = base;
    for (uint32_t j = 0; j < C; ++j) {
        uint64_t h = qgen1_sm64(r + j + 1);
        int target = base - (int)(((uint64_t)slope * j) / C);
        unsigned u = (unsigned)(h & 15);
        int step = u < 6 ? 0 : (u < 10 ? 1 : (u < 14 ? -1 : (u == 14 ? -2 : 2)));
        int pull = 0;
        if (((h >> 4) & 3) == 0) pull = (target > q) - (target < q);
        q = qgen1_clamp(q + step + pull);
        int drop = 0;
        if (((h >> 6) & 63) == 0) drop = -(int)((h >> 12) % 20);
        out[j] = (uint8_t)qgen1_clamp(q + drop);
    }
}
#endif
