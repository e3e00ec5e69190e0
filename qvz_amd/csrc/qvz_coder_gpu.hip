// qvz_coder_gpu.hip - the arithmetic coder on gfx950 for sharded output (include/qvz_coder.h, SURVEY.md section 8e option B).
//
// The reference's stream (src/arith.c:24-116, src/qv_stream.c:9-25) is one serial chain of interval updates: a single stream cannot
// be parallelised bit-exactly. S contiguous read ranges give S independent chains, each coded exactly as the reference would code a
// file holding only those reads - same integer interval arithmetic, same E1/E2/E3 renormalisation, same adaptive counts (init 1,
// step 8, halve-and-+1 when n > 2^19). The adaptive model runs per (stream, column) in k_model_ranks, the chain with a lane per
// stream in k_code_records; DESIGN.md section 4 has the layout and the numbers, HISTORY.md what was tried and dropped.
#include <hip/hip_runtime.h>

#include <mutex>
#include <algorithm>
#include <vector>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_coder.h"
#include "qvz_internal.h"

#define A QVZ_A
#define M_BITS 22u
#define MSB (1u << (M_BITS - 1))
#define MASK_M ((1u << M_BITS) - 1)
#define MASK_M1 ((1u << (M_BITS - 1)) - 1)
#define RESCALE (1u << (M_BITS - 3))
#define STEP 8u
#define NO_TABLE 0xFFFFFFFFu

struct LaneCoder {
    uint32_t l, u, scale3;
    uint32_t nbits;          // valid bits in acc (< 32 between symbols)
    uint64_t acc;            // MSB-first accumulator
    uint8_t *out;
    uint64_t pos, cap;
    bool overflow;
};

__device__ __forceinline__ void lc_flush32(LaneCoder &c) {
    // emit the top 32 pending bits as 4 big-endian bytes (pos stays 4-byte aligned)
    const uint32_t w = (uint32_t)(c.acc >> (c.nbits - 32));
    const bool fits = c.pos + 4 <= c.cap;                        // on overflow the word lands (harmlessly) in the last slot
    *reinterpret_cast<uint32_t *>(c.out + (fits ? c.pos : c.cap - 4)) = __builtin_bswap32(w);
    c.overflow = c.overflow || !fits;
    c.pos += 4;
    c.nbits -= 32;
}

__device__ __forceinline__ void lc_put(LaneCoder &c, uint32_t value, uint32_t count) {   // count <= 32
    c.acc = (c.acc << count) | (uint64_t)(count == 32 ? value : (value & ((1u << count) - 1)));
    c.nbits += count;
    if (c.nbits >= 32) lc_flush32(c);
}

__device__ __forceinline__ void lc_run(LaneCoder &c, uint32_t bit, uint32_t count) {
    const uint32_t pat = bit ? 0xFFFFFFFFu : 0u;
    while (count) {
        const uint32_t k = count > 32 ? 32 : count;
        lc_put(c, pat, k);
        count -= k;
    }
}

// ---- column-parallel model + streaming coder ------------------------------------------------------------------------
// A lane that walks its stream's reads column by column with the whole model in an arena of its own (rounds 1-2; removed in round
// 5, HISTORY.md) touches a different table - a different 64-128 B line of 0.5 MB - every symbol: measured, 52 % of the wave cycles
// parked on s_waitcnt, 49 % L2 misses, ~1.1 TB fetched + 0.6 TB written for 15 G symbols; the "latency-bound" chain was random HBM
// traffic. But the adaptive model of column s is a function of the column-s symbols only (a table belongs to one column, update_stats
// touches nothing else, qv_stream.c:9-25): only the interval arithmetic is a serial chain over the whole read. Hence two kernels:
//   k_model_columns  one lane per (stream, column) - 1.2 M lanes at 8192 x 150 instead of 8192. The lane walks its
//                    stream's reads for that column, so its working set is the handful of tables of one column
//                    (L1/L2 resident) and qi/st are read along the read axis. It emits the 8-byte record
//                    (n, cum_lo, freq) of every symbol in the stream's read-major coding order.
//   k_code_records   one lane per stream: streams its records (plain sequential loads, prefetched two blocks ahead)
//                    through the interval arithmetic + renormalisation + bit output (arith.c:24-97). No tables.
// Records go through a caller-sized workspace in segments of `seg_reads` reads per stream; the coder state of a
// stream survives between segments in CoderState.
struct CoderState {
    uint32_t l, u, scale3, nbits;
    uint64_t acc, pos;
    uint32_t overflow, pad;
};

// Record of one symbol: x = cum_lo / n / 1024 and y = (n - cum_hi) / n / 1024 as correctly rounded doubles; (0, 0) codes
// nothing. The coder keeps l and ~u left-aligned in 32 bits (l32 = l << 10, nu32 = (~u & MASK) << 10), so that
// R * 1024 = 2^32 - (l32 + nu32) and
//     l  += floor(R * cum_lo / n)       = trunc(fma(R * 1024, x, 2^-24))
//     ~u += R - floor(R * cum_hi / n)   = ceil(R * (n - cum_hi) / n) = trunc(fma(R * 1024, y, 1 - 2^-24))     (arith.c:47-48)
// exactly: R <= 2^22 and n <= 2^19, so |R*x*1024 - R*cum/n| <= 2^22 * 2^-54, the fma rounds by at most 2^-31, and the
// computed value is within 2^-30 of the true quotient plus the constant; a quotient that is an integer Q lands in
// [Q + c - 2^-30, Q + c + 2^-30] and truncates to Q for either constant c, one that is not has a fractional part in
// [2^-19, 1 - 2^-19] and truncates to floor (+1 with the second constant, i.e. the ceiling). The division by n thus
// leaves the serial chain - the model lanes do it - and the all-zero record is a natural no-op (both steps add 0, the
// interval is already renormalised), so padding needs no branch.
#define QUOT_LO 5.9604644775390625e-08    /* 2^-24 */
#define QUOT_HI 0.99999994039535522461    /* 1 - 2^-24 */
typedef double2 Record;
// A record is written once by the model kernel and read once by the coder kernel, a segment (~25 ms and gigabytes) later: the
// stores carry the non-temporal hint, so the lines do not displace what the chip's caches hold for both kernels' other traffic.
// Alternating builds on one box, 100M x 150: coder phase 154.5 -> 152.4 ms with one cluster (model kernels 127.9 -> 126.4, the
// coder kernels beside them 136.0 -> 134.2), 227.7 -> 224.1 with three. The same hint on the coder kernel's LOADS of the
// records makes it 45 % slower (137 -> 198 ms): its loader asks for half lines, and the other half then comes from HBM again.
typedef double rk_d2 __attribute__((ext_vector_type(2)));
#define RK_STORE(ptr, val) do { const Record v_ = (val); rk_d2 w_; w_.x = v_.x; w_.y = v_.y; __builtin_nontemporal_store(w_, reinterpret_cast<rk_d2 *>(ptr)); } while (0)
// ---- packed records (round 4): the integer triple in 8 bytes, the division on the coder's side ------------------------------
// What the model knows about a symbol is three integers below 2^21 (n <= 2^19 + 8 * 64: a table passes the rescale threshold
// inside a chunk at most; cum_lo < n; n - cum_hi <= n): cum_lo in bits [0, 21), n - cum_hi in [21, 42), n in [42, 63);
// PREC_NOOP (n = 1, the rest 0) codes nothing. The model kernel then spends 7 integer instructions per symbol on its record instead of a
// reciprocal, two Newton steps, three conversions and two products in FP64, its transposition buffer and its stores halve, and
// so does the record traffic in both directions (HISTORY.md section 6 item 10). The two quotients are made where they are
// consumed: by the two loader waves of k_code_records<true>, which share a SIMD of their own - throughput work with eight
// independent records per block - so that nothing is added to the interval wave's chain. Same quotients to within 2^-52
// relative (v_rcp_f64 + two Newton steps on n * 2^10, which is exact in a double), the bound Record's proof needs is 2^-47.
typedef unsigned long long PRec;
#define PREC_NOOP (1ull << 42)   /* n = 1, cum_lo = 0, n - cum_hi = 0: both quotients 0, nothing is coded; n is never 0 in a record (no guard in the unpacking) */
__device__ __forceinline__ PRec pack_record(uint32_t n, uint32_t cum_lo, uint32_t fx) {
    const uint32_t nc = n - cum_lo - fx;
    const uint32_t lo = cum_lo | (nc << 21), hi = (nc >> 11) | (n << 10);
    return ((PRec)hi << 32) | lo;
}
// Four records at a time, step by step across the four (the loaders' SIMD is the busiest of a coder workgroup's: written
// record after record the compiler left every record's nine dependent instructions in a row).
// NEWTON = 1: v_rcp_f64 of gfx950 is good to 2^-24.4 over every n * 2^10 a record can hold (n = 1 .. 2^19 + 1024), one Newton step
// leaves 2^-48.8 - measured over ALL of them (tools/unpack_bench.hip, tests/test_gpu_kernels.py) and checked again once per
// process and device before the one-step kernel is used (qvz_coder_rcp_one_step_ok); the bound Record's proof needs is 2^-46
// (2^-24 of slack over quotients up to 2^22). Two steps (2^-53) otherwise.
template <int NEWTON>
__device__ __forceinline__ void unpack_records4(const uint4 &a, const uint4 &b, Record *out) {
    const uint32_t lo[4] = {a.x, a.z, b.x, b.z}, hi[4] = {a.y, a.w, b.y, b.w};
    double nd[4], rn[4], e[4], cd[4], qd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) nd[i] = (double)(hi[i] & 0xFFFFFC00u);        // n * 2^10 (bit 63 of a record is zero), exact; n >= 1
#pragma unroll
    for (int i = 0; i < 4; ++i) rn[i] = __builtin_amdgcn_rcp(nd[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) cd[i] = (double)(lo[i] & 0x1FFFFFu);
#pragma unroll
    for (int i = 0; i < 4; ++i) qd[i] = (double)(__builtin_amdgcn_alignbit(hi[i], lo[i], 21) & 0x1FFFFFu);
#pragma unroll
    for (int k = 0; k < NEWTON; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_fma(-nd[i], rn[i], 1.0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rn[i] = __builtin_fma(e[i], rn[i], rn[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = make_double2(cd[i] * rn[i], qd[i] * rn[i]);
}
// two records (one 16-byte piece): the third loader's share of a block
template <int NEWTON>
__device__ __forceinline__ void unpack_records2(const uint4 &a, Record *out) {
    const uint32_t lo[2] = {a.x, a.z}, hi[2] = {a.y, a.w};
    double nd[2], rn[2], e[2], cd[2], qd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) nd[i] = (double)(hi[i] & 0xFFFFFC00u);
#pragma unroll
    for (int i = 0; i < 2; ++i) rn[i] = __builtin_amdgcn_rcp(nd[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) cd[i] = (double)(lo[i] & 0x1FFFFFu);
#pragma unroll
    for (int i = 0; i < 2; ++i) qd[i] = (double)(__builtin_amdgcn_alignbit(hi[i], lo[i], 21) & 0x1FFFFFu);
#pragma unroll
    for (int k = 0; k < NEWTON; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) e[i] = __builtin_fma(-nd[i], rn[i], 1.0);
#pragma unroll
        for (int i = 0; i < 2; ++i) rn[i] = __builtin_fma(e[i], rn[i], rn[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) out[i] = make_double2(cd[i] * rn[i], qd[i] * rn[i]);
}
// worst |1 - n 2^10 rn| over n = 1 .. nmax for v_rcp_f64 + NEWTON steps (the residual is exact in an fma): the relative error of rn
template <int NEWTON>
__global__ __launch_bounds__(256) void k_rcp_accuracy(unsigned long long *worst, uint32_t nmax) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x + 1;
    if (n > nmax) return;
    const double nd = (double)(n << 10);
    double rn = __builtin_amdgcn_rcp(nd);
    for (int k = 0; k < NEWTON; ++k) { const double e = __builtin_fma(-nd, rn, 1.0); rn = __builtin_fma(e, rn, rn); }
    const double res = __builtin_fabs(__builtin_fma(-nd, rn, 1.0));
    atomicMax(worst, (unsigned long long)__double_as_longlong(res));   // non-negative doubles order like their bit patterns
}
// Packed-record area of one segment: units of 4 reads x 4 columns (16 records, 128 bytes: [read & 3][column & 3]), the 64
// streams' units of the same (read quad, column group) side by side - a model workgroup (4 adjacent columns, 64 consecutive
// reads) still writes whole 128-byte lines, and a lane of the coder finds the 8 symbols of a block as two 32-byte pieces of
// two neighbouring units. Record (row, col) of stream g is at
//   (((g / 64) * nunit + (row / 4) * (Cp / 4) + col / 4) * 64 + g % 64) * 16 + (row % 4) * 4 + col % 4,  nunit = seg_reads * Cp / 16.
__device__ __forceinline__ uint64_t rec8_index(uint32_t g, uint64_t row, uint32_t col, uint32_t Cp, uint64_t nunit) {
    return (((uint64_t)(g >> 6) * nunit + (row >> 2) * (Cp >> 2) + (col >> 2)) * 64 + (g & 63)) * 16 + (row & 3) * 4 + (col & 3);
}
#define RK_STORE8(ptr, val) __builtin_nontemporal_store((PRec)(val), reinterpret_cast<PRec *>(ptr))

// The record for the HOST's interval chain (strict single-stream mode, qvz_coder_encode_records): two 64-bit integers in
// the record's 16 bytes, X(c) = floor(2^62 * c / n) + 1 for c = cum_lo and c = cum_hi. With rd = (u - l + 1) * 1024 <= 2^32
//     floor((u - l + 1) * c / n) = (rd * X(c)) >> 72                                                  (arith.c:47-48)
// exactly: rd * X / 2^72 = R * X / 2^62 lies in (R c / n, R c / n + 2^-40] (R <= 2^22), and R c / n is an integer or at
// least 1 / n >= 2^-19.1 below the next one. One 64 x 64 -> 128 bit multiply per bound on the host instead of a
// conversion, a multiply-add and a conversion back: the host's dependent chain per symbol is what the strict mode's
// speed is. The quotient of the 82-bit numerator comes from two 64-bit divisions (long division in base 2^31; the
// remainder of the first is below n < 2^20). This kernel variant has the GPU to itself (one stream: a few dozen
// workgroups), the divisions cost nothing that matters.
__device__ __forceinline__ uint64_t frac62(uint32_t c, uint32_t n) {
    const uint64_t a = (uint64_t)c << 31;
    const uint64_t hi = a / n, rem = a - hi * n;
    const uint64_t lo = (rem << 31) / n;
    return ((hi << 31) | lo) + 1;                                // hi < 2^32 (c <= n), lo < 2^31
}
__device__ __forceinline__ Record make_record_fixed(uint32_t n, uint32_t cum_lo, uint32_t fx) {
    return make_double2(__longlong_as_double((long long)frac62(cum_lo, n)), __longlong_as_double((long long)frac62(cum_lo + fx, n)));
}

// Record area of one segment. A read has Cp = Cx rounded up to 8 records (the pad records are zero = no-ops), streams
// go in groups of 64 (one coder workgroup). A stream's records are cut into units of 2 reads x 4 columns (8 records,
// 128 bytes: [read parity][column & 3]) and the 64 streams' units of the same (read pair, column group) lie side by
// side: record (row, col) of stream g is at
//   (((g / 64) * nblk8 + (row / 2) * (Cp / 4) + col / 4) * 64 + g % 64) * 8 + (row % 2) * 4 + col % 4,  nblk8 = seg_reads * Cp / 8.
// A model workgroup (4 adjacent columns, 64 consecutive reads) therefore writes whole 128-byte lines - two reads' four
// records - and the coder's loader still finds a lane's 8 symbols of a block in two 64-byte halves of two units whose
// other halves it reads a read later (out of L2). (With 8 columns of one read per unit the model kernel wrote half
// lines, and those writes slowed the coder's reads beside them.)
#define QVZ_CID_SLOTS 8   // several clusters: record positions 0 (cluster id) .. 7 (zero records) precede the columns (k_model_ranks)

template <bool HAVE_LABELS>
__global__ __launch_bounds__(256) void k_model_columns(const uint8_t *__restrict__ qi, const uint8_t *__restrict__ st, uint64_t ld,
                                                       const uint8_t *__restrict__ labels, uint32_t C, uint32_t K,
                                                       const uint64_t *__restrict__ first, uint32_t shards,
                                                       const uint32_t *__restrict__ off, uint32_t *__restrict__ arena,
                                                       uint32_t arena_words, uint64_t seg_lo, uint32_t seg_reads,
                                                       Record *__restrict__ rec, int32_t *__restrict__ status) {
    const uint32_t Cw = C + (HAVE_LABELS ? 1u : 0u);             // lanes per stream: [cluster id,] column 0 .. C-1
    const uint32_t Cx = C + (HAVE_LABELS ? (uint32_t)QVZ_CID_SLOTS : 0u);   // record positions per read (see k_model_ranks)
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (uint64_t)shards * Cw) return;
    const uint32_t g = (uint32_t)(id / Cw), sx = (uint32_t)(id % Cw);
    const uint32_t pos = !HAVE_LABELS ? sx : sx == 0 ? 0u : sx - 1 + QVZ_CID_SLOTS;
    const uint64_t r0 = first[g] + seg_lo, rend = first[g + 1];  // r0 is a multiple of 4
    if (r0 >= rend) return;
    const uint64_t r1 = rend < r0 + seg_reads ? rend : r0 + seg_reads;
    uint32_t *ar = arena + (uint64_t)g * arena_words;
    const uint32_t Cp = (Cx + 7) & ~7u;
    const uint64_t nunit = (uint64_t)seg_reads * Cp / 16;
    PRec *const rec8 = reinterpret_cast<PRec *>(rec);
    const bool cluster_lane = HAVE_LABELS && sx == 0;
    const uint32_t s = HAVE_LABELS ? sx - 1 : sx;
    bool bad = false;
    for (uint64_t r = r0; r < r1; r += 4) {
        uint32_t qw = 0, sw = 0;
        if (!cluster_lane) {
            qw = *reinterpret_cast<const uint32_t *>(qi + (uint64_t)s * ld + r);   // 4 reads of this column
            sw = *reinterpret_cast<const uint32_t *>(st + (uint64_t)s * ld + r);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (r + j >= r1) break;
            uint32_t cl = 0;
            if (HAVE_LABELS) {
                const uint32_t raw = labels[r + j];
                if (raw >= K) bad = true; else cl = raw;
            }
            uint32_t o, x;
            if (cluster_lane) {                                  // qv_write_cluster: table 0 of the arena
                o = 0; x = cl;
            } else {
                const uint32_t q = (qw >> (8 * j)) & 0xFF;
                x = (sw >> (8 * j)) & 0xFF;
                o = q < 2 * A ? off[((uint64_t)cl * C + s) * 2 * A + q] : NO_TABLE;
            }
            Record w = make_double2(0.0, 0.0);                   // nothing to code (bad feed, reported below)
            PRec w8 = PREC_NOOP;
            if (o == NO_TABLE) {
                bad = true;
            } else {
                uint32_t *t = ar + o;
                const uint4 ha = reinterpret_cast<const uint4 *>(t)[0];            // n, card, c0, c1
                const uint32_t n = ha.x, card = ha.y;
                if (x >= card) {
                    bad = true;
                } else {
                    uint32_t cum_lo, fx;
                    if (x < 2) {
                        cum_lo = x ? ha.z : 0u;
                        fx = x ? ha.w : ha.z;
                    } else {
                        const uint4 hb = reinterpret_cast<const uint4 *>(t)[1];    // c2 .. c5 (may run past a short table: unused then)
                        if (x < 6) {
                            cum_lo = ha.z + ha.w + (x > 2 ? hb.x : 0u) + (x > 3 ? hb.y : 0u) + (x > 4 ? hb.z : 0u);
                            fx = x == 2 ? hb.x : x == 3 ? hb.y : x == 4 ? hb.z : hb.w;
                        } else {
                            cum_lo = ha.z + ha.w + hb.x + hb.y + hb.z + hb.w;
                            for (uint32_t i = 6; i < x; ++i) cum_lo += t[2 + i];
                            fx = t[2 + x];
                        }
                    }
                    t[2 + x] = fx + STEP;                        // update_stats, qv_stream.c:12-24
                    uint32_t nn = n + STEP;
                    if (nn > RESCALE) {
                        nn = 0;
                        for (uint32_t i = 0; i < card; ++i) {
                            const uint32_t v = t[2 + i];
                            if (v) { const uint32_t hv = (v >> 1) + 1; t[2 + i] = hv; nn += hv; }
                        }
                    }
                    t[0] = nn;
                    w8 = pack_record(n, cum_lo, fx);
                }
            }
            rec8[rec8_index(g, r + j - r0, pos, Cp, nunit)] = w8;
            if (cluster_lane)                                    // the reserved positions behind the cluster id
                for (uint32_t pc = 1; pc < QVZ_CID_SLOTS; ++pc) rec8[rec8_index(g, r + j - r0, pc, Cp, nunit)] = PREC_NOOP;
            if (sx + 1 == Cw)                                    // the read's pad records
                for (uint32_t pc = Cx; pc < Cp; ++pc) rec8[rec8_index(g, r + j - r0, pc, Cp, nunit)] = PREC_NOOP;
        }
    }
    if (bad) atomicOr(status, 1);
}

// encoder_last_step (arith.c:99-116) + stream_finish_byte (os_stream.c:105-110); returns the stream length
__device__ __forceinline__ uint64_t lc_finish(LaneCoder &c) {
    const uint32_t bb = c.l >> (M_BITS - 1);
    lc_put(c, bb, 1);
    if (c.scale3) { lc_run(c, !bb, c.scale3); c.scale3 = 0; }
    lc_put(c, c.l & MASK_M1, M_BITS - 1);
    uint64_t total_bits = c.pos * 8 + c.nbits;
    if (total_bits % 8) { const uint32_t pad = 8 - (uint32_t)(total_bits % 8); lc_put(c, 0, pad); total_bits += pad; }
    else { lc_put(c, 0, 8); total_bits += 8; }                   // aligned streams still get one 0x00 byte
    while (c.nbits) {                                            // spill the remaining whole bytes
        const uint32_t take = c.nbits >= 8 ? 8 : c.nbits;
        const uint8_t byte = (uint8_t)((c.acc >> (c.nbits - take)) << (8 - take));
        if (c.pos < c.cap) c.out[c.pos] = byte; else c.overflow = true;
        c.pos += 1;
        c.nbits -= take;
    }
    return total_bits / 8;
}

// ---- the model as prefix counts: one WAVE per (stream, column), one lane per read ---------------------------------
// k_model_columns above still walks reads serially in every lane and keeps its tables in HBM/L2; measured, it is
// bound by that traffic. But between two rescales a table is nothing but counts: after e earlier hits of which E
// had the same state and L a smaller one,  n = n0 + 8e,  freq = c0_x + 8E,  cum_lo = cum0_x + 8L  (qv_stream.c:9-14).
// So a wave takes 64 consecutive reads of its column at once: the counts before the chunk come from the column's
// tables in LDS, the in-chunk ranks (e, E, L) from lane bit-masks: every lane ORs its bit into a 64-bit mask beside
// its table head and beside its count (ds_or_b64: commutative, so no ordering is assumed), and "earlier lanes with
// ..." is v_mbcnt of such a mask. (A first version looped over the distinct (table, state) pairs of the chunk with
// ballots: ~35 pairs per chunk on the 100M x 150 workload, 675 VALU instructions per chunk; this one needs ~100.)
// The tables are then brought up to date with LDS atomic adds. A chunk in which some table would pass the
// rescale threshold (n > 2^19, once per 65536 hits of a table) is replayed serially, symbol by symbol, with the
// reference's halve-and-+1 rescale. Records leave through an LDS transposition so that the workgroup (W waves = W
// adjacent columns of the same stream) writes W * 16 contiguous bytes per read of the stream's read-major record area.
// Table state crosses segment boundaries through `save` (one slice per (stream, column)).
#define RK_NONE 0xFFFFu
#ifdef QVZ_PROFILE          // development build (make HIPFLAGS+=-DQVZ_PROFILE): shader-clock cycles per section of the chunk loop, summed over waves
__device__ unsigned long long g_rk_prof[16];   // [7] prologue, [8] table save, [9] passes, [10] whole pass, [11] .. [13] prologue: to the tables in LDS, the barrier, the first look-up; [14] first passes' prologue
__device__ unsigned long long g_cr_prof[16];
__device__ int g_rk_drain;                     // QVZ_RK_DRAIN=1: a pass drains what is in flight before it asks for its tables, timed
__device__ uint32_t g_cr_simd[8];   // SIMD of every wave of one k_code_records workgroup (HW_ID bits 5:4)
#define RK_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define RK_T(i) do { } while (0)
#endif
// Read lists per (stream, cluster) for k_model_ranks<true>: list[cstart[g*K + l] + j] = number (relative to the stream's
// first read) of the j-th read of stream g that belongs to cluster l. One workgroup of four waves per stream; wave w owns
// the w-th quarter of the stream's reads, pass 1 counts, pass 2 writes in ascending order (a chunk of 64 reads is taken
// cluster by cluster: ballot of the first remaining lane's label).
__global__ __launch_bounds__(256) void k_cluster_lists(const uint8_t *__restrict__ labels, const uint64_t *__restrict__ first,
                                                       uint32_t K, uint32_t *__restrict__ clist, uint64_t *__restrict__ cstart,
                                                       int32_t *__restrict__ status) {
    __shared__ uint32_t cntw[4][256], basew[4][256];
    const uint32_t g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t f0 = first[g], f1 = first[g + 1];
    const uint64_t len = f1 - f0, per = ((len + 3) / 4 + 63) & ~(uint64_t)63;
    const uint64_t lo = per * wave < len ? per * wave : len, hi = lo + per < len ? lo + per : len;
    for (uint32_t i = threadIdx.x; i < 4 * 256; i += 256) (&cntw[0][0])[i] = 0;
    __syncthreads();
    bool bad = false;
    for (uint64_t j = lo + lane; j < hi; j += 64) {
        const uint32_t l = labels[f0 + j];
        if (l < K) atomicAdd(&cntw[wave][l], 1u); else bad = true;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t l = 0; l < K; ++l) {
            cstart[(uint64_t)g * K + l] = f0 + run;
            for (uint32_t w = 0; w < 4; ++w) { basew[w][l] = run; run += cntw[w][l]; }
        }
    }
    __syncthreads();
    for (uint64_t j0 = lo; j0 < hi; j0 += 64) {
        const uint64_t j = j0 + lane;
        const bool valid = j < hi;
        const uint32_t l = valid ? labels[f0 + j] : 0xFFFFFFFFu;
        unsigned long long rest = __ballot(valid && l < K);
        while (rest) {
            const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)l, __builtin_ctzll(rest));
            const unsigned long long m = __ballot(l == lead);
            const uint32_t base = __hip_atomic_load(&basew[wave][lead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // wave-uniform
            if (l == lead) clist[f0 + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)j;
            if (lane == 0) __hip_atomic_store(&basew[wave][lead], base + (uint32_t)__builtin_popcountll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            rest &= ~m;
        }
    }
    if (bad) atomicOr(status, 1);
}

// segpos[i * S*K + g*K + l] = number of entries of list (g, l) below segment boundary i (bounds[i] reads into the stream)
__global__ __launch_bounds__(256) void k_cluster_segpos(const uint32_t *__restrict__ clist, const uint64_t *__restrict__ cstart,
                                                        const uint64_t *__restrict__ first, uint32_t shards, uint32_t K,
                                                        const uint32_t *__restrict__ bounds, uint32_t nbounds,
                                                        uint32_t *__restrict__ segpos) {
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t SK = (uint64_t)shards * K;
    if (id >= SK * nbounds) return;
    const uint32_t i = (uint32_t)(id / SK), gl = (uint32_t)(id % SK), g = gl / K, l = gl % K;
    const uint64_t a = cstart[gl], e = l + 1 < K ? cstart[gl + 1] : first[g + 1];   // the lists of a stream lie back to back
    const uint32_t *lst = clist + a;
    uint32_t lo = 0, hi = (uint32_t)(e - a);
    const uint32_t b = bounds[i];
    while (lo < hi) {                                            // first entry >= b
        const uint32_t mid = (lo + hi) / 2;
        if (lst[mid] < b) lo = mid + 1; else hi = mid;
    }
    segpos[id] = lo;
}

// With several clusters (HAVE_LABELS) a read's record group is [cluster id, 7 zero records, column 0 .. C-1, zero pad]
// (QVZ_CID_SLOTS reserved positions, so that no workgroup mixes the two kinds of work below), and a workgroup of the
// column positions serves ONE cluster (blockIdx.z): it walks the list of that cluster's reads in the stream
// (k_cluster_lists) 64 at a time, so a wave needs the tables of one cluster only - the LDS footprint, and with it the
// number of waves a CU holds, is that of the one-cluster case - and every lane of a chunk has a symbol. The workgroups
// of the reserved positions (cluster z = 0 only) take the reads in file order: the cluster id of every read is a symbol
// of its own table (qv_compressor.c:86), the other reserved positions are zero records.
template <bool HAVE_LABELS, bool DENSE, bool REC8>
__global__ __launch_bounds__(1024) void k_model_ranks(const uint8_t *__restrict__ qi, const uint8_t *__restrict__ st, uint64_t ld,
                                                      const uint8_t *__restrict__ labels, uint32_t C, uint32_t K,
                                                      const uint64_t *__restrict__ first, const uint32_t *__restrict__ coff,
                                                      const uint32_t *__restrict__ colbase, const uint32_t *__restrict__ tmpl,
                                                      uint32_t *__restrict__ save, uint32_t total_words, uint32_t slice_cap,
                                                      uint64_t seg_lo, uint32_t seg_reads, int is_first, int is_last,
                                                      uint32_t log2w, Record *__restrict__ rec, int32_t *__restrict__ status,
                                                      const uint32_t *__restrict__ clist, const uint64_t *__restrict__ cstart,
                                                      const uint32_t *__restrict__ segpos_lo, const uint32_t *__restrict__ segpos_hi,
                                                      const uint32_t *__restrict__ cgmap) {
    // records between consecutive units of a stream: 512 = 8 records x 64 interleaved streams (rec_index, the coder
    // kernel's layout); DENSE: one stream alone, its units back to back (qvz_hip_model_session_records)
    // REC8: packed 8-byte records, units of 16 (rec8_index): 1024 records between consecutive units of a stream
    static_assert(DENSE != REC8, "packed records for the GPU coder, 16-byte fixed-point records for the host's chain (DENSE)");
    constexpr uint32_t unit_recs = DENSE ? 8u : REC8 ? 1024u : 512u;
    extern __shared__ uint4 rk_smem[];
    // the wave index is wave-uniform, which the compiler cannot see: readfirstlane keeps everything derived from it
    // (column, table slice, activity) in scalar registers and its tests in scalar branches
    const uint32_t W = 1u << log2w, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t Cx = C + (HAVE_LABELS ? (uint32_t)QVZ_CID_SLOTS : 0u);
    // Several clusters: the K workgroups of one (stream, column group) read the same lines of the qi / st planes (each
    // walks its own cluster's reads, which interleave). In grid order they would run a third of the launch apart and
    // each fetch the lines from HBM again (PMC: 160 GB fetched against 52 with one cluster). The linear workgroup number
    // is re-read so that they are dispatched 8 apart: close in time and - workgroups go round-robin over the 8 XCDs -
    // behind the same L2.
    uint32_t g = blockIdx.x, cg = blockIdx.y, z = HAVE_LABELS ? blockIdx.z : 0u;
    if (HAVE_LABELS && (gridDim.x & 7) == 0) {
        const uint64_t lin = blockIdx.x + (uint64_t)gridDim.x * (blockIdx.y + (uint64_t)gridDim.y * blockIdx.z);
        const uint32_t span = 8 * gridDim.z, within = (uint32_t)(lin % span);
        const uint64_t pair = lin / span * 8 + (within & 7);     // (stream, column group) number
        z = within >> 3;
        g = (uint32_t)(pair % gridDim.x);
        cg = (uint32_t)(pair / gridDim.x);
    }
    // cgmap: this launch serves the column groups whose tables fit the LDS it was given (the host's size classes, see
    // qvz_hip_encode_planes). Entry i of the launch: the column group, then per wave (where its part of the LDS starts, in words) |
    // (words of its largest slice) << 16 - a wave has room for ITS column's tables, not for the largest column's.
    uint32_t wave_off = wave * (3 * slice_cap + 2 * A), wave_cap = slice_cap;   // (no map: every wave has room for the largest slice)
    if (cgmap) {
        const uint32_t *ent = cgmap + (size_t)cg * (1 + W);
        cg = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[0]);
        const uint32_t lay = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[1 + wave]);
        wave_off = lay & 0xFFFFu;
        wave_cap = lay >> 16;
    }
    const uint32_t sx = cg * W + wave;                           // position in the read's record group
    const bool lists = HAVE_LABELS && cg * W >= QVZ_CID_SLOTS;   // this workgroup walks cluster z's reads
    if (HAVE_LABELS && !lists && z) return;                      // the reserved positions belong to the z = 0 workgroups
    const uint32_t Cp = (Cx + 7) & ~7u;
    const uint64_t nblk8 = REC8 ? (uint64_t)seg_reads * Cp / 16 : (uint64_t)seg_reads * Cp / 8;   // units of a stream in the segment
    auto uniform64 = [](uint64_t v) -> uint64_t {               // same value in every lane: keep it in scalar registers
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t f0 = uniform64(first[g]);
    const uint64_t r0 = f0 + seg_lo, rend = uniform64(first[g + 1]);
    if (r0 >= rend) return;                                      // whole workgroup
    bool bad = false;
    // Several clusters, gridDim.z == 1: the workgroup of a (stream, column group) serves the clusters one after the other (tables
    // in, the cluster's read list walked, tables out) instead of one workgroup per cluster: a third as many workgroups to dispatch
    // and start per segment, the same lines of the planes out of the same L2.
    const uint32_t nz = HAVE_LABELS && lists && gridDim.z == 1 ? K : 1u;
    // ... and what a pass needs before its first chunk is asked for during the pass before it (PF: the instantiations that walk
    // lists): the list entries of its first eight chunks at that pass's start, the bytes of its first four chunks and its tables
    // at the first `park` of that pass whose own request would lie beyond the pass's reads - into the very registers the request
    // would have filled (the bytes). Measured before this (cycle counters of a
    // -DQVZ_PROFILE build, three clusters at 100M x 150): a pass waited 10 100 cycles - two dependent memory round trips: list
    // entries, then the bytes of those reads - of its 79 300 before its first chunk.
    constexpr bool PF = HAVE_LABELS && REC8 && !DENSE;
    typedef uint32_t rk_u32x4 __attribute__((ext_vector_type(4)));
    uint32_t q_nx[4], x_nx[4];                                   // the bytes requested for the four chunks after the current four
    uint32_t r_pfa[4] = {0, 0, 0, 0}, r_pfb[4] = {0, 0, 0, 0};   // PF: list entries of the next pass's chunks 0-3 / 4-7
    bool pf_lists = false, pf_rest = false;                      // (wave-uniform) what has been asked for on behalf of the next pass
    for (uint32_t zi = 0; zi < nz; ++zi) {
    if (zi) __syncthreads();                                     // the pass before has drained its last chunk (recbuf, row slots)
#ifdef QVZ_PROFILE
    const unsigned long long tpass = __builtin_readcyclecounter();
#endif
    if (nz > 1) z = zi;
    const bool from_lists = PF && pf_lists, from_rest = PF && pf_rest;   // what the pass before has asked for on behalf of this one
    pf_lists = pf_rest = false;
    uint32_t nr = rend - r0 < seg_reads ? (uint32_t)(rend - r0) : seg_reads;   // reads of this stream in this segment
    const uint32_t *lp = nullptr;                                // list entries of this segment (stream-relative read numbers)
    if (lists) {
        const uint32_t j0 = segpos_lo[g * K + z], j1 = segpos_hi[g * K + z];
        nr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(j1 - j0));
        lp = clist + uniform64(cstart[g * K + z]) + j0;
    }
    const uint32_t rb_stride = W + 1;                            // +1: conflict-free transposition
    Record *recbuf = reinterpret_cast<Record *>(rk_smem);        // [64][W + 1]
    // REC8: the transposition buffer holds packed records, [64][W + 1] of 8 bytes (the odd stride keeps the 8-byte accesses
    // of 32 lanes on 64 different banks), and the list rows below have 1 KB of their own behind it - 2.5 + 1 KB where the
    // 16-byte records take 5
    PRec *recbuf8 = reinterpret_cast<PRec *>(rk_smem);
    uint32_t *rows8 = reinterpret_cast<uint32_t *>(recbuf8 + 64 * rb_stride);   // [64][4], several clusters only
    // lists: the segment rows of the reads of a block of four chunks live in the padding record of the read's row of
    // recbuf (16 bytes = four row numbers per lane): no LDS of their own - with three clusters on 100M x 150 that is what
    // lets a third workgroup fit a CU
    auto rowslot = [&](uint32_t rd, uint32_t j) -> uint32_t * {
        return REC8 ? rows8 + rd * 4 + j : reinterpret_cast<uint32_t *>(recbuf + rd * rb_stride + W) + j;
    };
    // per wave: lane masks [slice_cap] u64, tables [slice_cap] u32, and per quantizer of the column its table's
    // offset in the slice | alphabet size << 16 (RK_NONE, size 0: no table), so that one LDS read resolves a symbol
    unsigned long long *msk = (REC8 ? reinterpret_cast<unsigned long long *>(rows8 + (HAVE_LABELS ? 256 : 0))
                                    : reinterpret_cast<unsigned long long *>(recbuf + 64 * rb_stride)) + wave_off / 2;
    uint32_t *tab = reinterpret_cast<uint32_t *>(msk + wave_cap);
    uint32_t *cof = tab + wave_cap;
    const bool cluster_wave = HAVE_LABELS && sx == 0;
    const bool active = HAVE_LABELS ? (cluster_wave || (lists && sx < Cx)) : sx < Cx;
    const uint32_t s = HAVE_LABELS ? sx - QVZ_CID_SLOTS : sx;
    const uint32_t slice = !HAVE_LABELS ? sx : cluster_wave ? 0u : 1u + s * K + z;   // tables of (column, cluster)
    uint32_t b0 = 0, sw = 0;
    if (active) {
        b0 = colbase[slice];
        sw = colbase[slice + 1] - b0;
    }
    if (nr == 0) {                                               // whole workgroup (lists only): no read of this cluster here, the state passes through
        if (active && is_first && !is_last) {
            const uint4 *src = reinterpret_cast<const uint4 *>(tmpl + b0);
            for (uint32_t i = lane; i < sw / 4; i += 64) reinterpret_cast<uint4 *>(save + (uint64_t)g * total_words + b0)[i] = src[i];
        }
        continue;
    }

    Record *const rec_g = rec + (((uint64_t)(g >> 6) * nblk8) * 64 + (g & 63)) * 8;   // record 0 of this stream (rec_index; g = 0 when unit_recs = 8)
    PRec *const rec8_g = reinterpret_cast<PRec *>(rec) + (((uint64_t)(g >> 6) * nblk8) * 64 + (g & 63)) * 16;   // REC8: the same by rec8_index
    uint32_t chunk = 0;
    // Input: every lane loads the bytes of its own read for the next four chunks (a wave instruction = 64 contiguous bytes
    // of a plane) and keeps them in registers until the current four are used up, so the HBM latency stays off the chunk
    // loop and nothing is staged through LDS. No bounds branch: a read index past the stream's last read is clamped to
    // it, idle waves read plane 0, and `on` below ignores what such lanes loaded.
    uint32_t q_cur = 0, x_cur = 0, lab_cur = 0, lab_nx[4] = {0, 0, 0, 0};
    uint32_t r_cur[4] = {0, 0, 0, 0}, r_nx[4] = {0, 0, 0, 0}, r_nn[4] = {0, 0, 0, 0};   // lists: stream-relative read numbers
    const uint8_t *const qplane = qi + (uint64_t)(active && !cluster_wave ? s : 0u) * ld + (lists ? f0 : r0);
    const uint8_t *const xplane = st + (uint64_t)(active && !cluster_wave ? s : 0u) * ld + (lists ? f0 : r0);
    auto fetch_list = [&](uint32_t ro) {                         // lists: the read numbers of chunks ro/64 .. ro/64 + 3
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t e = ro + j * 64 + lane;
            r_nn[j] = lp[e < nr ? e : nr - 1];
        }
    };
    uint32_t r_n2[4] = {0, 0, 0, 0};                             // the second batch of list entries of a workgroup's start (both are asked for at once)
    auto fetch_list_second = [&](uint32_t ro) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t e = ro + j * 64 + lane;
            r_n2[j] = lp[e < nr ? e : nr - 1];
        }
    };
    auto fetch = [&](uint32_t ro) {                              // ro = read (or list entry) offset in the segment
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t r = ro + j * 64 + lane, rc = lists ? r_nx[j] : (r < nr ? r : nr - 1);
            q_nx[j] = qplane[rc];
            x_nx[j] = xplane[rc];
            if (HAVE_LABELS && cluster_wave) lab_nx[j] = labels[r0 + rc];
        }
    };
    // Table look-up of a chunk: one LDS read (table offset and alphabet size) that does not depend on the counts; it is
    // issued one chunk ahead. Written without per-lane branches (selects and safe indices instead): every exec-mask
    // change costs a wave several scalar instructions and their latencies.
    uint32_t tb_nx = 0, x_lk = 0, card_nx = 1;
    bool ok_nx = false;
    auto lookup = [&](uint32_t ro, uint32_t ch) {
        const bool on = active && ro + lane < nr;
        const uint32_t sh = (ch & 3) * 8;
        const uint32_t q = (q_cur >> sh) & 0xFF, xs = (x_cur >> sh) & 0xFF;
        const uint32_t raw = HAVE_LABELS && cluster_wave ? (lab_cur >> sh) & 0xFF : 0u;
        const uint32_t cl = raw < K ? raw : 0u;
        const uint32_t ent = cof[q < 2 * A ? q : 0u];
        const uint32_t tb = cluster_wave ? 0u : (q < 2 * A ? ent & 0xFFFFu : (uint32_t)RK_NONE);
        const uint32_t card = cluster_wave ? K : ent >> 16;
        const uint32_t x = cluster_wave ? cl : xs;
        bool ok = on && tb != RK_NONE && raw < K;
        const uint32_t tbs = ok ? tb : 0u;                       // lanes without a symbol work on table 0 and contribute nothing
        ok = ok && x < card;
        bad = bad || (on && !ok);
        tb_nx = tbs; x_lk = ok ? x : 0u; card_nx = card; ok_nx = ok;
    };
    // PF: the pass after this one - its reads, its tables
    uint32_t nrN = 0;
    const uint32_t *lpN = nullptr;
    auto park = [&](uint32_t ro) {                               // the requested bytes become the current block; request the next
        q_cur = q_nx[0] | q_nx[1] << 8 | q_nx[2] << 16 | q_nx[3] << 24;
        x_cur = x_nx[0] | x_nx[1] << 8 | x_nx[2] << 16 | x_nx[3] << 24;
        if (HAVE_LABELS && cluster_wave) lab_cur = lab_nx[0] | lab_nx[1] << 8 | lab_nx[2] << 16 | lab_nx[3] << 24;
        if (lists) {                                             // read numbers: current <- next <- the list entries requested a block ago
#pragma unroll
            for (int j = 0; j < 4; ++j) { r_cur[j] = r_nx[j]; r_nx[j] = r_nn[j]; }
        }
        if (PF && pf_lists && ro + 256 >= nr) {
            // nothing of this pass is left to ask for: the first bytes of the next pass instead (its list entries were asked for when
            // this pass began). No `park` follows in this pass (the chunk loop skips them: q_nx / x_nx stay).
            // (The tables: there are no registers to hold a slice through the chunks that are left - the loop's 115 and these 16
            // spill -, and a load that only pulled them into L2 - one word of every line, an atomic load of wavefront scope, which is
            // a plain global_load_dword that the compiler neither moves nor drops and whose return it counts itself - changed
            // nothing: a later pass waits ~7 700 cycles for its tables like a workgroup's first.)
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                q_nx[j] = qplane[r_pfa[j]];
                x_nx[j] = xplane[r_pfa[j]];
            }
            pf_rest = true;
            return;
        }
        if (lists) fetch_list(ro + 512);
        fetch(ro + 256);
    };
    // Before the first chunk a workgroup needs its tables in LDS and the bytes of its first reads in registers: the list entries
    // (several clusters) and the input bytes are asked for first, the tables next, and everything is in flight together - one
    // memory round trip where the order tables, barrier, list entries, bytes made three or four. A workgroup lives for some twenty
    // chunks, so what it does before the first one counts.
    if (lists) {
        if (!from_lists) {
            fetch_list(0);
            fetch_list_second(256);
        }
    } else {
        fetch(0);
    }
#ifdef QVZ_PROFILE          // how long what the pass before left in flight (its records, its table save) takes to drain: the tables' loads queue behind it
    unsigned long long tdrain = 0;
    if (g_rk_drain) {
        const unsigned long long td0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tdrain = __builtin_readcyclecounter() - td0;
    }
#endif
    if (active) {
        // the slice comes in as 16-byte loads issued back to back, the offset table is ready-made on the host (no dependent
        // second load). (Written as `for (i ..) tab[i] = src[i]` the compiler made a loop of load, s_waitcnt vmcnt(0), store:
        // one memory round trip per KB of tables and three more for the offsets, one after the other. Unconditional loads -
        // the index clamped instead of the lane masked -, issued by hand in batches of four per lane, the offsets first:
        // model kernels 130.7 -> 127.9 ms at 100M x 150 with one cluster, 228 -> 219 with three, 18.0 -> 16.2 at 10M x 150.)
        const uint4 *src = reinterpret_cast<const uint4 *>(is_first ? tmpl + b0 : save + (uint64_t)g * total_words + b0);
        const uint32_t n4 = sw / 4;
        uint32_t cv[(2 * A + 63) / 64];
        if (!cluster_wave) {
#pragma unroll
            for (uint32_t u = 0; u < (2 * A + 63) / 64; ++u) {
                const uint32_t i = lane + 64 * u;
                cv[u] = coff[(uint64_t)slice * 2 * A + (i < 2 * A ? i : 2 * A - 1)];
            }
        }
        for (uint32_t i0 = 0; i0 < n4; i0 += 256) {
            rk_u32x4 v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {                 // issued by hand: the compiler sinks a plain load to its store and waits there
                const uint32_t i = i0 + lane + 64 * u;
                const uint4 *pa = src + (i < n4 ? i : n4 - 1);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u]) : "v"(pa) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t i = i0 + lane + 64 * u;
                if (i < n4) {
                    reinterpret_cast<uint4 *>(tab)[i] = make_uint4(v[u].x, v[u].y, v[u].z, v[u].w);
                    reinterpret_cast<uint4 *>(msk)[2 * i] = make_uint4(0u, 0u, 0u, 0u);
                    reinterpret_cast<uint4 *>(msk)[2 * i + 1] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        if (!cluster_wave) {
#pragma unroll
            for (uint32_t u = 0; u < (2 * A + 63) / 64; ++u) {
                const uint32_t i = lane + 64 * u;
                if (i < 2 * A) cof[i] = cv[u];
            }
        }
    }
    if (lists) {                                                 // the list entries have landed (the tables' wait): now the bytes of those reads
#pragma unroll
        for (int j = 0; j < 4; ++j) r_nx[j] = from_lists ? r_pfa[j] : r_nn[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) r_nn[j] = from_lists ? r_pfb[j] : r_n2[j];
        if (!from_rest) fetch(0);                                // (from_rest: q_nx / x_nx hold them)
    }
    if (PF && nz > 1 && zi + 1 < nz) {
        // the next pass: its first list entries are asked for now
        const uint32_t zN = zi + 1;
        const uint32_t j0 = segpos_lo[g * K + zN], j1 = segpos_hi[g * K + zN];
        nrN = (uint32_t)__builtin_amdgcn_readfirstlane((int)(j1 - j0));
        lpN = clist + uniform64(cstart[g * K + zN]) + j0;
        if (nrN) {                                               // (a next pass without a read passes its tables through by itself)
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t e = j * 64 + lane, e2 = 256 + e;
                r_pfa[j] = lpN[e < nrN ? e : nrN - 1];
                r_pfb[j] = lpN[e2 < nrN ? e2 : nrN - 1];
            }
            pf_lists = true;
        }
    }
#ifdef QVZ_PROFILE
    const unsigned long long tp1 = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef QVZ_PROFILE
    const unsigned long long tp2 = __builtin_readcyclecounter();
#endif
    park(0);
    lookup(0, 0);
#ifdef QVZ_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    const unsigned long long tloop = tprev;
#endif
    for (uint32_t rb = 0; rb < nr; rb += 64, ++chunk) {       // rb = read offset in the segment
        Record outr = make_double2(0.0, 0.0);
        PRec outp = PREC_NOOP;
        const uint32_t tb = tb_nx, x = x_lk, card = card_nx;
        const bool ok = ok_nx;
        // where in the segment the reads of this block of four chunks lie. Everybody has drained the chunk before the last
        // (barrier 1 of the last one), so the first three row numbers can be replaced here; the fourth, which the last
        // chunk's drain may still be reading, after this chunk's first barrier
        const bool new_rows = lists && wave == 0 && (chunk & 3) == 0;
        if (new_rows) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *rowslot(lane, j) = r_cur[j] - (uint32_t)seg_lo;
        }
        if (rb + 64 < nr) lookup(rb + 64, chunk + 1);
        RK_T(0);
        if (active) {
            // every lane marks its table and its (table, state) entry (lanes without a symbol OR in nothing). An LDS atomic costs
            // two cycles for every lane of the instruction that hits the SAME address (tools/lds_bench.hip: 9 cycles for 64
            // different addresses, 129 for one), and most lanes of a chunk share one or two tables - but the masks need the OR.
            // (Tried: the table mask as four 16-bit pieces in the dwords of msk[tb], msk[tb + 1], one per quarter of the wave,
            // and the (table, state) mask ORed into by halves - a quarter and a half of the lanes per address: the ten extra
            // instructions cost more than the LDS cycles gave back, model kernels 130.8 -> 136 ms at 100M x 150 -f 0.5.)
            {
                const unsigned long long bit = ok ? 1ull << lane : 0ull;
                __hip_atomic_fetch_or(&msk[tb], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(&msk[tb + 2 + x], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // counts before this chunk, and ranks inside it: earlier lanes with the same table (e), the same table and
            // state (E), a smaller state (L). cum_lo and L come from whichever end of the table is nearer
            // (sum over y < x directly, or n - c_x - sum over y > x).
            uint32_t cum_b = 0, L = 0;
            auto earlier = [](unsigned long long m, uint32_t add) -> uint32_t {     // add + popcount(m & lanes below me)
                return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, add));
            };
            const uint32_t n_b = tab[tb], c_x = tab[tb + 2 + x];
            const unsigned long long mt = msk[tb];
            const unsigned long long mx = msk[tb + 2 + x];
            const uint32_t e = earlier(mt, 0u), hits = (uint32_t)__builtin_popcountll(mt);
            const uint32_t E = earlier(mx, 0u), same = (uint32_t)__builtin_popcountll(mx);
            {
                const bool up = 2 * x < card;
                uint32_t y = up ? 0u : x + 1, yend = up ? x : card, acc = 0, accm = 0;
                if (!ok) yend = y;
                // four entries per LDS round trip; all four are loaded whatever yend is (a read past the table stays
                // inside the wave's LDS: msk runs into tab, tab into cof) and dropped by selects -
                // no exec-mask changes
                const uint32_t *tp = tab + tb + 2 + y;
                const unsigned long long *mp = msk + tb + 2 + y;
                int32_t left = (int32_t)(yend - y);
                // Most chunks need one or two entries per lane (the quantizers of a -f codebook have a handful of states and the sum
                // runs from the nearer end): such a chunk loads one or two entries per lane instead of four - a quarter or half of
                // the round trip's 48 bytes per lane, of the selects and of the mask counts. (Timing experiment, 100M x 150 -f 0.5:
                // without this loop the model kernels take 30 % less.) Wave-uniform branches.
                if (!__ballot(left > 2)) {
                    if (__ballot(left > 1)) {
                        uint32_t c0 = tp[0], c1 = tp[1];
                        unsigned long long m0 = mp[0], m1 = mp[1];
                        c0 = left > 0 ? c0 : 0u; m0 = left > 0 ? m0 : 0ull;
                        c1 = left > 1 ? c1 : 0u; m1 = left > 1 ? m1 : 0ull;
                        acc = c0 + c1;
                        accm = earlier(m1, earlier(m0, 0u));
                    } else if (__ballot(left > 0)) {
                        uint32_t c0 = tp[0];
                        unsigned long long m0 = mp[0];
                        c0 = left > 0 ? c0 : 0u; m0 = left > 0 ? m0 : 0ull;
                        acc = c0;
                        accm = earlier(m0, 0u);
                    }
                    left = 0;
                }
                while (__ballot(left > 0)) {
                    uint32_t c0 = tp[0], c1 = tp[1], c2 = tp[2], c3 = tp[3];
                    unsigned long long m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3];
                    c0 = left > 0 ? c0 : 0u; m0 = left > 0 ? m0 : 0ull;
                    c1 = left > 1 ? c1 : 0u; m1 = left > 1 ? m1 : 0ull;
                    c2 = left > 2 ? c2 : 0u; m2 = left > 2 ? m2 : 0ull;
                    c3 = left > 3 ? c3 : 0u; m3 = left > 3 ? m3 : 0ull;
                    acc += c0 + c1 + c2 + c3;
                    accm = earlier(m3, earlier(m2, earlier(m1, earlier(m0, accm))));
                    tp += 4; mp += 4; left -= 4;
                }
                cum_b = up ? acc : n_b - c_x - acc;
                L = up ? accm : e - E - accm;
            }
            RK_T(1);
            const uint32_t key = ok ? ((tb << 8) | x) : 0xFFFFFFFFu;
            uint32_t n = n_b + STEP * e, fx = c_x + STEP * E, cum_lo = cum_b + STEP * L;
            const bool rescale = __ballot(ok && n_b + STEP * hits > RESCALE) != 0;
            msk[tb] = 0ull;                                      // leave the masks clean for the next chunk
            msk[tb + 2 + x] = 0ull;
            if (!rescale) {
                // update_stats for the whole chunk: every lane knows what its table's total and its state's count become
                // (before + 8 x the lanes of the chunk that share them) and stores it - the lanes that share an entry store
                // the same value, and plain stores to one address do not take turns as atomic adds do (model kernels 132.6 ->
                // 130.8 ms at 100M x 150 with one cluster, 234.8 -> 228.5 with three). Lanes without a symbol sit on table 0,
                // state 0 and store what the lanes that do have them store. (One store per entry by the first lane that holds it
                // needs two exec-mask regions: measured 4 % slower than the atomics had been.)
                {
                tab[tb + 2 + x] = c_x + STEP * same;
                tab[tb] = n_b + STEP * hits;
                }
            }
            if (rescale) {
                // a table rescales inside this chunk: replay the chunk serially (qv_stream.c:9-25), every lane doing the
                // same uniform work on the LDS tables, lane i keeping symbol i's values
                unsigned long long rest = __ballot(ok);
                while (rest) {
                    const int i = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, i);
                    const uint32_t t = kk >> 8, xx = kk & 0xFF;
                    const uint32_t nn0 = tab[t], card = tab[t + 1];
                    uint32_t cum = 0;
                    for (uint32_t y = 0; y < xx; ++y) cum += tab[t + 2 + y];
                    const uint32_t f = tab[t + 2 + xx];
                    if (lane == (uint32_t)i) { n = nn0; cum_lo = cum; fx = f; }
                    uint32_t nn = nn0 + STEP;
                    if (lane == 0) tab[t + 2 + xx] = f + STEP;
                    if (nn > RESCALE) {
                        nn = 0;
                        for (uint32_t y = 0; y < card; ++y) {
                            const uint32_t v = tab[t + 2 + y];
                            if (v) { const uint32_t hv = (v >> 1) + 1; if (lane == 0) tab[t + 2 + y] = hv; nn += hv; }
                        }
                    }
                    if (lane == 0) tab[t] = nn;
                }
            }
            if (REC8) {
                outp = ok ? pack_record(n, cum_lo, fx) : PREC_NOOP;   // (ok: n >= the alphabet size >= 1)
            } else {
                const Record rr = make_record_fixed(n ? n : 1u, cum_lo, fx);   // (DENSE: the host's chain)
                outr = make_double2(ok ? rr.x : 0.0, ok ? rr.y : 0.0);
            }
        }
        Record *buf = recbuf;
        RK_T(2);
        __syncthreads();                                         // the previous chunk has been drained
        RK_T(3);
        if (REC8) recbuf8[lane * rb_stride + wave] = outp;
        else buf[lane * rb_stride + wave] = outr;
        if (new_rows) *rowslot(lane, 3) = r_cur[3] - (uint32_t)seg_lo;
        __syncthreads();
        RK_T(4);
        {                                                        // W adjacent columns of one read = W * 16 contiguous bytes
            const uint32_t rd = threadIdx.x >> log2w, cw = threadIdx.x & (W - 1);
            const uint32_t col = cg * W + cw;                    // rec_index with the stream's part hoisted out of the loop
            const uint32_t row = lists ? *rowslot(rd, chunk & 3) : rb + rd;
            // The input bytes of chunks c+2 .. c+5 are taken over here (and the following four requested): the wait for
            // them also waits for every store issued so far, and at this point the latest one is a whole chunk old.
            if (((chunk + 2) & 3) == 0 && !(PF && pf_rest)) park(rb + 128);
            if (rb + rd < nr && col < Cp) {                      // columns Cx .. Cp-1 belong to idle waves: zero records
                {
                if (REC8) RK_STORE8(&rec8_g[((uint64_t)(row >> 2) * (Cp >> 2) + (col >> 2)) * unit_recs + (row & 3) * 4 + (col & 3)], recbuf8[rd * rb_stride + cw]);
                else RK_STORE(&rec_g[((uint64_t)(row >> 1) * (Cp >> 2) + (col >> 2)) * unit_recs + (row & 1) * 4 + (col & 3)], buf[rd * rb_stride + cw]);
                }
            }
        }
        RK_T(5);
    }
#ifdef QVZ_PROFILE
    if (lane == 0 && (g & 127) == 5 && (cg & 7) == 3) {             // a sample of the workgroups: same-address atomics are slow
        for (int i = 0; i < 6; ++i) atomicAdd(&g_rk_prof[i], prof[i]);
        atomicAdd(&g_rk_prof[6], (unsigned long long)chunk);
        atomicAdd(&g_rk_prof[7], tloop - tpass);
        atomicAdd(&g_rk_prof[11], tp1 - tpass); atomicAdd(&g_rk_prof[12], tp2 - tp1); atomicAdd(&g_rk_prof[13], tloop - tp2);
        if (zi == 0) atomicAdd(&g_rk_prof[14], tloop - tpass);
        atomicAdd(&g_rk_prof[15], tdrain);
        atomicAdd(&g_rk_prof[9], 1ull);
    }
    const unsigned long long tsave = __builtin_readcyclecounter();
#endif
    if (active && !is_last)
        for (uint32_t i = lane; i < sw / 4; i += 64) {           // (read again a segment later: the non-temporal hint, as for the records)
            typedef uint32_t rk_u32x4s __attribute__((ext_vector_type(4)));
            const uint4 v = reinterpret_cast<const uint4 *>(tab)[i];
            rk_u32x4s w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
            __builtin_nontemporal_store(w, reinterpret_cast<rk_u32x4s *>(save + (uint64_t)g * total_words + b0) + i);
        }
#ifdef QVZ_PROFILE
    if (lane == 0 && (g & 127) == 5 && (cg & 7) == 3) {
        const unsigned long long tend = __builtin_readcyclecounter();
        atomicAdd(&g_rk_prof[8], tend - tsave);
        atomicAdd(&g_rk_prof[10], tend - tpass);
    }
#endif
    }                                                            // clusters of this workgroup
    if (bad) atomicOr(status, 1);
}

// Workgroup = four waves serving the same 64 streams, one lane per stream in each. A lone wave issues a dependent
// VALU instruction every 8 cycles and an independent one every 5 (tools/lat_bench.hip), pays tens of cycles for every
// exec-mask change, and a load whose 64 lanes touch 64 different lines holds the wave for about as many cycles; so a
// stream's speed is the work of its busiest wave, and the work is cut by wave and kept branch-free:
//   wave 3 (loader)    streams every lane's records (128-byte runs, four blocks in flight in registers) into an LDS
//                      ring; it is the only wave that loads from global memory;
//   wave 0 (interval)  the serial chain proper (arith.c:24-97): l/u from the record, E1/E2 shift count k, E3 count j,
//                      renormalised l/u - 24 dependent instructions. What the symbol contributes to the output
//                      depends only on (top k bits of the new l, k, j), parked as one word in a second ring;
//   wave 1 (bits)      pending-bit bookkeeping (scale3): turns a word into (value, bit count) in a third ring;
//   wave 2 (bytes)     bit accumulator and the byte stream (the only wave that stores).
// One raw s_barrier per block of CR_B symbols (no fence, nobody but the loader waits on vmcnt): before barrier b the
// loader has parked records b; after it the interval wave codes block b, the bits wave block b-1, the bytes wave
// block b-2. (Measured with a single emitter wave: emitter busy 1440 cycles per 8 symbols, loader 1190, chain 1070.)
__global__ __launch_bounds__(64) void k_coder_spin_us(uint32_t us) {   // see the segment loop of qvz_hip_encode_planes
    const uint64_t t0 = wall_clock64();                           // 100 MHz
    while (wall_clock64() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(8);
}

// NB = symbols per block (barrier) and lane: 8 (2 x 8 x 64 records = 16 KB of ring; the whole kernel keeps 31.5 KB of LDS) or, with
// packed records, 16 (58 KB: half as many barriers; a coder workgroup then leaves room for two model workgroups, which is what the
// LDS padding of the 8-symbol kernel arranges on purpose)
#define CR_B NB
#define CR_RARE 0x80000000u   // ebits entry of a symbol with more than 16 bits: CR_RARE | pending run (the word itself is still in `words`)
// REC8 (round 4): the records come packed (PRec, rec8_index) and a FIFTH wave joins: waves 0 and 4 - which a round-robin
// placement of the workgroup's waves puts on one SIMD - are two loaders that each take half of every block (symbols 0-3 /
// 4-7: the 32-byte piece of one unit per lane), turn the integer triples into the two quotients (unpack_record) and park them
// in the ring; waves 1, 2, 3 are interval, bits, bytes as before. The loaders are throughput work (eight independent
// records per block and lane); the three chain waves keep a SIMD each.
template <int NB, int NEWTON>
__global__ __launch_bounds__(384) void k_code_records(const Record *__restrict__ rec, uint32_t Cx, const uint64_t *__restrict__ first,
                                                      uint32_t shards, uint64_t seg_lo, uint32_t seg_reads, int is_first, int is_last,
                                                      CoderState *__restrict__ cs, uint8_t *__restrict__ out, uint64_t cap,
                                                      uint64_t *__restrict__ sizes, int32_t *__restrict__ status, int high_prio) {
    constexpr int LAG = 1;                                       // blocks the interval wave (and the two behind it) run behind the loaders
    static_assert(NB == 8 || NB == 16, "the record layout has 8 records per lane and sub-block");
    __shared__ Record ring[2][CR_B][64];
    __shared__ uint32_t words[3][CR_B][64];                      // three slots: the bytes wave re-reads a rare symbol's word two blocks later
    __shared__ uint32_t ebits[2][CR_B][64];                      // value (16 bits) | bit count << 16, or CR_RARE | pending run
    __shared__ uint32_t final_l[64], final_s3[64];
    __shared__ uint4 oring[64 * 5];                              // the bytes wave's output ring: 16 words per lane, 80-byte stride
    const uint32_t hw_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // role: 0 interval, 1 bits, 2 bytes, 3 loader of symbols 0-3 of every sub-block, 4 loader of symbols 4-7, 5 the third loader.
    // Six waves on four SIMDs: the placement goes round (hardware waves 0 and 4 share a SIMD, 1 and 5 another), so the two big
    // loaders are waves 0 and 4 as before, the bytes wave - the lightest of the chain's three - is wave 1 and gets the third
    // loader (wave 5) for company, and bits and interval keep a SIMD each (waves 2 and 3).
    const uint32_t wave = hw_wave == 0 ? 3u : hw_wave == 4 ? 4u : hw_wave == 5 ? 5u : hw_wave == 1 ? 2u : hw_wave == 2 ? 1u : 0u;
    const uint32_t g = blockIdx.x * 64 + lane;
    const bool live = g < shards;
    const uint32_t Cp = (Cx + 7) & ~7u;                          // records per read (rec_index)
    if (high_prio) {
        // all waves of the latency chain issue ahead of the throughput kernels sharing the CU; REC8: the two loaders, which are
        // throughput work themselves, one step below the three chain waves
        if (wave >= 3) asm volatile("s_setprio 2"); else asm volatile("s_setprio 3");
    }
    uint64_t T = 0;
    if (live) {
        const uint64_t rend = first[g + 1], a = first[g] + seg_lo, b = a + seg_reads;
        if (a < rend) T = ((b < rend ? b : rend) - a) * Cp;
    }
    uint64_t Tmax = T, Tmin = T;                                 // lanes past the last stream have T = 0 and are padded
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t v = __shfl_xor(Tmax, o, 64), m = __shfl_xor(Tmin, o, 64);
        Tmax = v > Tmax ? v : Tmax;
        Tmin = m < Tmin ? m : Tmin;
    }
    // (the two are the same in every lane after the exchange, which the compiler cannot see: out of a scalar register every loop
    // bound and every "is this block whole in all lanes" below is a scalar compare and branch instead of a vector compare and an
    // exec-mask region)
    Tmax = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(Tmax >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)Tmax);
    Tmin = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(Tmin >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)Tmin);
    const uint64_t nblk = (Tmax + CR_B - 1) / CR_B;              // the same in all four waves
    const bool fresh = is_first || !live;
#ifdef QVZ_PROFILE          // busy / waiting cycles of each role between the block barriers
    unsigned long long cr_busy = 0, cr_wait = 0, cr_drain = 0, cr_t = __builtin_readcyclecounter();   // cr_drain: the part of cr_wait spent in s_waitcnt lgkmcnt(0)
#define CR_BARRIER() do { const unsigned long long a_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                          const unsigned long long m_ = __builtin_readcyclecounter(); asm volatile("s_barrier" ::: "memory"); \
                          const unsigned long long b_ = __builtin_readcyclecounter(); cr_busy += a_ - cr_t; cr_wait += b_ - a_; cr_drain += m_ - a_; cr_t = b_; } while (0)
#define CR_REPORT() do { if (lane == 0 && (blockIdx.x & 15) == 3 && wave < 4) { atomicAdd(&g_cr_prof[wave * 2], cr_busy); atomicAdd(&g_cr_prof[wave * 2 + 1], cr_wait); \
                                                                    atomicAdd(&g_cr_prof[9 + wave], cr_drain); \
                                                                    if (wave == 0) atomicAdd(&g_cr_prof[8], (unsigned long long)nblk); } \
                         if (lane == 0 && (blockIdx.x & 15) == 3 && wave == 4) { atomicAdd(&g_cr_prof[13], cr_busy); atomicAdd(&g_cr_prof[14], cr_wait); } \
                         if (lane == 0 && (blockIdx.x & 15) == 3 && wave == 5) { atomicAdd(&g_cr_prof[15], cr_busy); } \
                         if (lane == 0 && (blockIdx.x & 15) == 3) { uint32_t hwid_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_)); g_cr_simd[hw_wave] = (hwid_ >> 4) & 3; } } while (0)
#else
#define CR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define CR_REPORT() do { } while (0)
#endif
    // barriers 0 .. nblk + 1 + LAG in every wave (+ one more on the last segment for the hand-over of l and scale3).
    // (Beside the model kernel the interval wave is busy 1084 cycles of a block's 1370, 74 of the rest drain its LDS writes and
    // 212 pass in the barrier - 57 with the CU to itself. Tried: no barrier, the waves handing blocks over through progress
    // counters in LDS - a counter written after a block's data, read before it; a wave's LDS instructions are carried out in
    // order -, only the wave that is behind waiting, polls bounded. Byte-identical streams and twice as slow, alone (210 against
    // 110 ms) and beside the model kernel (223 against 139): a hand-over costs a write's and a read's LDS latency plus the
    // poll's sleep where the barrier costs tens of cycles, and the polling waves take LDS slots from the one that works.)

    // The two loaders' SIMD is the busiest of the workgroup (the second loader was busy 2175 of a block's 2327 cycles, the interval
    // wave 1823: profiles/r05_coder_cycles.txt), and what they do is wave-wide instructions: a block is eight wave-loads of 64
    // sixteen-byte pieces, four per loader. The third loader takes ONE of the second loader's four - the piece of streams
    // 32 .. 63 in the second half of a block's second sub-block - on the bytes wave's SIMD.
    if (wave == 5) {
        // ------------------------------------ the third loader ------------------------------------
        static_assert(NB == 16, "the third loader's share is the odd sub-block of a 16-symbol block");
        const uint32_t sB = 32 + (lane >> 1), hp = lane & 1;
        const uint64_t nunit = (uint64_t)seg_reads * Cp / 16;
        const uint64_t nblk8 = (uint64_t)seg_reads * Cp / 8;
        const uint4 *p0 = reinterpret_cast<const uint4 *>(reinterpret_cast<const PRec *>(rec) + ((uint64_t)blockIdx.x * nunit * 64) * 16);
        const uint4 *pB = p0 + sB * 8 + hp;
        const uint64_t TB = __shfl(T, (int)sB, 64);
        const uint64_t nfull = Tmin / 8;
        const uint32_t per_row = Cp >> 3;                        // sub-blocks per read
        uint64_t l_blk = 1;                                      // the sub-block the next load asks for: 1, 3, 5, ...
        uint32_t l_row = 0, l_m = 1;
        while (l_m >= per_row) { l_m -= per_row; ++l_row; }
        if (l_blk >= nblk8) { l_blk = 0; l_row = 0; l_m = 0; }   // (an area of one sub-block: its odd sub-blocks are padding, any address inside it will do)
        auto load1 = [&]() -> uint4 {
            const uint64_t at = (((uint64_t)(l_row >> 2) * (Cp >> 2) + 2 * l_m + 1) * 64 * 8) + (l_row & 3) * 2;
            const uint4 v = pB[at];
            if (l_blk + 2 < nblk8) {                             // (past the area's last odd sub-block the same one is loaded again)
                l_blk += 2;
                l_m += 2;
                while (l_m >= per_row) { l_m -= per_row; ++l_row; }
            }
            return v;
        };
        auto park1 = [&](uint64_t t, const uint4 &d) {           // block t's piece -> slot t & 1, then the block's barrier
            const uint64_t sb = 2 * t + 1;
            const uint32_t slot = (uint32_t)(t & 1), sym = 12 + 2 * hp, tw = hp << 3;
            Record r2[2];
            unpack_records2<NEWTON>(d, r2);
            if (sb < nfull) {
                ring[slot][sym][sB ^ tw] = r2[0];
                ring[slot][sym + 1][sB ^ tw] = r2[1];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint64_t ts = sb * 8 + 4 + 2 * hp + i;
                    ring[slot][sym + i][sB ^ tw] = make_double2(ts >= TB ? 0.0 : r2[i].x, ts >= TB ? 0.0 : r2[i].y);
                }
            }
            CR_BARRIER();
        };
        uint4 q0 = load1(), q1 = load1(), q2 = load1(), q3 = load1();   // four blocks' pieces on their way
        for (uint64_t t = 0; t < nblk; t += 4) {
            park1(t, q0); q0 = load1();
            if (t + 1 >= nblk) break;
            park1(t + 1, q1); q1 = load1();
            if (t + 2 >= nblk) break;
            park1(t + 2, q2); q2 = load1();
            if (t + 3 >= nblk) break;
            park1(t + 3, q3); q3 = load1();
        }
        CR_BARRIER();                                            // barriers nblk .. nblk + 2: the other waves run behind
        CR_BARRIER();
        CR_BARRIER();
        if (is_last) CR_BARRIER();
        CR_REPORT();
        return;
    }
    if (wave >= 3) {
        // ------------------------------------ the two loaders of packed records ------------------------------------
        constexpr int SUB = NB / 8;                              // sub-blocks of 8 symbols per block
        const uint32_t half = wave - 3;                          // symbols 4 * half .. 4 * half + 3 of every sub-block
        const uint64_t nunit = (uint64_t)seg_reads * Cp / 16;    // units of 16 records in a stream's area
        const uint64_t nblk8 = (uint64_t)seg_reads * Cp / 8;     // sub-blocks of the area (what a lane can load)
        // Two lanes share a line: a load instruction whose 64 lanes touch 64 different lines holds the wave about as many cycles, and
        // the loaders' SIMD is the busiest of the workgroup's. So lanes 2k and 2k + 1 take the two 16-byte halves (hp) of stream k's
        // 32-byte piece with the first load of a sub-block and of stream 32 + k's with the second - 32 lines per instruction instead
        // of 64 -, every lane unpacks the four records it loaded (two of either stream) and parks them where they belong by address:
        // nothing is shuffled. The ring's stream index is XORed with 8 for symbols 2, 3 (mod 4), so that an even and an odd lane -
        // same stream, symbols two apart: 2 KB apart - write to different banks; the interval wave reads with the same twist.
        const uint32_t sA = lane >> 1, sB = 32 + (lane >> 1), hp = lane & 1;
        const uint4 *p0 = reinterpret_cast<const uint4 *>(reinterpret_cast<const PRec *>(rec) + ((uint64_t)blockIdx.x * nunit * 64) * 16);   // unit 0 of the workgroup's streams (rec8_index), 8 uint4 per unit and stream
        const uint4 *pA = p0 + sA * 8 + hp, *pB = p0 + sB * 8 + hp;
        const uint64_t TA = __shfl(T, (int)sA, 64), TB = __shfl(T, (int)sB, 64);
        const uint64_t nfull = Tmin / 8;                         // sub-blocks that are whole in every lane: no padding test
        const uint64_t nsub = nblk * SUB;                        // sub-blocks to park (those past the last symbol as no-op records)
        // sub-block sb of a lane = columns 8m .. 8m+7 of read `row` (sb = row * Cp/8 + m): this wave's four are the quarter
        // [row & 3] of unit (row / 4, 2m + half): 32 bytes. A register stage holds LD_B sub-blocks' pieces; five stages in
        // rotation keep four of them in flight. Past the last sub-block of the area the last one is loaded again (unconditional
        // loads keep vmcnt countable).
        constexpr int LD_B = 2;
        uint64_t l_blk = 0;
        uint32_t l_row = 0, l_m = 0;
        auto load_stage = [&](uint4 *d) {
#pragma unroll
            for (int k = 0; k < LD_B; ++k) {
                const uint64_t at = (((uint64_t)(l_row >> 2) * (Cp >> 2) + 2 * l_m + half) * 64 * 8) + (l_row & 3) * 2;   // in uint4: a unit is 8, the 64 streams' units lie side by side
                d[k * 2] = pA[at];
                // (a block's last piece of streams 32 .. 63 is the third loader's: the second loader asks for one line in all its
                // lanes instead - a load all the same, so that the loads in flight stay countable, but one line of work for the memory path)
                const uint4 *const srcB = (half == 1 && k == LD_B - 1) ? p0 : pB + at;
                d[k * 2 + 1] = *srcB;
                if (l_blk + 1 < nblk8) {
                    ++l_blk;
                    if (++l_m == (Cp >> 3)) { l_m = 0; ++l_row; }
                }
            }
        };
        auto park = [&](uint64_t sb, const uint4 *d) {           // this wave's half of sub-block sb -> slot (block & 1); the block's barrier behind its last sub-block
            const uint32_t slot = (uint32_t)((sb / SUB) & 1), at = (uint32_t)(sb % SUB) * 8 + half * 4;
            const uint32_t sym = at + 2 * hp, tw = hp << 3;      // (at is a multiple of 4: bit 1 of the symbol number is hp)
            if (half == 1 && sb % SUB == SUB - 1) {              // the second loader's short piece: stream sA only
                Record r2[2];
                unpack_records2<NEWTON>(d[0], r2);
                if (sb < nfull) {
                    ring[slot][sym][sA ^ tw] = r2[0];
                    ring[slot][sym + 1][sA ^ tw] = r2[1];
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const uint64_t t = sb * 8 + half * 4 + 2 * hp + i;
                        ring[slot][sym + i][sA ^ tw] = make_double2(t >= TA ? 0.0 : r2[i].x, t >= TA ? 0.0 : r2[i].y);
                    }
                }
                CR_BARRIER();
                return;
            }
            Record r4[4];                                        // records 2 hp, 2 hp + 1 of the piece of stream sA, then of stream sB
            unpack_records4<NEWTON>(d[0], d[1], r4);
            if (sb < nfull) {
                ring[slot][sym][sA ^ tw] = r4[0];
                ring[slot][sym + 1][sA ^ tw] = r4[1];
                ring[slot][sym][sB ^ tw] = r4[2];
                ring[slot][sym + 1][sB ^ tw] = r4[3];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint64_t t = sb * 8 + half * 4 + 2 * hp + i;   // beyond a stream's symbols: the no-op record
                    ring[slot][sym + i][sA ^ tw] = make_double2(t >= TA ? 0.0 : r4[i].x, t >= TA ? 0.0 : r4[i].y);
                    ring[slot][sym + i][sB ^ tw] = make_double2(t >= TB ? 0.0 : r4[2 + i].x, t >= TB ? 0.0 : r4[2 + i].y);
                }
            }
            if (sb % SUB == SUB - 1) CR_BARRIER();
        };
        auto park_stage = [&](uint64_t stg, const uint4 *d) -> bool {   // false: that was the last sub-block
#pragma unroll
            for (int k = 0; k < LD_B; ++k) {
                const uint64_t sb = stg * LD_B + k;
                if (sb >= nsub) return false;
                park(sb, d + k * 2);
            }
            return (stg + 1) * LD_B < nsub;
        };
        // five register stages in rotation: four (eight sub-blocks = four blocks) on their way while one is unpacked; with three
        // stages the coder kernels took 105.5-106.2 ms against 104.7 on the same box (round 5)
        uint4 s0[LD_B * 2], s1[LD_B * 2], s2[LD_B * 2], s3[LD_B * 2], s4[LD_B * 2];
        load_stage(s0); load_stage(s1); load_stage(s2); load_stage(s3);
        for (uint64_t t = 0; t * LD_B < nsub; t += 5) {
            load_stage(s4);
            if (!park_stage(t, s0)) break;
            load_stage(s0);
            if (!park_stage(t + 1, s1)) break;
            load_stage(s1);
            if (!park_stage(t + 2, s2)) break;
            load_stage(s2);
            if (!park_stage(t + 3, s3)) break;
            load_stage(s3);
            if (!park_stage(t + 4, s4)) break;
        }
        CR_BARRIER();                                            // barriers nblk .. nblk + 2: the other waves run behind
        CR_BARRIER();
        CR_BARRIER();
        if (is_last) CR_BARRIER();
        CR_REPORT();
        return;
    }

    if (wave == 0) {
        // ------------------------------------ interval chain ------------------------------------
        uint32_t l32 = 0, nu32 = 0;                              // l = 0, u = 2^22 - 1
        if (!fresh) { l32 = cs[g].l << 10; nu32 = (~cs[g].u & MASK_M) << 10; }
        uint32_t wslot = 0;                                      // (block coded) % 3
        // Packed-record kernel: the wave runs one block behind the loaders - after barrier t it asks LDS for block t's
        // records (into the register set the block before last has left) and codes block t - 1 from the registers filled
        // a block ago, so no symbol waits for an LDS round trip. Two ring slots still do: block t + 1 is parked between
        // barriers t and t + 1 in the slot whose reads were complete before barrier t.
        // (The interval wave is ISSUE-bound, not latency-bound: ~22 VALU instructions per symbol at the 5-6 cycles a lone wave
        // needs per instruction are the 135 cycles per symbol it is busy, whatever the depth of the dependent chain. Tried:
        // fma(2^32 - s, x, c) as fma(-s, x, c') with c' = 2^32 x + c made off the chain - one dependent instruction less, five
        // instructions more (two v_fmac and the moves of their constants): coder kernels 116.0 -> 117.6 ms. What does pay is
        // an instruction less: the E3 count as ffbh(~(l & u') & 0x7FFFFFFF) - 1 - one v_bitop3 with the mask as third operand
        // instead of nand and a shift by one; the count is used as it comes (one too many): shifting left by it and right by
        // one is the old shift and mask, and the bits wave subtracts the one.)
        Record ra[CR_B], rb2[CR_B];
        auto load = [&](Record *r, uint32_t slot) {         // (symbols 2, 3 mod 4 sit at stream ^ 8: the loaders' bank twist)
#pragma unroll
            for (int jx = 0; jx < CR_B; ++jx) r[jx] = ring[slot][jx][lane ^ (((jx >> 1) & 1) << 3)];
        };
        auto code = [&](const Record *r) {
#pragma unroll
            for (int jx = 0; jx < CR_B; ++jx) {
                const double rd = 4294967296.0 - (double)(l32 + nu32);       // (u - l + 1) * 1024
                const uint32_t ql = (uint32_t)__builtin_fma(rd, r[jx].x, QUOT_LO);
                const uint32_t qc = (uint32_t)__builtin_fma(rd, r[jx].y, QUOT_HI);
                l32 += ql << 10;
                nu32 += qc << 10;
                const uint32_t k = (uint32_t)__builtin_clz(~(l32 ^ nu32));   // E1/E2: leading bits shared by l and u (low 10 bits stop it at 22)
                const uint32_t lk = l32;                                      // its top k bits go out
                l32 <<= k;
                nu32 <<= k;
                // E3: j positions below the MSB with l = 1, u = 0. Both MSBs are 0 here, so ~(l & nu) has its MSB set and the
                // leading zeros of the rest number j + 1 (never all 32: the ten low bits of l32 are zero)
                const uint32_t j1 = (uint32_t)__builtin_clz(~(l32 & nu32) & 0x7FFFFFFFu);
                l32 = (l32 << j1) >> 1;
                nu32 = (nu32 << j1) >> 1;
                words[wslot][jx][lane] = lk | (k << 5) | j1;                 // (j + 1 <= 22 fits the five bits)
            }
            wslot = wslot == 2 ? 0u : wslot + 1;
        };
        CR_BARRIER();                                        // barrier 0
        if (nblk > 0) load(ra, 0);
        uint64_t t = 1;
        for (; t + 1 <= nblk; t += 2) {
            CR_BARRIER();                                    // barrier t (odd): block t in, block t - 1 coded
            load(rb2, 1);
            code(ra);
            CR_BARRIER();                                    // barrier t + 1
            if (t + 1 < nblk) load(ra, 0);
            code(rb2);
        }
        if (t == nblk) { CR_BARRIER(); code(ra); }           // an odd number of blocks: the last one
        CR_BARRIER();
        CR_BARRIER();
        if (is_last) {
            final_l[lane] = l32 >> 10;
            CR_BARRIER();
        } else if (live) {
            cs[g].l = l32 >> 10; cs[g].u = ~(nu32 >> 10) & MASK_M;
        }
        CR_REPORT();
        return;
    }

    if (wave == 1) {
        // ------------------------------------ bits ------------------------------------
        // A symbol puts out k + s3 bits: the first of the k shared bits, the s3 pending E3 bits (its complement), then
        // the other k-1. As a number that is top + ((2^s3 - 1) << (k-1)) whatever the first bit is (top = the k bits):
        // one v_bfm + one add. More than 16 bits (about one symbol in 2^16): left to the bytes wave, piecewise.
        uint32_t scale3 = fresh ? 0u : cs[g].scale3;
        uint32_t wslot = 0;                                      // (b - 1) % 3
        for (uint64_t b = 0; b <= nblk + LAG; ++b) {
            CR_BARRIER();
            if (b < 1 + LAG) continue;
            const uint32_t slot = (uint32_t)((b - 1 - LAG) & 1);
            uint32_t w[CR_B];
#pragma unroll
            for (int jx = 0; jx < CR_B; ++jx) w[jx] = words[wslot][jx][lane];
#pragma unroll
            for (int jx = 0; jx < CR_B; ++jx) {
                const uint32_t k = (w[jx] >> 5) & 31, j = (w[jx] & 31) - (LAG ? 1u : 0u);   // (the packed-record kernel's interval wave leaves j + 1)
                const uint32_t top = __builtin_amdgcn_ubfe(w[jx], 32 - k, k);   // the k leading bits of l (0 when k = 0)
                const uint32_t s3 = k ? scale3 : 0u;             // the pending run goes out only when a bit is decided
                const uint32_t nb = k + s3;
                // (2^s3 - 1) << (k - 1): nothing when k = 0 (s3 = 0 then), so the shift count may wrap
                uint32_t normal = (top + (((1u << (s3 & 31)) - 1) << ((k - 1) & 31))) | (nb << 16);
                asm volatile("" : "+v"(normal));                 // computed in every lane: a select below, not an exec-mask region
                ebits[slot][jx][lane] = nb > 16 ? (CR_RARE | s3) : normal;
                scale3 = scale3 - s3 + j;
            }
            wslot = wslot == 2 ? 0u : wslot + 1;
        }
        CR_BARRIER();                                            // barrier nblk + 1
        if (is_last) {
            final_s3[lane] = scale3;
            CR_BARRIER();
        } else if (live) {
            cs[g].scale3 = scale3;
        }
        CR_REPORT();
        return;
    }

    // ------------------------------------ bytes ------------------------------------
    // Symbols are taken in pairs (at most 16 bits each on top of fewer than 32 pending ones: the 64-bit accumulator
    // holds them). After a pair the word at the top of the accumulator is WRITTEN UNCONDITIONALLY at the lane's write
    // index and the index advances only if the word was complete (nbits >= 32): an incomplete word is overwritten
    // later. No test, no exec-mask change. The words go to a 16-word ring per lane in LDS, not to memory: a store
    // whose 64 lanes hit 64 different lines costs the wave (and the memory pipeline it shares with the model kernel)
    // about as many cycles, four of them per block made this wave the slowest of the four. Once per block the lanes
    // that have four complete words move them out as one 16-byte store (about five lanes per block at 1.4 bits per
    // symbol). A block adds at most four complete words, so at most seven are ever waiting.
    // A pair with a long symbol in some lane goes through a wave-uniform branch to the code with exact bounds (the
    // ring is emptied first). Room is checked every fourth block, which makes a stream that comes within 256 bytes of
    // cap_per_shard count as overflowed. Lanes past the last stream produce zero bits and never store.
    uint8_t *const base = live ? out + (uint64_t)g * cap : out + (uint64_t)shards * cap - 4;
    uint8_t *const end = live ? base + cap : base + 4 + 256;
    uint8_t *rb = base;                                          // where word 0 of the ring numbering lies
    uint32_t wi = 0, fi = 0;                                     // words completed / moved out since rb (fi a multiple of 4)
    uint32_t nbits = 0;
    uint64_t acc = 0;
    bool ovf = false;
    if (!fresh) {
        const CoderState s = cs[g];
        nbits = s.nbits; acc = s.acc; rb = base + s.pos; ovf = s.overflow != 0;
    }
    uint32_t *const myring = reinterpret_cast<uint32_t *>(oring) + lane * 20;   // 80-byte lane stride: 16-byte aligned, spreads the banks
    auto drain = [&]() {                                         // everything complete goes out, word by word; the numbering restarts
        for (uint32_t t = fi; __ballot(t < wi); ++t)
            if (t < wi) *reinterpret_cast<uint32_t *>(rb + (uint64_t)t * 4) = myring[t & 15];
        rb += (uint64_t)wi * 4;
        wi = fi = 0;
    };
    auto piecewise = [&](uint32_t e, uint32_t w) {               // one symbol, any number of bits, exact bounds; ring empty
        LaneCoder c;
        c.out = base; c.cap = cap; c.pos = (uint64_t)(rb - base); c.acc = acc; c.nbits = nbits; c.overflow = ovf;
        c.l = 0; c.u = 0; c.scale3 = 0;
        if (e & CR_RARE) {
            const uint32_t k = (w >> 5) & 31, km1 = k - 1, top = w >> (32 - k);   // k >= 1 here
            const uint32_t bit = top >> km1;
            lc_put(c, bit, 1);
            lc_run(c, !bit, e & ~CR_RARE);
            if (km1) lc_put(c, top & ((1u << km1) - 1), km1);
        } else if (e >> 16) {
            lc_put(c, e & 0xFFFF, e >> 16);
        }
        acc = c.acc; nbits = c.nbits; ovf = c.overflow;
        rb = base + c.pos;
    };
    auto room = [&]() {                                          // out of room: keep the stores inside the stream's area
        const bool tight = rb + (uint64_t)wi * 4 + 256 > end;    // (looked at every fourth block: 4 x 8 symbols x 16 bits = 64 bytes + the ring)
        ovf = ovf || tight;
        rb = tight ? base : rb;
        wi = tight ? 0u : wi;
        fi = tight ? 0u : fi;
    };
    uint32_t wslot = 0;                                          // (b - 2) % 3
    for (uint64_t b = 0; b <= nblk + 1 + LAG; ++b) {
        CR_BARRIER();
        if (b < 2 + LAG) continue;
        const uint32_t slot = (uint32_t)((b - 2 - LAG) & 1);
        uint32_t e[CR_B];
#pragma unroll
        for (int jx = 0; jx < CR_B; ++jx) e[jx] = ebits[slot][jx][lane];
        if ((b - 2 - LAG) % (32 / NB) == 0) room();                    // every 32 symbols (b starts at 2; the exact-bounds path below looks again itself)
        auto pair = [&](uint32_t e0, uint32_t e1) {
            // the pair first, in 32 bits (at most 16 + 16), then one 64-bit shift for both
            const uint32_t n2 = e1 >> 16, pn = (e0 >> 16) + n2;
            const uint32_t pv = ((e0 & 0xFFFF) << n2) | (e1 & 0xFFFF);
            acc = (acc << pn) | (uint64_t)pv;
            nbits += pn;                                         // < 32 + 32
            myring[wi & 15] = __builtin_bswap32((uint32_t)(acc >> ((nbits - 32) & 63)));
            wi += nbits >> 5;                                    // keep the word only if it was complete
            nbits &= 31;
        };
        // one wave-uniform test per block (a scalar branch on a vector compare stalls a lone wave for tens of cycles; per
        // pair it was four of them): a block with a long symbol in some lane - about one in 2^13 - goes pair by pair
#pragma unroll
        for (int h = 0; h < CR_B; h += 8) {                      // eight symbols at a time: at most four complete words, then the ring is looked at
        if (__ballot((e[h] | e[h + 1] | e[h + 2] | e[h + 3] | e[h + 4] | e[h + 5] | e[h + 6] | e[h + 7]) & CR_RARE)) {
#pragma unroll
            for (int jx = h; jx < h + 8; jx += 2) {
                if (__ballot((e[jx] | e[jx + 1]) & CR_RARE)) {
                    drain();
                    piecewise(e[jx], words[wslot][jx][lane]);
                    piecewise(e[jx + 1], words[wslot][jx + 1][lane]);
                    room();
                } else {
                    pair(e[jx], e[jx + 1]);
                }
            }
        } else {
#pragma unroll
            for (int jx = h; jx < h + 8; jx += 2) pair(e[jx], e[jx + 1]);
        }
        if (__ballot(wi - fi >= 4)) {                            // wave-uniform; then only the lanes with four words store
            if (wi - fi >= 4) {
                *reinterpret_cast<uint4 *>(rb + (uint64_t)fi * 4) = *reinterpret_cast<const uint4 *>(myring + (fi & 15));
                fi += 4;
            }
        }
        }
        wslot = wslot == 2 ? 0u : wslot + 1;
    }
    drain();
    uint8_t *const p = rb;
    CR_REPORT();
    if (is_last) {
        CR_BARRIER();
        if (!live) return;
        LaneCoder c;
        c.out = base; c.cap = cap; c.pos = (uint64_t)(p - base); c.acc = acc; c.nbits = nbits; c.overflow = ovf;
        c.scale3 = final_s3[lane]; c.u = 0;
        c.l = final_l[lane];
        sizes[g] = lc_finish(c);
        if (c.overflow) atomicOr(status, 2);
    } else if (live) {
        cs[g].nbits = nbits; cs[g].acc = acc; cs[g].pos = (uint64_t)(p - base); cs[g].overflow = ovf ? 1u : 0u;
    }
#undef CR_BARRIER
#undef CR_REPORT
}

// every shard starts from the same initial model
__global__ __launch_bounds__(256) void k_arena_fill(const uint32_t *__restrict__ tmpl, uint32_t words, uint32_t shards,
                                                    uint32_t *__restrict__ arena) {
    const uint64_t total = (uint64_t)words * shards;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256)
        arena[i] = tmpl[i % words];
}

extern "C" int qvz_hip_encode_planes_bounds(const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                                            uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                            const uint64_t *h_bounds, uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes,
                                            void *d_work, uint64_t work_bytes, void *stream);
// The buffers a call needs beside the caller's (table images, the per-stream table saves between segments - gigabytes -,
// read lists, coder states) are carved out of ONE allocation per device that is kept between calls: a dozen hipMalloc /
// hipFree pairs per encode were, once in several hundred encodes, 2.6 - 4.8 SECONDS of it (measured: the whole stall inside
// the set-up part of the call, kernels and everything else as always). A call that needs more than the area holds takes the
// excess from hipMalloc and leaves a larger area behind. qvz_hip_encode_planes_release() gives the memory back.
struct KeptArea {
    std::mutex mu;
    char *base = nullptr;
    size_t cap = 0, used = 0, want = 0;
    std::vector<void *> extra;
    void begin() {
        used = 0; want = 0;
    }
    hipError_t get(void **p, size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        want += bytes;
        if (base && used + bytes <= cap) { *p = base + used; used += bytes; return hipSuccess; }
        const hipError_t e = hipMalloc(p, bytes ? bytes : 256);
        if (e == hipSuccess) extra.push_back(*p);
        return e;
    }
    void finish() {
        const bool grow = !extra.empty();
        for (void *p : extra) (void)hipFree(p);
        extra.clear();
        if (grow) {
            if (base) (void)hipFree(base);
            base = nullptr; cap = 0;
            const size_t room = want + want / 8;
            if (hipMalloc((void **)&base, room) == hipSuccess) cap = room; else base = nullptr;
        }
        used = want = 0;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = used = want = 0; }
};
static KeptArea g_coder_area[64];

extern "C" void qvz_hip_encode_planes_release(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lk(g_coder_area[dev].mu);
    g_coder_area[dev].release();
}

// Bytes of record workspace that let qvz_hip_encode_planes code streams of up to `longest_reads` reads in one segment
// (8 bytes per record, streams padded to 64, record positions per read padded to 8). A caller that
// allocates the workspace itself - early, beside other work - asks this, caps it (96 GiB, what is free) and passes it in.
// (Sized for 16-byte records whichever kind a call uses: packed ones need half.)
extern "C" uint64_t qvz_hip_encode_planes_work_bytes(uint64_t longest_reads, uint32_t columns, uint32_t clusters, uint32_t shards) {
    const uint32_t Cx = columns + (clusters > 1 ? (uint32_t)QVZ_CID_SLOTS : 0u);
    const uint32_t Cp = (Cx + 7) & ~7u;
    return (uint64_t)((shards + 63) & ~63u) * Cp * sizeof(PRec) * ((longest_reads + 3) & ~(uint64_t)3);
}

// 1: v_rcp_f64 + ONE Newton step is within 2^-47 of 1 / (n 2^10) for every n a packed record can hold on the current device (the
// loaders of k_code_records then save two of the unpacking's fourteen instructions per record); 0: it is not (two steps are used);
// measured once per process and device over all n = 1 .. 2^19 + 1024. QVZ_CODER=newton=2 (test hook) forces the two steps.
// *worst_out (may be NULL) receives the worst relative error found.
extern "C" int qvz_coder_rcp_one_step_ok_ex(double *worst_out) {
    static std::mutex mu;
    static int verdict[64];                                      // 0 unknown, 1 good, 2 not good enough
    static double worst_seen[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!verdict[dev]) {
        const uint32_t nmax = (1u << 19) + 1024;
        unsigned long long *d = nullptr, h = ~0ull;
        if (hipMalloc(&d, sizeof(h)) == hipSuccess && hipMemset(d, 0, sizeof(h)) == hipSuccess) {
            hipLaunchKernelGGL(k_rcp_accuracy<1>, dim3((nmax + 255) / 256), dim3(256), 0, 0, d, nmax);
            // a launch that failed leaves the memset's zero behind - which would read as "exact": only a residual the kernel
            // really wrote counts (some divisor of 2^19 always leaves a non-zero one)
            if (hipGetLastError() != hipSuccess || hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || h == 0) h = ~0ull;
        }
        if (d) (void)hipFree(d);
        double w;
        memcpy(&w, &h, sizeof(w));
        worst_seen[dev] = h == ~0ull ? 1.0 : w;
        verdict[dev] = worst_seen[dev] <= 0x1p-47 ? 1 : 2;
    }
    if (worst_out) *worst_out = worst_seen[dev];
    { double steps = 0.0; if (qvz_env_number("QVZ_CODER", "newton", &steps) && steps == 2.0) return 0; }   // (test hook: the two-step kernel)
    return verdict[dev] == 1 ? 1 : 0;
}
extern "C" int qvz_coder_rcp_one_step_ok(void) { return qvz_coder_rcp_one_step_ok_ex(nullptr); }

// The model / coder stream pair of the current device: made once, tested for running side by side when it is made
// (qvz_hip_streams_overlap: a partner on the first stream's hardware queue is put aside and another one made, a few times at
// most), and kept. Returns 1 with the pair, 0 when the ordinal is beyond the table (the caller makes its own), < 0 on an error.
static int coder_stream_pair(hipStream_t *pair) {
    static std::mutex mu;
    static hipStream_t kept[64][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return qvz_set_error(QVZ_E_HIP, "no current device");
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!kept[dev][0]) {
        hipStream_t first = nullptr, second = nullptr, aside[8];
        int naside = 0, ok = 0;
        if (hipStreamCreateWithFlags(&first, hipStreamNonBlocking) != hipSuccess) return qvz_set_error(QVZ_E_HIP, "hipStreamCreate failed");
        while (true) {
            if (hipStreamCreateWithFlags(&second, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(first); return qvz_set_error(QVZ_E_HIP, "hipStreamCreate failed"); }
            if (qvz_hip_streams_overlap(first, second, &ok) != QVZ_OK || ok || naside == 8) break;
            aside[naside++] = second;                            // kept alive until the choice is made: destroying it would hand its queue to the next one
        }
        for (int i = 0; i < naside; ++i) (void)hipStreamDestroy(aside[i]);
        if (getenv("QVZ_CODER_DEBUG")) fprintf(stderr, "qvz_hip_encode_planes: model / coder streams made for device %d, %d candidate(s) put aside, side by side: %d\n", dev, naside, ok);
        kept[dev][0] = first;
        kept[dev][1] = second;
    }
    pair[0] = kept[dev][0];
    pair[1] = kept[dev][1];
    return 1;
}

static thread_local double g_planes_ms[2] = {0.0, 0.0};   // model kernels, coder kernels of the calling thread's last qvz_hip_encode_planes call
extern "C" void qvz_hip_encode_planes_last_ms(double *model_ms, double *coder_ms) {
    if (model_ms) *model_ms = g_planes_ms[0];
    if (coder_ms) *coder_ms = g_planes_ms[1];
}

extern "C" int qvz_hip_encode_planes(const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                                     uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                     uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes, void *d_work, uint64_t work_bytes,
                                     void *stream) {
    return qvz_hip_encode_planes_bounds(d_qi, d_st, ld, d_labels, n, columns, clusters, cbs, shards, nullptr, d_out, cap_per_shard,
                                        h_sizes, d_work, work_bytes, stream);
}

// Segment list of a call: (first read, reads) relative to a stream's start, for the longest stream. When the two kernels
// overlap, the first model kernel and the last coder kernel still run alone: those two segments are a quarter of the others.
// first_div: the first segment is seg_reads / first_div reads. 4 where the coder kernels bound the phase (one cluster: they idle
// until the first model kernel has ended, so that one is short); 1 where the model kernels do (several clusters: a short first
// segment is one more table save and load per workgroup - ~10 ms at 100M x 150 with three clusters - and buys nothing, the model
// kernels run back to back from the start anyway). The last segment is a quarter either way: the coder codes it alone.
// (The last segment's coder kernel runs after the last model kernel has ended; a SHORTER last segment - an eighth, a sixteenth - was
// measured slower with three clusters, 197 against 192 ms: the coder kernel of the segment before it, 3776 reads long, then outlasts
// the last model kernel by more than the short tail saves.)
static void coder_segments(uint64_t longest, uint64_t seg_reads, bool overlap, std::vector<std::pair<uint64_t, uint64_t>> &segs, uint32_t first_div = 4) {
    if (overlap && seg_reads >= 256 && longest > seg_reads) {
        const uint64_t q_last = (seg_reads / 4) & ~(uint64_t)63;
        uint64_t q = (seg_reads / (first_div ? first_div : 1)) & ~(uint64_t)63;
        const uint64_t tail_lo = (longest - q_last) & ~(uint64_t)63;             // segment starts stay multiples of 64
        if (q > tail_lo) q = tail_lo;
        const uint64_t mid = tail_lo - q;
        // the middle in nm segments of whole 64-read chunks that differ by one chunk at most (cut as "each, each, .., the rest"
        // the rest could be a fifth of a segment: a table save and load per workgroup for a handful of chunks)
        const uint64_t units = mid / 64;                                          // q and tail_lo are multiples of 64
        uint64_t nm = (units + seg_reads / 64 - 1) / (seg_reads / 64);           // (whole chunks: a seg_reads that is no multiple of 64 is not exceeded either)
        if (nm == 0) nm = 1;
        if (q) segs.push_back({0, q});
        uint64_t lo = q;
        if (first_div == 1 && units >= 4) {
            // Where the model kernels bound the phase the coder kernel of the last middle segment must not outlast the (short) last
            // model kernel, or the last coder kernel starts late: the last middle segment is 0.72 of the others. Three clusters at
            // 100M x 150 (reads per segment, coder phase): 4096 3520 3520 1072: 181.4 ms; 4096 4096 2944 1072: 178.5;
            // 4096 4096 2432 1584: 178.9; 4096 3072 3072 1968: 180.2; 4096 4096 4016: 186.2 (model kernels 149.5 instead of 163 -
            // a segment costs them 14 ms whatever its length - but the last coder kernel then runs alone for 35 ms).
            const uint64_t S = seg_reads / 64;                                    // in 64-read chunks: nm - 1 segments of `each`, the last the rest (~0.72 each)
            nm = (units * 100 + 28 * S + 100 * S - 1) / (100 * S);                // ceil(units / S + 0.28)
            if (nm == 0) nm = 1;
            uint64_t each = nm > 1 ? (units * 100 + (nm * 100 - 28) - 1) / (nm * 100 - 28) : units;
            if (each > S) each = S;
            while (nm > 1 && (nm - 1) * each >= units) --nm;                      // (the last one would be empty)
            for (uint64_t i = 0; i < nm; ++i) {
                const uint64_t len = i + 1 == nm ? tail_lo - lo : each * 64;
                segs.push_back({lo, len});
                lo += len;
            }
        } else {
            const uint64_t base = units / nm, extra = units % nm;                // `extra` segments get one chunk more
            for (uint64_t i = 0; i < nm && lo < tail_lo; ++i) {
                uint64_t len = (base + (i < extra ? 1 : 0)) * 64;
                if (i + 1 == nm || lo + len > tail_lo) len = tail_lo - lo;        // (mid is a multiple of 64: the last one ends on tail_lo)
                if (len) segs.push_back({lo, len});
                lo += len;
            }
        }
        segs.push_back({tail_lo, longest - tail_lo});
        // (Halving the last full segment - and its second half again - so that less of the coder's work is left when the last
        // model kernel ends was tried: every extra segment costs the model kernel a table save and load per workgroup,
        // 3 ms at 100M x 150 with one cluster and 8-10 ms with three, more than the shorter tail gives back:
        // 220.7 -> 221.9 -> 224.3 ms and 349 -> 361 -> 363 ms per step.)
    } else {
        for (uint64_t lo = 0; lo < longest; lo += seg_reads) segs.push_back({lo, lo + seg_reads <= longest ? seg_reads : longest - lo});
    }
}

// The rule above on its own (no device needed): tests and tools look at the cut a call would make. Returns the number of segments;
// the first min(count, cap) are written.
extern "C" uint32_t qvz_coder_segment_list2(uint64_t longest, uint64_t seg_reads, int overlap, uint32_t first_div, uint64_t *first, uint64_t *reads, uint32_t cap);
extern "C" uint32_t qvz_coder_segment_list(uint64_t longest, uint64_t seg_reads, int overlap, uint64_t *first, uint64_t *reads, uint32_t cap) {
    return qvz_coder_segment_list2(longest, seg_reads, overlap, 4, first, reads, cap);
}
extern "C" uint32_t qvz_coder_segment_list2(uint64_t longest, uint64_t seg_reads, int overlap, uint32_t first_div, uint64_t *first, uint64_t *reads, uint32_t cap) {
    std::vector<std::pair<uint64_t, uint64_t>> segs;
    if (!longest || seg_reads < 4) return 0;
    coder_segments(longest, seg_reads, overlap != 0, segs, first_div);
    for (uint32_t i = 0; i < segs.size() && i < cap; ++i) {
        if (first) first[i] = segs[i].first;
        if (reads) reads[i] = segs[i].second;
    }
    return (uint32_t)segs.size();
}

static int encode_planes_impl(bool dry, const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                              uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                              const uint64_t *h_bounds, uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes,
                              void *d_work, uint64_t work_bytes, void *stream);

extern "C" int qvz_hip_encode_planes_bounds(const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                                            uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                            const uint64_t *h_bounds, uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes,
                                            void *d_work, uint64_t work_bytes, void *stream) {
    return encode_planes_impl(false, d_qi, d_st, ld, d_labels, n, columns, clusters, cbs, shards, h_bounds, d_out, cap_per_shard, h_sizes,
                              d_work, work_bytes, stream);
}

// Everything a first qvz_hip_encode_planes(_bounds) call of a process does before its first kernel - the code object of
// this file loaded, the kept buffers sized and allocated, the model / coder stream pair made and probed - without coding
// anything: a one-shot process (the qvz binary) runs this on a host thread beside the quantizer, where a resident process
// simply keeps what its previous encode left. Same n, columns, clusters, codebooks, shards, bounds and workspace size as
// the call that follows.
extern "C" int qvz_hip_encode_planes_prepare(uint64_t n, uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs,
                                             uint32_t shards, const uint64_t *h_bounds, uint64_t work_bytes) {
    uint64_t sizes_dummy = 0;
    return encode_planes_impl(true, nullptr, nullptr, qvz_hip_ld(n), nullptr, n, columns, clusters, cbs, shards, h_bounds, nullptr, 8, &sizes_dummy,
                              work_bytes ? (void *)(uintptr_t)256 : nullptr, work_bytes, nullptr);
}

// What a first coder call of a process needs and no codebook: the stream pair made and probed (~20 ms), the code object of
// this file loaded, the reciprocal check of the loaders. A one-shot process runs it beside its upload (the rest of the set-up -
// qvz_hip_encode_planes_prepare: the kept buffers, which are sized by the codebooks - then fits beside the quantizer).
extern "C" int qvz_hip_encode_planes_warmup(void) {
    hipStream_t pair[2];
    const int how = coder_stream_pair(pair);
    if (how < 0) return how;
    (void)qvz_coder_rcp_one_step_ok();
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_code_records<16, 1>));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_model_ranks<false, false, true>));
    return QVZ_OK;
}

static int encode_planes_impl(bool dry, const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld, const uint8_t *d_labels, uint64_t n,
                              uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                              const uint64_t *h_bounds, uint8_t *d_out, uint64_t cap_per_shard, uint64_t *h_sizes,
                              void *d_work, uint64_t work_bytes, void *stream) {
    if (((!d_qi || !d_st || !d_out) && !dry) || !n || !columns || !clusters || !shards || !h_sizes || ld < n || (ld & 3))
        return qvz_set_error(QVZ_E_ARG, "bad plane-coder arguments");
    if (h_bounds) {                                              // the caller's cut of these n reads (a device's part of a larger file)
        if (h_bounds[0] != 0 || h_bounds[shards] != n) return qvz_set_error(QVZ_E_ARG, "shard bounds must run from 0 to n");
        for (uint32_t g = 0; g < shards; ++g)
            if ((h_bounds[g] & 3) || h_bounds[g + 1] <= h_bounds[g]) return qvz_set_error(QVZ_E_ARG, "shard %u: bounds must ascend in multiples of 4 reads", g);
    }
    if (clusters > 1 && !d_labels && !dry) return qvz_set_error(QVZ_E_ARG, "labels required for clusters > 1");
    if ((cap_per_shard & 3) || cap_per_shard < 8) return qvz_set_error(QVZ_E_ARG, "cap_per_shard must be a multiple of 4, at least 8");
    auto wall_ms = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_entry = wall_ms();
    int area_dev = 0;
    if (hipGetDevice(&area_dev) != hipSuccess || area_dev < 0 || area_dev >= 64) area_dev = 0;
    KeptArea &area = g_coder_area[area_dev];
    std::unique_lock<std::mutex> area_lock(area.mu);             // one call per device at a time uses the kept buffers
    area.begin();
    double t_first_launch = 0.0, t_enqueued = 0.0, t_synced = 0.0, t_tables = 0.0, t_work = 0.0, t_area = 0.0;
    uint32_t spin_us = 0;
    bool odd_call = false;
    const int coder_prio = 1;                                    // the chain waves issue ahead of the throughput kernels that share their CU
    hipStream_t st = (hipStream_t)stream;
    const uint32_t Cx = columns + (clusters > 1 ? (uint32_t)QVZ_CID_SLOTS : 0u);   // record positions per read (k_model_ranks)
    uint32_t *h_arena = nullptr, *h_off = nullptr, words = 0;
    if (qvz_model_build(clusters, columns, cbs, QVZ_MODEL_LIVE_ONLY | QVZ_MODEL_ALIGN16, &h_arena, &h_off, &words))
        return qvz_set_error(QVZ_E_NOMEM, "model build failed");
    uint64_t *h_first = (uint64_t *)malloc((shards + 1) * sizeof(uint64_t));
    if (!h_first) { free(h_arena); free(h_off); return qvz_set_error(QVZ_E_NOMEM, "host alloc"); }
    if (h_bounds) memcpy(h_first, h_bounds, ((size_t)shards + 1) * sizeof(uint64_t));
    else qvz_shard_bounds(n, shards, h_first);
    uint64_t longest = 0;
    for (uint32_t g = 0; g < shards; ++g) if (h_first[g + 1] - h_first[g] > longest) longest = h_first[g + 1] - h_first[g];
    longest = (longest + 3) & ~(uint64_t)3;

    uint32_t *d_tmpl = nullptr, *d_off = nullptr, *d_arena = nullptr;
    uint64_t *d_first = nullptr, *d_sizes = nullptr;
    int32_t *d_status = nullptr;
    CoderState *d_cs = nullptr;
    void *own_work = nullptr;
    int rc = QVZ_OK;
    int32_t status = 0;
    const size_t offn = (size_t)clusters * columns * 2 * A;
    const uint32_t Cp = (Cx + 7) & ~7u;                          // records per read incl. pad, streams padded to 64 (rec_index)
    // packed 8-byte records, the division on the coder's side; 16 symbols per barrier of the coder workgroup
    const uint64_t per_read = (uint64_t)((shards + 63) & ~63u) * Cp * sizeof(PRec);  // record bytes for one read of every stream
    const bool one_step = qvz_coder_rcp_one_step_ok() == 1;      // (hardware reciprocal good enough for one Newton step: checked once per device)
    const void *const cr_fn = one_step ? reinterpret_cast<const void *>(&k_code_records<16, 1>) : reinterpret_cast<const void *>(&k_code_records<16, 2>);
    uint64_t seg_reads = 0, seg_lo = 0;

    // Slices for k_model_ranks: the tables of one (column, cluster) packed back to back, coff[slice][quantizer] = word
    // offset inside the slice. Slice numbering: one cluster: slice = column; several: 0 = the cluster-id table,
    // 1 + column * clusters + cluster. W = columns per workgroup.
    const uint32_t nslices = clusters > 1 ? 1 + columns * clusters : columns;
    std::vector<uint32_t> h_slices, h_colbase(nslices + 1, 0);
    std::vector<uint32_t> h_coff((size_t)nslices * 2 * A, (uint32_t)RK_NONE);   // offset | alphabet size << 16; RK_NONE (size 0): no table
    uint32_t slice_cap = 0, rk_waves = 0;
    size_t rk_lds = 0;
    uint32_t *d_slices = nullptr, *d_colbase = nullptr, *d_save = nullptr;
    uint32_t *d_coff = nullptr;
    uint32_t *d_clist = nullptr, *d_segpos = nullptr, *d_bounds = nullptr;   // several clusters: read lists per (stream, cluster)
    struct RkClass { uint32_t off, count, cap; size_t lds; };    // several clusters: column groups by the LDS their tables need
    std::vector<RkClass> rk_classes;
    std::vector<uint32_t> h_cgmap;
    uint32_t *d_cgmap = nullptr;
    uint64_t *d_cstart = nullptr;
    bool overlap = false;
    hipStream_t s_model = nullptr, s_coder = nullptr;
    bool own_streams = true;                                    // false: the device's kept pair (not destroyed here)
    hipEvent_t ev_start = nullptr;
    std::vector<hipEvent_t> evs;                                // per segment: model start, model end, coder start, coder end
    uint32_t nseg = 0;
    size_t cr_pad = 0;
    std::vector<std::pair<uint64_t, uint64_t>> segs;            // (first read, reads) of every segment, relative to a stream's start
    {
        bool fits = true;
        for (uint32_t sl = 0; sl < nslices; ++sl) {
            h_colbase[sl] = (uint32_t)h_slices.size();
            if (clusters > 1 && sl == 0) {                       // cluster_stats: n = card = k, counts 1
                h_slices.push_back(clusters); h_slices.push_back(clusters);
                for (uint32_t c = 0; c < clusters; ++c) h_slices.push_back(1);
            } else {
                const uint32_t s = clusters > 1 ? (sl - 1) / clusters : sl, c = clusters > 1 ? (sl - 1) % clusters : 0;
                for (uint32_t j = 0; j < 2 * A; ++j) {
                    const uint32_t o = h_off[((size_t)c * columns + s) * 2 * A + j];
                    if (o == NO_TABLE) continue;
                    const size_t rel = h_slices.size() - h_colbase[sl];
                    if (rel >= RK_NONE) { fits = false; continue; }
                    h_coff[(size_t)sl * 2 * A + j] = (uint32_t)rel | h_arena[o + 1] << 16;
                    for (uint32_t i = 0; i < 2 + h_arena[o + 1]; ++i) h_slices.push_back(h_arena[o + i]);
                }
            }
            while (h_slices.size() & 3) h_slices.push_back(0);   // slices start on 16 bytes: they are copied as uint4
            const uint32_t len = (uint32_t)h_slices.size() - h_colbase[sl];
            if (len > slice_cap) slice_cap = len;
        }
        h_colbase[nslices] = (uint32_t)h_slices.size();
        slice_cap = (slice_cap + 3) & ~3u;
        if (fits && !qvz_env_flag("QVZ_CODER", "columns")) {      // (test hook: the lane-per-column kernel, what tables beyond the LDS get)
            // W = columns (waves) per workgroup: records leave as W * 16-byte runs. W = 4 wherever a workgroup fits
            // beside k_code_records' 32 KB of LDS (measured with three clusters, where LDS holds one such workgroup
            // per CU: W = 4 0.40 s, W = 2 0.42 s, W = 1 0.67 s for the model kernels; W = 8 never paid).
            auto need_cap = [&](uint32_t w, uint32_t cap) {      // packed records (+ 1 KB of list rows with several clusters), per wave masks + tables + offsets
                return (size_t)64 * (w + 1) * sizeof(PRec) + (clusters > 1 ? 1024u : 0u) + (size_t)w * (3 * cap + 2 * A) * 4;
            };
            auto need_of = [&](uint32_t w) { return need_cap(w, slice_cap); };
            const size_t budget = 160 * 1024 - 32 * 1024;
            const uint32_t order[3] = {4, 2, 1};
            for (uint32_t w : order) {
                const size_t need = need_of(w);
                if (need > (w == 1 ? (size_t)160 * 1024 : budget)) continue;
                rk_waves = w; rk_lds = need;
                break;
            }
            // (At 100M x 150 -f 0.5 four such workgroups fit a CU, 4 x 40.4 KB: steady beside the 16-symbol coder workgroup, whose 58 KB a
            // brim-full CU could never host anyway and which its head start therefore secures first.)
            // Several clusters - the model kernels bound the phase -: the launch's LDS per workgroup is sized by the LARGEST slice
            // of any (column, cluster), and at 100M x 150 -c 3 -r 1.0 that is an outlier (920 words; 94 % of the 451 slices have
            // 740 or fewer, half of them 364 or fewer): 50 KB, three workgroups to a CU, two beside a coder workgroup. So the column
            // groups are sorted into classes by what their own tables need and every class gets a launch of its own with its own
            // LDS size (`cgmap` tells a workgroup which column group it serves). The class bounds are where one more workgroup fits:
            // a quarter of a CU's 160 KB (four on a CU without a coder workgroup - the 128 registers of the list walk allow no
            // fifth), and a third / a quarter of what a coder workgroup leaves.
            if (rk_waves && clusters > 1) {
                const uint32_t W = rk_waves, ncg = (Cp + W - 1) / W;
                std::vector<uint32_t> capw((size_t)ncg * W, 4);                      // words of the largest slice a wave meets (over the clusters)
                std::vector<size_t> need(ncg, 0);                                    // LDS bytes of the column group's workgroup
                for (uint32_t cg = 0; cg < ncg; ++cg) {
                    uint32_t sum = 0;
                    for (uint32_t w = 0; w < W; ++w) {
                        const uint32_t sx = cg * W + w;
                        uint32_t words = 4;
                        if (clusters == 1) { if (sx < columns) words = h_colbase[sx + 1] - h_colbase[sx]; }
                        else if (sx == 0) words = h_colbase[1] - h_colbase[0];
                        else if (sx >= QVZ_CID_SLOTS && sx - QVZ_CID_SLOTS < columns)
                            for (uint32_t c = 0; c < clusters; ++c) {
                                const uint32_t sl = 1 + (sx - (uint32_t)QVZ_CID_SLOTS) * clusters + c;
                                if (h_colbase[sl + 1] - h_colbase[sl] > words) words = h_colbase[sl + 1] - h_colbase[sl];
                            }
                        capw[(size_t)cg * W + w] = words;                             // (slices are multiples of 4 words)
                        sum += 3 * words + 2 * A;
                    }
                    need[cg] = need_cap(W, 0) - (size_t)W * 2 * A * 4 + (size_t)sum * 4;
                }
                hipFuncAttributes fa;
                const size_t lds_cu = 160 * 1024;
                size_t beside = lds_cu - 58 * 1024;
                if (hipFuncGetAttributes(&fa, cr_fn) == hipSuccess && fa.sharedSizeBytes < lds_cu) beside = lds_cu - fa.sharedSizeBytes;
                const size_t bound[3] = {beside / 4, beside / 3, lds_cu / 4};   // ascending
                std::vector<uint32_t> order(ncg);
                for (uint32_t cg = 0; cg < ncg; ++cg) order[cg] = cg;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return need[a] > need[b]; });   // largest first
                auto class_of = [&](uint32_t cg) { const size_t l = need[cg]; return l <= bound[0] ? 0 : l <= bound[1] ? 1 : l <= bound[2] ? 2 : 3; };
                for (uint32_t i = 0; i < ncg;) {
                    uint32_t j = i;
                    while (j < ncg && class_of(order[j]) == class_of(order[i])) ++j;
                    uint32_t cap = 0;
                    for (uint32_t e = i; e < j; ++e)
                        for (uint32_t w = 0; w < W; ++w) if (capw[(size_t)order[e] * W + w] > cap) cap = capw[(size_t)order[e] * W + w];
                    // (sorted: the class's first needs the most)
                    rk_classes.push_back({i * (1 + W), j - i, cap, need[order[i]]});
                    i = j;
                }
                for (uint32_t i = 0; i < ncg; ++i) {                                  // a launch's entries: column group, then every wave's (offset | words << 16)
                    const uint32_t cg = order[i];
                    h_cgmap.push_back(cg);
                    uint32_t off = 0;
                    for (uint32_t w = 0; w < W; ++w) {
                        const uint32_t words = capw[(size_t)cg * W + w];
                        h_cgmap.push_back(off | words << 16);
                        off += 3 * words + 2 * A;
                    }
                    if (off > 0xFFFFu) { rk_classes.clear(); break; }                 // (cannot happen: 160 KB of LDS are 40 960 words)
                }
            }
        }
    }
#define TRY(expr)                                                                                                    \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) { rc = qvz_set_error(QVZ_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
    t_tables = wall_ms();
    if (!d_work) {                                               // no workspace given: our own, up to 96 GiB of what is free
        work_bytes = per_read * longest;
        size_t mem_free = 0, mem_total = 0;
        uint64_t limit = 4ull << 30;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && (uint64_t)mem_free / 5 * 4 > limit) limit = (uint64_t)mem_free / 5 * 4;
        if (limit > (96ull << 30)) limit = 96ull << 30;
        if (work_bytes > limit) work_bytes = limit;
        if (work_bytes < per_read * 4) work_bytes = per_read * 4;
        TRY(hipMalloc(&own_work, work_bytes));
        d_work = own_work;
    }
    t_work = wall_ms();
    // Segments. With room for two record buffers the model kernel of segment i + 1 runs beside the coder kernel of segment i on a
    // second HIP stream. A segment costs the model kernel a table load and save per workgroup, and the coder kernels idle while the
    // pipeline fills: an eighth of a stream with one cluster (first and last a quarter of that), a third with several - where the model
    // kernels bound the phase and a segment costs them ~10 ms at 100M x 150 - with a FULL first segment (measurements: HISTORY.md).
    seg_reads = (work_bytes / per_read) & ~(uint64_t)3;
    if (seg_reads > longest) seg_reads = longest;
    {
        const uint64_t half = (work_bytes / 2 / per_read) & ~(uint64_t)63;
        if (longest >= 1024 && half >= 64) {
            const uint64_t div = clusters > 1 ? 3 : 8;
            uint64_t target = ((longest + div - 1) / div + 63) & ~(uint64_t)63;
            // (short streams - 10M x 150 over 8192 streams: 1224 reads - are not cut below 512 reads a segment: an eighth would be
            // 192 reads, ten kernel pairs of a millisecond each whose starts and table saves outweigh the shorter fill: 17.1 -> 16.1 ms)
            if (target < 512) target = 512;
            seg_reads = half < target ? half : target;
            overlap = true;
        }
    }
    if (seg_reads < 4 || seg_reads > 0xFFFFFFFCull) { rc = qvz_set_error(QVZ_E_ARG, "workspace of %llu bytes holds fewer than 4 reads of %u streams", (unsigned long long)work_bytes, shards); goto done; }
#ifdef QVZ_PROFILE
    {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rk_prof), z, sizeof(z));
        { const int dr = getenv("QVZ_RK_DRAIN") ? 1 : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rk_drain), &dr, sizeof(dr)); }
        { unsigned long long zc[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cr_prof), zc, sizeof(zc)); }
    }
#endif
    if (getenv("QVZ_CODER_DEBUG"))
        fprintf(stderr, "qvz_hip_encode_planes: %u streams, longest %llu reads, %llu reads/segment, overlap %d, model waves %u, LDS %zu B, slice_cap %u words, total table words %zu\n",
                shards, (unsigned long long)longest, (unsigned long long)seg_reads, (int)overlap, rk_waves, rk_lds, slice_cap, h_slices.size());
    if (getenv("QVZ_CODER_DEBUG"))
        for (size_t ci = 0; ci < rk_classes.size(); ++ci)
            fprintf(stderr, "  model launch %zu: %u column groups, LDS %zu B, largest slice %u words\n", ci, rk_classes[ci].count, rk_classes[ci].lds, rk_classes[ci].cap);
    TRY(area.get((void **)&d_first, (shards + 1) * sizeof(uint64_t)));
    TRY(area.get((void **)&d_sizes, shards * sizeof(uint64_t)));
    TRY(area.get((void **)&d_status, sizeof(int32_t)));
    TRY(area.get((void **)&d_cs, shards * sizeof(CoderState)));
    TRY(hipMemcpyAsync(d_first, h_first, (shards + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    if (rk_waves) {
        TRY(area.get((void **)&d_slices, h_slices.size() * 4));
        TRY(area.get((void **)&d_colbase, h_colbase.size() * 4));
        TRY(area.get((void **)&d_coff, h_coff.size() * 4));
        if (seg_reads < longest) TRY(area.get((void **)&d_save, h_slices.size() * 4 * (size_t)shards));   // table state between segments
        TRY(hipMemcpyAsync(d_slices, h_slices.data(), h_slices.size() * 4, hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(d_colbase, h_colbase.data(), h_colbase.size() * 4, hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(d_coff, h_coff.data(), h_coff.size() * 4, hipMemcpyHostToDevice, st));
        if (!rk_classes.empty()) {
            TRY(area.get((void **)&d_cgmap, h_cgmap.size() * 4));
            TRY(hipMemcpyAsync(d_cgmap, h_cgmap.data(), h_cgmap.size() * 4, hipMemcpyHostToDevice, st));
        }
        TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_model_ranks<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rk_lds));
        TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_model_ranks<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rk_lds));
    } else {                                                     // lane-per-column kernel with per-stream arenas in HBM
        TRY(area.get((void **)&d_tmpl, (size_t)words * 4));
        TRY(area.get((void **)&d_off, offn * 4));
        TRY(area.get((void **)&d_arena, ((size_t)words * shards + 8) * 4));   // + slack: a header read covers 32 bytes
        TRY(hipMemcpyAsync(d_tmpl, h_arena, (size_t)words * 4, hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(d_off, h_off, offn * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_arena_fill, dim3(2048), dim3(256), 0, st, d_tmpl, words, shards, d_arena);
    }
    coder_segments(longest, seg_reads, overlap, segs, clusters > 1 ? 1u : 4u);
    nseg = (uint32_t)segs.size();
    if (rk_waves && clusters > 1) {                              // read lists per (stream, cluster) and where the segments cut them
        std::vector<uint32_t> h_bounds(nseg + 1);
        for (uint32_t i = 0; i < nseg; ++i) h_bounds[i] = (uint32_t)segs[i].first;
        h_bounds[nseg] = 0xFFFFFFFFu;
        const size_t SK = (size_t)shards * clusters;
        TRY(area.get((void **)&d_clist, (size_t)n * sizeof(uint32_t)));
        TRY(area.get((void **)&d_cstart, SK * sizeof(uint64_t)));
        TRY(area.get((void **)&d_segpos, SK * (nseg + 1) * sizeof(uint32_t)));
        TRY(area.get((void **)&d_bounds, (nseg + 1) * sizeof(uint32_t)));
        TRY(hipMemcpyAsync(d_bounds, h_bounds.data(), (nseg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        if (!dry) {
            hipLaunchKernelGGL(k_cluster_lists, dim3(shards), dim3(256), 0, st, d_labels, (const uint64_t *)d_first, clusters, d_clist, d_cstart, d_status);
            hipLaunchKernelGGL(k_cluster_segpos, dim3((unsigned)((SK * (nseg + 1) + 255) / 256)), dim3(256), 0, st, (const uint32_t *)d_clist,
                               (const uint64_t *)d_cstart, (const uint64_t *)d_first, shards, clusters, (const uint32_t *)d_bounds, nseg + 1, d_segpos);
        }
        TRY(hipGetLastError());
        TRY(hipStreamSynchronize(st));                           // h_bounds goes out of scope; the lists are ready for both streams
    }
    t_area = wall_ms();
    evs.assign((size_t)nseg * 4, nullptr);
    for (size_t i = 0; i < evs.size(); ++i) TRY(hipEventCreate(&evs[i]));
    g_planes_ms[0] = g_planes_ms[1] = 0.0;
    s_model = s_coder = st;
    if (overlap) {
        // The two streams are made once per device and kept: the runtime spreads streams over a handful of hardware queues
        // (four by default) and a queue runs its kernels in order, so a pair of streams made afresh for every call lands,
        // every few calls, on the queue of some other stream of the process that is busy at the time - or on one and the
        // same queue, and the model kernel of a segment then waits for the coder kernel of the one before it.
        // (A high-priority queue for the coder kernel was tried: its segments got slower, 225 ms against 167 ms of kernel
        // time per step with three clusters at 100M x 150, and the step with them.)
        {
            hipStream_t pair[2] = {nullptr, nullptr};
            const int how = coder_stream_pair(pair);   // 1: the device's kept streams, 0: none kept for this ordinal, < 0: error (message set)
            if (how < 0) { rc = how; goto done; }
            if (how == 1) {
                s_model = pair[0];
                s_coder = pair[1];
                own_streams = false;
            } else {
                TRY(hipStreamCreateWithFlags(&s_model, hipStreamNonBlocking));
                TRY(hipStreamCreateWithFlags(&s_coder, hipStreamNonBlocking));
            }
        }
        TRY(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
        TRY(hipEventRecord(ev_start, st));                       // uploads and whatever the caller queued before us
        TRY(hipStreamWaitEvent(s_model, ev_start, 0));
        TRY(hipStreamWaitEvent(s_coder, ev_start, 0));
    }
    // The coder kernel is the longer of the two once they overlap, and what slows it is the model waves it shares its CUs
    // with (0.11 s alone, 0.16-0.17 s beside three model workgroups per CU, 0.15 s beside two). So when three or more
    // model workgroups would fit beside a coder workgroup, the coder asks for just enough LDS it never touches to keep
    // the last of them away (measured: -8 ms per step at 100M x 150; more padding than that delays the coder's own
    // dispatch and costs 30 ms).
    if (overlap && rk_waves) {
        hipFuncAttributes fa;
        TRY(hipFuncGetAttributes(&fa, cr_fn));
        const size_t lds_cu = 160 * 1024, beside = (lds_cu - fa.sharedSizeBytes) / rk_lds;
        if (beside >= 3) cr_pad = ((lds_cu - fa.sharedSizeBytes - beside * rk_lds) + 1024) & ~(size_t)255;
    }
    if (cr_pad) TRY(hipFuncSetAttribute(cr_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cr_pad));
    if (dry) {                                                   // qvz_hip_encode_planes_prepare: everything is in place for the call that codes
        TRY(hipStreamSynchronize(st));
        goto done;
    }
    t_first_launch = wall_ms();
    // the coder kernel's head start over the next model kernel (below). 40 us are enough where the model's workgroups take a CU's
    // LDS between them slowly (three clusters); with one cluster the model kernel's workgroups - since they ask for their
    // tables and their first bytes at once - fill the chip faster than 128 coder workgroups find their CUs: at 10M x 150 the coder
    // kernels took 15 or 18-20 ms from step to step with 40 us, 15.5 with 200 (step 43-44 -> 41.7-42.1 ms; 100M x 150: no change)
    spin_us = clusters > 1 ? 40u : 200u;
    for (uint32_t i = 0; i < nseg; ++i) {
        seg_lo = segs[i].first;
        const uint32_t seg_len = (uint32_t)segs[i].second;       // both kernels lay the segment's records out for this many reads
        const uint32_t mblocks = (uint32_t)(((uint64_t)shards * (columns + (clusters > 1 ? 1u : 0u)) + 255) / 256);
        const int is_first = i == 0, is_last = i + 1 == nseg;
        Record *buf = (Record *)((char *)d_work + (overlap ? (size_t)(i & 1) * seg_reads * per_read : 0));
        hipEvent_t *e = &evs[(size_t)i * 4];
        // Head start for the coder kernel of the segment before: it is released by an event of the model kernel that has just ended,
        // across queues, while the next model kernel follows in its own queue at once - and a model kernel that has filled the CUs
        // first leaves the coder workgroups no room. A one-wave kernel that spins for a few tens of microseconds in front of the model
        // kernel - BEHIND the wait for the buffer, or both kernels are released at the same instant - lets them onto the CUs first.
        if (overlap && i >= 2) TRY(hipStreamWaitEvent(s_model, evs[(size_t)(i - 2) * 4 + 3], 0));   // the coder has drained this buffer
        if (overlap && i >= 1 && spin_us) hipLaunchKernelGGL(k_coder_spin_us, dim3(1), dim3(64), 0, s_model, spin_us);
        TRY(hipEventRecord(e[0], s_model));
        if (rk_waves) {
            // one workgroup serves all clusters of its (stream, column group), one after the other (gridDim.z == 1)
            const dim3 block(64 * rk_waves);
            const size_t SK = (size_t)shards * clusters;
            // (several clusters: one launch per size class of the column groups, the largest tables first)
            const size_t nlaunch = rk_classes.empty() ? 1 : rk_classes.size();
            for (size_t ci = 0; ci < nlaunch; ++ci) {
                const uint32_t ny = rk_classes.empty() ? (Cp + rk_waves - 1) / rk_waves : rk_classes[ci].count;
                const uint32_t cap_l = rk_classes.empty() ? slice_cap : rk_classes[ci].cap;
                const size_t lds_l = rk_classes.empty() ? rk_lds : rk_classes[ci].lds;
                const uint32_t *map_l = rk_classes.empty() ? nullptr : d_cgmap + rk_classes[ci].off;
                const dim3 grid(shards, ny, 1);
#define RK_ARGS(LISTS) d_qi, d_st, ld, d_labels, columns, clusters, d_first, d_coff, d_colbase, d_slices, d_save, (uint32_t)h_slices.size(), cap_l, seg_lo, \
                       seg_len, is_first, is_last, (uint32_t)__builtin_ctz(rk_waves), buf, d_status, \
                       (const uint32_t *)(LISTS ? d_clist : nullptr), (const uint64_t *)(LISTS ? d_cstart : nullptr), \
                       (const uint32_t *)(LISTS ? d_segpos + (size_t)i * SK : nullptr), (const uint32_t *)(LISTS ? d_segpos + (size_t)(i + 1) * SK : nullptr), map_l
                if (clusters > 1) hipLaunchKernelGGL((k_model_ranks<true, false, true>), grid, block, lds_l, s_model, RK_ARGS(true));
                else hipLaunchKernelGGL((k_model_ranks<false, false, true>), grid, block, lds_l, s_model, RK_ARGS(false));
#undef RK_ARGS
            }
        } else {
#define MC_ARGS d_qi, d_st, ld, d_labels, columns, clusters, d_first, shards, d_off, d_arena, words, seg_lo, seg_len, buf, d_status
            if (clusters > 1) hipLaunchKernelGGL((k_model_columns<true>), dim3(mblocks), dim3(256), 0, s_model, MC_ARGS);
            else hipLaunchKernelGGL((k_model_columns<false>), dim3(mblocks), dim3(256), 0, s_model, MC_ARGS);
#undef MC_ARGS
        }
        TRY(hipEventRecord(e[1], s_model));
        if (overlap) TRY(hipStreamWaitEvent(s_coder, e[1], 0));
        TRY(hipEventRecord(e[2], s_coder));
#define CR_ARGS (const Record *)buf, Cx, d_first, shards, seg_lo, seg_len, is_first, is_last, d_cs, d_out, cap_per_shard, d_sizes, d_status, coder_prio
        if (one_step) hipLaunchKernelGGL((k_code_records<16, 1>), dim3((shards + 63) / 64), dim3(384), cr_pad, s_coder, CR_ARGS);
        else hipLaunchKernelGGL((k_code_records<16, 2>), dim3((shards + 63) / 64), dim3(384), cr_pad, s_coder, CR_ARGS);
#undef CR_ARGS
        TRY(hipEventRecord(e[3], s_coder));
    }
    TRY(hipGetLastError());
    t_enqueued = wall_ms();
    if (overlap) TRY(hipStreamWaitEvent(st, evs[(size_t)(nseg - 1) * 4 + 3], 0));   // the last coder kernel ends after everything else
    TRY(hipMemcpyAsync(h_sizes, d_sizes, shards * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    TRY(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // the wait polls an event (a load of the completion signal every 20 us): it does not depend on interrupt delivery
    {
        hipEvent_t ev_done = nullptr;
        TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        hipError_t qe = hipEventRecord(ev_done, st);
        while (qe == hipSuccess) {
            qe = hipEventQuery(ev_done);
            if (qe != hipErrorNotReady) break;
            struct timespec nap = {0, 20000};                     // 20 us
            nanosleep(&nap, nullptr);
            qe = hipSuccess;
        }
        (void)hipEventDestroy(ev_done);
        if (qe != hipSuccess) { rc = qvz_set_error(QVZ_E_HIP, "waiting for the coder kernels -> %s", hipGetErrorString(qe)); goto done; }
    }
    TRY(hipStreamSynchronize(st));                               // (returns at once: everything on st has completed)
    t_synced = wall_ms();
    {
        // A call that took much longer than its kernels (seen a few times in a thousand: 0.1 - 3.9 s, kernel times as always)
        // prints its time line whether asked to or not, so that the gap can be placed
        float span = 0.f, busy_m = 0.f, busy_c = 0.f;
        TRY(hipEventElapsedTime(&span, evs[0], evs[(size_t)(nseg - 1) * 4 + 3]));
        for (uint32_t i = 0; i < nseg; ++i) {
            float ta = 0.f, tb = 0.f;
            TRY(hipEventElapsedTime(&ta, evs[(size_t)i * 4], evs[(size_t)i * 4 + 1]));
            TRY(hipEventElapsedTime(&tb, evs[(size_t)i * 4 + 2], evs[(size_t)i * 4 + 3]));
            busy_m += ta; busy_c += tb;
        }
        const double host_span = t_synced - t_first_launch;
        odd_call = host_span > 1.25 * (busy_m + busy_c) + 30.0;
        if (odd_call)
            fprintf(stderr, "qvz_hip_encode_planes: this call took %.1f ms on the host (%.1f ms from the first kernel's start to the last one's end) for %.1f + %.1f ms of model and coder kernels; its time line:\n",
                    host_span, span, busy_m, busy_c);
    }
    for (uint32_t i = 0; i < nseg; ++i) {                        // busy time of either kernel family (they overlap in time)
        float ta = 0.f, tb = 0.f;
        TRY(hipEventElapsedTime(&ta, evs[(size_t)i * 4], evs[(size_t)i * 4 + 1]));
        TRY(hipEventElapsedTime(&tb, evs[(size_t)i * 4 + 2], evs[(size_t)i * 4 + 3]));
        g_planes_ms[0] += ta; g_planes_ms[1] += tb;
        if (odd_call || getenv("QVZ_CODER_DEBUG")) {             // the segment's kernels on the time line of the call
            float m0 = 0.f, c0 = 0.f;
            TRY(hipEventElapsedTime(&m0, evs[0], evs[(size_t)i * 4]));
            TRY(hipEventElapsedTime(&c0, evs[0], evs[(size_t)i * 4 + 2]));
            fprintf(stderr, "  segment %u (%llu reads): model %.2f .. %.2f ms, coder %.2f .. %.2f ms\n", i, (unsigned long long)segs[i].second,
                    m0, m0 + ta, c0, c0 + tb);
        }
    }
#ifdef QVZ_PROFILE
    {
        unsigned long long z[16] = {0};
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rk_prof), sizeof(z));
        const double ch = z[6] ? (double)z[6] : 1.0;
        unsigned long long cr[16] = {0};
        (void)hipMemcpyFromSymbol(cr, HIP_SYMBOL(g_cr_prof), sizeof(cr));
        const double nb = cr[8] ? (double)cr[8] : 1.0;
        fprintf(stderr, "k_code_records cycles per block of symbols, busy/waiting: interval %.0f/%.0f  bits %.0f/%.0f  bytes %.0f/%.0f  loader %.0f/%.0f\n",
                cr[0] / nb, cr[1] / nb, cr[2] / nb, cr[3] / nb, cr[4] / nb, cr[5] / nb, cr[6] / nb, cr[7] / nb);
        fprintf(stderr, "  cycles of that waiting spent in s_waitcnt lgkmcnt(0) before the barrier: interval %.0f  bits %.0f  bytes %.0f  loader %.0f\n", cr[9] / nb, cr[10] / nb, cr[11] / nb, cr[12] / nb);
        {
            uint32_t simd[8] = {0};
            (void)hipMemcpyFromSymbol(simd, HIP_SYMBOL(g_cr_simd), sizeof(simd));
            fprintf(stderr, "  second loader %.0f/%.0f, third loader busy %.0f; SIMD of the workgroup's waves 0..5 (loader, bytes, bits, interval, second loader, third loader): %u %u %u %u %u %u\n", cr[13] / nb, cr[14] / nb, cr[15] / nb,
                    simd[0], simd[1], simd[2], simd[3], simd[4], simd[5]);
        }
        fprintf(stderr, "k_model_ranks cycles per wave and chunk: lookup %.0f  ranks %.0f  update+record %.0f  barrier1 %.0f  park+barrier2 %.0f  drain %.0f  (%.3g wave-chunks)\n",
                z[0] / ch, z[1] / ch, z[2] / ch, z[3] / ch, z[4] / ch, z[5] / ch, ch);
        if (z[9]) fprintf(stderr, "  per pass of a workgroup (a cluster's reads of a stream in a segment): %.0f cycles before the first chunk, %.0f for the table save, %.0f in all, %.1f chunks\n",
                          (double)z[7] / z[9], (double)z[8] / z[9], (double)z[10] / z[9], ch / z[9]);
        if (z[9]) fprintf(stderr, "    before the first chunk: %.0f until the wave's tables are in LDS, %.0f at the barrier, %.0f for the first bytes and look-up (sums over all passes / the passes; a workgroup's first passes alone: %.0f in all, summed over every pass)\n",
                          (double)z[11] / z[9], (double)z[12] / z[9], (double)z[13] / z[9], (double)z[14] / z[9]);
        if (z[9] && z[15]) fprintf(stderr, "    QVZ_RK_DRAIN: of the cycles until the tables are in LDS, %.0f drain what was in flight before the tables were asked for\n", (double)z[15] / z[9]);
    }
#endif
    if (status & 1) rc = qvz_set_error(QVZ_E_DEAD_CTX, "coder input addresses a quantizer without a model table (dead context or bad state)");
    else if (status & 2) rc = qvz_set_error(QVZ_E_NOMEM, "a shard's stream exceeded cap_per_shard=%llu bytes", (unsigned long long)cap_per_shard);
done:
#undef TRY
    for (size_t i = 0; i < evs.size(); ++i) if (evs[i]) (void)hipEventDestroy(evs[i]);
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (s_model && s_model != st) { (void)hipStreamSynchronize(s_model); if (own_streams) (void)hipStreamDestroy(s_model); }
    if (s_coder && s_coder != st) { (void)hipStreamSynchronize(s_coder); if (own_streams) (void)hipStreamDestroy(s_coder); }
    (void)hipFree(own_work);
    area.finish();                                               // (the buffers of this call stay allocated for the next one)
    area_lock.unlock();
    if (t_synced > 0.0 && (getenv("QVZ_CODER_DEBUG") || wall_ms() - t_entry > 1.25 * (g_planes_ms[0] + g_planes_ms[1]) + 30.0))
        fprintf(stderr, "  host side of the call: set-up %.2f ms (model tables on the host %.2f, workspace %.2f, buffers + uploads %.2f, events + streams %.2f), launches enqueued in %.2f ms, waited %.2f ms, tear-down %.2f ms\n",
                t_first_launch - t_entry, t_tables - t_entry, t_work - t_tables, t_area - t_work, t_first_launch - t_area,
                t_enqueued - t_first_launch, t_synced - t_enqueued, wall_ms() - t_synced);
    free(h_arena); free(h_off); free(h_first);
    return rc;
}

// ---- the adaptive model of ONE stream on the GPU, chunk by chunk (strict single-stream mode) ---------------------------
// The reference's .q holds one serial arithmetic stream (qv_stream.c:109-110): the interval chain (arith.c:35-97) cannot
// be spread over lanes, but its model can: update_stats touches one table, a table belongs to one column
// (qv_stream.c:9-25,46-53), so k_model_ranks runs the model of the whole file with a wave per column, and what is left to
// the one host core that owns the chain is two multiply-adds, the renormalisation and the bit output per symbol
// (host/qvz_coder.c, qvz_coder_encode_records) - no table look-ups, no cumulative sums, no division.
// A session carries the tables from chunk to chunk (`save`); records of a chunk are laid out for one stream alone:
//   record (row, pos) at ((row / 2) * (Cp / 4) + pos / 4) * 8 + (row % 2) * 4 + pos % 4,   Cp = positions per read rounded up to 8
// (positions: one cluster: the columns; several: 0 = cluster id, 1..7 empty, 8 + column).
struct qvz_model_session {
    uint32_t columns, clusters, Cx, Cp, nslices, slice_cap, rk_waves, total_words;
    size_t rk_lds;
    uint32_t *d_slices, *d_colbase, *d_coff, *d_save;
    uint64_t *d_first, *d_cstart;
    uint32_t *d_clist, *d_segpos, *d_bounds;
    int32_t *d_status;
    uint64_t list_cap;           // reads d_clist holds
    int started;                 // 0 until the first chunk has run (it starts from the template tables)
};

extern "C" void qvz_hip_model_session_close(qvz_model_session *s) {
    if (!s) return;
    (void)hipFree(s->d_slices); (void)hipFree(s->d_colbase); (void)hipFree(s->d_coff); (void)hipFree(s->d_save);
    (void)hipFree(s->d_first); (void)hipFree(s->d_cstart); (void)hipFree(s->d_clist); (void)hipFree(s->d_segpos);
    (void)hipFree(s->d_bounds); (void)hipFree(s->d_status);
    delete s;
}

extern "C" int qvz_hip_model_session_open(uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs,
                                          qvz_model_session **out) {
    if (!columns || !clusters || !cbs || !out) return qvz_set_error(QVZ_E_ARG, "bad model-session arguments");
    uint32_t *h_arena = nullptr, *h_off = nullptr, words = 0;
    if (qvz_model_build(clusters, columns, cbs, QVZ_MODEL_ALIGN16, &h_arena, &h_off, &words))   // every context: a single stream may meet any
        return qvz_set_error(QVZ_E_NOMEM, "model build failed");
    qvz_model_session *s = new qvz_model_session();
    memset(s, 0, sizeof(*s));
    s->columns = columns; s->clusters = clusters;
    s->Cx = columns + (clusters > 1 ? (uint32_t)QVZ_CID_SLOTS : 0u);
    s->Cp = (s->Cx + 7) & ~7u;
    s->nslices = clusters > 1 ? 1 + columns * clusters : columns;
    std::vector<uint32_t> h_slices, h_colbase(s->nslices + 1, 0);
    std::vector<uint32_t> h_coff((size_t)s->nslices * 2 * A, (uint32_t)RK_NONE);
    bool fits = true;
    for (uint32_t sl = 0; sl < s->nslices; ++sl) {              // the slices of qvz_hip_encode_planes
        h_colbase[sl] = (uint32_t)h_slices.size();
        if (clusters > 1 && sl == 0) {
            h_slices.push_back(clusters); h_slices.push_back(clusters);
            for (uint32_t c = 0; c < clusters; ++c) h_slices.push_back(1);
        } else {
            const uint32_t col = clusters > 1 ? (sl - 1) / clusters : sl, c = clusters > 1 ? (sl - 1) % clusters : 0;
            for (uint32_t j = 0; j < 2 * A; ++j) {
                const uint32_t o = h_off[((size_t)c * columns + col) * 2 * A + j];
                if (o == NO_TABLE) continue;
                const size_t rel = h_slices.size() - h_colbase[sl];
                if (rel >= RK_NONE) { fits = false; continue; }
                h_coff[(size_t)sl * 2 * A + j] = (uint32_t)rel | h_arena[o + 1] << 16;
                for (uint32_t i = 0; i < 2 + h_arena[o + 1]; ++i) h_slices.push_back(h_arena[o + i]);
            }
        }
        while (h_slices.size() & 3) h_slices.push_back(0);
        const uint32_t len = (uint32_t)h_slices.size() - h_colbase[sl];
        if (len > s->slice_cap) s->slice_cap = len;
    }
    h_colbase[s->nslices] = (uint32_t)h_slices.size();
    s->slice_cap = (s->slice_cap + 3) & ~3u;
    s->total_words = (uint32_t)h_slices.size();
    free(h_arena); free(h_off);
    auto need_of = [&](uint32_t w) { return (size_t)64 * (w + 1) * sizeof(Record) + (size_t)w * (3 * s->slice_cap + 2 * A) * 4; };
    const uint32_t order[3] = {4, 2, 1};
    for (uint32_t w : order)
        if (fits && need_of(w) <= (size_t)160 * 1024) { s->rk_waves = w; s->rk_lds = need_of(w); break; }
    if (!s->rk_waves) {
        delete s;
        return qvz_set_error(QVZ_E_NOMEM, "a column's tables (%u words) do not fit LDS", s->slice_cap);
    }
    int rc = QVZ_OK;
#define TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess && rc == QVZ_OK) rc = qvz_set_error(QVZ_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); } while (0)
    TRY(hipMalloc(&s->d_slices, h_slices.size() * 4));
    TRY(hipMalloc(&s->d_colbase, h_colbase.size() * 4));
    TRY(hipMalloc(&s->d_coff, h_coff.size() * 4));
    TRY(hipMalloc(&s->d_save, h_slices.size() * 4));
    TRY(hipMalloc(&s->d_first, 2 * sizeof(uint64_t)));
    TRY(hipMalloc(&s->d_status, sizeof(int32_t)));
    if (clusters > 1) {
        TRY(hipMalloc(&s->d_cstart, clusters * sizeof(uint64_t)));
        TRY(hipMalloc(&s->d_segpos, (size_t)clusters * 2 * sizeof(uint32_t)));
        TRY(hipMalloc(&s->d_bounds, 2 * sizeof(uint32_t)));
    }
    if (rc == QVZ_OK) {
        const uint32_t bounds[2] = {0u, 0xFFFFFFFFu};
        TRY(hipMemcpy(s->d_slices, h_slices.data(), h_slices.size() * 4, hipMemcpyHostToDevice));
        TRY(hipMemcpy(s->d_colbase, h_colbase.data(), h_colbase.size() * 4, hipMemcpyHostToDevice));
        TRY(hipMemcpy(s->d_coff, h_coff.data(), h_coff.size() * 4, hipMemcpyHostToDevice));
        TRY(hipMemset(s->d_status, 0, sizeof(int32_t)));
        TRY(hipMemset(s->d_first, 0, 2 * sizeof(uint64_t)));
        if (clusters > 1) TRY(hipMemcpy(s->d_bounds, bounds, sizeof(bounds), hipMemcpyHostToDevice));
        TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_model_ranks<true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->rk_lds));
        TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_model_ranks<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->rk_lds));
    }
#undef TRY
    if (rc != QVZ_OK) { qvz_hip_model_session_close(s); return rc; }
    *out = s;
    return QVZ_OK;
}

extern "C" uint32_t qvz_hip_model_session_pitch(const qvz_model_session *s) { return s ? s->Cp : 0; }

// The records of the next n reads of the stream (their planes start at index 0 of d_qi / d_st / d_labels) into d_rec
// (n rounded up to 2, times pitch, times 16 bytes). Asynchronous on `stream`; chunks must be issued in file order on
// one stream. *d_status_out (device, may be NULL): |= 1 if a symbol had no table.
extern "C" int qvz_hip_model_session_records(qvz_model_session *s, const uint8_t *d_qi, const uint8_t *d_st, uint64_t ld,
                                             const uint8_t *d_labels, uint64_t n, void *d_rec, void *stream) {
    if (!s || !d_qi || !d_st || !n || !d_rec || ld < n || n > 0xFFFFFFF0ull) return qvz_set_error(QVZ_E_ARG, "bad model-session chunk");
    if (s->clusters > 1 && !d_labels) return qvz_set_error(QVZ_E_ARG, "labels required for clusters > 1");
    hipStream_t st = (hipStream_t)stream;
    // first[] = {0, n} for the kernels: the value travels inside the command (n < 2^32: the high word stays zero)
    QVZ_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(s->d_first + 1), (int)(uint32_t)n, 1, st));
    const dim3 grid(1, (s->Cp + s->rk_waves - 1) / s->rk_waves, s->clusters), block(64 * s->rk_waves);
    const int is_first = s->started ? 0 : 1;
    if (s->clusters > 1) {
        if (n > s->list_cap) {
            if (s->d_clist) { QVZ_HIP_CHECK(hipStreamSynchronize(st)); (void)hipFree(s->d_clist); s->d_clist = nullptr; }
            QVZ_HIP_CHECK(hipMalloc(&s->d_clist, (size_t)n * sizeof(uint32_t)));
            s->list_cap = n;
        }
        hipLaunchKernelGGL(k_cluster_lists, dim3(1), dim3(256), 0, st, d_labels, (const uint64_t *)s->d_first, s->clusters, s->d_clist, s->d_cstart, s->d_status);
        hipLaunchKernelGGL(k_cluster_segpos, dim3((s->clusters * 2 + 255) / 256), dim3(256), 0, st, (const uint32_t *)s->d_clist,
                           (const uint64_t *)s->d_cstart, (const uint64_t *)s->d_first, 1u, s->clusters, (const uint32_t *)s->d_bounds, 2u, s->d_segpos);
        hipLaunchKernelGGL((k_model_ranks<true, true, false>), grid, block, s->rk_lds, st, d_qi, d_st, ld, d_labels, s->columns, s->clusters,
                           (const uint64_t *)s->d_first, s->d_coff, s->d_colbase, s->d_slices, s->d_save, s->total_words, s->slice_cap,
                           (uint64_t)0, (uint32_t)n, is_first, 0, (uint32_t)__builtin_ctz(s->rk_waves), (Record *)d_rec, s->d_status,
                           (const uint32_t *)s->d_clist, (const uint64_t *)s->d_cstart, (const uint32_t *)s->d_segpos,
                           (const uint32_t *)(s->d_segpos + s->clusters), (const uint32_t *)nullptr);
    } else {
        hipLaunchKernelGGL((k_model_ranks<false, true, false>), grid, block, s->rk_lds, st, d_qi, d_st, ld, d_labels, s->columns, s->clusters,
                           (const uint64_t *)s->d_first, s->d_coff, s->d_colbase, s->d_slices, s->d_save, s->total_words, s->slice_cap,
                           (uint64_t)0, (uint32_t)n, is_first, 0, (uint32_t)__builtin_ctz(s->rk_waves), (Record *)d_rec, s->d_status,
                           (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    }
    QVZ_HIP_CHECK(hipGetLastError());
    s->started = 1;
    return QVZ_OK;
}

// status bits gathered so far (1: a symbol addressed a quantizer without a table / a state outside its alphabet); synchronises `stream`
extern "C" int qvz_hip_model_session_status(qvz_model_session *s, void *stream) {
    int32_t v = 0;
    if (!s) return QVZ_E_ARG;
    QVZ_HIP_CHECK(hipMemcpyAsync(&v, s->d_status, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    QVZ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return v ? qvz_set_error(QVZ_E_DEAD_CTX, "coder input addresses a quantizer without a model table (dead context or bad state)") : QVZ_OK;
}
