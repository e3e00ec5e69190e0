// qvz_decoder_gpu.hip - decoder of the sharded streams on gfx950: one LANE per stream.
//
// Reference: start_qv_decompression (src/qv_compressor.c:145-231), arithmetic_decoder_step (src/arith.c:118-188),
// decoder_last_step (src/arith.c:190-205), update_stats (src/qv_stream.c:9-25), stream_read_bit (src/os_stream.c:36-52).
//
// Unlike the encoder, the decoder cannot take the adaptive model out of the serial chain: which table a symbol is
// decoded with depends on the previous decoded value (the context) and the table's counts decide the symbol. So this
// is the plain "one lane = one stream, model in the loop" design: every lane walks its stream's reads, looks up the
// context of the previous value, picks lo/hi with the same WELL draw the encoder used (the draw matrix of
// qvz_hip_well_draws), decodes one state with the table of (column, 2*ctx+b) in its private model arena (HBM),
// updates that table and maps the state back to a quality value with the quantizer's output alphabet. Per-symbol time
// is a chain of dependent memory round trips (context -> table -> output alphabet); throughput comes from the number
// of streams. Decoded values are collected in LDS, 4 reads x all columns per lane, and leave as dwords of the
// column-major matrix qv[s*ld + i] (the layout of the encoder's `qv` plane, so qvz_hip_untranspose_text gives the text).
//
// Same integers as the reference: sub = ((t - l + 1) * n - 1) / range by exact division, symbol = first k with
// cum_k > sub, l/u by floor(range * cum / n), E1/E2 and E3 renormalisation batched by count-leading-zeros (the
// reference's loop does all E1/E2 steps, then all E3 steps, never back), the tag follows the same shifts with fresh
// bits from the stream (zero past its end), and the last symbol of a stream goes through decoder_last_step: no state
// change, 64-bit comparison.
#include <hip/hip_runtime.h>

#include <vector>

#include <stdint.h>
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

// 1 / d to the last bit or two: v_rcp_f64 (good to 2^-23 of the value: a seed) and two Newton steps, the residual exact in an fma
__device__ __forceinline__ double dec_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(e, r, r);
}
#define DEC_QUOT_LO 5.9604644775390625e-08    /* 2^-24 */
#define DEC_QUOT_HI 0.99999994039535522461    /* 1 - 2^-24 */

#ifdef QVZ_PROFILE          // development build: shader cycles of one wave per section of the symbol chain
__device__ unsigned long long g_dec_prof[8];
#define DP_NOW() __builtin_readcyclecounter()
#else
#define DP_NOW() 0ull
#endif

typedef uint32_t dec_u32x4 __attribute__((ext_vector_type(4)));

// The interval is kept as the encoder's chain keeps it (qvz_coder_gpu.hip, Record): l and ~u left-aligned in 32 bits - l32 = l << 10,
// nu32 = (~u & MASK) << 10 - so that (u - l + 1) * 1024 = 2^32 - (l32 + nu32), the shared leading bits of l and u are the leading
// zeros of ~(l32 ^ nu32), the E3 positions those of ~(l32 & nu32) below the top bit, and a shift keeps the invariants without
// masks; the tag the same way (t32 = t << 10). Every quotient is a product with a reciprocal made to the last bits:
//   sub = floor((gap n - 1) / R)       = trunc((gap n - 1/2) / R): the true value is a multiple of 1 / R (R <= 2^22), the shifted one
//                                        at least 2^-23 away from an integer on either side, the computed one within 2^-30 of it
//   l  += floor(R cum_lo / n),  ~u += ceil(R (n - cum_hi) / n): the encoder's two fma, its constants and its proof (there)
// instead of three divisions with exact integer remainders (35 instructions each, of the ~450 a symbol took).
struct LaneDecoder {
    uint32_t l32, nu32, t32;
    double rd, rr;           // (u - l + 1) * 1024 and its reciprocal: made at the end of a symbol's interval update - behind the next
                             // table's round trip -, not at the top of the next symbol, where that table's search waits for them
    uint64_t buf;            // next bits of the stream, left-aligned
    uint32_t avail;          // valid bits in buf
    const uint8_t *src;
    uint64_t pos, size;      // next byte to load (multiple of 4), stream length
    // the dword behind `buf` is requested when its predecessor is consumed and looked at a refill later (~23 symbols at 1.4 bits
    // per symbol): a load whose value is needed at once would also wait - loads return in order - for the next symbol's table
    // header, which is requested before the interval update so that its round trip passes behind that work
    uint32_t next_raw, next_left;   // the raw dword and how many of its bytes belong to the stream (0: past the end, nothing loaded)
#ifdef QVZ_PROFILE
    unsigned long long c_head, c_rest, c_n;
#endif
};

__device__ __forceinline__ void ld_request(LaneDecoder &d) {     // the stream's next dword -> next_raw (zeros past the end, os_stream.c:36-52)
    const bool in = d.pos < d.size;
    const uint64_t left = d.size - d.pos;
    d.next_left = !in ? 0u : left < 4 ? (uint32_t)left : 4u;
    // one unconditional load (past the end: the stream's first dword, which nobody looks at): a load under a condition was a second
    // register for the dword and a copy - a use, so a wait - right behind the load. (The area behind a stream is readable.)
    d.next_raw = *reinterpret_cast<const uint32_t *>(d.src + (in ? d.pos : 0));
    d.pos += 4;
}

// Called once per symbol, behind the request for the next symbol's table header; the dword it consumes was requested at the refill
// before - before this symbol's header was waited for, and loads return in order -: no wait. Afterwards avail >= 32: enough for the E1/E2 bits of the symbol (22 at most); the E3
// bits (21 at most) find avail >= 21 unless E1/E2 took twelve or more, and ld_back refills then (with a wait) before it takes them.
__device__ __forceinline__ void ld_refill(LaneDecoder &d) {
    if (d.avail <= 32) {
        uint32_t w = __builtin_bswap32(d.next_raw);
        w = d.next_left >= 4 ? w : d.next_left ? w & (~0u << (8 * (4 - d.next_left))) : 0u;
        d.buf |= (uint64_t)w << (32 - d.avail);
        d.avail += 32;
        __builtin_amdgcn_sched_barrier(0);                       // the old dword is used up: the new one may take its register
        ld_request(d);
    }
}

__device__ __forceinline__ uint32_t ld_take(LaneDecoder &d, uint32_t nb) {   // nb <= 22 <= avail
    const uint32_t v = nb ? (uint32_t)(d.buf >> (64 - nb)) : 0u;
    d.buf = nb ? d.buf << nb : d.buf;
    d.avail -= nb;
    return v;
}

// Decoder tables are laid out, 64 bytes aligned,
//   [n][card][recon 0..7, a byte each][counts 0 .. card-1, zeros up to twelve][counts 12 ..][recon 8 .. card-1]
// so that the 64 bytes the decode needs anyway - four independent 16-byte loads, ONE memory round trip for a lane whose
// table lies in HBM or the Infinity Cache, one half of a 128-byte line - also hold the quality values of the first eight states:
// with the output alphabet behind the counts, in another cache line, mapping the state back to a value was a second dependent
// round trip for every symbol (4.2 s for 100M x 150 reads; HISTORY.md section 6b). The zeros behind a short table's counts let
// the search run over twelve counts without asking where the table ends.
struct DecHeader { dec_u32x4 a, b, c, d; };
__device__ __forceinline__ DecHeader ld_header(const uint32_t *t) {
    DecHeader h;
    h.a = reinterpret_cast<const dec_u32x4 *>(t)[0]; h.b = reinterpret_cast<const dec_u32x4 *>(t)[1];
    h.c = reinterpret_cast<const dec_u32x4 *>(t)[2]; h.d = reinterpret_cast<const dec_u32x4 *>(t)[3];
    return h;
}
__device__ __forceinline__ uint32_t dec_counts_end(uint32_t card) { return 4 + (card > 12 ? card : 12u); }   // where recon 8.. begins, in words

// A symbol is decoded in three parts so that the NEXT symbol's table header can be requested as soon as this symbol's value -
// the next context - is known, and its memory round trip (2 300 of a symbol's 4 800 cycles when it was waited for at the top of
// the next symbol: profiles/r05_decoder_cycles.txt) passes behind the interval update:
//   ld_front   the state x of the tag under the table (arith.c:134-139), its value, the counts below it (lo) and its own (fx)
//   ld_update  update_stats (qv_stream.c:12-24): the table's two stores, the halving when n passes the limit
//   ld_back    the interval and the tag after the symbol (arith.c:146-181)
// All three are straight-line code for every lane - a lane without a symbol to decode works on the tables of a stream that
// does not exist (the arena's last slot), its results go nowhere -: per-lane conditions around them cost more scalar mask
// instructions than they saved vector ones (a lone wave pays every instruction with its issue latency). Branches are left where
// lanes rarely go: tables of more than twelve states, states beyond the eighth, the halving, the stream's next dword.
// Returns the state; *ok = false when the tag runs off the table (the counts of a table add up to n and the tag lies inside the
// interval whatever the stream's bits are, so this is a table or an interval nobody made).
__device__ __forceinline__ uint32_t ld_front(const LaneDecoder &d, const DecHeader &h, const uint32_t *__restrict__ t,
                                             uint32_t *val, uint32_t *lo_out, uint32_t *fx_out, bool *ok) {
    const uint32_t n = h.a.x, card = h.a.y;
    const uint32_t cn[12] = {h.b.x, h.b.y, h.b.z, h.b.w, h.c.x, h.c.y, h.c.z, h.c.w, h.d.x, h.d.y, h.d.z, h.d.w};
    const double nd = (double)n;
    const double num = __builtin_fma((double)(d.t32 - d.l32), nd, __builtin_fma(1024.0, nd, -512.0));   // ((t - l + 1) n - 1/2) * 1024, exact (t - l + 1 may be 2^22)
    const uint32_t sub = (uint32_t)(num * d.rr);                // arith.c:134 / :199 (below n)
    // arith.c:136-139: the state x whose counts straddle sub - lo = c_0 + .. + c_(x-1) <= sub < lo + c_x. The first twelve counts
    // came with the header (walking them one dependent load at a time cost several cache round trips per symbol); behind a
    // short table's counts they are zeros, the running sum stays n there and n > sub.
    uint32_t lo = 0, fx = cn[0], x = 0, p = cn[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) {
        const bool take = p <= sub;
        lo = take ? p : lo;
        fx = take ? cn[i] : fx;
        x += take ? 1u : 0u;
        p += cn[i];
    }
    if (card > 12 && p <= sub) {                                 // long tables: the rest from memory
        uint32_t k = 12;
        do { lo = p; fx = t[4 + k]; p += fx; x = k++; } while (k < card && p <= sub);
    }
    *ok = sub - lo < fx;                                         // (lo <= sub by construction)
    // the state's value: one of the eight bytes that came with the header, or - long tables, rare states - behind the counts
    uint32_t v = ((x & 4 ? h.a.w : h.a.z) >> (8 * (x & 3))) & 0xFF;
    if (x >= 8) v = (t[dec_counts_end(card) + ((x - 8) >> 2)] >> (8 * (x & 3))) & 0xFF;
    *val = v;
    *lo_out = lo;
    *fx_out = fx;
    return x;
}

__device__ __forceinline__ void ld_update(uint32_t *__restrict__ t, uint32_t x, uint32_t fx, uint32_t n, uint32_t card) {
    t[4 + x] = fx + STEP;                                        // update_stats, qv_stream.c:12-24
    uint32_t nn = n + STEP;
    if (nn > RESCALE) {
        nn = 0;
        for (uint32_t i = 0; i < card; ++i) {
            const uint32_t v = t[4 + i];
            if (v) { const uint32_t hv = (v >> 1) + 1; t[4 + i] = hv; nn += hv; }
        }
    }
    t[0] = nn;
}

__device__ __forceinline__ void ld_back(LaneDecoder &d, uint32_t lo, uint32_t fx, uint32_t n) {
    const double rd = d.rd;
    const double rn = dec_rcp((double)(n << 10));                // n <= 2^19 + 8: exact
    const uint32_t ql = (uint32_t)__builtin_fma(rd, (double)lo * rn, DEC_QUOT_LO);            // arith.c:146-147
    const uint32_t qc = (uint32_t)__builtin_fma(rd, (double)(n - lo - fx) * rn, DEC_QUOT_HI);
    uint32_t l32 = d.l32 + (ql << 10), nu32 = d.nu32 + (qc << 10);
    // E1/E2 (arith.c:158-170): k leading bits shared by l and u leave (the low ten bits stop the count at 22), k stream bits enter the tag
    const uint32_t k = (uint32_t)__builtin_clz(~(l32 ^ nu32));
    l32 <<= k;
    nu32 <<= k;
    // E3 (arith.c:171-181): j positions below the MSB with l = 1, u = 0 (both MSBs are 0 here: the leading zeros of the rest number
    // j + 1, never all 32); the tag's new MSB is flipped once
    const uint32_t j1 = (uint32_t)__builtin_clz(~(l32 & nu32) & 0x7FFFFFFFu);
    d.l32 = (l32 << j1) >> 1;
    d.nu32 = (nu32 << j1) >> 1;
    const uint32_t total = k + j1 - 1;                           // bits the tag takes from the stream: 43 at most
    uint32_t t32 = d.t32;
    if (total < 32) {                                            // (avail >= 32 at the top of a symbol: ld_refill)
        const uint32_t bits = ((uint32_t)(d.buf >> 32) >> 1) >> (31 - total);
        t32 = (t32 << total) | (bits << 10);
        d.buf <<= total;
        d.avail -= total;
    } else {                                                     // one symbol in 2^31
        t32 = (t32 << k) | (ld_take(d, k) << 10);
        ld_refill(d);
        t32 = (t32 << (j1 - 1)) | (ld_take(d, j1 - 1) << 10);
    }
    d.t32 = t32 ^ (j1 > 1 ? 0x80000000u : 0u);
    d.rd = 4294967296.0 - (double)(d.l32 + d.nu32);
    d.rr = dec_rcp(d.rd);
}

// The per-symbol chain is context -> table -> output alphabet. Two of its three memory round trips are shortened:
//  * `ctx2[cluster][column][previous value]` = {lo table offset | qratio << 24, hi table offset}: one look-up instead
//    of context index, then table offset; it lives in LDS when it fits beside the staging area (CTX_LDS);
//  * the output alphabet (reconstruction points) of a quantizer is appended to its model table in the arena
//    ([n][card][counts...][recon bytes...]), so the value comes out of the cache line the decode just used.
template <bool HAVE_LABELS, bool CTX_LDS>
__global__ __launch_bounds__(64) void k_decode_shards(const uint8_t *__restrict__ in, uint64_t cap, const uint64_t *__restrict__ sizes,
                                                      const uint64_t *__restrict__ first, uint32_t shards, uint32_t C, uint32_t K,
                                                      const uint2 *__restrict__ ctx2, uint32_t *__restrict__ arena,
                                                      uint32_t arena_words, const uint8_t *__restrict__ draws, uint64_t ld,
                                                      uint8_t *__restrict__ qv, uint8_t *__restrict__ labels,
                                                      int32_t *__restrict__ status, uint32_t lanes) {
    extern __shared__ uint32_t stage[];                          // [C + 1][lanes] dwords: 4 reads of a column per lane (+ a row nobody reads); then ctx2 (CTX_LDS)
    uint2 *lctx = reinterpret_cast<uint2 *>(stage + ((size_t)C + 1) * lanes);
    if (CTX_LDS) {
        for (uint32_t i = threadIdx.x; i < K * C * A; i += 64) lctx[i] = ctx2[i];
        __syncthreads();
    }
    const uint2 *cx = CTX_LDS ? lctx : ctx2;
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * lanes + lane;                 // `lanes` streams per wave (64, or 32 when that puts a workgroup on every CU)
    const bool live = lane < lanes && g < shards;
    // where this lane's dword of column s lies in the staging area: s * st_stride + st_base; the lanes beyond `lanes` share the
    // last row (whatever they put there, nobody reads)
    const uint32_t st_stride = lane < lanes ? lanes : 0u, st_base = lane < lanes ? lane : C * lanes + (lane - lanes) % lanes;
    const uint64_t r0 = live ? first[g] : 0, nreads = live ? first[g + 1] - r0 : 0;
    uint64_t most = nreads;
    for (int o = 32; o > 0; o >>= 1) { const uint64_t v = __shfl_xor(most, o, 64); most = v > most ? v : most; }
    uint32_t *const nobody = arena + (uint64_t)shards * arena_words;   // the tables of a stream that does not exist
    uint32_t *const ar = live ? arena + (uint64_t)g * arena_words : nobody;
    LaneDecoder d;
    d.l32 = 0; d.nu32 = 0; d.t32 = 0; d.buf = 0; d.avail = 0; d.pos = 0;   // l = 0, u = 2^22 - 1
    d.rd = 4294967296.0; d.rr = dec_rcp(4294967296.0);
    d.src = in + (uint64_t)(live ? g : 0) * cap;
    d.size = live ? sizes[g] : 0;
#ifdef QVZ_PROFILE
    d.c_head = d.c_rest = d.c_n = 0;
    const unsigned long long dp_start = DP_NOW();
#endif
    ld_request(d);
    ld_refill(d);
    d.t32 = ld_take(d, M_BITS) << 10;                            // qv_stream.c:112-113
    ld_refill(d);
    bool bad = false;
    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
    // where the table of (cluster cl, column s) after the value pv lies, with this lane's draw of the symbol (codebook.c:165-170);
    // NO_TABLE when pv is no context of the column or the context is dead (no table: such a stream is not of this model)
    auto table_of = [&](uint32_t cl, uint32_t s, uint32_t pv, uint32_t draw) -> uint32_t {
        const uint2 cv = cx[((size_t)cl * C + s) * A + pv];
        const uint32_t o = draw >= (cv.x >> 24) ? cv.y : (cv.x & 0xFFFFFFu);
        return cv.x == NO_TABLE || o == 0xFFFFFFu ? NO_TABLE : o;
    };

    for (uint64_t i = 0; i < most; ++i) {
        if ((i & 3) == 0 && i < nreads) {
            // the WELL draws of the next 4 reads, all columns, in one burst of independent dword loads (fetched one by
            // one inside the symbol chain each of them was a cache miss of its own); a draw sits in the staging byte
            // that the decoded value of the same (read, column) overwrites
            for (uint32_t s = 0; s < C; ++s)
                stage[s * st_stride + st_base] = *reinterpret_cast<const uint32_t *>(draws + (uint64_t)s * ld + r0 + i);
        }
        const bool mine = i < nreads;                            // this lane has a read i; it decodes it unless the stream has failed (`bad`)
        const uint32_t bsel = (uint32_t)(i & 3);
        uint32_t cl = 0;
        if (HAVE_LABELS) {                                       // qv_read_cluster, qv_compressor.c:36-43
            uint32_t *tl = mine && !bad ? ar : nobody;
            DecHeader h = ld_header(tl);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(h.a), "+v"(h.b), "+v"(h.c), "+v"(h.d) : : "memory");
            uint32_t unused, lo, fx;
            bool ok;
            ld_refill(d);
            const uint32_t x = ld_front(d, h, tl, &unused, &lo, &fx, &ok);
            ld_update(tl, x, fx, h.a.x, h.a.y);
            ld_back(d, lo, fx, h.a.x);
            if (mine && !bad) {
                if (ok && x < K) { cl = x; labels[r0 + i] = (uint8_t)x; }
                else bad = true;
            }
        }
        // column 0's table (the previous value is 0 there)
        uint32_t *tA = nobody, *tB = nobody;
        {
            const uint32_t o = table_of(cl, 0, 0, sb[st_base * 4 + bsel] & 127);    // well.c:33-46
            bad = bad || (mine && o == NO_TABLE);
            if (mine && !bad) tA = ar + o;
        }
        // One column: the header hc of table tc has been requested; the next column's table is known as soon as this symbol's value
        // is, and its header (hn, tn) is requested BEFORE the table's update and the interval's: its round trip passes behind
        // them (and the update's stores, which nobody waits for, queue behind it). Two header register sets in turn (a copy
        // would be a use: a wait). (The last symbol of a stream changes no state in the reference - decoder_last_step,
        // qv_compressor.c:222-225 -; here it does, and nobody looks at that state again.)
        auto column = [&](uint32_t s, DecHeader &hc, uint32_t *tc, DecHeader &hn, uint32_t *&tn) {
            // (the draw of the next column, before this column's value goes into the staging area: another byte, but the compiler cannot know)
            const uint32_t draw_next = s + 1 < C ? sb[((s + 1) * st_stride + st_base) * 4 + bsel] & 127u : 0u;
#ifdef QVZ_PROFILE
            const unsigned long long dp0 = DP_NOW();
#endif
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc.a), "+v"(hc.b), "+v"(hc.c), "+v"(hc.d) : : "memory");   // the header - and every load before it
#ifdef QVZ_PROFILE
            const unsigned long long dp1 = DP_NOW();
            d.c_head += dp1 - dp0; d.c_n += 1;
#endif
            uint32_t value, lo, fx;
            bool ok;
            const uint32_t x = ld_front(d, hc, tc, &value, &lo, &fx, &ok);
            sb[(s * st_stride + st_base) * 4 + bsel] = (uint8_t)value;                         // the quantizer's output alphabet
            bad = bad || (mine && !ok);
            // the next column's table is known now: its header is on its way while the interval is updated
            tn = nobody;
            if (s + 1 < C) {
                const uint32_t o = table_of(cl, s + 1, value, draw_next);
                bad = bad || (mine && o == NO_TABLE);                                // the value is no context of the next column
                if (mine && !bad) tn = ar + o;
            }
            hn = ld_header(tn);
            ld_refill(d);                                        // (the search needs no stream bits: the refill waits until the header is on its way)
            ld_update(tc, x, fx, hc.a.x, hc.a.y);
            ld_back(d, lo, fx, hc.a.x);
#ifdef QVZ_PROFILE
            d.c_rest += DP_NOW() - dp1;
#endif
        };
        DecHeader hA = ld_header(tA), hB;
        for (uint32_t s = 0; s < C; s += 2) {
            column(s, hA, tA, hB, tB);
            if (s + 1 < C) column(s + 1, hB, tB, hA, tA);
        }
        if ((i & 3) == 3 || i + 1 == most) {                     // 4 reads of every column collected: out as dwords
            const uint64_t i0 = i & ~(uint64_t)3;
            if (i0 < nreads) {
                const uint32_t cnt = nreads - i0 < 4 ? (uint32_t)(nreads - i0) : 4u;
                for (uint32_t s = 0; s < C; ++s) {
                    const uint32_t w = stage[s * st_stride + st_base];
                    uint8_t *dst = qv + (uint64_t)s * ld + r0 + i0;                  // r0 and i0 are multiples of 4
                    if (cnt == 4) *reinterpret_cast<uint32_t *>(dst) = w;
                    else for (uint32_t b = 0; b < cnt; ++b) dst[b] = (uint8_t)(w >> (8 * b));
                }
            }
        }
    }
    if (bad) atomicOr(status, 1);
#ifdef QVZ_PROFILE
    if (lane == 0 && (blockIdx.x & 15) == 3) {
        atomicAdd(&g_dec_prof[0], d.c_head); atomicAdd(&g_dec_prof[1], d.c_rest); atomicAdd(&g_dec_prof[2], d.c_n);
        atomicAdd(&g_dec_prof[3], DP_NOW() - dp_start);
    }
#endif
}

__global__ __launch_bounds__(256) void k_dec_arena_fill(const uint32_t *__restrict__ tmpl, uint32_t words, uint32_t shards,
                                                        uint32_t *__restrict__ arena) {
    const uint64_t total = (uint64_t)words * shards;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256)
        arena[i] = tmpl[i % words];
}

// words of one decoder table of `card` states (the layout above), and of all the tables of one stream
static uint32_t dec_table_words(uint32_t card) {
    const uint32_t w = 4 + (card > 12 ? card : 12u) + (card > 8 ? (card - 8 + 3) / 4 : 0u);
    return (w + 15) & ~15u;
}
extern "C" uint64_t qvz_hip_decode_tables_bytes(uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs) {
    if (!cbs || !columns || !clusters) return 0;
    uint64_t words = dec_table_words(clusters);                  // the cluster-id table
    for (uint32_t c = 0; c < clusters; ++c)
        for (uint32_t s = 0; s < columns; ++s)
            for (uint32_t j = 0; j < 2u * cbs[c]->nctx[s]; ++j)
                if (cbs[c]->live[(size_t)s * A + j / 2]) words += dec_table_words(cbs[c]->nstates[(size_t)s * 2 * A + j]);   // (qvz_model_build, QVZ_MODEL_LIVE_ONLY)
    return words * 4;
}

extern "C" int qvz_hip_decode_shards(const uint8_t *d_in, uint64_t cap_per_shard, const uint64_t *h_sizes, uint64_t n,
                                     uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                     const uint8_t *d_draws, uint64_t ld, uint8_t *d_qv, uint8_t *d_labels, void *stream) {
    return qvz_hip_decode_shards_bounds(d_in, cap_per_shard, h_sizes, n, columns, clusters, cbs, shards, nullptr, d_draws, ld, d_qv, d_labels, stream);
}

extern "C" int qvz_hip_decode_shards_bounds(const uint8_t *d_in, uint64_t cap_per_shard, const uint64_t *h_sizes, uint64_t n,
                                            uint32_t columns, uint32_t clusters, const qvz_codebook_t *const *cbs, uint32_t shards,
                                            const uint64_t *h_bounds, const uint8_t *d_draws, uint64_t ld, uint8_t *d_qv,
                                            uint8_t *d_labels, void *stream) {
    if (h_bounds) {                                              // a cut somebody else made: whole WELL words, in order, all of the reads
        if (h_bounds[0] != 0 || h_bounds[shards ? shards : 0] != n) return qvz_set_error(QVZ_E_ARG, "shard bounds do not cover the %llu reads", (unsigned long long)n);
        for (uint32_t g = 0; g < shards; ++g)
            if (h_bounds[g] > h_bounds[g + 1] || (h_bounds[g] & 3)) return qvz_set_error(QVZ_E_ARG, "shard %u does not start on a multiple of 4 reads or runs backwards", g);
    }
    if (!d_in || !h_sizes || !n || !columns || !clusters || !cbs || !shards || !d_draws || !d_qv || ld < n || (ld & 3) || (cap_per_shard & 3))
        return qvz_set_error(QVZ_E_ARG, "bad shard-decoder arguments");
    if (clusters > 1 && !d_labels) return qvz_set_error(QVZ_E_ARG, "a label buffer is required for clusters > 1");
    if (((size_t)columns + 1) * 32 * 4 > 150 * 1024) return qvz_set_error(QVZ_E_ARG, "%u columns exceed the decoder's LDS staging (1199 at most)", columns);
    for (uint32_t g = 0; g < shards; ++g)
        if (h_sizes[g] > cap_per_shard) return qvz_set_error(QVZ_E_ARG, "stream %u is longer than cap_per_shard", g);
    hipStream_t st = (hipStream_t)stream;
    // A workgroup is one wave (its LDS - staging, context table - lets one or two live on a CU). 8192 streams as 128 waves of 64 lanes
    // leave every other CU idle; as 256 waves of 32 lanes every CU has one and the decode of 100M x 150 takes 3.77 instead of 4.05 s
    // (fewer lanes than that need more workgroups than there are CUs and run in rounds: 16 lanes 6.5 s).
    uint32_t dec_lanes = 64;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            (shards + 31) / 32 <= (uint32_t)cus) dec_lanes = 32;
        if (((size_t)columns + 1) * 64 * 4 > 150 * 1024) dec_lanes = 32;   // long reads: a staging row of 64 lanes would not fit the LDS
    }
    uint32_t *h_arena = nullptr, *h_off = nullptr, words = 0;
    if (qvz_model_build(clusters, columns, cbs, QVZ_MODEL_LIVE_ONLY | QVZ_MODEL_ALIGN16, &h_arena, &h_off, &words))
        return qvz_set_error(QVZ_E_NOMEM, "model build failed");
    // decoder model: every table of qvz_model_build followed by its quantizer's output alphabet, 4 values per word
    const size_t ctxn = (size_t)clusters * columns * A, offn = (size_t)clusters * columns * 2 * A;
    std::vector<uint32_t> h_dar;
    std::vector<uint32_t> h_doff(offn, NO_TABLE);
    std::vector<uint2> h_ctx(ctxn);
    uint64_t *h_first = (uint64_t *)malloc((shards + 1) * sizeof(uint64_t));
    uint32_t *d_tmpl = nullptr, *d_arena = nullptr;
    uint2 *d_ctx = nullptr;
    uint64_t *d_first = nullptr, *d_sizes = nullptr;
    int32_t *d_status = nullptr;
    int rc = QVZ_OK;
    int32_t status = 0;
    uint32_t dwords = 0, dwords_stride = 0;
    const size_t lds_stage = ((size_t)columns + 1) * dec_lanes * 4, lds_ctx = ctxn * sizeof(uint2);
    const bool ctx_lds = lds_stage + lds_ctx <= 150 * 1024;
    const size_t lds = lds_stage + (ctx_lds ? lds_ctx : 0);
#define TRY(expr)                                                                                                    \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) { rc = qvz_set_error(QVZ_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
    if (!h_first) { rc = qvz_set_error(QVZ_E_NOMEM, "host alloc"); goto done; }
    {
        // every table of qvz_model_build ([n][card][counts]) re-laid as [n][card][recon 0..7][counts, twelve at least][recon 8..], 64-byte aligned;
        // the cluster-id table first (its "values" are the ids themselves and are not used)
        auto put_table = [&](const uint32_t *tb, const uint8_t *rc8) {
            const uint32_t card = tb[1];
            h_dar.push_back(tb[0]); h_dar.push_back(card);
            for (uint32_t w0 = 0; w0 < 8; w0 += 4) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4 && w0 + b < card; ++b) w |= (uint32_t)(rc8 ? rc8[w0 + b] : w0 + b) << (8 * b);
                h_dar.push_back(w);
            }
            for (uint32_t i = 0; i < card || i < 12; ++i) h_dar.push_back(i < card ? tb[2 + i] : 0u);
            for (uint32_t i = 8; i < card; i += 4) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4 && i + b < card; ++b) w |= (uint32_t)(rc8 ? rc8[i + b] : i + b) << (8 * b);
                h_dar.push_back(w);
            }
            while (h_dar.size() & 15) h_dar.push_back(0);
        };
        put_table(h_arena, nullptr);                             // offset 0: [n][clusters][counts of the cluster ids]
        for (uint32_t c = 0; c < clusters; ++c)
            for (uint32_t s = 0; s < columns; ++s)
                for (uint32_t j = 0; j < 2 * A; ++j) {
                    const uint32_t o = h_off[((size_t)c * columns + s) * 2 * A + j];
                    if (o == NO_TABLE) continue;
                    h_doff[((size_t)c * columns + s) * 2 * A + j] = (uint32_t)h_dar.size();
                    put_table(h_arena + o, cbs[c]->recon + (((size_t)s * 2 * A) + j) * A);
                }
        if (h_dar.size() >= 0xFFFFFFu) { rc = qvz_set_error(QVZ_E_ARG, "model too large for the decoder's 24-bit table offsets"); goto done; }
        dwords = dwords_stride = (uint32_t)h_dar.size();
        if ((uint64_t)dwords * 4 != qvz_hip_decode_tables_bytes(columns, clusters, cbs)) { rc = qvz_set_error(QVZ_E_ARG, "decoder tables: %u words built, %llu bytes announced", dwords, (unsigned long long)qvz_hip_decode_tables_bytes(columns, clusters, cbs)); goto done; }
        for (uint32_t c = 0; c < clusters; ++c)
            for (uint32_t s = 0; s < columns; ++s)
                for (uint32_t pv = 0; pv < A; ++pv) {
                    const uint8_t ctx = cbs[c]->ctx_idx[(size_t)s * A + pv];
                    uint2 e = make_uint2(NO_TABLE, NO_TABLE);
                    if (ctx != QVZ_NOT_FOUND) {
                        const uint32_t lo = h_doff[((size_t)c * columns + s) * 2 * A + 2 * ctx];
                        const uint32_t hi = h_doff[((size_t)c * columns + s) * 2 * A + 2 * ctx + 1];
                        e.x = (lo == NO_TABLE ? 0xFFFFFFu : lo) | ((uint32_t)cbs[c]->qratio[(size_t)s * A + ctx] << 24);
                        e.y = hi == NO_TABLE ? 0xFFFFFFu : hi;
                    }
                    h_ctx[((size_t)c * columns + s) * A + pv] = e;
                }
    }
    if (h_bounds) memcpy(h_first, h_bounds, ((size_t)shards + 1) * sizeof(uint64_t));
    else qvz_shard_bounds(n, shards, h_first);
    TRY(hipMalloc(&d_tmpl, (size_t)dwords * 4));
    TRY(hipMalloc(&d_arena, (size_t)dwords * ((size_t)shards + 1) * 4));   // + the tables of a stream that does not exist (lanes without a symbol to decode)
    TRY(hipMalloc(&d_ctx, ctxn * sizeof(uint2)));
    TRY(hipMalloc(&d_first, (shards + 1) * sizeof(uint64_t)));
    TRY(hipMalloc(&d_sizes, shards * sizeof(uint64_t)));
    TRY(hipMalloc(&d_status, sizeof(int32_t)));
    TRY(hipMemcpyAsync(d_tmpl, h_dar.data(), (size_t)dwords * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_ctx, h_ctx.data(), ctxn * sizeof(uint2), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_first, h_first, (shards + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_sizes, h_sizes, shards * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_dec_arena_fill, dim3(2048), dim3(256), 0, st, d_tmpl, dwords, shards + 1, d_arena);
#ifdef QVZ_PROFILE
    { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dec_prof), z, sizeof(z)); }
#endif
#define DEC_LAUNCH(LBL, CLDS)                                                                                                       \
    do {                                                                                                                            \
        TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode_shards<LBL, CLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_decode_shards<LBL, CLDS>), dim3((shards + dec_lanes - 1) / dec_lanes), dim3(64), lds, st, d_in, cap_per_shard, d_sizes, \
                           d_first, shards, columns, clusters, d_ctx, d_arena, dwords_stride, d_draws, ld, d_qv, d_labels, d_status, dec_lanes); \
    } while (0)
#ifdef QVZ_PROFILE          // QVZ_DEC_SHARED_ARENA=1: every stream on one set of tables (wrong output; what a symbol costs when its table comes from the cache)
    if (getenv("QVZ_DEC_SHARED_ARENA")) dwords_stride = 0;
#endif
    if (clusters > 1) { if (ctx_lds) DEC_LAUNCH(true, true); else DEC_LAUNCH(true, false); }
    else { if (ctx_lds) DEC_LAUNCH(false, true); else DEC_LAUNCH(false, false); }
#undef DEC_LAUNCH
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRY(hipStreamSynchronize(st));
    if (status) rc = qvz_set_error(QVZ_E_DATA, "a stream does not decode with these codebooks (unknown context, table or state)");
#ifdef QVZ_PROFILE
    {
        unsigned long long dp[8];
        (void)hipMemcpyFromSymbol(dp, HIP_SYMBOL(g_dec_prof), sizeof(dp));
        if (dp[2]) fprintf(stderr, "k_decode_shards cycles per symbol (lane 0 of every 16th wave): table header wait %.0f, rest of the symbol %.0f, all %.0f (%llu symbols; %u words of tables per stream, %u lanes per wave)\n",
                           (double)dp[0] / dp[2], (double)dp[1] / dp[2], (double)dp[3] / dp[2], dp[2], dwords, dec_lanes);
    }
#endif
done:
#undef TRY
    (void)hipFree(d_tmpl); (void)hipFree(d_arena); (void)hipFree(d_ctx);
    (void)hipFree(d_first); (void)hipFree(d_sizes); (void)hipFree(d_status);
    free(h_arena); free(h_off); free(h_first);
    return rc;
}
