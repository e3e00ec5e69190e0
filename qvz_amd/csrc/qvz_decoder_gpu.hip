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

// floor(prod / d) for prod < 2^53, quotient < 2^32: estimate by v_rcp_f64, one correction against the exact remainder
__device__ __forceinline__ uint32_t dec_div(uint64_t prod, uint32_t d) {
    const double pd = (double)(uint32_t)(prod >> 32) * 4294967296.0 + (double)(uint32_t)prod;
    uint32_t q = (uint32_t)(pd * __builtin_amdgcn_rcp((double)d));
    const int64_t r = (int64_t)prod - (int64_t)((uint64_t)q * d);
    if (r < 0) q -= 1;
    else if (r >= (int64_t)d) q += 1;
    return q;
}

struct LaneDecoder {
    uint32_t l, u, t;
    uint64_t buf;            // next bits of the stream, left-aligned
    uint32_t avail;          // valid bits in buf
    const uint8_t *src;
    uint64_t pos, size;      // next byte to load (multiple of 4), stream length
};

__device__ __forceinline__ void ld_refill(LaneDecoder &d) {     // afterwards avail > 32 (zeros past the end, os_stream.c:36-52)
    if (d.avail <= 32) {
        uint32_t w = 0;
        if (d.pos < d.size) {
            w = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(d.src + d.pos));   // the area behind a stream is readable
            const uint64_t left = d.size - d.pos;
            if (left < 4) w &= ~0u << (8 * (4 - (uint32_t)left));
        }
        d.buf |= (uint64_t)w << (32 - d.avail);
        d.avail += 32;
        d.pos += 4;
    }
}

__device__ __forceinline__ uint32_t ld_take(LaneDecoder &d, uint32_t nb) {   // nb <= 22 <= avail
    const uint32_t v = nb ? (uint32_t)(d.buf >> (64 - nb)) : 0u;
    d.buf = nb ? d.buf << nb : d.buf;
    d.avail -= nb;
    return v;
}

// One symbol with table t; returns the state, or 0xFFFFFFFF on a corrupt stream. Decoder tables are laid out
//   [n][card][recon 0..7, a byte each][counts 0 .. card-1][recon 8 .. card-1]
// so that the 64 bytes the decode needs anyway - four independent 16-byte loads, ONE memory round trip for a lane whose
// table lies in HBM or the Infinity Cache - also hold the quality values of the first eight states: with the output
// alphabet behind the counts, in another cache line, mapping the state back to a value was a second dependent round trip
// for every symbol (4.2 s for 100M x 150 reads; HISTORY.md section 6b). *val: the value of the state (x < card).
__device__ __forceinline__ uint32_t ld_decode(LaneDecoder &d, uint32_t *__restrict__ t, bool last, uint32_t *val) {
    // The first 12 counts come in with the header, so the cumulative search runs in registers: walking the counts one
    // dependent load at a time cost several cache round trips per symbol.
    const uint4 h0 = reinterpret_cast<const uint4 *>(t)[0], h1 = reinterpret_cast<const uint4 *>(t)[1],
                h2 = reinterpret_cast<const uint4 *>(t)[2], h3 = reinterpret_cast<const uint4 *>(t)[3];
    const uint32_t n = h0.x, card = h0.y;
    const uint32_t cn[12] = {h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
    const uint32_t range = d.u - d.l + 1;
    const uint64_t gap = (uint64_t)(d.t - d.l) + 1;
    const uint32_t sub = dec_div(gap * n - 1, range);           // arith.c:134 / :199 (below n, so 32 bits hold it either way)
    uint32_t cum = 0, k = 0, fxr = 0;                            // arith.c:136-139: first k with c_0 + .. + c_k > sub
    bool found = false;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const bool take = !found && (uint32_t)i < card && sub >= cum;
        cum += take ? cn[i] : 0u;
        fxr = take ? cn[i] : fxr;
        k += take ? 1u : 0u;
        found = found || !take;
    }
    while (k >= 12 && k < card && sub >= cum) { fxr = t[4 + k++]; cum += fxr; }   // long tables: the rest from memory
    if (sub >= cum) return 0xFFFFFFFFu;                          // ran off the table: not a stream of this model
    const uint32_t x = k - 1;
    // the state's value: one of the eight bytes that came with the header, or - long tables, rare states - behind the counts
    *val = x < 8 ? ((x < 4 ? h0.z : h0.w) >> (8 * (x & 3))) & 0xFF
                 : (t[4 + card + ((x - 8) >> 2)] >> (8 * (x & 3))) & 0xFF;
    if (last) return x;                                          // decoder_last_step: nothing else changes
    const uint32_t fx = fxr, cum_lo = cum - fx;
    uint32_t u = d.l + dec_div((uint64_t)range * cum, n) - 1;    // arith.c:146-147
    uint32_t l = d.l + dec_div((uint64_t)range * cum_lo, n);
    uint32_t tg = d.t;
    // E1/E2 (arith.c:158-170): k1 leading bits shared by l and u leave, k1 stream bits enter the tag
    const uint32_t diff = l ^ u;
    const uint32_t k1 = diff ? (uint32_t)__clz((int)diff) - (32 - M_BITS) : M_BITS;
    l = (l << k1) & MASK_M;
    u = ((u << k1) & MASK_M) | ((1u << k1) - 1);
    tg = ((tg << k1) & MASK_M) | ld_take(d, k1);
    ld_refill(d);
    // E3 (arith.c:171-181): j positions below the MSB with l = 1, u = 0; the tag's new MSB is flipped once
    const uint32_t xx = (~l | u) & MASK_M1;
    const uint32_t j = xx ? (uint32_t)__clz((int)xx) - (32 - (M_BITS - 1)) : M_BITS - 1;
    l = (l << j) & MASK_M1;
    u = ((u << j) & MASK_M1) | MSB | ((1u << j) - 1);
    tg = (((tg << j) & MASK_M) | ld_take(d, j)) ^ (j ? MSB : 0u);
    ld_refill(d);
    d.l = l; d.u = u; d.t = tg;
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
    return x;
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
    extern __shared__ uint32_t stage[];                          // [C][64] dwords: 4 reads of a column per lane; then ctx2 (CTX_LDS)
    uint2 *lctx = reinterpret_cast<uint2 *>(stage + (size_t)C * 64);
    if (CTX_LDS) {
        for (uint32_t i = threadIdx.x; i < K * C * A; i += 64) lctx[i] = ctx2[i];
        __syncthreads();
    }
    const uint2 *cx = CTX_LDS ? lctx : ctx2;
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * lanes + lane;                 // `lanes` streams per wave (64, or 32 when that puts a workgroup on every CU)
    const bool live = lane < lanes && g < shards;
    const uint64_t r0 = live ? first[g] : 0, nreads = live ? first[g + 1] - r0 : 0;
    uint64_t most = nreads;
    for (int o = 32; o > 0; o >>= 1) { const uint64_t v = __shfl_xor(most, o, 64); most = v > most ? v : most; }
    uint32_t *ar = arena + (uint64_t)(live ? g : 0) * arena_words;
    LaneDecoder d;
    d.l = 0; d.u = MASK_M; d.t = 0; d.buf = 0; d.avail = 0; d.pos = 0;
    d.src = in + (uint64_t)(live ? g : 0) * cap;
    d.size = live ? sizes[g] : 0;
    ld_refill(d);
    d.t = ld_take(d, M_BITS);                                    // qv_stream.c:112-113
    ld_refill(d);
    bool bad = false;
    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);

    for (uint64_t i = 0; i < most; ++i) {
        const bool on = i < nreads && !bad;
        if ((i & 3) == 0 && i < nreads) {
            // the WELL draws of the next 4 reads, all columns, in one burst of independent dword loads (fetched one by
            // one inside the symbol chain each of them was a cache miss of its own); a draw sits in the staging byte
            // that the decoded value of the same (read, column) overwrites
            for (uint32_t s = 0; s < C; ++s)
                stage[s * 64 + lane] = *reinterpret_cast<const uint32_t *>(draws + (uint64_t)s * ld + r0 + i);
        }
        if (on) {
            uint32_t cl = 0;
            if (HAVE_LABELS) {                                   // qv_read_cluster, qv_compressor.c:36-43
                uint32_t unused;
                cl = ld_decode(d, ar, false, &unused);
                if (cl >= K) { bad = true; cl = 0; }                                 // (0xFFFFFFFF, ran off the table, included)
                labels[r0 + i] = (uint8_t)cl;
            }
            uint32_t prev = 0;
            for (uint32_t s = 0; s < C && !bad; ++s) {
                const uint2 cv = cx[((size_t)cl * C + s) * A + prev];
                if (cv.x == NO_TABLE) { bad = true; break; }                         // previous value is no context of this column
                const uint32_t draw = sb[(s * 64 + lane) * 4 + (uint32_t)(i & 3)] & 127;   // well.c:33-46
                const uint32_t o = draw >= (cv.x >> 24) ? cv.y : (cv.x & 0xFFFFFFu); // hi : lo quantizer, codebook.c:165-170
                if (o == 0xFFFFFFu) { bad = true; break; }                           // dead context: no table
                const bool last = i + 1 == nreads && s + 1 == C;                     // qv_compressor.c:222-225
                uint32_t value = 0;
                const uint32_t x = ld_decode(d, ar + o, last, &value);
                if (x == 0xFFFFFFFFu) { bad = true; break; }                         // ran off the table
                prev = value;                                                        // the quantizer's output alphabet
                sb[(s * 64 + lane) * 4 + (uint32_t)(i & 3)] = (uint8_t)prev;
            }
        }
        if ((i & 3) == 3 || i + 1 == most) {                     // 4 reads of every column collected: out as dwords
            const uint64_t i0 = i & ~(uint64_t)3;
            if (i0 < nreads) {
                const uint32_t cnt = nreads - i0 < 4 ? (uint32_t)(nreads - i0) : 4u;
                for (uint32_t s = 0; s < C; ++s) {
                    const uint32_t w = stage[s * 64 + lane];
                    uint8_t *dst = qv + (uint64_t)s * ld + r0 + i0;                  // r0 and i0 are multiples of 4
                    if (cnt == 4) *reinterpret_cast<uint32_t *>(dst) = w;
                    else for (uint32_t b = 0; b < cnt; ++b) dst[b] = (uint8_t)(w >> (8 * b));
                }
            }
        }
    }
    if (bad) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void k_dec_arena_fill(const uint32_t *__restrict__ tmpl, uint32_t words, uint32_t shards,
                                                        uint32_t *__restrict__ arena) {
    const uint64_t total = (uint64_t)words * shards;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256)
        arena[i] = tmpl[i % words];
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
    if ((size_t)columns * 64 * 4 > 150 * 1024) return qvz_set_error(QVZ_E_ARG, "%u columns exceed the decoder's LDS staging (600 at most)", columns);
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
    uint32_t dwords = 0;
    const size_t lds_stage = (size_t)columns * 64 * 4, lds_ctx = ctxn * sizeof(uint2);
    const bool ctx_lds = lds_stage + lds_ctx <= 150 * 1024;
    const size_t lds = lds_stage + (ctx_lds ? lds_ctx : 0);
#define TRY(expr)                                                                                                    \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) { rc = qvz_set_error(QVZ_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
    if (!h_first) { rc = qvz_set_error(QVZ_E_NOMEM, "host alloc"); goto done; }
    {
        // every table of qvz_model_build ([n][card][counts]) re-laid as [n][card][recon 0..7][counts][recon 8..], 16-byte aligned;
        // the cluster-id table first (its "values" are the ids themselves and are not used)
        auto put_table = [&](const uint32_t *tb, const uint8_t *rc8) {
            const uint32_t card = tb[1];
            h_dar.push_back(tb[0]); h_dar.push_back(card);
            for (uint32_t w0 = 0; w0 < 8; w0 += 4) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4 && w0 + b < card; ++b) w |= (uint32_t)(rc8 ? rc8[w0 + b] : w0 + b) << (8 * b);
                h_dar.push_back(w);
            }
            for (uint32_t i = 0; i < card; ++i) h_dar.push_back(tb[2 + i]);
            for (uint32_t i = 8; i < card; i += 4) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4 && i + b < card; ++b) w |= (uint32_t)(rc8 ? rc8[i + b] : i + b) << (8 * b);
                h_dar.push_back(w);
            }
            while (h_dar.size() & 3) h_dar.push_back(0);
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
        dwords = (uint32_t)h_dar.size();
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
    TRY(hipMalloc(&d_arena, ((size_t)dwords * shards + 16) * 4));   // + slack: a header read covers 64 bytes
    TRY(hipMalloc(&d_ctx, ctxn * sizeof(uint2)));
    TRY(hipMalloc(&d_first, (shards + 1) * sizeof(uint64_t)));
    TRY(hipMalloc(&d_sizes, shards * sizeof(uint64_t)));
    TRY(hipMalloc(&d_status, sizeof(int32_t)));
    TRY(hipMemcpyAsync(d_tmpl, h_dar.data(), (size_t)dwords * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_ctx, h_ctx.data(), ctxn * sizeof(uint2), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_first, h_first, (shards + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_sizes, h_sizes, shards * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_dec_arena_fill, dim3(2048), dim3(256), 0, st, d_tmpl, dwords, shards, d_arena);
#define DEC_LAUNCH(LBL, CLDS)                                                                                                       \
    do {                                                                                                                            \
        TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode_shards<LBL, CLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_decode_shards<LBL, CLDS>), dim3((shards + dec_lanes - 1) / dec_lanes), dim3(64), lds, st, d_in, cap_per_shard, d_sizes, \
                           d_first, shards, columns, clusters, d_ctx, d_arena, dwords, d_draws, ld, d_qv, d_labels, d_status, dec_lanes); \
    } while (0)
    if (clusters > 1) { if (ctx_lds) DEC_LAUNCH(true, true); else DEC_LAUNCH(true, false); }
    else { if (ctx_lds) DEC_LAUNCH(false, true); else DEC_LAUNCH(false, false); }
#undef DEC_LAUNCH
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRY(hipStreamSynchronize(st));
    if (status) rc = qvz_set_error(QVZ_E_DATA, "a stream does not decode with these codebooks (unknown context, table or state)");
done:
#undef TRY
    (void)hipFree(d_tmpl); (void)hipFree(d_arena); (void)hipFree(d_ctx);
    (void)hipFree(d_first); (void)hipFree(d_sizes); (void)hipFree(d_status);
    free(h_arena); free(h_off); free(h_first);
    return rc;
}
