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

// One symbol with table t ([n][card][counts...]); returns the state, or card on a corrupt stream.
__device__ __forceinline__ uint32_t ld_decode(LaneDecoder &d, uint32_t *__restrict__ t, bool last) {
    const uint32_t n = t[0], card = t[1];
    const uint32_t range = d.u - d.l + 1;
    const uint64_t gap = (uint64_t)(d.t - d.l) + 1;
    const uint32_t sub = dec_div(gap * n - 1, range);           // arith.c:134 / :199 (below n, so 32 bits hold it either way)
    uint32_t cum = 0, k = 0;
    while (k < card && sub >= cum) cum += t[2 + k++];            // arith.c:136-139
    if (sub >= cum) return card;                                 // ran off the table: not a stream of this model
    const uint32_t x = k - 1;
    if (last) return x;                                          // decoder_last_step: nothing else changes
    const uint32_t fx = t[2 + x], cum_lo = cum - fx;
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
    t[2 + x] = fx + STEP;                                        // update_stats, qv_stream.c:12-24
    uint32_t nn = n + STEP;
    if (nn > RESCALE) {
        nn = 0;
        for (uint32_t i = 0; i < card; ++i) {
            const uint32_t v = t[2 + i];
            if (v) { const uint32_t hv = (v >> 1) + 1; t[2 + i] = hv; nn += hv; }
        }
    }
    t[0] = nn;
    return x;
}

template <bool HAVE_LABELS>
__global__ __launch_bounds__(64) void k_decode_shards(const uint8_t *__restrict__ in, uint64_t cap, const uint64_t *__restrict__ sizes,
                                                      const uint64_t *__restrict__ first, uint32_t shards, uint32_t C, uint32_t K,
                                                      const uint16_t *__restrict__ ct, const uint8_t *__restrict__ recon,
                                                      const uint32_t *__restrict__ off, uint32_t *__restrict__ arena,
                                                      uint32_t arena_words, const uint8_t *__restrict__ draws, uint64_t ld,
                                                      uint8_t *__restrict__ qv, uint8_t *__restrict__ labels,
                                                      int32_t *__restrict__ status) {
    extern __shared__ uint32_t stage[];                          // [C][64] dwords: 4 reads of a column per lane
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * 64 + lane;
    const bool live = g < shards;
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
        if (on) {
            uint32_t cl = 0;
            if (HAVE_LABELS) {                                   // qv_read_cluster, qv_compressor.c:36-43
                cl = ld_decode(d, ar, false);
                if (cl >= K) { bad = true; cl = 0; }
                labels[r0 + i] = (uint8_t)cl;
            }
            uint32_t prev = 0;
            for (uint32_t s = 0; s < C && !bad; ++s) {
                const uint32_t cv = ct[((uint64_t)cl * C + s) * QVZ_CT_STRIDE + prev];
                const uint32_t ctx = cv & 0xFF;
                if (ctx == QVZ_NOT_FOUND) { bad = true; break; }
                const uint32_t draw = draws[(uint64_t)s * ld + r0 + i] & 127;        // well.c:33-46
                const uint32_t qi = 2 * ctx + (draw >= (cv >> 8) ? 1u : 0u);        // codebook.c:165-170
                const uint32_t o = off[((uint64_t)cl * C + s) * 2 * A + qi];
                if (o == NO_TABLE) { bad = true; break; }
                const bool last = i + 1 == nreads && s + 1 == C;                     // qv_compressor.c:222-225
                const uint32_t x = ld_decode(d, ar + o, last);
                if (x >= ar[o + 1]) { bad = true; break; }
                prev = recon[(((uint64_t)cl * C + s) * 2 * A + qi) * A + x];
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
    if (!d_in || !h_sizes || !n || !columns || !clusters || !cbs || !shards || !d_draws || !d_qv || ld < n || (ld & 3) || (cap_per_shard & 3))
        return qvz_set_error(QVZ_E_ARG, "bad shard-decoder arguments");
    if (clusters > 1 && !d_labels) return qvz_set_error(QVZ_E_ARG, "a label buffer is required for clusters > 1");
    if ((size_t)columns * 64 * 4 > 160 * 1024) return qvz_set_error(QVZ_E_ARG, "%u columns exceed the decoder's LDS staging (640 at most)", columns);
    for (uint32_t g = 0; g < shards; ++g)
        if (h_sizes[g] > cap_per_shard) return qvz_set_error(QVZ_E_ARG, "stream %u is longer than cap_per_shard", g);
    hipStream_t st = (hipStream_t)stream;
    uint32_t *h_arena = nullptr, *h_off = nullptr, words = 0;
    if (qvz_model_build(clusters, columns, cbs, QVZ_MODEL_LIVE_ONLY | QVZ_MODEL_ALIGN16, &h_arena, &h_off, &words))
        return qvz_set_error(QVZ_E_NOMEM, "model build failed");
    const size_t ctn = (size_t)clusters * columns * QVZ_CT_STRIDE, rcn = (size_t)clusters * columns * 2 * A * A;
    const size_t offn = (size_t)clusters * columns * 2 * A;
    uint16_t *h_ct = (uint16_t *)malloc(ctn * 2), *h_qs = (uint16_t *)malloc((size_t)columns * QVZ_QS_STRIDE * 2);
    uint8_t *h_rc = (uint8_t *)malloc(rcn);
    uint64_t *h_first = (uint64_t *)malloc((shards + 1) * sizeof(uint64_t));
    uint32_t *d_tmpl = nullptr, *d_off = nullptr, *d_arena = nullptr;
    uint16_t *d_ct = nullptr;
    uint8_t *d_rc = nullptr;
    uint64_t *d_first = nullptr, *d_sizes = nullptr;
    int32_t *d_status = nullptr;
    int rc = QVZ_OK;
    int32_t status = 0;
    const size_t lds = (size_t)columns * 64 * 4;
#define TRY(expr)                                                                                                    \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) { rc = qvz_set_error(QVZ_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
    if (!h_ct || !h_qs || !h_rc || !h_first) { rc = qvz_set_error(QVZ_E_NOMEM, "host alloc"); goto done; }
    for (uint32_t c = 0; c < clusters; ++c) {
        qvz_codebook_tables(cbs[c], h_ct + (size_t)c * columns * QVZ_CT_STRIDE, h_qs);
        memcpy(h_rc + (size_t)c * columns * 2 * A * A, cbs[c]->recon, (size_t)columns * 2 * A * A);
    }
    qvz_shard_bounds(n, shards, h_first);
    TRY(hipMalloc(&d_tmpl, (size_t)words * 4));
    TRY(hipMalloc(&d_off, offn * 4));
    TRY(hipMalloc(&d_arena, ((size_t)words * shards + 8) * 4));
    TRY(hipMalloc(&d_ct, ctn * 2));
    TRY(hipMalloc(&d_rc, rcn));
    TRY(hipMalloc(&d_first, (shards + 1) * sizeof(uint64_t)));
    TRY(hipMalloc(&d_sizes, shards * sizeof(uint64_t)));
    TRY(hipMalloc(&d_status, sizeof(int32_t)));
    TRY(hipMemcpyAsync(d_tmpl, h_arena, (size_t)words * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_off, h_off, offn * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_ct, h_ct, ctn * 2, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_rc, h_rc, rcn, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_first, h_first, (shards + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_sizes, h_sizes, shards * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_dec_arena_fill, dim3(2048), dim3(256), 0, st, d_tmpl, words, shards, d_arena);
    TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode_shards<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode_shards<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (clusters > 1)
        hipLaunchKernelGGL(k_decode_shards<true>, dim3((shards + 63) / 64), dim3(64), lds, st, d_in, cap_per_shard, d_sizes, d_first,
                           shards, columns, clusters, d_ct, d_rc, d_off, d_arena, words, d_draws, ld, d_qv, d_labels, d_status);
    else
        hipLaunchKernelGGL(k_decode_shards<false>, dim3((shards + 63) / 64), dim3(64), lds, st, d_in, cap_per_shard, d_sizes, d_first,
                           shards, columns, clusters, d_ct, d_rc, d_off, d_arena, words, d_draws, ld, d_qv, d_labels, d_status);
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRY(hipStreamSynchronize(st));
    if (status) rc = qvz_set_error(QVZ_E_DATA, "a stream does not decode with these codebooks (unknown context, table or state)");
done:
#undef TRY
    (void)hipFree(d_tmpl); (void)hipFree(d_off); (void)hipFree(d_arena); (void)hipFree(d_ct); (void)hipFree(d_rc);
    (void)hipFree(d_first); (void)hipFree(d_sizes); (void)hipFree(d_status);
    free(h_arena); free(h_off); free(h_ct); free(h_qs); free(h_rc); free(h_first);
    return rc;
}
