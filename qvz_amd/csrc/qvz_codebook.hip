// qvz_codebook.hip - codebook generation for one cluster on gfx950 (C-ABI: include/qvz_codebook.h).
//
// Reference path restated: calculate_statistics' marginal pass (src/codebook.c:208-219) and generate_codebooks
// (src/codebook.c:355-468) with optimize_for_entropy (:230-269), compute_qpmf_quan_list (:274-289),
// compute_qpmf_list (:291-330), compute_xpmf_list (:332-349), generate_quantizer / apply_quantizer
// (src/quantizer.c:34-161).
//
// The work is N-independent, FP64 and decision-bearing: codebook bytes depend on `<` comparisons of sums, so
// every sum below is evaluated in the reference's index order with separately rounded multiply and add
// (this file is compiled with -ffp-contract=off). Parallelism comes from the independent dimensions only:
//   k_cond_pmf      one wave per (column, previous symbol) row: counts -> P(x | prev) by true division
//   k_marginals     one block, sequential over columns and over j (pmf.c:200-202 order), lanes over x
//   k_ctx_ptemp / k_ctx_propagate / k_ctx_xpmf  one column of the context recursion: thread per output element, each a sequential 72-term sum;
//                   p_temp(k,j) hoisted out of the idx loop (same per-term order => bit-identical)
//   k_lm_window     one wave per (context, state count): Lloyd-Max with a lane per candidate reconstruction point, exact
//                   (value, index) arg-min by wave shuffles, ballot masks + count-trailing-zeros for the
//                   boundary walk; the number-of-states search runs on device, steered by device log2 inside
//                   a guard band; host libm takes every decision that falls inside the band and computes the
//                   entropies that become `ratio`.
// Columns are sequentially dependent (column s needs the quantizers of column s-1), so the host drives one
// column at a time: ~2 small launches and 2 small copies per column.
#include <hip/hip_runtime.h>

#include <cmath>

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "qvz_codebook.h"
#include "qvz_internal.h"
#include "qvz_log2_glibc.h"

#define A QVZ_A
// The kernels of the column loop are short dependent chains, a wave or two per design: on a SIMD they share with the waves of
// a throughput kernel on another stream (the WELL draws run beside the codebooks) they would get every n-th issue slot; raised
// priority lets them issue whenever they can.
#define CB_PRIO() asm volatile("s_setprio 3")

// (the host-side codebook container, the distortion matrices and the serialisation live in host/qvz_codebook_host.c)

// ------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------
// P[row][x] = cnt[row][x] / total(row)   (recalculate_pmf, pmf.c:219-230; all zero when total == 0)
// CountT: u32 as in the reference (pmf.h:35-36: a file of 2^32 reads or more wraps them there), or u64 for the counters of
// such a file summed over its parts (same doubles whenever the u32 ones do not wrap)
template <typename CountT>
__global__ __launch_bounds__(64) void k_cond_pmf(const CountT *__restrict__ cnt, double *__restrict__ P) {
    const uint64_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const CountT c1 = cnt[row * A + lane];
    const CountT c2 = lane < A - 64 ? cnt[row * A + 64 + lane] : 0;
    CountT tot = c1 + c2;                                        // pmf_t.total (pmf.h:36)
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    const double t = (double)tot;
    P[row * A + lane] = tot ? ((double)c1) / t : 0.0;
    if (lane < A - 64) P[row * A + 64 + lane] = tot ? ((double)c2) / t : 0.0;
}

// marg[0] = 1.0*P[0][0] + 0.0*0 ; marg[s][x] = sum_j (1.0*marg[s][x] + marg[s-1][j]*P[s][j][x]) in j order
// (codebook.c:213-218, combine_pmfs pmf.c:200-202)
__global__ __launch_bounds__(256) void k_marginals(const double *__restrict__ P, uint32_t C, double *__restrict__ marg) {
    // Columns are a chain (marg[s] needs marg[s-1]) and a column is a chain of 72 multiply-adds per symbol: ~0.5 us of arithmetic.
    // What a column cost beyond that was the round trip of its 72 x 72 tile P(X_s | X_{s-1}): requested one column ahead into
    // registers it still took 4.2 us per column (0.64 ms per codebook in round 5's kernel statistics) - __syncthreads() waits
    // for every outstanding global load, so each column paid the whole round trip at its first barrier. Now the barriers order
    // LDS only and TWO tiles are on their way (two register sets, even and odd columns) while a third is summed out of LDS.
#define MG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    __shared__ double prev[A];
    __shared__ double tile[A * A];                               // P_s[j][x], the column being summed
    constexpr uint32_t PER = (A * A + 255) / 256;                // tile elements per thread
    const uint32_t x = threadIdx.x;
    if (x < A) {
        const double m0 = 1.0 * P[x] + 0.0 * 0.0;
        prev[x] = m0;
        marg[x] = m0;
    }
    double ra[PER], rb[PER];
    auto request = [&](double *r, uint32_t s) {
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) { const uint32_t e = u * 256 + x; r[u] = P[(uint64_t)s * A * A + (e < A * A ? e : 0u)]; }
    };
    auto park = [&](const double *r) {
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) { const uint32_t e = u * 256 + x; if (e < A * A) tile[e] = r[u]; }
    };
    // One column: tile s is in LDS, tile s + 1 in `r1` (requested two columns ago), tile s + 2 on its way in the other set.
    auto column = [&](uint32_t s, double *r1) {
        double m = 0.0;
        if (x < A)
            for (int j = 0; j < A; ++j) m = 1.0 * m + prev[j] * tile[j * A + x];   // combine_pmfs, pmf.c:200-202: the same terms in the same order
        MG_BARRIER();                                            // everybody has read prev and the tile
        if (x < A) {
            prev[x] = m;
            marg[(uint64_t)s * A + x] = m;
        }
        if (s + 1 < C) park(r1);
        if (s + 3 < C) request(r1, s + 3);
        MG_BARRIER();
    };
    if (C > 1) { request(ra, 1); park(ra); }
    if (C > 2) request(ra, 2);                                   // even tiles travel in ra, odd ones in rb
    if (C > 3) request(rb, 3);
    MG_BARRIER();
    for (uint32_t s = 1; s < C; s += 2) {
        column(s, ra);
        if (s + 1 < C) column(s + 1, rb);
    }
#undef MG_BARRIER
}

// what the Lloyd-Max window kernel is told per context (k_lm_window)
#define LM_JUMP_BAND 1e-9   // how far below its target a row's entropy must lie for its walk to be known in advance (k_ctx_xpmf)
struct LmIn {
    double target;
    uint8_t mode;      // 0 skip, 1 window [startS, startS + W), 2 the single design startS
    uint8_t startS;
    uint8_t pad[6];
};

// One column of the context recursion, in three launches. Inputs (device): P, marg, previous column's qpmf, quantizer
// tables and ratios of column s-1, context symbols of column s. Outputs: qpmf[72][72] (row x, entry idx) and
// xpmf[nu][72] (row idx).
// The two 72-term sums (p_temp, and the xpmf rows) gather three operands per term from L1/L2; in one workgroup - one
// CU - they were bound by that CU's load path (64 us and 30 us of the 96 us per column). They are spread over the
// chip: a thread per output element as before, the same terms in the same order.
// (p_temp(k, j) = sum_x prev_qpmf[x][j] * P_{s-1}(k|x) * marg_{s-2}[x], codebook.c:318-321, of column s + 1 is computed by extra
// rows of column s's design window launch - k_lm_window's rider -, which needs column s's qpmf only.)

// qpmf rows: one workgroup per symbol k (rows are renormalised on their own), a thread per context idx
__global__ __launch_bounds__(128) void k_ctx_propagate(uint32_t s, uint32_t nprev, uint32_t nu,
                                                       const uint8_t *__restrict__ ctx_sym,   // [nu] symbols of column s
                                                       const uint8_t *__restrict__ qprev,     // [144][72] tables of column s-1
                                                       const double *__restrict__ rprev,      // [72] ratios of column s-1
                                                       const double *__restrict__ ptemp,      // [72][72] (s >= 2)
                                                       double *__restrict__ qpmf, uint32_t *__restrict__ stop) {
    __shared__ double pt[A], raw[A], sr[A];
    __shared__ uint8_t slo[A], shi[A];
    CB_PRIO();
    const uint32_t k = blockIdx.x, idx = threadIdx.x;
    // (ctx_sym, qprev and rprev are mapped HOST memory: a load is a PCIe round trip. The context symbol is asked for here, with the
    // tables and ratios below - behind the barrier it was a second round trip in a row.)
    const uint32_t sym_of_idx = idx < nu ? ctx_sym[idx] : 0u;
    if (k == 0 && idx < A) stop[idx] = 0xFFFFFFFFu;              // the design window that follows starts with no stop marks
    if (s == 1) {
        // compute_qpmf_quan_list (codebook.c:274-289): not renormalised
        if (idx < nu) {
            const double r0 = rprev[0];
            const uint32_t sym = sym_of_idx;
            double v = 0.0;
            if (qprev[k] == sym) v += r0;
            if (qprev[A + k] == sym) v += (1 - r0);
            qpmf[k * A + idx] = v;
        }
        return;
    }
    if (idx < nprev) {                                           // row k of p_temp, the ratios and what the quantizers map k to
        pt[idx] = ptemp[k * A + idx];
        sr[idx] = rprev[idx];
        slo[idx] = qprev[(2 * idx) * A + k];
        shi[idx] = qprev[(2 * idx + 1) * A + k];
    }
    __syncthreads();
    // qpmf[k][idx] += p_q_xq * p_temp over j                                          (codebook.c:301-322)
    if (idx < nu) {
        const uint32_t sym = sym_of_idx;
        double acc = 0.0;
#pragma unroll 8
        for (uint32_t j = 0; j < nprev; ++j) {                   // (eight terms' operands asked of LDS at a time; added in j order)
            double p_q_xq = 0.0;
            const double rj = sr[j];
            if (slo[j] == sym) p_q_xq += rj;                     // q_lo->ratio
            if (shi[j] == sym) p_q_xq += (1 - rj);               // q_hi->ratio
            acc += p_q_xq * pt[j];
        }
        raw[idx] = acc;
    }
    __syncthreads();
    // renormalize_pmf of the row (pmf.c:235-254): sequential total, true division when total > 0
    if (idx < nu) {
        double total = 0;
#pragma unroll 8
        for (uint32_t i = 0; i < nu; ++i) total += raw[i];
        const double v = raw[idx];
        qpmf[k * A + idx] = total > 0 ? v / total : v;
    }
}

// compute_xpmf_list (codebook.c:332-349): one workgroup per context idx, a thread per symbol k
// It also writes the plan of the first design window (LmIn), so that the window can be launched without a host round
// trip: whether the row is all zero (no design), whether its target is exactly 0 (entropy 0 <=> every entry is 0 or 1;
// or a target of 0 was asked for), else a target from the device's own log2 - in the kernel that value only steers
// which designs give up early, by more than the guard band; every decision is the host's, from libm values.
__global__ __launch_bounds__(128) void k_ctx_xpmf(const double *__restrict__ P, const double *__restrict__ marg, uint32_t s,
                                                  const double *__restrict__ qpmf, double *__restrict__ xpmf, double *__restrict__ xpmf_host,
                                                  LmIn *__restrict__ lin, int ratio_mode, double opt_ratio, int exact,
                                                  LmIn *__restrict__ lin_host) {
    __shared__ double row[A], nrow[A], ent[A], qcol[A], mcol[A];
    CB_PRIO();
    const uint32_t idx = blockIdx.x, k = threadIdx.x;
    const double *Ps = P + (uint64_t)s * A * A;
    const double *m1 = marg + (uint64_t)(s - 1) * A;
    // A thread per symbol k adds its 72 terms qpmf[x][idx] * P_s[x][k] * marg_{s-1}[x] in x order. Its 72 values of P_s are asked
    // for AT ONCE (one memory round trip; gathered term by term, eight at a time, the sum waited nine round trips in a row - beside
    // the WELL generator's stores most of the kernel's 19 us); the two operands every thread shares come through LDS. The workgroup
    // stays small in registers-times-threads and LDS: it has to find room on a CU beside the generator's workgroups.
    double pk[A];
    const uint32_t kk = k < A ? k : 0u;
#pragma unroll
    for (int x = 0; x < A; ++x) pk[x] = Ps[x * A + kk];
    if (k < A) {
        qcol[k] = qpmf[k * A + idx];
        mcol[k] = m1[k];
    }
    __syncthreads();
    if (k < A) {
        double acc = 0.0;
#pragma unroll
        for (int x = 0; x < A; ++x) acc += qcol[x] * pk[x] * mcol[x];
        row[k] = acc;
    }
    __syncthreads();
    if (k < A) {
        double total = 0;
        for (int i = 0; i < A; ++i) total += row[i];
        const double v = row[k];
        const double nv = total > 0 ? v / total : v;
        xpmf[idx * A + k] = nv;
        if (xpmf_host) xpmf_host[idx * A + k] = nv;               // the host's copy (mapped memory): its libm entropy is the target
        nrow[k] = nv;
        // every thread its own term; `exact`: with glibc's log2 restated (qvz_log2_glibc.h) the sum below IS the reference's
        // get_entropy of this row (pmf.c:141-155: the same terms, subtracted in index order), and its product with the -f
        // ratio is the target the host would compute - the host then takes it from here instead of calling libm 72 x 72 times
        ent[k] = lin && nv > 0.0 ? nv * (exact ? qvz_log2_glibc(nv) : log2(nv)) : 0.0;
    }
    __syncthreads();
    if (k == 0 && lin) {
        bool zero = true, degenerate = true;
        double h = 0.0;
        for (int i = 0; i < A; ++i) {
            const double x = nrow[i];
            if (x != 0.0) zero = false;
            if (x != 0.0 && x != 1.0) degenerate = false;
            h -= ent[i];
        }
        const bool target0 = ratio_mode ? (degenerate || opt_ratio == 0.0) : opt_ratio == 0.0;
        LmIn I;
        I.target = target0 ? 0.0 : ratio_mode ? h * opt_ratio : opt_ratio;
        I.mode = zero ? 0 : target0 ? 2 : 1;
        // A row whose own entropy is below the target: no quantizer's output has more (the output is a function of the symbol,
        // H(Q(X)) <= H(X); the sums differ by rounding, ~1e-14), so every step of the walk (codebook.c:256) says "below" and it
        // ends at the alphabet's size with the designs A - 1 and A - the only two of the row's A that are made.
        I.startS = I.mode == 1 && h < I.target - LM_JUMP_BAND ? A - 1 : 1;
        for (int i = 0; i < 6; ++i) I.pad[i] = 0;
        lin[idx] = I;
        if (exact && lin_host) lin_host[idx] = I;                  // mapped host memory: the host's copy of the plan and the exact target
    }
}

// ---- Lloyd-Max search: a window of state counts per context, one wave per design ----------------------
// The reference walks S = 1, 2, 3 ... until the design's entropy reaches the target (codebook.c:244-258). The
// designs for different S do not depend on each other, only the stopping decision is sequential, so a launch
// computes a window of consecutive S per context on separate CUs and the host replays the walk over the results.

struct LmDesign {
    double o[A];       // output pmf (apply_quantizer, quantizer.c:139-161)
    double hdev;       // device-side entropy (steering only; decisions inside the guard band are the host's)
    uint8_t q[A];
    uint8_t recon[A];
    uint8_t S;
    uint8_t valid;     // 0: not computed (S > 72, or a smaller S of this context already reached the target)
    uint8_t pad[6];
};

// What the host's walk over a window reads of every design - the entropy, the state count, whether it was computed - once more
// in 16 bytes of a dense array: the walk then reads a window's summaries line after line (a dozen cache lines that the
// prefetcher sees coming) instead of two lines out of every 736-byte LmDesign the device has just written to host memory.
struct LmSum {
    double hdev;
    uint8_t S, valid, pad[6];
};

// Region index of two symbols under a partition: the number of interior bounds (t = 1 .. S-1) not above the symbol.
// The bounds are read sixteen at a time (one LDS round trip) and compared from registers.
template <bool BOTH = true>
__device__ __forceinline__ void lm_region_of(const uint8_t *bounds, uint32_t S, uint32_t xa, uint32_t xb, uint32_t &ja,
                                             uint32_t &jb) {
    for (uint32_t t0 = 0; t0 < S; t0 += 16) {
        const uint4 w = *reinterpret_cast<const uint4 *>(bounds + t0);
        // every lane read the same sixteen bounds: kept in scalar registers, so that picking them apart is scalar work and
        // a lane spends two vector instructions per bound and symbol (compare, add) instead of six
        const uint32_t ww[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)w.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)w.y),
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)w.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)w.w)};
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            const uint32_t t = t0 + u;
            // t = 0 never counts; bounds past S - 1 are made unreachable (a scalar select) instead of masking two vector results
            const uint32_t bt = (t >= 1 && t < S) ? (ww[u >> 2] >> (8 * (u & 3))) & 0xFFu : 0xFFFFu;
            ja += bt <= xa ? 1u : 0u;
            if (BOTH) jb += bt <= xb ? 1u : 0u;
        }
    }
}

struct LmLds {
    alignas(16) uint8_t bounds[80];      // read sixteen at a time (lm_region_of)
    uint8_t q[2][A];
    uint8_t recon[2][A];
    double mv[A];            // expected distortion of every candidate reconstruction point within its own region
    uint32_t iters;          // Lloyd-Max iterations of the last design (diagnostics: QVZ_CB_DEBUG)
};

// One Lloyd-Max design (quantizer.c:34-132) for S states into slot `b`. All 64 lanes call it together.
// `stop` (may be null) holds the smallest S of this context known to reach the target: a design with a larger S
// is not needed by the walk and gives up (returns false).
// TOEP: the distortion is a function of |i - r| only (all of the reference's -d choices): `dl` then holds those 72
// values instead of the 72 x 72 matrix - 0.6 KB of LDS per design instead of 41 KB, i.e. twenty designs per CU instead
// of three, and no 41 KB copy at the start of every design.
#ifdef QVZ_PROFILE          // development build: shader cycles per phase of a Lloyd-Max design, summed over the designs
__device__ unsigned long long g_lm_prof[8];
#define LM_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); lmp[i] += t_ - lmt; lmt = t_; } while (0)
#else
#define LM_T(i) do { } while (0)
#endif
// A design is ONE wave (k_lm_window: 64-thread blocks): its lanes run in lockstep and its LDS operations complete in issue
// order, so what a block-wide barrier would order is ordered once the LDS counter has drained. __syncthreads() would also
// wait for every outstanding global access - here the look at `stop`, ~1000 cycles that are meant to overlap an iteration.
#define LM_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <bool TOEP, bool MONO>
__device__ bool lm_design(const double *__restrict__ p, const double *__restrict__ dl, uint32_t S, LmLds *L,
                          double *__restrict__ o, int b, const uint32_t *stop) {
    auto D = [&](uint32_t i, uint32_t r) -> double { return TOEP ? dl[i > r ? i - r : r - i] : dl[i * A + r]; };
    const uint32_t lane = threadIdx.x & 63;
    uint8_t *bounds = L->bounds, *recon = L->recon[b], *q = L->q[b];
    if (lane == 0) { bounds[0] = 0; bounds[S] = A; }                                  // :45-46
    for (uint32_t j = lane; j < S; j += 64) if (j >= 1) bounds[j] = (uint8_t)((j * A) / S);   // :47-49
    LM_SYNC();
    for (uint32_t j = lane; j < S; j += 64) recon[j] = (uint8_t)((bounds[j] + bounds[j + 1] - 1) / 2);   // :50-52
    LM_SYNC();

#ifdef QVZ_PROFILE
    unsigned long long lmp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lmt = __builtin_readcyclecounter();
#endif
    uint32_t changed = 1, iter = 0;
    // Symbols below zlo and from zhi on have p = 0 (real quality values use 40 of the 72): their terms in the sums of (:62-87)
    // are p * d = +0.0 exactly (TOEP: d is finite, the host checks) and adding +0.0 leaves a sum of non-negative terms bit
    // for bit what it was, so the candidate sums run over [zlo, zhi) only. The last region typically reaches from ~40 to
    // 72: the longest loop of an iteration shrinks from 8 batches of four terms to 1.
    uint32_t zlo = 0, zhi = A;
    if (TOEP) {
        const unsigned long long nzA = __ballot(p[lane] != 0.0), nzB = __ballot(lane + 64 < A && p[lane + 64 < A ? lane + 64 : A - 1] != 0.0);
        zlo = nzA ? (uint32_t)__builtin_ctzll(nzA) : nzB ? 64u + (uint32_t)__builtin_ctzll(nzB) : 0u;
        zhi = nzB ? 128u - (uint32_t)__builtin_clzll(nzB) : nzA ? 64u - (uint32_t)__builtin_clzll(nzA) : 0u;
    }
    while (changed && iter < 100) {                                                  // :57, quantizer.h:10
        changed = 0;
        iter += 1;
        // `stop` lives in global memory: looking at it costs this lone wave ~1100 cycles, a tenth of an iteration (asking an
        // iteration ahead did not hide it). Every fourth iteration is enough: a design learns of a smaller S that reached
        // the target at most three iterations late. (Also tried, end of round 3: the look as a hand-issued VECTOR load - the
        // uniform address makes it a scalar one, which every lgkmcnt wait between an iteration's phases waits for - asked for
        // at the top of every iteration and read at the top of the next: same bytes, the same 18 ms per codebook.)
        if (stop && (iter & 3) == 1 && __atomic_load_n(stop, __ATOMIC_RELAXED) < S) return false;   // wave-uniform
        LM_T(0);
        // New reconstruction points (:62-87). The regions of one iteration depend on the bounds only, which do not
        // move in this phase, so when the bounds form a partition (they always do unless an earlier walk left stale
        // ones behind) every candidate r is evaluated in its own region at once: the loop is as long as the widest
        // region instead of the whole alphabet. Same sums in the same order, first minimum wins.
        bool mono = true;
        for (uint32_t t = lane; t < S; t += 64) mono = mono && bounds[t] <= bounds[t + 1];
        if (!__ballot(!mono)) {
            {
                // 72 candidates on 64 lanes: lanes 0..7 carry a second one (rr + 64) through the same loop iterations
                // instead of a second pass of the whole wave. A term outside a lane's region is added as +0.0, which
                // leaves a sum of non-negative terms bit for bit what it was.
                const uint32_t rrA = lane, rrB = lane + 64 < A ? lane + 64 : A - 1;
                const bool hasB = lane + 64 < A;
                // MONO and no mass from symbol 64 on (real quality values end near 41): a candidate r >= 64 costs more than
                // r - 1 - every term p_i f(r - i) with i < 64 <= r grows with r, f strictly increasing - or, in a region without
                // mass, as much as every other one, where the first candidate wins: it is never the first minimum of its region
                // unless it is the region's first candidate, which the arg-min below starts from whatever its cost. So the
                // eight second candidates are not summed (they doubled the loop for all 64 lanes) and their cost reads DBL_MAX.
                const bool skipB = MONO && zhi <= 64;
                uint32_t jA = 0, jB = 0;
                if (skipB) lm_region_of<false>(bounds, S, rrA, rrB, jA, jB); else lm_region_of<true>(bounds, S, rrA, rrB, jA, jB);
                uint32_t a0 = bounds[jA], a1 = bounds[jA + 1];
                a0 = a0 > zlo ? a0 : zlo; a1 = a1 < zhi ? a1 : zhi; a1 = a1 > a0 ? a1 : a0;   // the region's part with mass
                const uint32_t lenA = a1 - a0;
                double mA = 0.0, mB = DBL_MAX;
                if (skipB) {
                    LM_T(1);
                    for (uint32_t k = 0; __ballot(k < lenA); k += 4) {
                        double vA[4];
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) {
                            const uint32_t iA = k + u < lenA ? a0 + k + u : a0;
                            vA[u] = p[iA] * D(iA, rrA);
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) mA += k + u < lenA ? vA[u] : 0.0;
                    }
                } else {
                    uint32_t c0 = bounds[jB], c1 = hasB ? bounds[jB + 1] : c0;
                    c0 = c0 > zlo ? c0 : zlo; c1 = c1 < zhi ? c1 : zhi; c1 = c1 > c0 ? c1 : c0;
                    const uint32_t lenB = c1 - c0;
                    LM_T(1);
                    // A design is one wave alone on its SIMD: every LDS access it waits for costs its full latency (~130
                    // cycles), so four terms are fetched and multiplied at a time and then added in index order.
                    mB = 0.0;
                    for (uint32_t k = 0; __ballot(k < lenA || k < lenB); k += 4) {
                        double vA[4], vB[4];
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) {
                            const uint32_t iA = k + u < lenA ? a0 + k + u : a0, iB = k + u < lenB ? c0 + k + u : c0;
                            vA[u] = p[iA] * D(iA, rrA);
                            vB[u] = p[iB] * D(iB, rrB);
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) {
                            mA += k + u < lenA ? vA[u] : 0.0;
                            mB += k + u < lenB ? vB[u] : 0.0;
                        }
                    }
                }
                L->mv[rrA] = mA;
                if (hasB) L->mv[rrB] = mB;
            }
            LM_SYNC();
            LM_T(2);
            uint32_t chg = 0;
            for (uint32_t j = lane; j < S; j += 64) {
                const uint32_t b0 = bounds[j], b1 = bounds[j + 1];
                double best_v = DBL_MAX;
                uint32_t best_r = b0;                                                // min_r = bounds[j]
                // MONO: a candidate below the first symbol with mass costs more than its right neighbour, one from the last
                // symbol with mass on more than its left neighbour (every term p_i f(|i - r|) moves the same way, f strictly
                // increasing), so the first minimum of a region with mass lies in [zlo, zhi); a region without mass either lies
                // outside that range (nothing is scanned, the first candidate stands) or begins inside it (the scan begins at its
                // first candidate, all costs are 0, the first one stands). The widest region is the last one - it runs to
                // symbol 71 whatever the data - and its scan ends at zhi (~41) now.
                const uint32_t s0 = MONO ? (b0 > zlo ? b0 : zlo) : b0, s1 = MONO ? (b1 < zhi ? b1 : zhi) : b1;
                for (uint32_t r = s0; r < s1; r += 8) {              // eight candidates per LDS round trip, compared in order
                    double v[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) v[u] = L->mv[r + u < s1 ? r + u : s0];
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u)
                        if (r + u < s1 && v[u] < best_v) { best_v = v[u]; best_r = r + u; }
                }
                if (best_r != recon[j]) { chg = 1; recon[j] = (uint8_t)best_r; }
            }
            changed = __ballot(chg) ? 1u : 0u;
            LM_T(3);
        } else {
            for (uint32_t j = 0; j < S; ++j) {
                const uint32_t b0 = bounds[j], b1 = bounds[j + 1];
                const uint32_t r1 = b0 + lane, r2 = r1 + 64;
                double best_v = DBL_MAX;
                uint32_t best_r = b0;
                if (r1 < b1) {
                    double m = 0.0;
                    for (uint32_t i = b0; i < b1; ++i) m += p[i] * D(i, r1);
                    if (m < best_v) { best_v = m; best_r = r1; }
                }
                if (r2 < b1) {
                    double m = 0.0;
                    for (uint32_t i = b0; i < b1; ++i) m += p[i] * D(i, r2);
                    if (m < best_v) { best_v = m; best_r = r2; }
                }
                // first minimum wins (strict '<' scan in r order) == lexicographic min of (value, r)
                for (int off = 32; off > 0; off >>= 1) {
                    const double ov = __shfl_xor(best_v, off, 64);
                    const uint32_t orr = __shfl_xor(best_r, off, 64);
                    if (ov < best_v || (ov == best_v && orr < best_r)) { best_v = ov; best_r = orr; }
                }
                if (best_r != recon[j]) {
                    changed = 1;
                    if (lane == 0) recon[j] = (uint8_t)best_r;
                }
            }
        }
        LM_SYNC();
        // Boundary walk (:93-107). flag(t, j) = d(j, recon[t+1]) < d(j, recon[t]) for j = 1..70 as two ballot masks;
        // "advance at the first flagged j after the previous boundary" is a count-trailing-zeros. The walk stops at
        // the first state without such a j; the bounds beyond it keep their old values, as in the reference.
        if (MONO) {
            // d(j, r) = f(|j - r|) with f strictly increasing: d(j, rb) < d(j, ra) <=> |j - rb| < |j - ra|, i.e. j beyond
            // the midpoint of ra and rb on rb's side. The first flagged j after the previous boundary is then arithmetic,
            // not a search: no LDS gathers, no ballots.
            // The reconstruction points are read eight at a time (recon[] is 8-byte aligned in both slots).
            // Every lane would compute the same walk: the eight reconstruction points of a step go to scalar registers and
            // the walk is scalar arithmetic with scalar branches (as vector code every `if` of it was an exec-mask region).
            if (S <= 64) {
                // ... and the walk itself is a scan: with c_t = (ra + rb) / 2 + 1 where rb > ra (0, which never binds, elsewhere),
                // jn_t = max(c_t, jn_{t-1} + 1) from jn_{-1} = 0 is t + max(1, max_{s <= t} (c_s - s)): lane t takes boundary t, a
                // prefix maximum over the lanes (six shuffles) replaces S - 1 dependent scalar steps of ten instructions each -
                // a fifth of an iteration of this lone wave. The walk stops at the first t whose own test fails (the same tests on
                // the same numbers: a lane's jn_t is exact whenever every earlier lane passed); the bounds beyond keep their values.
                const uint32_t t = lane;
                const bool in = t + 1 < S;
                const uint32_t ra = recon[in ? t : 0], rb = recon[in ? t + 1 : 0];
                int32_t m = in ? (rb > ra ? (int32_t)((ra + rb) / 2 + 1) - (int32_t)t : -(int32_t)t) : -(int32_t)(1 << 20);
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int32_t o = __shfl_up(m, off, 64);
                    m = (int)lane >= off && o > m ? o : m;
                }
                const uint32_t jn = t + (uint32_t)(m > 1 ? m : 1);
                const bool stop_here = in && ((rb <= ra && (rb == ra || 2 * jn >= ra + rb)) || jn > A - 2);
                const unsigned long long stops = __ballot(stop_here);
                const uint32_t first = stops ? (uint32_t)__builtin_ctzll(stops) : 64u;
                if (in && t < first) bounds[t + 1] = (uint8_t)jn;
            } else {
            uint32_t jprev = 0;
            bool walking = true;
            for (uint32_t t0 = 0; t0 + 1 < S && walking; t0 += 8) {
                const uint2 wv = *reinterpret_cast<const uint2 *>(recon + t0);
                const uint32_t nxv = recon[t0 + 8 < A ? t0 + 8 : A - 1];
                const uint32_t wx = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv.x), wy = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv.y);
                const uint32_t nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxv);
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t t = t0 + u;
                    if (!walking || t + 1 >= S) { walking = walking && t + 1 < S; continue; }
                    const uint32_t ra = ((u < 4 ? wx : wy) >> (8 * (u & 3))) & 0xFFu;
                    const uint32_t rb = u == 7 ? nx : ((u + 1 < 4 ? wx : wy) >> (8 * ((u + 1) & 3))) & 0xFFu;
                    uint32_t jn = jprev + 1;
                    if (rb > ra) { const uint32_t c = (ra + rb) / 2 + 1; jn = c > jn ? c : jn; }     // j > (ra + rb) / 2
                    else if (rb == ra || 2 * jn >= ra + rb) { walking = false; continue; }            // j < (ra + rb) / 2
                    if (jn > A - 2) { walking = false; continue; }
                    if (lane == 0) bounds[t + 1] = (uint8_t)jn;
                    jprev = jn;
                }
            }
            }
        } else {
            uint32_t jprev = 0;
            for (uint32_t t = 0; t + 1 < S; ++t) {
                const uint32_t ra = recon[t], rb = recon[t + 1];
                const uint32_t j1 = lane, j2 = lane + 64;
                const int f1 = (j1 >= 1) && (D(j1, rb) < D(j1, ra));
                const int f2 = (j2 <= A - 2) && (D(j2 <= A - 1 ? j2 : A - 1, rb) < D(j2 <= A - 1 ? j2 : A - 1, ra));
                const unsigned long long m1 = __ballot(f1), m2 = __ballot(f2);
                const unsigned long long a1 = jprev < 63 ? (m1 & (~0ull << (jprev + 1))) : 0ull;
                uint32_t jn;
                if (a1) {
                    jn = (uint32_t)__builtin_ctzll(a1);
                } else {
                    const uint32_t lo = jprev >= 64 ? jprev - 64 + 1 : 0;
                    const unsigned long long a2 = lo < 64 ? (m2 & (~0ull << lo)) : 0ull;
                    if (!a2) break;
                    jn = 64 + (uint32_t)__builtin_ctzll(a2);
                }
                if (lane == 0) bounds[t + 1] = (uint8_t)jn;
                jprev = jn;
            }
        }
        LM_SYNC();
        LM_T(4);
    }

    for (uint32_t i = lane; i < A; i += 64) q[i] = 0;                                 // calloc, quantizer.c:14
    LM_SYNC();
    bool part = true;                                                                // the bounds are a partition of 0 .. 71
    for (uint32_t t = lane; t < S; t += 64) part = part && bounds[t] <= bounds[t + 1];
    if (!__ballot(!part)) {
        // :112-116 with regions that do not overlap: every symbol takes the reconstruction point of its own region
        const uint32_t iA = lane, iB = lane + 64 < A ? lane + 64 : A - 1;
        uint32_t jA = 0, jB = 0;
        lm_region_of(bounds, S, iA, iB, jA, jB);
        q[iA] = recon[jA];
        if (lane + 64 < A) q[iB] = recon[jB];
        LM_SYNC();
    } else {
        for (uint32_t j = 0; j < S; ++j) {                                           // :112-116, later regions overwrite
            const uint32_t b0 = bounds[j], b1 = bounds[j + 1];
            const uint8_t rj = recon[j];
            if (b0 + lane < b1) q[b0 + lane] = rj;
            if (b0 + lane + 64 < b1) q[b0 + lane + 64] = rj;
            LM_SYNC();
        }
    }
    {                                                                                // apply_quantizer :155-157
        // both output symbols of a lane (v, v + 64) in one sweep over i, eight inputs per LDS round trip; what does
        // not map to the lane's symbol is added as +0.0 (the sums are of non-negative terms, in index order)
        const uint32_t vA = lane, vB = lane + 64;
        double accA = 0.0, accB = 0.0;
        for (uint32_t i = 0; i < A; i += 8) {
            uint32_t qq[8];
            double pp[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) { qq[u] = q[i + u]; pp[u] = p[i + u]; }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                accA += qq[u] == vA ? pp[u] : 0.0;
                accB += qq[u] == vB ? pp[u] : 0.0;
            }
        }
        o[vA] = accA;
        if (vB < A) o[vB] = accB;
    }
    if (lane == 0) L->iters = iter;
    LM_SYNC();
    LM_T(5);
#ifdef QVZ_PROFILE
    if (lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_lm_prof[i], lmp[i]);
        atomicAdd(&g_lm_prof[6], (unsigned long long)iter);
        atomicAdd(&g_lm_prof[7], 1ull);
    }
#endif
    return true;
}

// get_entropy (pmf.c:141-155) of a design's output pmf, bit for bit: the 72 terms p * log2(p) by the lanes (glibc's log2
// restated, one IEEE multiply), then ONE lane subtracts them in index order as the reference's loop does. `term` is LDS.
__device__ double lm_entropy_exact(const double *o, double *term) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t v = lane; v < A; v += 64) {
        const double x = o[v];
        term[v] = x > 0.0 ? x * qvz_log2_glibc(x) : 0.0;         // e - 0.0 == e: a skipped term changes nothing
    }
    LM_SYNC();                                                   // one wave: its LDS operations complete in order
    double e = 0.0;
    if (lane == 0)
        for (uint32_t v = 0; v < A; ++v) e -= term[v];
    return __shfl(e, 0, 64);
}

__device__ double lm_entropy_dev(const double *o) {   // steering value only (device log2, tree order)
    const uint32_t lane = threadIdx.x & 63;
    double t = 0.0;
    for (uint32_t v = lane; v < A; v += 64) {
        const double x = o[v];
        if (x > 0.0) t += x * log2(x);
    }
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    return -t;
}

struct LmRider {           // k_ctx_ptemp of the next column riding on a window launch (rows == 0: none)
    const double *P, *marg, *qpmf;
    double *ptemp;
    uint32_t s, nprev, rows;
};

// grid (contexts, window); block (ctx, w) designs S = startS + w. distT[i*72 + r] = d(i, r) (lanes walk r: no LDS
// bank conflicts). blockIdx.x runs fastest, so the small S of every context are dispatched first.
template <bool TOEP, bool MONO>
__global__ __launch_bounds__(64) void k_lm_window(const double *__restrict__ pm, const double *__restrict__ distT,
                                                  const LmIn *__restrict__ in, LmDesign *__restrict__ out, LmSum *__restrict__ sum,
                                                  uint32_t *__restrict__ stop, double guard, int exact, LmRider rider) {
    CB_PRIO();
    // Rows of the grid beyond the window (blockIdx.y >= rider.rows) carry k_ctx_ptemp of the NEXT column: it needs this
    // column's qpmf only, and as blocks of this launch it runs on CUs the designs leave idle instead of being one more
    // dependent launch in the stream (a dependent launch costs ~11 us on this runtime, the kernel itself 15-23 us).
    if (rider.rows && blockIdx.y >= rider.rows) {
        const uint32_t e = ((blockIdx.y - rider.rows) * gridDim.x + blockIdx.x) * 64 + threadIdx.x;
        if (e >= A * rider.nprev) return;
        const double *Pc = rider.P + (uint64_t)(rider.s - 1) * A * A;
        const double *mm = rider.marg + (uint64_t)(rider.s - 2) * A;
        const uint32_t k = e / rider.nprev, j = e % rider.nprev;
        double pt = 0;
#pragma unroll 8
        for (int x = 0; x < A; ++x) pt += rider.qpmf[x * A + j] * Pc[x * A + k] * mm[x];   // k_ctx_ptemp's sum, term for term
        rider.ptemp[k * A + j] = pt;
        return;
    }
    extern __shared__ double sm[];
    double *dl = sm;
    double *p = sm + (TOEP ? A : A * A);
    double *o = p + A;
    LmLds *L = reinterpret_cast<LmLds *>(o + A);
    const uint32_t ctx = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    const LmIn I = in[ctx];
    LmDesign *O = out + (size_t)ctx * (rider.rows ? rider.rows : gridDim.y) + w;
    LmSum *Q = sum + (size_t)ctx * (rider.rows ? rider.rows : gridDim.y) + w;
    const uint32_t S = (uint32_t)I.startS + w;
    // (`out` is mapped host memory: the results need no copy; the k_cb_flag launched behind this kernel tells the host.
    // Letting the last block raise the flag instead needs a system-scope fence in every block: measured, with the 72 x 72
    // block windows of -r 1.0 that cost 13 ms per codebook.)
    const unsigned long long t_begin = __builtin_readcyclecounter();
    const bool wanted = I.mode == 1 ? S <= A : (I.mode == 2 && w == 0);
    auto not_computed = [&]() {
        if (lane == 0) {
            O->S = (uint8_t)S; O->valid = 0;
            LmSum q; q.hdev = 0.0; q.S = (uint8_t)S; q.valid = 0; for (int i = 0; i < 6; ++i) q.pad[i] = 0;
            *Q = q;
        }
    };
    if (!wanted || (I.mode == 1 && __atomic_load_n(stop + ctx, __ATOMIC_RELAXED) < S)) { not_computed(); return; }
    for (uint32_t e = lane; e < (TOEP ? A : A * A); e += 64) dl[e] = distT[e];   // TOEP: row 0 = d(0, r) = f(r)
    for (uint32_t e = lane; e < A; e += 64) p[e] = pm[(uint64_t)ctx * A + e];
    __syncthreads();
    if (!lm_design<TOEP, MONO>(p, dl, S, L, o, 0, I.mode == 1 ? stop + ctx : nullptr)) { not_computed(); return; }
    // exact: hdev IS the libm entropy of o and every decision of the walk can be taken from it (no guard band);
    // L->mv (72 doubles of LDS) is free once the design is done
    const double H = exact ? lm_entropy_exact(o, L->mv) : lm_entropy_dev(o);
    for (uint32_t e = lane; e < A; e += 64) {
        O->o[e] = o[e];
        O->q[e] = L->q[0][e];
        O->recon[e] = e < S ? L->recon[0][e] : 0;
    }
    if (lane == 0) {
        O->S = (uint8_t)S; O->valid = 1; O->hdev = H;
        {
            LmSum q; q.hdev = H; q.S = (uint8_t)S; q.valid = 1; for (int i = 0; i < 6; ++i) q.pad[i] = 0;
            *Q = q;
        }
        O->pad[0] = (uint8_t)(L->iters > 255 ? 255 : L->iters);                       // diagnostics (QVZ_CB_DEBUG): iterations,
        const unsigned long long dt = (__builtin_readcyclecounter() - t_begin) >> 8;   // shader cycles / 256
        O->pad[1] = (uint8_t)(dt > 65535 ? 255 : dt >> 8);
        O->pad[2] = (uint8_t)dt;
        // surely >= target (exact: simply not below it, codebook.c:249-258): the walk ends here or earlier
        // (exact mode: the host asks its libm again when H lies within QVZ_EXACT_BAND of the target - see there -, so a design
        // that close to it must not tell the larger ones to give up either)
        // (`guard` is that band then)
        if (I.mode == 1 && S > 1 && H - I.target > guard) atomicMin(stop + ctx, S);
    }
}

// ---- the per-column hand-over between device and host without stream synchronisation ------------------------------------
// A column costs one exchange with the host (its libm entropies decide, section "host driver" below). Through
// hipMemcpyAsync + hipStreamSynchronize that exchange took ~140 us of a column's ~250; the kernels' own time is ~110 us.
// So the results go straight into mapped, coherent host memory (the kernels' output pointers are host memory), a last
// one-thread kernel raises a sequence number there, and the host spins on that word; what the next column needs from the
// host (quantizer tables, ratios, context symbols) is read from mapped host memory by a first small kernel.
__global__ __launch_bounds__(64) void k_cb_flag(volatile uint32_t *flag, uint32_t seq) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        *flag = seq;
    }
}

__global__ __launch_bounds__(256) void k_cb_copy_f64(const double *__restrict__ src, double *__restrict__ dst, uint32_t count) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------
static double entropy_libm(const double *p) {                   // get_entropy, pmf.c:141-155
    double e = 0.0;
    for (int i = 0; i < A; ++i)
        if (p[i] > 0.0) e -= p[i] * log2(p[i]);
    return e;
}

// alphabet_union (pmf.c:312-357): a merge of two symbol lists
static uint32_t merge_lists(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb, uint8_t *out) {
    uint8_t tmp[4 * A];
    uint32_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) tmp[k] = a[i++];
        else if (a[i] == b[j]) { tmp[k] = a[i]; ++i; ++j; }
        else tmp[k] = b[j++];
        ++k;
    }
    while (i < na) tmp[k++] = a[i++];
    while (j < nb) tmp[k++] = b[j++];
    memcpy(out, tmp, k);
    return k;
}

struct ZeroDesign { uint8_t q[A], recon[A]; };

struct CbWork {
    double *P = nullptr, *marg = nullptr, *qpmf[2] = {nullptr, nullptr}, *xpmf = nullptr, *dist = nullptr, *zero_rows = nullptr;
    double *ptemp = nullptr;
    LmIn *lin = nullptr;
    LmDesign *hout = nullptr;                      // the design windows' results: mapped host memory the kernels write
    LmSum *hsum = nullptr;                         // ... and their summaries (entropy, state count, computed or not), densely
    uint32_t *stop = nullptr;
    uint8_t *pin = nullptr;                        // pinned staging for everything else that crosses PCIe per column
    uint32_t *flag = nullptr;                      // mapped host word the device raises when a column's results are in host memory
    uint32_t seq = 0;
    uint32_t capC = 0;                             // columns the buffers were sized for (0: nothing allocated)
    int device = -1;
    void release() {
        void *ptrs[] = {P, marg, qpmf[0], qpmf[1], xpmf, dist, zero_rows, ptemp, lin, stop};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        if (hout) (void)hipHostFree(hout);
        if (hsum) (void)hipHostFree(hsum);
        if (pin) (void)hipHostFree(pin);
        if (flag) (void)hipHostFree(flag);
        *this = CbWork();
    }
};

// Workspaces are kept between calls (one per concurrent caller): hipMalloc / hipFree of a dozen small buffers per
// codebook costs milliseconds, and hipFree waits for every stream of the device - including kernels of other work the
// caller overlaps with the generation.
static std::mutex g_cbw_mu;
static std::vector<CbWork *> g_cbw_free;
static CbWork *cbw_acquire() {
    std::lock_guard<std::mutex> lk(g_cbw_mu);
    if (g_cbw_free.empty()) return new CbWork();
    CbWork *w = g_cbw_free.back();
    g_cbw_free.pop_back();
    return w;
}
static void cbw_give_back(CbWork *w) {
    std::lock_guard<std::mutex> lk(g_cbw_mu);
    g_cbw_free.push_back(w);
}

static thread_local uint64_t g_deferred = 0;   // per calling thread: clusters may be generated by concurrent host threads
extern "C" uint64_t qvz_hip_codebook_deferred_decisions(void) { return g_deferred; }

#define LM_LDS_BYTES(toep) (((toep ? A : A * A) + A + A) * sizeof(double) + sizeof(LmLds))

static void store_design(qvz_codebook_t *cb, uint32_t s, uint32_t qi, const uint8_t *q, const uint8_t *recon, uint32_t S) {
    const size_t t = ((size_t)s * 2 * A + qi) * A;
    memcpy(cb->q + t, q, A);
    memset(cb->recon + t, 0, A);
    memcpy(cb->recon + t, recon, S);
    memset(cb->out_idx + t, QVZ_NOT_FOUND, A);
    for (uint32_t j = 0; j < S; ++j) cb->out_idx[t + recon[j]] = (uint8_t)j;       // alphabet_compute_index, pmf.c:365-382
    cb->nstates[(size_t)s * 2 * A + qi] = (uint8_t)S;
}

static double cb_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static thread_local unsigned long long cb_stat[8];   // QVZ_CB_DEBUG: designs run, their iterations, per-column maxima summed over columns; [6] windows after a column's first; [7] walks known in advance
static thread_local double cb_wait[6];   // [4], [5]: host time between a hand-over seen and the next column's first launch call, and in that column's launch calls
   // seconds spent waiting for the context propagation / the design windows (QVZ_CB_DEBUG)
// exact mode: |entropy - target| at or below this is re-decided by the host's libm (a one-bit miss of the device's log2 restatement
// moves an entropy of at most ~6.2 bits by a few ulps of 8.9e-16; the band is a thousand times that and is met about once in 1e12 decisions)
#define QVZ_EXACT_BAND 1e-12   /* QVZ_CB_LOG2=band=<x> overrides it (tests: a huge band makes the libm re-decide everything) */
static std::atomic<unsigned long long> g_exact_rechecked{0}, g_exact_mismatch{0};
extern "C" void qvz_hip_codebook_exact_recheck_counts(unsigned long long *rechecked, unsigned long long *mismatched) {
    if (rechecked) *rechecked = g_exact_rechecked.load();
    if (mismatched) *mismatched = g_exact_mismatch.load();
}
static std::atomic<int> g_cb_active{0};   // qvz_hip_generate_codebook calls in flight (one per cluster when a caller runs them concurrently)
// ---- is the device's restatement of glibc's log2 THIS host's log2? ---------------------------------------------------------
// The exact path (device entropies decide, the host calls libm no more) is taken only after the device function has agreed
// with the host's log2, bit for bit, on 2^20 arguments of the kinds the pmfs hold (count / total, products, values next to
// 1, powers of two, arbitrary mantissas): once per process and device, on a host thread of its own when the caller says so
// early (qvz_hip_codebook_warmup), ~15 ms of libm calls. One differing bit - another libm, another glibc, a compiler that
// contracts - and the generation falls back to the hand-over in which the host's libm takes every decision.
// QVZ_CB_LOG2=host forces that path (the tests run both and compare codebook bytes).
__global__ __launch_bounds__(256) void k_log2_selftest(const double *__restrict__ x, double *__restrict__ y, uint32_t count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = qvz_log2_glibc(x[i]);
}

struct Log2Check {
    std::mutex mu;
    std::condition_variable cv;
    int state = 0;          // 0 not started, 1 running, 2 verified, 3 differs / could not run
    std::thread th;         // the checking thread: joined before the process leaves (log2_join_all)
};
// one per device ordinal, made on first use and never destroyed. A process may end, or never generate a codebook, while the
// check runs (bad input refused during the upload, a Python process that only builds the backend): exit() must not reach the
// HIP runtime's teardown with that thread still inside hipMalloc or a kernel launch, so the thread is joinable and an atexit
// hook - registered after the runtime's own, hence run before it - and the library's destructor wait for it (~15 ms at worst).
static std::mutex g_log2_mu;
static Log2Check *g_log2[64];
static void log2_join_all() {
    for (int dev = 0; dev < 64; ++dev) {
        Log2Check *L;
        {
            std::lock_guard<std::mutex> g(g_log2_mu);
            L = g_log2[dev];
        }
        if (!L) continue;
        std::thread t;
        {
            std::lock_guard<std::mutex> lk(L->mu);
            t.swap(L->th);
        }
        if (t.joinable()) t.join();
    }
}
__attribute__((destructor)) static void log2_join_at_unload() { log2_join_all(); }

static void log2_selftest_run(int dev, Log2Check *L) {
    const uint32_t M = 1u << 20;
    int verdict = 3;
    std::vector<double> hx(M), hy(M);
    uint64_t st = 0x243F6A8885A308D3ull;
    auto rnd = [&]() { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (uint32_t i = 0; i < M; ++i) {
        const uint64_t r = rnd();
        double x;
        switch (i & 7) {
            case 0: case 1: case 2: { const uint64_t t = 1 + (rnd() >> (rnd() % 40 + 24)), c = 1 + r % t; x = (double)c / (double)t; break; }   // count / total
            case 3: x = qvz_log2_from_bits((r >> 12) | ((uint64_t)(1023 - (rnd() % 64)) << 52)); break;                                          // any mantissa, exponent 0 .. -63
            case 4: x = 1.0 - ldexp((double)(r >> 11), -53 - (int)(rnd() % 30)); break;                                                         // just below 1
            case 5: x = qvz_log2_from_bits(0x3FEEA4AF00000000ull + r % 0x210AA00000000ull + (rnd() % 5) - 2); break;                             // the close-to-1 branch and its edges
            case 6: x = ldexp(1.0, -(int)(r % 64)) * (1.0 + (double)(rnd() % 3) * 0x1p-52); break;                                              // powers of two and neighbours
            default: x = (double)(1 + r % 1000000) * (double)(1 + rnd() % 1000000) * 0x1p-60; break;                                            // products
        }
        if (!(x > 0x1p-1000) || !(x < 0x1p+1000)) x = 0.5;
        hx[i] = x;
    }
    double *dx = nullptr, *dy = nullptr;
    hipStream_t s = nullptr;
    if (hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
        hipMalloc(&dx, M * sizeof(double)) == hipSuccess && hipMalloc(&dy, M * sizeof(double)) == hipSuccess &&
        hipMemcpyAsync(dx, hx.data(), M * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess) {
        hipLaunchKernelGGL(k_log2_selftest, dim3(M / 256), dim3(256), 0, s, (const double *)dx, dy, M);
        if (hipMemcpyAsync(hy.data(), dy, M * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
            verdict = 2;
            for (uint32_t i = 0; i < M; ++i) {
                const double ref = log2(hx[i]);                  // this host's libm: what the reference would call here
                if (memcmp(&ref, &hy[i], sizeof(double)) != 0) {
                    verdict = 3;
                    if (getenv("QVZ_CB_DEBUG")) fprintf(stderr, "qvz: device log2 differs from this host's: x=%a libm=%a device=%a -> host hand-over\n", hx[i], ref, hy[i]);
                    break;
                }
            }
        }
    }
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (s) (void)hipStreamDestroy(s);
    {
        std::lock_guard<std::mutex> lk(L->mu);
        L->state = verdict;
    }
    L->cv.notify_all();
}

static Log2Check *log2_check_start() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Log2Check *L;
    {
        std::lock_guard<std::mutex> g(g_log2_mu);
        if (!g_log2[dev]) g_log2[dev] = new Log2Check();
        L = g_log2[dev];
    }
    std::lock_guard<std::mutex> lk(L->mu);
    if (L->state == 0) {
        static std::once_flag hook;
        std::call_once(hook, [] { atexit(log2_join_all); });
        L->state = 1;
        L->th = std::thread(log2_selftest_run, dev, L);
    }
    return L;
}

extern "C" int qvz_hip_codebook_warmup(void) {                   // starts the check on a thread of its own; returns at once
    if (qvz_env_flag("QVZ_CB_LOG2", "host")) return QVZ_OK;
    return log2_check_start() ? QVZ_OK : qvz_set_error(QVZ_E_HIP, "no current device");
}

static bool log2_exact_on_this_device();
extern "C" int qvz_hip_codebook_exact_entropies(void) { return log2_exact_on_this_device() ? 1 : 0; }

static bool log2_exact_on_this_device() {
    if (qvz_env_flag("QVZ_CB_LOG2", "host")) return false;
    Log2Check *L = log2_check_start();
    if (!L) return false;
    std::unique_lock<std::mutex> lk(L->mu);
    L->cv.wait(lk, [&] { return L->state >= 2; });
    return L->state == 2;
}

static int generate(const void *d_cnt, bool wide, uint32_t C, const double *h_dist, int mode, double opt_ratio, double guard,
                    qvz_codebook_t *cb, hipStream_t st, CbWork &W) {
    const size_t tile = (size_t)A * A;
    int dev = 0;
    QVZ_HIP_CHECK(hipGetDevice(&dev));
    if (W.capC < C || W.device != dev) {
        W.release();
        QVZ_HIP_CHECK(hipMalloc(&W.P, (size_t)C * tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.marg, (size_t)C * A * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.qpmf[0], tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.qpmf[1], tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.xpmf, tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.ptemp, tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.dist, tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.zero_rows, tile * sizeof(double)));
        QVZ_HIP_CHECK(hipMalloc(&W.lin, A * sizeof(LmIn)));
        QVZ_HIP_CHECK(hipHostMalloc(&W.hout, 2 * tile * sizeof(LmDesign), hipHostMallocMapped));   // two halves, a column each in turn
        QVZ_HIP_CHECK(hipHostMalloc(&W.hsum, 2 * tile * sizeof(LmSum), hipHostMallocMapped));
        QVZ_HIP_CHECK(hipMalloc(&W.stop, A * sizeof(uint32_t)));
        QVZ_HIP_CHECK(hipHostMalloc(&W.pin, tile * sizeof(double) + A * sizeof(LmIn) + 2 * (2 * tile + A * sizeof(double) + 4 * A) + A * sizeof(LmIn), hipHostMallocMapped));
        QVZ_HIP_CHECK(hipHostMalloc(&W.flag, 64, hipHostMallocMapped));
        *W.flag = 0;
        W.seq = 0;
        W.capC = C;
        W.device = dev;
    }
    bool toep = true;                              // d(i, r) a function of |i - r|: the Lloyd-Max kernel keeps 72 values
    for (int i = 0; i < A && toep; ++i)
        for (int r = 0; r < A; ++r)
            if (memcmp(&h_dist[i + A * r], &h_dist[(i > r ? i - r : r - i)], sizeof(double)) != 0) { toep = false; break; }
    for (int d = 0; d < A && toep; ++d) toep = std::isfinite(h_dist[d]) && h_dist[d] >= 0.0;   // (the zero-mass terms are skipped: 0 * d must be +0.0)
    bool mono = toep;                              // ... and strictly increasing in |i - r|: the boundary walk is arithmetic
    for (int d = 1; d < A && mono; ++d) mono = h_dist[d] > h_dist[d - 1];
    {
        double hT[A * A];                          // d(i, r) = dist[i + 72 r] laid out with r fastest
        for (int i = 0; i < A; ++i)
            for (int r = 0; r < A; ++r) hT[i * A + r] = h_dist[i + A * r];
        QVZ_HIP_CHECK(hipMemcpyAsync(W.dist, hT, tile * sizeof(double), hipMemcpyHostToDevice, st));
        QVZ_HIP_CHECK(hipStreamSynchronize(st));
    }
    QVZ_HIP_CHECK(hipMemsetAsync(W.zero_rows, 0, tile * sizeof(double), st));

    if (wide) hipLaunchKernelGGL(k_cond_pmf<unsigned long long>, dim3(C * A), dim3(64), 0, st, (const unsigned long long *)d_cnt, W.P);
    else hipLaunchKernelGGL(k_cond_pmf<uint32_t>, dim3(C * A), dim3(64), 0, st, (const uint32_t *)d_cnt, W.P);
    hipLaunchKernelGGL(k_marginals, dim3(1), dim3(256), 0, st, W.P, C, W.marg);
    QVZ_HIP_CHECK(hipGetLastError());

    // One exchange with the host per column (two when a context's walk runs off the first design window). Everything that
    // crosses PCIe sits in mapped host memory that the kernels read and write themselves (see k_cb_flag); what the next
    // column's propagation needs from this one is double-buffered there.
    double *hx = reinterpret_cast<double *>(W.pin);
    LmIn *hin = reinterpret_cast<LmIn *>(hx + tile);
    double *hr2 = reinterpret_cast<double *>(hin + A);                  // [2][A]
    uint8_t *hq2 = reinterpret_cast<uint8_t *>(hr2 + 2 * A);            // [2][2 * tile]
    uint8_t *hu2 = hq2 + 2 * 2 * tile;                                  // [2][4 * A]
    LmIn *hplan = reinterpret_cast<LmIn *>(hu2 + 2 * 4 * A);            // [A] exact mode: the device's plan of a column with its exact targets
    LmDesign *hout = W.hout;                                            // (the column loop below alternates between the area's two halves)
    uint8_t *pin_d = nullptr;                                           // the same areas as the device sees them
    LmDesign *hout_d = nullptr, *hout_d0 = nullptr;
    LmSum *hsum_d = nullptr, *hsum_d0 = nullptr;
    uint32_t *flag_d = nullptr;
    QVZ_HIP_CHECK(hipHostGetDevicePointer((void **)&pin_d, W.pin, 0));
    QVZ_HIP_CHECK(hipHostGetDevicePointer((void **)&hout_d0, W.hout, 0));
    hout_d = hout_d0;
    QVZ_HIP_CHECK(hipHostGetDevicePointer((void **)&hsum_d0, W.hsum, 0));
    hsum_d = hsum_d0;
    QVZ_HIP_CHECK(hipHostGetDevicePointer((void **)&flag_d, W.flag, 0));
    double *hx_d = reinterpret_cast<double *>(pin_d);
    LmIn *hin_d = reinterpret_cast<LmIn *>(hx_d + tile);
    double *hr2_d = reinterpret_cast<double *>(hin_d + A);
    uint8_t *hq2_d = reinterpret_cast<uint8_t *>(hr2_d + 2 * A);
    uint8_t *hu2_d = hq2_d + 2 * 2 * tile;
    LmIn *hplan_d = reinterpret_cast<LmIn *>(hu2_d + 2 * 4 * A);
    // Entropies: the device's own (glibc's log2 restated, qvz_log2_glibc.h) once they have been checked against this host's
    // libm; then the targets and every `entropy < target` come from the device and the host calls libm no more. Otherwise
    // the device values only steer and the host's libm decides inside the guard band, as before.
    const bool exact = log2_exact_on_this_device();
    if (exact) { guard = QVZ_EXACT_BAND; (void)qvz_env_number("QVZ_CB_LOG2", "band", &guard); }   // the band inside which the host's libm is asked again (test hook: band=<x>)
    const bool cb_debug = getenv("QVZ_CB_DEBUG") != nullptr;
    // Waits for hand-over number `seq`, raised by a k_cb_flag (flag_after): everything enqueued before it has run and its
    // results are in host memory.
    auto await = [&](uint32_t seq, int which) -> int {
        const double t_ = cb_now();
        uint64_t polls = 0;
        while (__atomic_load_n(W.flag, __ATOMIC_ACQUIRE) != seq) {
            if ((++polls & 0xFFFF) == 0 && cb_now() - t_ > 5.0) {        // something is wrong: let the runtime say what
                QVZ_HIP_CHECK(hipStreamSynchronize(st));
                if (__atomic_load_n(W.flag, __ATOMIC_ACQUIRE) != seq) return qvz_set_error(QVZ_E_HIP, "codebook hand-over flag never arrived");
            }
        }
        cb_wait[which] += cb_now() - t_;
        if (cb_debug && cb_now() - t_ > 5e-3) fprintf(stderr, "  long hand-over wait: %.1f ms for flag %u (kind %d)\n", (cb_now() - t_) * 1e3, seq, which);
        return QVZ_OK;
    };
    auto flag_after = [&]() -> uint32_t {
        const uint32_t seq = ++W.seq;
        hipLaunchKernelGGL(k_cb_flag, dim3(1), dim3(64), 0, st, (volatile uint32_t *)flag_d, seq);
        return seq;
    };
    auto launch_window = [&](const double *d_rows, const LmIn *in, uint32_t nu, uint32_t win, LmRider rider = LmRider{nullptr, nullptr, nullptr, nullptr, 0, 0, 0}) -> uint32_t {
        const uint32_t extra = rider.rows ? (A * rider.nprev + 64 * nu - 1) / (64 * nu) : 0;     // rows of nu blocks x 64 threads for A * nprev sums
        if (mono) hipLaunchKernelGGL((k_lm_window<true, true>), dim3(nu, win + extra), dim3(64), LM_LDS_BYTES(true), st, d_rows, W.dist, in, hout_d, hsum_d, W.stop, guard, exact ? 1 : 0, rider);
        else if (toep) hipLaunchKernelGGL((k_lm_window<true, false>), dim3(nu, win + extra), dim3(64), LM_LDS_BYTES(true), st, d_rows, W.dist, in, hout_d, hsum_d, W.stop, guard, exact ? 1 : 0, rider);
        else hipLaunchKernelGGL((k_lm_window<false, false>), dim3(nu, win + extra), dim3(64), LM_LDS_BYTES(false), st, d_rows, W.dist, in, hout_d, hsum_d, W.stop, guard, exact ? 1 : 0, rider);
        return flag_after();
    };
    LmDesign lo_d[A], hi_d[A];                     // chosen pair per context of the current column

    // designs for the all-zero pmf depend only on (distortion, S): one launch, 72 waves
    ZeroDesign Z[A + 1];
    memset(hin, 0, A * sizeof(LmIn));
    for (int S = 1; S <= A; ++S) { hin[S - 1].mode = 2; hin[S - 1].startS = (uint8_t)S; }
    {
        const uint32_t seq = launch_window(W.zero_rows, hin_d, A, 1);
        QVZ_HIP_CHECK(hipGetLastError());
        const int rc_ = await(seq, 1);
        if (rc_) return rc_;
    }
    for (int S = 1; S <= A; ++S) { memcpy(Z[S].q, hout[S - 1].q, A); memcpy(Z[S].recon, hout[S - 1].recon, A); }
    uint32_t window0 = 16;                         // first window of a column: the previous column's largest S + 3

    // What the host does between a column's hand-over and the next column's first launch is on the chain of 150 dependent
    // columns (it was ~21 us of a column's ~116: struct copies of the chosen designs out of the results area, the codebook's
    // bookkeeping, 86 list merges, a 10 KB table copy). So only what the next launch needs is done first - the walk over the
    // window's entropies, the chosen tables and ratios into the mapped staging slot, the next column's context symbols - and
    // the column's entries in the codebook are written (finish_column) once the next column's kernels are on their way. The
    // results area has two halves in turn, so that a column's chosen designs stay where the kernels left them until then.
    struct Pick { const uint8_t *q, *recon; uint32_t S; };
    Pick pick[2 * A];                              // the chosen quantizer pair of every context of the column last decided
    double ratio_s[A];
    bool zero[A];
    uint32_t nu_fin = 0;
    auto finish_column = [&](uint32_t s) {          // store_cond_quantizers_indexed (codebook.c:152-157) and the context list of column s
        const uint8_t *uni = hu2 + (size_t)(s & 1) * 4 * A;
        cb->nctx[s] = (uint8_t)nu_fin;
        memcpy(cb->ctx_sym + (size_t)s * A, uni, nu_fin);
        for (uint32_t i = 0; i < nu_fin; ++i) cb->ctx_idx[(size_t)s * A + uni[i]] = (uint8_t)i;
        for (uint32_t j = 0; j < nu_fin; ++j) {
            store_design(cb, s, 2 * j, pick[2 * j].q, pick[2 * j].recon, pick[2 * j].S);
            store_design(cb, s, 2 * j + 1, pick[2 * j + 1].q, pick[2 * j + 1].recon, pick[2 * j + 1].S);
            cb->ratio[(size_t)s * A + j] = ratio_s[j];
            cb->qratio[(size_t)s * A + j] = (uint8_t)(ratio_s[j] * 128.);
            cb->live[(size_t)s * A + j] = zero[j] ? 0 : 1;
        }
    };
    uint32_t nprev = 0, nu_next = 1;
    double t_prev_seen = cb_now();
    int pq = 0;   // index of the qpmf buffer holding the previous column
    for (uint32_t s = 0; s < C; ++s) {
        uint32_t nu, win0 = 0, seq0;                                                // win0: width of a first window already run (s >= 1)
        double t_launch = cb_now();
        uint8_t *uni = hu2 + (size_t)(s & 1) * 4 * A;
        LmDesign *const hout = W.hout + (size_t)(s & 1) * tile;                     // this column's half of the results area
        hout_d = hout_d0 + (size_t)(s & 1) * tile;
        const LmSum *const hsum = W.hsum + (size_t)(s & 1) * tile;
        hsum_d = hsum_d0 + (size_t)(s & 1) * tile;
        const double *d_rows;
        if (s == 0) {                                                               // codebook.c:392-411
            nu = 1;
            uni[0] = 0;
            d_rows = W.P;                                                           // get_cond_pmf(list, 0, 0)
            hipLaunchKernelGGL(k_cb_copy_f64, dim3(1), dim3(256), 0, st, (const double *)W.P, hx_d, (uint32_t)A);
            seq0 = flag_after();
        } else {                                                                    // codebook.c:419-438
            nu = nu_next;                                                           // (the context symbols are in `uni` already)
            // (k_ctx_ptemp of this column ran behind the previous column's window, while the host was busy: it needs nothing
            // from the host.) The propagation reads the previous column's tables and ratios (slot (s - 1) & 1) and this
            // column's context symbols straight from mapped host memory: a few hundred bytes per workgroup.
            t_launch = cb_now();
            if (cb_debug) cb_wait[4] += t_launch - t_prev_seen;
            hipLaunchKernelGGL(k_ctx_propagate, dim3(A), dim3(128), 0, st, s, nprev, nu, (const uint8_t *)(hu2_d + (size_t)(s & 1) * 4 * A),
                               (const uint8_t *)(hq2_d + (size_t)((s - 1) & 1) * 2 * tile), (const double *)(hr2_d + (size_t)((s - 1) & 1) * A),
                               W.ptemp, W.qpmf[pq ^ 1], W.stop);
            // k_ctx_xpmf plans the first design window on the device (all-zero rows and zero targets are exact on both
            // sides; its log2-based target only steers which designs give up early), so the window follows at once
            hipLaunchKernelGGL(k_ctx_xpmf, dim3(nu), dim3(128), 0, st, W.P, W.marg, s, W.qpmf[pq ^ 1], W.xpmf, hx_d,
                               W.lin, mode == QVZ_MODE_RATIO ? 1 : 0, opt_ratio, exact ? 1 : 0, hplan_d);
            pq ^= 1;
            d_rows = W.xpmf;
            win0 = window0;
            // p_temp of column s + 1 needs this column's qpmf only: it rides on the window launch
            seq0 = s + 1 < C ? launch_window(d_rows, W.lin, nu, win0, LmRider{W.P, W.marg, W.qpmf[pq], W.ptemp, s + 1, nu, win0})
                             : launch_window(d_rows, W.lin, nu, win0);
        }
        QVZ_HIP_CHECK(hipGetLastError());
        if (cb_debug) cb_wait[5] += cb_now() - t_launch;
        if (s > 0) finish_column(s - 1);                                            // while this column's kernels run
        { const int rc_ = await(seq0, 0); if (rc_) return rc_; }
        const double t_seen = cb_now();
        if (cb_debug && win0) {                                                     // the first window's designs: how long, how many iterations
            uint32_t max_it = 0, max_cyc = 0, nvalid = 0, sum_it = 0, at_s = 0;
            for (uint32_t e = 0; e < nu * win0; ++e)
                if (hout[e].valid) {
                    const uint32_t cyc = ((uint32_t)hout[e].pad[1] << 8 | hout[e].pad[2]) << 8;
                    nvalid += 1; sum_it += hout[e].pad[0];
                    if (hout[e].pad[0] > max_it) max_it = hout[e].pad[0];
                    if (cyc > max_cyc) { max_cyc = cyc; at_s = hout[e].S; }
                }
            cb_stat[0] += nvalid; cb_stat[1] += sum_it; cb_stat[2] += max_it; cb_stat[3] += max_cyc; cb_stat[4] += nu * win0; cb_stat[5] += at_s;
        }

        // targets (libm) and the per-context plan
        double target[A];
        bool any_gpu = false;
        memset(hin, 0, A * sizeof(LmIn));
        const double t_tg = cb_now();
        for (uint32_t j = 0; j < nu; ++j) {
            if (exact && s > 0) {
                // the device's plan of this column: all-zero row (mode 0), target exactly 0 (mode 2), else the exact target -
                // get_entropy of the row times the -f ratio, or the -r rate (codebook.c:444-447) - computed beside the row
                zero[j] = hplan[j].mode == 0;
                target[j] = hplan[j].target;
                if (!zero[j]) { hin[j] = hplan[j]; any_gpu = true; }
                continue;
            }
            const double *pj = hx + (size_t)j * A;
            zero[j] = true;
            for (int i = 0; i < A; ++i) if (pj[i] != 0.0) { zero[j] = false; break; }
            const double hrow = zero[j] ? 0.0 : entropy_libm(pj);
            target[j] = mode == QVZ_MODE_RATIO ? hrow * opt_ratio : opt_ratio;     // codebook.c:444-447
            if (!zero[j]) {
                hin[j].target = target[j];
                hin[j].mode = target[j] == 0.0 ? 2 : 1;                            // :236-242 vs the search
                hin[j].startS = hin[j].mode == 1 && hrow < target[j] - LM_JUMP_BAND ? A - 1 : 1;   // as k_ctx_xpmf plans it
                any_gpu = true;
            }
        }
        cb_wait[2] += cb_now() - t_tg;
        bool jumped[A];                                                             // the plan starts at A - 1: the row's entropy is below its target
        for (uint32_t j = 0; j < nu; ++j) { jumped[j] = hin[j].mode == 1 && hin[j].startS == A - 1; cb_stat[7] += jumped[j] ? 1 : 0; }
        const LmDesign *lo_p[A], *hi_p[A];                                          // chosen pair per context: in the results area, or a copy (lo_d / hi_d)
        bool fresh[A];                                                              // chosen by the window just replayed: still in the results area
        uint32_t win = window0, smax = 1, round = 0;
        while (any_gpu) {
            if (round == 0 && win0) {
                // this window has been run already, with the same plan (all-zero rows and zero targets are exact on
                // both sides; the device's target differs from the libm one in the last bits, which only steers)
                ++round;
            } else {
            if (s == 0 || round) QVZ_HIP_CHECK(hipMemsetAsync(W.stop, 0xff, A * sizeof(uint32_t), st));   // (k_ctx_propagate resets it otherwise)
            ++round;
            cb_stat[6] += 1;
            const uint32_t seqw = launch_window(d_rows, hin_d, nu, win);         // the plan is read from host memory
            QVZ_HIP_CHECK(hipGetLastError());
            { const int rc_ = await(seqw, 1); if (rc_) return rc_; }
            }
            any_gpu = false;
            for (uint32_t j = 0; j < nu; ++j) {
                const LmDesign *d = hout + (size_t)j * win;
                const LmSum *q = hsum + (size_t)j * win;                            // the same designs' summaries: what the walk reads
                fresh[j] = false;
                if (hin[j].mode == 2) {                                             // target 0: one state, lo == hi
                    if (!q[0].valid) return qvz_set_error(QVZ_E_HIP, "column %u context %u: design missing", s, j);
                    lo_p[j] = &d[0]; hi_p[j] = &d[0]; fresh[j] = true;
                    hin[j].mode = 0;
                    continue;
                }
                if (hin[j].mode != 1) continue;
                // replay of the walk (codebook.c:249-258) over the window; d[0] is the last design already rejected
                // (or S = 1 at the start), so the walk resumes at d[1]
                hin[j].mode = 0;
                bool done = false;
                uint32_t w = 1;
                for (; w < win && (uint32_t)hin[j].startS + w <= A; ++w) {
                    if (!q[w].valid || !q[w - 1].valid)
                        return qvz_set_error(QVZ_E_HIP, "column %u context %u: design %u missing", s, j, hin[j].startS + w);
                    const double diff = q[w].hdev - target[j];
                    bool below;
                    if (exact && fabs(diff) <= guard) {
                        // The run-time check of the device's log2 is a sample: it catches another libm, it cannot prove equality
                        // on the arguments this very pmf produces. A single differing bit in one term moves an entropy by a few
                        // ulps (~1e-15), so a decision this close to its target - one in ~1e12 - is taken by the host's libm,
                        // target included, and a miss cannot flip it.
                        g_exact_rechecked += 1;
                        const double t_libm = mode == QVZ_MODE_RATIO && s > 0 ? entropy_libm(hx + (size_t)j * A) * opt_ratio : target[j];
                        const double h_libm = entropy_libm(d[w].o);
                        if (memcmp(&t_libm, &target[j], sizeof(double)) != 0 || memcmp(&h_libm, &q[w].hdev, sizeof(double)) != 0) g_exact_mismatch += 1;
                        below = h_libm < t_libm;
                    }
                    else if (exact) below = q[w].hdev < target[j];                  // hdev is get_entropy(d[w].o): codebook.c:249-258 as it stands
                    else if (fabs(diff) <= guard) { g_deferred += 1; below = entropy_libm(d[w].o) < target[j]; }
                    else below = diff < 0.0;
                    if (!below || q[w].S >= A) { done = true; break; }
                }
                if (done) {
                    lo_p[j] = &d[w - 1]; hi_p[j] = &d[w]; fresh[j] = true;
                    if (q[w].S > smax && !jumped[j]) smax = q[w].S;                // (a walk known in advance needs two designs, not a window up to A)
                } else {                                                            // keep walking from the last design
                    hin[j].mode = 1;
                    hin[j].startS = (uint8_t)(hin[j].startS + w - 1);
                    any_gpu = true;
                }
            }
            if (any_gpu) {
                win = win * 2 > A ? A : win * 2;
                for (uint32_t j = 0; j < nu; ++j)                                   // the next window overwrites the results area: what this one decided is copied out
                    if (fresh[j]) { lo_d[j] = *lo_p[j]; hi_d[j] = *hi_p[j]; lo_p[j] = &lo_d[j]; hi_p[j] = &hi_d[j]; }
            }
        }
        {
            // (a wider margin, or the largest of the last few columns instead of the last one's, saves some of the second windows
            // that -r targets need in a third of the columns and costs as much in wider first windows: HISTORY.md, round 5)
            const uint32_t margin = 3u;
            window0 = smax + margin < 8 ? 8 : (smax + margin > A ? A : smax + margin);
        }

        // ratios from the entropies (optimize_for_entropy :263-268), the chosen pairs
        const double t_rt = cb_now();
        for (uint32_t j = 0; j < nu; ++j) {
            double ratio;
            if (zero[j]) {
                // every design of a zero pmf has entropy 0: replay the search loop on the host
                if (target[j] == 0.0) {
                    pick[2 * j] = Pick{Z[1].q, Z[1].recon, 1};
                    pick[2 * j + 1] = Pick{Z[1].q, Z[1].recon, 1};
                    ratio = 1.0;
                } else {
                    uint32_t states = 1;
                    do { states += 1; } while (0.0 < target[j] && states < A);
                    pick[2 * j] = Pick{Z[states - 1].q, Z[states - 1].recon, states - 1};
                    pick[2 * j + 1] = Pick{Z[states].q, Z[states].recon, states};
                    if (0.0 < target[j]) ratio = 0.0;
                    else ratio = 1.0;                                               // lo_entropy (0) >= target
                }
            } else {
                pick[2 * j] = Pick{lo_p[j]->q, lo_p[j]->recon, lo_p[j]->S};
                pick[2 * j + 1] = Pick{hi_p[j]->q, hi_p[j]->recon, hi_p[j]->S};
                if (target[j] == 0.0) {
                    ratio = 1.0;
                } else {
                    const double lo_entropy = exact ? lo_p[j]->hdev : entropy_libm(lo_p[j]->o), hi_entropy = exact ? hi_p[j]->hdev : entropy_libm(hi_p[j]->o);
                    if (hi_entropy < target[j]) ratio = 0.0;
                    else if (lo_entropy >= target[j] || hi_entropy == lo_entropy) ratio = 1.0;
                    else ratio = (target[j] - hi_entropy) / (lo_entropy - hi_entropy);
                }
            }
            ratio_s[j] = ratio;
        }
        nu_fin = nu;
        // hand this column's tables and ratios to the next column's propagation, and make that column's context symbols:
        // the union of the chosen quantizers' output alphabets (alphabet_union over all of them, codebook.c:419-438)
        uint8_t *hq = hq2 + (size_t)(s & 1) * 2 * tile;   // slot s & 1: its last reader (the propagation of column s - 1) finished before that column's hand-over
        double *hr = hr2 + (size_t)(s & 1) * A;
        for (uint32_t t = 0; t < 2 * nu; ++t) memcpy(hq + (size_t)t * A, pick[t].q, A);
        for (uint32_t j = 0; j < A; ++j) hr[j] = j < nu ? ratio_s[j] : 0.0;
        if (s + 1 < C) {
            uint8_t *nxt = hu2 + (size_t)((s + 1) & 1) * 4 * A;
            // Merging sorted lists without duplicates one after the other (merge_lists, pmf.c:312-357) gives the sorted set of
            // everything in them: a mark per symbol. A list that is not strictly increasing (bounds an earlier walk left in
            // disorder can produce one) takes the merges themselves.
            uint8_t mark[A];
            memset(mark, 0, A);
            bool sorted = true;
            for (uint32_t t = 0; t < 2 * nu && sorted; ++t)
                for (uint32_t i = 0; i < pick[t].S; ++i) {
                    const uint8_t r = pick[t].recon[i];
                    if (r >= A || (i && r <= pick[t].recon[i - 1])) { sorted = false; break; }
                    mark[r] = 1;
                }
            uint32_t nn = 0;
            if (sorted) {
                for (uint32_t v = 0; v < A; ++v) if (mark[v]) nxt[nn++] = (uint8_t)v;
            } else {
                nn = pick[0].S;
                memcpy(nxt, pick[0].recon, nn);
                for (uint32_t t = 1; t < 2 * nu; ++t) nn = merge_lists(nxt, nn, pick[t].recon, pick[t].S, nxt);
            }
            if (nn > A) return qvz_set_error(QVZ_E_DATA, "column %u: %u contexts exceed the alphabet", s + 1, nn);
            nu_next = nn;
        }
        cb_wait[3] += cb_now() - t_rt;
        nprev = nu;
        t_prev_seen = t_seen;
    }
    finish_column(C - 1);
    return QVZ_OK;
}

static int generate_codebook(const void *d_cnt, bool wide, uint32_t C, const double *h_dist, int mode, double ratio,
                             double guard, qvz_codebook_t *out, void *stream);

extern "C" int qvz_hip_generate_codebook(const uint32_t *d_cnt, uint32_t C, const double *h_dist, int mode, double ratio,
                                         double guard, qvz_codebook_t *out, void *stream) {
    return generate_codebook(d_cnt, false, C, h_dist, mode, ratio, guard, out, stream);
}

extern "C" int qvz_hip_generate_codebook64(const uint64_t *d_cnt, uint32_t C, const double *h_dist, int mode, double ratio,
                                           double guard, qvz_codebook_t *out, void *stream) {
    return generate_codebook(d_cnt, true, C, h_dist, mode, ratio, guard, out, stream);
}

static int generate_codebook(const void *d_cnt, bool wide, uint32_t C, const double *h_dist, int mode, double ratio,
                             double guard, qvz_codebook_t *out, void *stream) {
    if (!d_cnt || !h_dist || !out || out->C != C || C == 0) return qvz_set_error(QVZ_E_ARG, "bad codebook arguments");
    if (mode != QVZ_MODE_RATIO && mode != QVZ_MODE_FIXED) return qvz_set_error(QVZ_E_ARG, "unknown mode %d", mode);
    if (guard <= 0.0) guard = 1e-9;
    g_deferred = 0;
    memset(out->ctx_idx, QVZ_NOT_FOUND, (size_t)C * A);
    CbWork *W = cbw_acquire();
    g_cb_active.fetch_add(1);
    cb_wait[0] = cb_wait[1] = cb_wait[2] = cb_wait[3] = cb_wait[4] = cb_wait[5] = 0.0;
    for (auto &v : cb_stat) v = 0;
    const double t_gen = cb_now();
    const int rc = generate(d_cnt, wide, C, h_dist, mode, ratio, guard, out, (hipStream_t)stream, *W);
    (void)hipStreamSynchronize((hipStream_t)stream);
    g_cb_active.fetch_sub(1);
    cbw_give_back(W);
    if (getenv("QVZ_CB_DEBUG"))                                  // development knob
        fprintf(stderr, "qvz_hip_generate_codebook: %.1f ms, of which waiting for the context propagation %.1f ms, for the design windows %.1f ms; host: targets %.1f ms, ratios and storing %.1f ms, hand-over seen to next launch call %.2f ms, launch calls %.2f ms\n",
                (cb_now() - t_gen) * 1e3, cb_wait[0] * 1e3, cb_wait[1] * 1e3, cb_wait[2] * 1e3, cb_wait[3] * 1e3, cb_wait[4] * 1e3, cb_wait[5] * 1e3);
#ifdef QVZ_PROFILE
    {
        unsigned long long z[8] = {0};
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_lm_prof), sizeof(z));
        const double it = z[6] ? (double)z[6] : 1.0;
        fprintf(stderr, "  lm_design cycles per iteration (cumulative over this process): stop check %.0f, mono check + region + bounds %.0f, candidate sums %.0f, arg-min + recon %.0f, walk %.0f; per design after the loop %.0f; %.0f designs, %.1f iterations each\n",
                z[0] / it, z[1] / it, z[2] / it, z[3] / it, z[4] / it, z[7] ? (double)z[5] / z[7] : 0.0, (double)z[7], z[7] ? it / z[7] : 0.0);
    }
#endif
    if (getenv("QVZ_CB_DEBUG") && C > 1)
        fprintf(stderr, "  first windows: %llu blocks, %llu designs completed, %.1f iterations on average; per column the longest design: %.1f iterations at most, %.0f shader cycles (S = %.1f on average)\n",
                cb_stat[4], cb_stat[0], cb_stat[0] ? (double)cb_stat[1] / cb_stat[0] : 0.0, (double)cb_stat[2] / (C - 1), (double)cb_stat[3] / (C - 1), (double)cb_stat[5] / (C - 1));
    if (getenv("QVZ_CB_DEBUG")) fprintf(stderr, "  windows after a column's first: %llu over %u columns; walks known in advance (row entropy below the target): %llu contexts\n", cb_stat[6], C, cb_stat[7]);
    return rc;
}

// Marginal pmfs of one cluster (second half of calculate_statistics, codebook.c:208-219) copied to the host:
// h_marg[s*72 + x]. Used by the compat layer to fill cond_pmf_list_t.marginal_pmfs; generate_codebook computes
// the same values internally.
extern "C" int qvz_hip_marginals(const uint32_t *d_cnt, uint32_t C, double *h_marg, void *stream) {
    if (!d_cnt || !h_marg || C == 0) return qvz_set_error(QVZ_E_ARG, "bad marginals arguments");
    hipStream_t st = (hipStream_t)stream;
    double *P = nullptr, *marg = nullptr;
    int rc = QVZ_OK;
    if (hipMalloc(&P, (size_t)C * A * A * sizeof(double)) != hipSuccess || hipMalloc(&marg, (size_t)C * A * sizeof(double)) != hipSuccess) {
        rc = qvz_set_error(QVZ_E_NOMEM, "device alloc");
    } else {
        hipLaunchKernelGGL(k_cond_pmf<uint32_t>, dim3(C * A), dim3(64), 0, st, d_cnt, P);
        hipLaunchKernelGGL(k_marginals, dim3(1), dim3(256), 0, st, P, C, marg);
        if (hipMemcpyAsync(h_marg, marg, (size_t)C * A * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            rc = qvz_set_error(QVZ_E_HIP, "marginals copy failed");
    }
    (void)hipFree(P);
    (void)hipFree(marg);
    return rc;
}
