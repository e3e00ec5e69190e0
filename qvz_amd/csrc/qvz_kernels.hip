// qvz_kernels.hip - gfx950 (MI355X / CDNA4) kernels for the N-dependent phases of qvz's encode path
// and their C-ABI (include/qvz_hip.h). Written for wave64, 256 CUs / 8 XCDs, 160 KB LDS per CU.
// The budget is HBM bytes, LDS atomics and latency; the one contraction on the path (the per-cluster column sums of a
// k-means iteration: one-hot labels times the symbol tile) is the one MFMA use (k_kmeans_step).
//
// Data layout: one column-major, offset-removed symbol matrix d_sym[s*ld + i] (see qvz_hip.h). Every
// kernel streams columns with 4- or 16-byte per-lane accesses, so each wave instruction moves 256 B-1 KiB.
//
// Reference behaviour restated by each kernel is cited as file:line of mikelhernaez/qvz.
#include <hip/hip_runtime.h>

#include <mutex>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_hip.h"
#include "qvz_internal.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" const char *qvz_hip_last_error(void) { return g_err; }

int qvz_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int qvz_hip_set_error(int code, const char *msg) { return qvz_set_error(code, "%s", msg ? msg : ""); }

extern "C" uint64_t qvz_hip_ld(uint64_t n) { return (n + 255) & ~(uint64_t)255; }

extern "C" int qvz_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline int check_matrix(uint64_t n, uint32_t c, uint64_t ld) {
    if (n == 0 || c == 0) return qvz_set_error(QVZ_E_ARG, "empty matrix (n=%llu c=%u)", (unsigned long long)n, c);
    if (ld < n || (ld & 255)) return qvz_set_error(QVZ_E_ARG, "ld=%llu must be a multiple of 256 and >= n=%llu",
                                                    (unsigned long long)ld, (unsigned long long)n);
    if (c > 1022) return qvz_set_error(QVZ_E_ARG, "c=%u exceeds MAX_READS_PER_LINE (reference lines.h:13)", c);
    return QVZ_OK;
}

// ------------------------------------------------------------------------------------------------
// QGEN-1: counter-based synthetic reads straight into the column-major layout.
// One thread produces 4 consecutive reads so that every column store is a full dword per lane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t sm64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ int clamp_q(int v) { return v < 2 ? 2 : (v > 41 ? 41 : v); }

__global__ __launch_bounds__(256) void k_qgen1(uint64_t seed, uint64_t first, uint64_t n4, uint32_t C,
                                               uint8_t *__restrict__ D, uint64_t ld) {
    uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    uint64_t r[4];
    int base[4], slope[4], q[4], tgt_acc[4], tgt_dec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = sm64(seed * 0xD1342543DE82EF95ull + first + 4 * t + k);
        unsigned cls = (unsigned)(r[k] % 10);
        base[k] = cls < 6 ? 38 : (cls < 9 ? 30 : 20);
        slope[k] = (int)((r[k] >> 8) % 13);
        q[k] = base[k];
        tgt_acc[k] = 0;   // slope*j mod C, kept incrementally
        tgt_dec[k] = 0;   // floor(slope*j / C)
    }
    for (uint32_t j = 0; j < C; ++j) {
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t h = sm64(r[k] + j + 1);
            int target = base[k] - tgt_dec[k];
            unsigned u = (unsigned)(h & 15);
            int step = u < 6 ? 0 : (u < 10 ? 1 : (u < 14 ? -1 : (u == 14 ? -2 : 2)));
            int pull = 0;
            if (((h >> 4) & 3) == 0) pull = (target > q[k]) - (target < q[k]);
            q[k] = clamp_q(q[k] + step + pull);
            int drop = 0;
            if (((h >> 6) & 63) == 0) drop = -(int)((h >> 12) % 20);
            packed |= (uint32_t)clamp_q(q[k] + drop) << (8 * k);
            tgt_acc[k] += slope[k];
            while (tgt_acc[k] >= (int)C) { tgt_acc[k] -= (int)C; tgt_dec[k] += 1; }
        }
        *reinterpret_cast<uint32_t *>(D + (uint64_t)j * ld + 4 * t) = packed;
    }
}

extern "C" int qvz_hip_qgen1(uint64_t seed, uint64_t first, uint64_t n, uint32_t c, uint8_t *d_sym, uint64_t ld,
                             void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    uint64_t n4 = (n + 3) / 4;  // the last thread may write into the ld padding: allowed by contract
    dim3 grid((unsigned)((n4 + 255) / 256));
    hipLaunchKernelGGL(k_qgen1, grid, dim3(256), 0, (hipStream_t)stream, seed, first, n4, c, d_sym, ld);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// ------------------------------------------------------------------------------------------------
// text rows <-> column-major symbols. Tile = 256 rows x 64 columns staged through LDS so that both the
// row-major side (64 contiguous bytes per row) and the column-major side (dword per lane) are coalesced.
// ------------------------------------------------------------------------------------------------
#define TT_ROWS 256
#define TT_COLS 64
#define TT_PITCH (TT_ROWS + 4)  // +4: lanes walking columns hit distinct banks (260/4 = 65 = 1 mod 32)

__global__ __launch_bounds__(256) void k_transpose_text(const uint8_t *__restrict__ text, uint64_t n, uint32_t C,
                                                        uint8_t *__restrict__ D, uint64_t ld,
                                                        uint32_t *__restrict__ bad) {
    __shared__ uint8_t tile[TT_COLS * TT_PITCH];
    const uint64_t row0 = (uint64_t)blockIdx.x * TT_ROWS;
    const uint32_t col0 = blockIdx.y * TT_COLS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t stride = (uint64_t)C + 1;
    uint32_t nbad = 0;
    const uint32_t col = col0 + lane;
    for (uint32_t rr = wave; rr < TT_ROWS; rr += 4) {
        uint64_t row = row0 + rr;
        uint8_t v = 0;
        if (row < n && col < C) {
            uint8_t b = text[row * stride + col];
            if (b < 33 || b >= 33 + QVZ_A) { nbad++; b = 33; }
            v = (uint8_t)(b - 33);
        }
        tile[lane * TT_PITCH + rr] = v;
        if (row < n && col0 == 0 && lane == 0 && text[row * stride + C] != '\n') nbad++;
    }
    __syncthreads();
    // 64 columns x 64 dwords of rows: thread -> (column = i / 64, dword = i % 64)
    for (uint32_t i = threadIdx.x; i < TT_COLS * (TT_ROWS / 4); i += 256) {
        uint32_t cc = i >> 6, dw = i & 63;
        if (col0 + cc < C && row0 + 4 * dw < ld)
            *reinterpret_cast<uint32_t *>(D + (uint64_t)(col0 + cc) * ld + row0 + 4 * dw) =
                *reinterpret_cast<const uint32_t *>(&tile[cc * TT_PITCH + 4 * dw]);
    }
    if (nbad) atomicAdd(bad, nbad);
}

// rows [first, first + n) of the matrix Q (ld reads per column) as text rows 0 .. n-1 (first must be a multiple of 4)
__global__ __launch_bounds__(256) void k_untranspose_text(const uint8_t *__restrict__ Q, uint64_t first, uint64_t n, uint32_t C,
                                                          uint64_t ld, uint8_t *__restrict__ text) {
    __shared__ uint8_t tile[TT_COLS * TT_PITCH];
    const uint64_t row0 = (uint64_t)blockIdx.x * TT_ROWS;
    const uint32_t col0 = blockIdx.y * TT_COLS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t stride = (uint64_t)C + 1;
    for (uint32_t i = threadIdx.x; i < TT_COLS * (TT_ROWS / 4); i += 256) {
        uint32_t cc = i >> 6, dw = i & 63;
        uint32_t v = 0;
        if (col0 + cc < C && first + row0 + 4 * dw < ld)
            v = *reinterpret_cast<const uint32_t *>(Q + (uint64_t)(col0 + cc) * ld + first + row0 + 4 * dw);
        *reinterpret_cast<uint32_t *>(&tile[cc * TT_PITCH + 4 * dw]) = v;
    }
    __syncthreads();
    const uint32_t col = col0 + lane;
    for (uint32_t rr = wave; rr < TT_ROWS; rr += 4) {
        uint64_t row = row0 + rr;
        if (row < n) {
            if (col < C) text[row * stride + col] = (uint8_t)(tile[lane * TT_PITCH + rr] + 33);
            if (col == C) text[row * stride + C] = '\n';   // lane just past the last column writes the newline
        }
    }
}

extern "C" int qvz_hip_transpose_text(const uint8_t *d_text, uint64_t n, uint32_t c, uint8_t *d_sym, uint64_t ld,
                                      uint32_t *d_bad, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    dim3 grid((unsigned)((n + TT_ROWS - 1) / TT_ROWS), (c + TT_COLS - 1) / TT_COLS);
    hipLaunchKernelGGL(k_transpose_text, grid, dim3(256), 0, (hipStream_t)stream, d_text, n, c, d_sym, ld, d_bad);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

extern "C" int qvz_hip_untranspose_rows(const uint8_t *d_qv, uint64_t first, uint64_t n, uint32_t c, uint64_t ld,
                                        uint8_t *d_text, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    if ((first & 3) || first + n > ld) return qvz_set_error(QVZ_E_ARG, "rows [%llu, +%llu) of a matrix with ld=%llu (the first row must be a multiple of 4)",
                                                            (unsigned long long)first, (unsigned long long)n, (unsigned long long)ld);
    // one extra column slot so that some lane owns column index C (the '\n')
    dim3 grid((unsigned)((n + TT_ROWS - 1) / TT_ROWS), (c + 1 + TT_COLS - 1) / TT_COLS);
    hipLaunchKernelGGL(k_untranspose_text, grid, dim3(256), 0, (hipStream_t)stream, d_qv, first, n, c, ld, d_text);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

extern "C" int qvz_hip_untranspose_text(const uint8_t *d_qv, uint64_t n, uint32_t c, uint64_t ld, uint8_t *d_text,
                                        void *stream) {
    return qvz_hip_untranspose_rows(d_qv, 0, n, c, ld, d_text, stream);
}

// ------------------------------------------------------------------------------------------------
// k-means (reference src/cluster.c). Integer-exact: distances fit u32 (c <= 1022, |diff| <= 71).
// ------------------------------------------------------------------------------------------------
// Assignment: one thread owns 4 consecutive reads (dword loads down the columns), K clusters per pass
// with the centroids of the pass in LDS. Passes keep (best distance, best label) so that the lowest
// cluster id wins ties exactly like the strict '<' scan of assign_cluster (cluster.c:155-163).
#define KM_KPASS 8

template <int K>
__global__ __launch_bounds__(256) void k_kmeans_assign(const uint8_t *__restrict__ D, uint64_t n, uint32_t C,
                                                       uint64_t ld, uint32_t k, const uint8_t *__restrict__ means,
                                                       uint8_t *__restrict__ labels, uint32_t *__restrict__ counts) {
    extern __shared__ uint8_t lds_means[];  // [min(k,..)][C] for the current pass
    __shared__ uint32_t blk_counts[256];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t i0 = 4 * t;
    const bool active = i0 < n;
    uint32_t best[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t lab[4] = {0, 0, 0, 0};
    for (uint32_t kb = 0; kb < k; kb += K) {
        const uint32_t kk = (k - kb) < (uint32_t)K ? (k - kb) : (uint32_t)K;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < kk * C; i += 256) lds_means[i] = means[(uint64_t)kb * C + i];
        __syncthreads();
        if (active) {
            uint32_t d[4][K];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < K; ++j) d[r][j] = 0;
            for (uint32_t s = 0; s < C; ++s) {
                const uint32_t x4 = *reinterpret_cast<const uint32_t *>(D + (uint64_t)s * ld + i0);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if ((uint32_t)j < kk) {
                        const int m = lds_means[j * C + s];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int diff = (int)((x4 >> (8 * r)) & 0xFF) - m;
                            d[r][j] += (uint32_t)(diff * diff);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < K; ++j)
                if ((uint32_t)j < kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d[r][j] < best[r]) { best[r] = d[r][j]; lab[r] = kb + j; }
        }
    }
    // labels + per-cluster counts (cluster->count, cluster.c:168)
    if (active) {
        const uint32_t packed = lab[0] | (lab[1] << 8) | (lab[2] << 16) | (lab[3] << 24);
        *reinterpret_cast<uint32_t *>(labels + i0) = packed;   // labels buffer is ld long: padding is ours
    }
    for (uint32_t i = threadIdx.x; i < k; i += 256) blk_counts[i] = 0;
    __syncthreads();
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i0 + r < n) atomicAdd(&blk_counts[lab[r]], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < k; i += 256)
        if (blk_counts[i]) atomicAdd(&counts[i], blk_counts[i]);
}

// The same assignment for 2 .. 4 clusters with 16 reads per thread. The distance of read x to centroid m is
// sum (x_s - m_s)^2 = sum x_s^2 - 2 sum x_s m_s + sum m_s^2, and sum x_s^2 is the same for every centroid: the order of the
// distances, ties included, is the order of c_j - 2 dot_j with c_j = sum m_js^2 and dot_j = sum x_s m_js (exact integers
// either way). So a symbol costs one multiply-add per centroid instead of a subtraction and a multiply-add, a thread asks
// for 16 bytes per column instead of 4 (the 4-read kernel ran at a third of the HBM rate: too few bytes in flight), and
// the centroid bytes of a column are the same in every lane: scalar operands.
template <int K>
__global__ __launch_bounds__(256) void k_kmeans_assign16(const uint8_t *__restrict__ D, uint64_t n, uint32_t C, uint64_t ld,
                                                         const uint8_t *__restrict__ means, uint8_t *__restrict__ labels,
                                                         uint32_t *__restrict__ counts) {
    extern __shared__ uint32_t lds_m[];                          // [C] the K centroid bytes of a column packed into a dword
    __shared__ uint32_t blk_counts[K];
    __shared__ int32_t csq[K];                                   // sum of m_js^2
    if (threadIdx.x < K) { blk_counts[threadIdx.x] = 0; csq[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < C; s += 256) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t m = means[(uint64_t)j * C + s];
            w |= m << (8 * j);
            atomicAdd(&csq[j], (int32_t)(m * m));
        }
        lds_m[s] = w;
    }
    __syncthreads();
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 < n) {                                                // (ld is a multiple of 256: the 16-byte loads stay inside the column)
        uint32_t dot[16][K];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < K; ++j) dot[r][j] = 0;
        const uint8_t *p = D + i0;
        uint4 cur = *reinterpret_cast<const uint4 *>(p);
        for (uint32_t s = 0; s < C; ++s) {
            const uint4 nxt = s + 1 < C ? *reinterpret_cast<const uint4 *>(p + (uint64_t)(s + 1) * ld) : cur;
            const uint32_t mw = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_m[s]);
            const uint32_t xw[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t m = (mw >> (8 * j)) & 0xFFu;
#pragma unroll
                for (int r = 0; r < 16; ++r) dot[r][j] = __umul24((xw[r >> 2] >> (8 * (r & 3))) & 0xFFu, m) + dot[r][j];
            }
            cur = nxt;
        }
        uint32_t lw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int32_t best = csq[0] - 2 * (int32_t)dot[r][0];
            uint32_t lab = 0;
#pragma unroll
            for (int j = 1; j < K; ++j) {
                const int32_t d = csq[j] - 2 * (int32_t)dot[r][j];
                if (d < best) { best = d; lab = j; }             // strict: the lowest cluster id wins ties (cluster.c:155-163)
            }
            lw[r >> 2] |= lab << (8 * (r & 3));
            if (i0 + r < n) atomicAdd(&blk_counts[lab], 1u);
        }
        *reinterpret_cast<uint4 *>(labels + i0) = make_uint4(lw[0], lw[1], lw[2], lw[3]);   // the label vector is ld long
    }
    __syncthreads();
    if (threadIdx.x < K && blk_counts[threadIdx.x]) atomicAdd(&counts[threadIdx.x], blk_counts[threadIdx.x]);
}

extern "C" int qvz_hip_kmeans_assign(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint32_t k,
                                     const uint8_t *d_means, uint8_t *d_labels, uint32_t *d_counts, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    if (k == 0 || k > 255) return qvz_set_error(QVZ_E_ARG, "k=%u out of range 1..255", k);
    uint64_t n4 = (n + 3) / 4;
    dim3 grid((unsigned)((n4 + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (k >= 2 && k <= 4) {
        dim3 g16((unsigned)((n + 4095) / 4096));
        const size_t lds = (size_t)c * sizeof(uint32_t);
        if (k == 2) hipLaunchKernelGGL(k_kmeans_assign16<2>, g16, dim3(256), lds, st, d_sym, n, c, ld, d_means, d_labels, d_counts);
        else if (k == 3) hipLaunchKernelGGL(k_kmeans_assign16<3>, g16, dim3(256), lds, st, d_sym, n, c, ld, d_means, d_labels, d_counts);
        else hipLaunchKernelGGL(k_kmeans_assign16<4>, g16, dim3(256), lds, st, d_sym, n, c, ld, d_means, d_labels, d_counts);
    } else if (k == 1) {
        size_t lds = (size_t)c;
        hipLaunchKernelGGL(k_kmeans_assign<1>, grid, dim3(256), lds, st, d_sym, n, c, ld, k, d_means, d_labels, d_counts);
    } else if (k <= 4) {
        size_t lds = (size_t)4 * c;
        hipLaunchKernelGGL(k_kmeans_assign<4>, grid, dim3(256), lds, st, d_sym, n, c, ld, k, d_means, d_labels, d_counts);
    } else {
        size_t lds = (size_t)KM_KPASS * c;
        hipLaunchKernelGGL(k_kmeans_assign<KM_KPASS>, grid, dim3(256), lds, st, d_sym, n, c, ld, k, d_means, d_labels,
                           d_counts);
    }
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// Accumulation: block = (column, row slab). Threads stream 16 rows per load (16 B of symbols + 16 B of
// labels), keep K partial sums in registers, then wave-reduce and add into the u64 accumulators.
#define KMA_SLABS 8

template <int K, bool HAVE_LABELS>
__global__ __launch_bounds__(256) void k_kmeans_accum(const uint8_t *__restrict__ D,
                                                      const uint8_t *__restrict__ labels, uint64_t n, uint32_t C,
                                                      uint64_t ld, uint32_t kb, uint32_t k,
                                                      unsigned long long *__restrict__ acc) {
    const uint32_t s = blockIdx.x;
    const uint64_t chunks = (n + 4095) / 4096;          // 4096 rows per block iteration
    const uint64_t per = (chunks + gridDim.y - 1) / gridDim.y;
    const uint64_t c0 = (uint64_t)blockIdx.y * per, c1 = (c0 + per < chunks) ? c0 + per : chunks;
    uint32_t sum[K];
#pragma unroll
    for (int j = 0; j < K; ++j) sum[j] = 0;
    unsigned long long total[K];
#pragma unroll
    for (int j = 0; j < K; ++j) total[j] = 0;
    const uint8_t *col = D + (uint64_t)s * ld;
    uint32_t iters = 0;
    for (uint64_t ch = c0; ch < c1; ++ch) {
        const uint64_t i = ch * 4096 + (uint64_t)threadIdx.x * 16;
        if (i < n) {
            const uint4 xv = *reinterpret_cast<const uint4 *>(col + i);
            uint4 lv = make_uint4(0, 0, 0, 0);
            if (HAVE_LABELS) lv = *reinterpret_cast<const uint4 *>(labels + i);
            const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w};
            const uint32_t ls[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (i + 4 * w + b < n) {
                        const uint32_t x = (xs[w] >> (8 * b)) & 0xFF;
                        const uint32_t l = (ls[w] >> (8 * b)) & 0xFF;
#pragma unroll
                        for (int j = 0; j < K; ++j) sum[j] += (!HAVE_LABELS || l == kb + j) ? x : 0u;
                    }
                }
        }
        if (++iters == 1u << 16) {   // 2^16 iterations x 16 rows x 71 < 2^32: spill to 64 bit before overflow
#pragma unroll
            for (int j = 0; j < K; ++j) { total[j] += sum[j]; sum[j] = 0; }
            iters = 0;
        }
    }
    __shared__ unsigned long long red[4][K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        unsigned long long v = total[j] + sum[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < K && kb + threadIdx.x < k) {
        unsigned long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v) atomicAdd(&acc[(uint64_t)(kb + threadIdx.x) * C + s], v);
    }
}

// The same sums for 2 .. 4 clusters with the reads' labels loaded once: a thread owns 16 consecutive reads, turns their
// labels into one byte mask per cluster and dword (0xFF where the read belongs to the cluster) and walks the columns; a
// column's 16 symbols are one 16-byte load, a cluster's share of a dword is one AND and one byte-wise dot product with
// 0x01010101. The sums of a wave go through a shuffle tree into the block's LDS accumulators, those to memory once.
// (The column-per-block kernel above reads the label vector again for every column - 150 times - out of L2.)
template <int K>
__global__ __launch_bounds__(256) void k_kmeans_accum16(const uint8_t *__restrict__ D, const uint8_t *__restrict__ labels,
                                                        uint64_t n, uint32_t C, uint64_t ld, unsigned long long *__restrict__ acc) {
    extern __shared__ uint32_t blk_acc[];                        // [K][C]; a block adds at most 4096 x 71 per entry
    for (uint32_t i = threadIdx.x; i < K * C; i += 256) blk_acc[i] = 0;
    __syncthreads();
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t mask[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) mask[j][w] = 0;
    const bool live = i0 < n;
    if (live) {
        const uint4 lv = *reinterpret_cast<const uint4 *>(labels + i0);
        const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t l = (lw[r >> 2] >> (8 * (r & 3))) & 0xFFu;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (l == (uint32_t)j && i0 + r < n) mask[j][r >> 2] |= 0xFFu << (8 * (r & 3));
        }
    }
    const uint8_t *p = D + (live ? i0 : 0);
    uint4 cur = *reinterpret_cast<const uint4 *>(p);
    for (uint32_t s = 0; s < C; ++s) {
        const uint4 nxt = s + 1 < C ? *reinterpret_cast<const uint4 *>(p + (uint64_t)(s + 1) * ld) : cur;
        const uint32_t xw[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t sum[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) t = __builtin_amdgcn_udot4(xw[w] & mask[j][w], 0x01010101u, t, false);
            sum[j] = t;                                          // (idle threads have empty masks)
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            uint32_t v = sum[j];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&blk_acc[j * C + s], v);
        }
        cur = nxt;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < K * C; i += 256)
        if (blk_acc[i]) atomicAdd(&acc[i], (unsigned long long)blk_acc[i]);
}

extern "C" int qvz_hip_kmeans_accumulate(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c,
                                         uint64_t ld, uint32_t k, uint64_t *d_acc, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    if (k == 0 || k > 255) return qvz_set_error(QVZ_E_ARG, "k=%u out of range 1..255", k);
    if (k > 1 && !d_labels) return qvz_set_error(QVZ_E_ARG, "labels required for k > 1");
    uint64_t chunks = (n + 4095) / 4096;
    unsigned slabs = (unsigned)(chunks < 2048 / c + 1 ? chunks : 2048 / c + 1);  // ~2048 blocks fill 256 CUs x 8
    dim3 grid(c, slabs);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(d_acc);
    if (k >= 2 && k <= 4) {
        dim3 g16((unsigned)((n + 4095) / 4096));
        const size_t lds = (size_t)k * c * sizeof(uint32_t);
        if (k == 2) hipLaunchKernelGGL(k_kmeans_accum16<2>, g16, dim3(256), lds, st, d_sym, d_labels, n, c, ld, acc);
        else if (k == 3) hipLaunchKernelGGL(k_kmeans_accum16<3>, g16, dim3(256), lds, st, d_sym, d_labels, n, c, ld, acc);
        else hipLaunchKernelGGL(k_kmeans_accum16<4>, g16, dim3(256), lds, st, d_sym, d_labels, n, c, ld, acc);
    } else if (k == 1) {
        hipLaunchKernelGGL((k_kmeans_accum<1, false>), grid, dim3(256), 0, st, d_sym, d_labels, n, c, ld, 0u, k, acc);
    } else {
        for (uint32_t kb = 0; kb < k; kb += 4)
            hipLaunchKernelGGL((k_kmeans_accum<4, true>), grid, dim3(256), 0, st, d_sym, d_labels, n, c, ld, kb, k, acc);
    }
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// One k-means iteration in ONE pass over the symbols (2 .. 8 clusters, up to 160 columns): assignment, counts and the
// per-cluster column sums. The two kernels above read the matrix twice per iteration, and each runs at the HBM rate
// already (2.8 + 3.3 ms for 15 GB on MI355X), so the only thing left to save is the second read: a workgroup keeps a tile of
// 256 reads x C columns in LDS (C x 272 bytes), and everything an iteration needs is computed from that copy.
//   distances   sum_s x_s m_js for the K centroids, the c_j - 2 dot_j algebra of k_kmeans_assign16. A thread takes four reads
//               and a quarter of the columns: four dwords (four reads of four adjacent columns) are transposed in registers
//               (8 v_perm_b32) into one dword per read holding its four columns, and v_dot4_u32_u8 against the four centroid
//               bytes does four multiply-adds per instruction; the partial sums of the four column quarters meet in LDS.
//   sums        acc[j][s] = sum_r [label_r = j] x_rs is a (clusters x reads) . (reads x columns) product of small integers:
//               the one-hot label bytes are the A operand, the tile as it lies in LDS (a column's reads are contiguous) is the
//               B operand of v_mfma_i32_16x16x64_i8; a wave owns 64 of the tile's reads and keeps its 16 x 16 accumulators in
//               registers over all the tiles of the workgroup (exact: int32, flushed long before they could wrap).
// The grid is persistent (three workgroups per CU), the next tile's loads are in flight while the current one is worked on,
// and a workgroup adds to the global counters once. Symbols must be below 128 (the i8 operand is signed; the encode path
// has rejected anything above 71 before it clusters). Exact integers throughout: same labels, counts and sums as the
// two-pass kernels, hence as cluster.c:75-98 and :104-128's inputs.
#define KMS_R 256
#define KMS_PITCH 272          // 68 dwords = 4 x odd: the 16-byte operand reads of 16 columns fall on 16 different bank quads
#define KMS_NT 10              // column tiles of 16: C <= 160
typedef int kms_v4i __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256, (K <= 4 ? 3 : 2)) void k_kmeans_step(const uint8_t *__restrict__ D, uint64_t n, uint32_t C, uint64_t ld,
                                                     const uint8_t *__restrict__ means, uint8_t *__restrict__ labels,
                                                     uint32_t *__restrict__ counts, unsigned long long *__restrict__ acc,
                                                     uint64_t ntiles) {
    extern __shared__ uint4 kms_smem[];
    const uint32_t CQ = (C + 3) / 4;
    constexpr int KP = K <= 4 ? 4 : 8;                           // centroid dwords kept per column quad (16-byte reads)
    uint8_t *tile = reinterpret_cast<uint8_t *>(kms_smem);                       // [C][KMS_PITCH]
    uint32_t *dots = reinterpret_cast<uint32_t *>(tile + (size_t)C * KMS_PITCH); // [K][4][64]: read 4q + i of the tile at [i][q]
    uint32_t *m4 = dots + K * KMS_R;                                             // [CQ][KP]: centroid j's bytes of columns 4cq .. 4cq+3
    uint32_t *blk = m4 + CQ * KP;                                                 // [K][KMS_NT * 16] sums of this workgroup
    int32_t *csq = reinterpret_cast<int32_t *>(blk + K * KMS_NT * 16);           // [8] sum of m_js^2
    uint32_t *cnt = reinterpret_cast<uint32_t *>(csq + 8);                       // [8]
    uint8_t *oh = reinterpret_cast<uint8_t *>(cnt + 8);                          // [K][KMS_R] one-hot labels (invalid reads: none)
    uint8_t *labsm = oh + K * KMS_R;                                             // [KMS_R]
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));

    if (tid < 8) { csq[tid] = 0; cnt[tid] = 0; }
    for (uint32_t i = tid; i < K * KMS_NT * 16; i += 256) blk[i] = 0;
    __syncthreads();
    for (uint32_t cq = tid; cq < CQ; cq += 256) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            uint32_t w = 0;
            if (j < K)
                for (uint32_t i = 0; i < 4; ++i) {
                    const uint32_t s = 4 * cq + i, m = s < C ? means[(uint64_t)j * C + s] : 0u;
                    w |= m << (8 * i);
                    if (m) atomicAdd(&csq[j], (int32_t)(m * m));
                }
            m4[cq * KP + j] = w;
        }
    }

    // loads of a tile: thread t takes bytes 16 (t & 15) .. +15 of columns (t >> 4) + 16 e
    const uint32_t v16 = (tid & 15) * 16, s0 = tid >> 4;
    uint4 pf[KMS_NT];
    auto load_tile = [&](uint64_t t) {
        const uint64_t r0 = t * KMS_R + v16;
#pragma unroll
        for (int e = 0; e < KMS_NT; ++e) {
            const uint32_t s = s0 + 16 * e;
            pf[e] = make_uint4(0u, 0u, 0u, 0u);
            if (s < C && r0 < ld) pf[e] = *reinterpret_cast<const uint4 *>(D + (uint64_t)s * ld + r0);
        }
    };
    kms_v4i accv[KMS_NT];
#pragma unroll
    for (int nt = 0; nt < KMS_NT; ++nt) accv[nt] = kms_v4i{0, 0, 0, 0};
    auto flush = [&]() {                                         // the workgroup's sums and counts -> global (all threads)
        if (lane < (K <= 4 ? 16u : 32u)) {                        // D rows 4 (lane >> 4) + reg = clusters, D columns lane & 15
#pragma unroll
            for (int nt = 0; nt < KMS_NT; ++nt) {
                const uint32_t s = 16 * nt + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t j = 4 * (lane >> 4) + r;
                    if (j < K && s < C && accv[nt][r]) atomicAdd(&blk[j * KMS_NT * 16 + s], (uint32_t)accv[nt][r]);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < KMS_NT; ++nt) accv[nt] = kms_v4i{0, 0, 0, 0};
        __syncthreads();
        for (uint32_t i = tid; i < K * KMS_NT * 16; i += 256) {
            const uint32_t j = i / (KMS_NT * 16), s = i % (KMS_NT * 16), v = blk[i];
            if (v) atomicAdd(&acc[(uint64_t)j * C + s], (unsigned long long)v);
            blk[i] = 0;
        }
        if (tid < K && cnt[tid]) { atomicAdd(&counts[tid], cnt[tid]); cnt[tid] = 0; }
        __syncthreads();
    };

    uint64_t t = blockIdx.x;
    if (t < ntiles) load_tile(t);
    uint32_t done = 0;
    for (; t < ntiles; t += gridDim.x) {
        __syncthreads();                                         // the previous tile is finished with (tile, dots, one-hot bytes)
#pragma unroll
        for (int e = 0; e < KMS_NT; ++e) {
            const uint32_t s = s0 + 16 * e;
            if (s < C) *reinterpret_cast<uint4 *>(tile + (size_t)s * KMS_PITCH + v16) = pf[e];
        }
        for (uint32_t i = tid; i < K * KMS_R; i += 256) dots[i] = 0;
        if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);    // in flight while this tile is worked on
        __syncthreads();
        // ---- distances: wave = a quarter of the column quads, lane = reads 4 lane .. 4 lane + 3 ----
        {
            const uint32_t per = (CQ + 3) / 4, c0 = wave * per, c1 = c0 + per < CQ ? c0 + per : CQ;
            uint32_t dot[4][K];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < K; ++j) dot[i][j] = 0;
            for (uint32_t cq = c0; cq < c1; ++cq) {
                uint32_t w[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                    const uint32_t s = 4 * cq + i < C ? 4 * cq + i : C - 1;   // a column past the last: its centroid bytes are 0
                    w[i] = *reinterpret_cast<const uint32_t *>(tile + (size_t)s * KMS_PITCH + 4 * lane);
                }
                // 4 x 4 byte transpose: r[i] = read i's bytes of the four columns
                const uint32_t lo01 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), hi01 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
                const uint32_t lo23 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), hi23 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
                const uint32_t r[4] = {__builtin_amdgcn_perm(lo23, lo01, 0x05040100u), __builtin_amdgcn_perm(lo23, lo01, 0x07060302u),
                                       __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), __builtin_amdgcn_perm(hi23, hi01, 0x07060302u)};
                const uint4 mm = *reinterpret_cast<const uint4 *>(m4 + cq * KP);  // the same address in every lane
                uint32_t mj[KP] = {mm.x, mm.y, mm.z, mm.w};
                if (KP > 4) {
                    const uint4 m2 = *reinterpret_cast<const uint4 *>(m4 + cq * KP + 4);
                    mj[KP - 4] = m2.x; mj[KP - 3] = m2.y; mj[KP - 2] = m2.z; mj[KP - 1] = m2.w;
                }
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot[i][j] = __builtin_amdgcn_udot4(r[i], mj[j], dot[i][j], false);
            }
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(&dots[j * KMS_R + i * 64 + lane], dot[i][j]);
        }
        __syncthreads();
        // ---- labels: thread = read ----
        {
            const uint64_t rd = t * KMS_R + tid;
            const uint32_t slot = (tid & 3) * 64 + (tid >> 2);
            int32_t best = csq[0] - 2 * (int32_t)dots[slot];
            uint32_t lab = 0;
#pragma unroll
            for (int j = 1; j < K; ++j) {
                const int32_t d = csq[j] - 2 * (int32_t)dots[j * KMS_R + slot];
                if (d < best) { best = d; lab = j; }             // strict: the lowest cluster id wins ties (cluster.c:155-163)
            }
            const bool valid = rd < n;
            labsm[tid] = (uint8_t)lab;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                oh[j * KMS_R + tid] = valid && lab == (uint32_t)j ? 1 : 0;
                const unsigned long long m = __ballot(valid && lab == (uint32_t)j);
                if (lane == 0 && m) atomicAdd(&cnt[j], (uint32_t)__builtin_popcountll(m));
            }
        }
        __syncthreads();
        if (tid < 16 && t * KMS_R + tid * 16 < ld)               // (the label vector is ld long)
            *reinterpret_cast<uint4 *>(labels + t * KMS_R + tid * 16) = *reinterpret_cast<const uint4 *>(labsm + tid * 16);
        // ---- sums: wave = reads 64 wave .. +63 (one k-step of 64), lane & 15 = cluster (A) / column of the tile (B) ----
        {
            const uint32_t i16 = lane & 15, q = lane >> 4, koff = 64 * wave + 16 * q;
            kms_v4i a = kms_v4i{0, 0, 0, 0};
            if (i16 < K) a = *reinterpret_cast<const kms_v4i *>(oh + i16 * KMS_R + koff);
#pragma unroll
            for (int nt = 0; nt < KMS_NT; ++nt) {
                if (16u * nt < C) {
                    const uint32_t s = 16 * nt + i16 < C ? 16 * nt + i16 : C - 1;   // past the last column: a copy nobody reads
                    const kms_v4i b = *reinterpret_cast<const kms_v4i *>(tile + (size_t)s * KMS_PITCH + koff);
                    accv[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, accv[nt], 0, 0, 0);
                }
            }
        }
        if (++done == 32768) { done = 0; __syncthreads(); flush(); }   // 2^15 tiles x 64 reads x 127 < 2^31
    }
    __syncthreads();
    flush();
}

extern "C" int qvz_hip_kmeans_step(const uint8_t *d_sym, uint64_t n, uint32_t c, uint64_t ld, uint32_t k, const uint8_t *d_means,
                                   uint8_t *d_labels, uint32_t *d_counts, uint64_t *d_acc, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    if (k == 0 || k > 255) return qvz_set_error(QVZ_E_ARG, "k=%u out of range 1..255", k);
    if (k < 2 || k > 8 || c > 16 * KMS_NT) {                     // shapes the one-pass kernel does not take: assignment, then sums
        rc = qvz_hip_kmeans_assign(d_sym, n, c, ld, k, d_means, d_labels, d_counts, stream);
        if (rc) return rc;
        return qvz_hip_kmeans_accumulate(d_sym, d_labels, n, c, ld, k, d_acc, stream);
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t ntiles = (n + KMS_R - 1) / KMS_R;
    const uint32_t cq = (c + 3) / 4;
    const size_t lds = (size_t)c * KMS_PITCH + ((size_t)k * KMS_R + cq * (k <= 4 ? 4 : 8) + (size_t)k * KMS_NT * 16 + 16) * 4 + (size_t)k * KMS_R + KMS_R;
    const uint64_t want = (uint64_t)cus * (k <= 4 ? 3 : 2);
    dim3 grid((unsigned)(ntiles < want ? ntiles : want));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(d_acc);
#define KMS_LAUNCH(KK) hipLaunchKernelGGL(k_kmeans_step<KK>, grid, dim3(256), lds, st, d_sym, n, c, ld, d_means, d_labels, d_counts, acc, ntiles)
    switch (k) {
    case 2: KMS_LAUNCH(2); break;
    case 3: KMS_LAUNCH(3); break;
    case 4: KMS_LAUNCH(4); break;
    case 5: KMS_LAUNCH(5); break;
    case 6: KMS_LAUNCH(6); break;
    case 7: KMS_LAUNCH(7); break;
    default: KMS_LAUNCH(8); break;
    }
#undef KMS_LAUNCH
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// New centroids and how far they moved (cluster.c:104-128). One block; exact integers throughout.
template <typename CountT>   // uint32_t as the reference keeps them (lines.h:41), or 64 bits for a file of more than 2^32 - 1 reads
__global__ __launch_bounds__(256) void k_kmeans_update(const unsigned long long *__restrict__ acc,
                                                       const CountT *__restrict__ counts, uint32_t k, uint32_t C,
                                                       uint8_t *__restrict__ means, double *__restrict__ moved_out,
                                                       int32_t *__restrict__ status) {
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long best;
    if (threadIdx.x == 0) best = 0;
    for (uint32_t j = 0; j < k; ++j) {
        const CountT cnt = counts[j];
        unsigned long long part = 0;
        if (cnt == 0) {
            if (threadIdx.x == 0) atomicOr(status, 1);
        } else {
            for (uint32_t s = threadIdx.x; s < C; s += 256) {
                const uint32_t nm = (uint32_t)(acc[(uint64_t)j * C + s] / cnt);
                const int diff = (int)(nm & 0xFF) - (int)means[(uint64_t)j * C + s];   // (uint8_t) cast, cluster.c:113
                part += (unsigned long long)(diff * diff);
                means[(uint64_t)j * C + s] = (uint8_t)nm;
            }
        }
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = red[0] + red[1] + red[2] + red[3];
            if (m > best) best = m;
            moved_out[1 + j] = (double)m;   // per-cluster value, printed by -v (cluster.c:126-127)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *moved_out = (double)best;
}

extern "C" int qvz_hip_kmeans_update(const uint64_t *d_acc, const uint32_t *d_counts, uint32_t k, uint32_t c,
                                     uint8_t *d_means, double *d_moved, int32_t *d_status, void *stream) {
    if (k == 0 || c == 0) return qvz_set_error(QVZ_E_ARG, "empty update");
    hipLaunchKernelGGL(k_kmeans_update<uint32_t>, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long *>(d_acc), d_counts, k, c, d_means, d_moved, d_status);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

extern "C" int qvz_hip_kmeans_update64(const uint64_t *d_acc, const uint64_t *d_counts, uint32_t k, uint32_t c,
                                       uint8_t *d_means, double *d_moved, int32_t *d_status, void *stream) {
    if (k == 0 || c == 0) return qvz_set_error(QVZ_E_ARG, "empty update");
    hipLaunchKernelGGL(k_kmeans_update<unsigned long long>, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long *>(d_acc), reinterpret_cast<const unsigned long long *>(d_counts), k, c,
                       d_means, d_moved, d_status);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// u32 counters of one part of a file -> u64, so that their sum over the parts cannot wrap (a file of 2^32 reads or more)
__global__ __launch_bounds__(256) void k_widen_u32(const uint32_t *__restrict__ src, unsigned long long *__restrict__ dst, uint64_t count) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int qvz_hip_widen_u32(const uint32_t *d_src, uint64_t *d_dst, uint64_t count, void *stream) {
    if (!d_src || !d_dst) return qvz_set_error(QVZ_E_ARG, "null argument");
    if (count == 0) return QVZ_OK;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_widen_u32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_src, reinterpret_cast<unsigned long long *>(d_dst), count);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// The k centroid reads of a file whose matrix is in HBM (cluster.c:198-201 copies the reads' bytes): out[j][s] = symbol s of read
// idx[j], or 0 for a read this part does not own (idx = ~0) - a SUM over the parts then holds every centroid.
__global__ __launch_bounds__(256) void k_gather_reads(const uint8_t *__restrict__ sym, uint64_t ld, uint32_t c, const unsigned long long *__restrict__ idx,
                                                      uint32_t *__restrict__ out) {
    const unsigned long long i = idx[blockIdx.x];
    for (uint32_t s = threadIdx.x; s < c; s += 256) out[(size_t)blockIdx.x * c + s] = i == ~0ull ? 0u : sym[(size_t)s * ld + i];
}

extern "C" int qvz_hip_gather_reads(const uint8_t *d_sym, uint64_t ld, uint32_t c, const uint64_t *d_idx, uint32_t k, uint32_t *d_out, void *stream) {
    if (!d_sym || !d_idx || !d_out || !c || !k) return qvz_set_error(QVZ_E_ARG, "bad gather arguments");
    hipLaunchKernelGGL(k_gather_reads, dim3(k), dim3(256), 0, (hipStream_t)stream, d_sym, ld, c, reinterpret_cast<const unsigned long long *>(d_idx), d_out);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// sum over the reads of err[i] / columns (the per-read term of qv_compressor.c:127) in a FIXED order - 1024 strided partial sums,
// each a sequential sum, then the partials in index order - so that the same input gives the same double on every run; the
// reference's own order (one sequential sum over the file) is what the file drivers keep on the host (qvz_sum_errors).
__global__ __launch_bounds__(256) void k_sum_errors(const double *__restrict__ err, uint64_t n, double cols, double *__restrict__ partial) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc += err[i] / cols;
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(64) void k_sum_partials(const double *__restrict__ partial, uint32_t count, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double acc = 0.0;
        for (uint32_t i = 0; i < count; ++i) acc += partial[i];
        out[0] = acc;
    }
}

extern "C" int qvz_hip_sum_errors(const double *d_err, uint64_t n, uint32_t c, double *d_sum /*[1 + 1024]: the sum, then scratch*/, void *stream) {
    if (!d_err || !d_sum || !c) return qvz_set_error(QVZ_E_ARG, "bad sum arguments");
    hipLaunchKernelGGL(k_sum_errors, dim3(1024), dim3(256), 0, (hipStream_t)stream, d_err, n, (double)c, d_sum + 1);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, (hipStream_t)stream, d_sum + 1, 1024u, d_sum);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// ------------------------------------------------------------------------------------------------
// Conditional-PMF counters (reference src/codebook.c:193-205).
//
// Grid = (column s, row slab). A block owns one column pair (s-1, s) for its slab of reads and keeps the whole
// 72x72 u32 counter tile of up to PMF_KT clusters in LDS (20 736 B per cluster); threads stream 16 reads per
// 16-byte load of each of the two columns and issue one LDS atomic per read. Quality scores are heavily skewed
// (a handful of (prev, cur) pairs carry most of the mass), so lanes of one wave instruction pile up on the same
// counter and the LDS serialises them: the tile is therefore kept in R replicas, lane l using replica l % R;
// consecutive replicas are offset by one dword so the same bin of different replicas sits in different banks.
// The replicas are summed and flushed once with global atomics (non-zero bins only). Columns s and s+1 of the
// same slab run side by side, so the second read of each column is served from L2 / Infinity Cache: HBM traffic
// stays ~1 B per symbol.
// ------------------------------------------------------------------------------------------------
#define PMF_TILE (QVZ_A * QVZ_A)
#define PMF_KT 7   // 7 x 20 736 B = 145 KB of the 160 KB LDS

// AA = side of the LDS tile: symbols < AA are counted in LDS, larger ones (none for a correct hint) go straight to
// the global counters, so the hint only affects speed. Real Illumina data and QGEN-1 stay below 42: a 42x42 tile
// is 7 KB instead of 20.7 KB, which buys three times the replicas at the same occupancy.
template <bool HAVE_LABELS>
__global__ void k_pmf_hist(const uint8_t *__restrict__ D, const uint8_t *__restrict__ labels, uint64_t n, uint32_t C,
                           uint64_t ld, uint32_t kb, uint32_t kt, uint32_t R, uint32_t AA, uint32_t *__restrict__ cnt) {
    extern __shared__ uint32_t tile[];   // [R][kt][AA][AA] (+1 dword skew per replica)
    const uint32_t s = blockIdx.x, T = blockDim.x;
    const uint32_t ts = AA * AA;                                 // one cluster's tile
    const uint32_t rs = kt * ts + 1;                             // replica stride in dwords
    for (uint32_t i = threadIdx.x; i < R * rs; i += T) tile[i] = 0;
    __syncthreads();
    uint32_t *mine = tile + (threadIdx.x % R) * rs;
    uint32_t *out = cnt + ((uint64_t)kb * C + s) * PMF_TILE;
    int miss = 0;

    const uint64_t rows = (uint64_t)T * 16;                      // rows per block iteration
    const uint64_t chunks = (n + rows - 1) / rows;
    const uint64_t per = (chunks + gridDim.y - 1) / gridDim.y;
    const uint64_t c0 = (uint64_t)blockIdx.y * per, c1 = (c0 + per < chunks) ? c0 + per : chunks;
    const uint8_t *cur = D + (uint64_t)s * ld;
    const uint8_t *prv = D + (uint64_t)(s ? s - 1 : 0) * ld;
    for (uint64_t ch = c0; ch < c1; ++ch) {
        const uint64_t i = ch * rows + (uint64_t)threadIdx.x * 16;
        if (i >= n) continue;
        const uint4 cv = *reinterpret_cast<const uint4 *>(cur + i);
        uint4 pv = make_uint4(0, 0, 0, 0);
        if (s) pv = *reinterpret_cast<const uint4 *>(prv + i);
        uint4 lv = make_uint4(0, 0, 0, 0);
        if (HAVE_LABELS) lv = *reinterpret_cast<const uint4 *>(labels + i);
        const uint32_t cs[4] = {cv.x, cv.y, cv.z, cv.w};
        const uint32_t ps[4] = {pv.x, pv.y, pv.z, pv.w};
        const uint32_t ls[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t x = (cs[w] >> (8 * b)) & 0xFF;
                const uint32_t p = (ps[w] >> (8 * b)) & 0xFF;
                uint32_t l = 0;
                bool ok = (i + 4 * w + b < n) && x < QVZ_A && p < QVZ_A;
                if (HAVE_LABELS) {
                    l = ((ls[w] >> (8 * b)) & 0xFF) - kb;
                    ok = ok && l < kt;
                }
                if (ok) {
                    if (x < AA && p < AA) atomicAdd(&mine[l * ts + p * AA + x], 1u);
                    else miss = 1;                               // outside the hinted range: second pass below
                }
            }
    }
    if (__syncthreads_or(miss)) {
        // Rare: the sampled hint was too small. Re-scan the slab and send only the out-of-range pairs straight to
        // the global counters; kept out of the hot loop so that it costs nothing when it does not happen.
        for (uint64_t ch = c0; ch < c1; ++ch) {
            const uint64_t i0 = ch * rows + (uint64_t)threadIdx.x * 16;
            for (uint32_t r = 0; r < 16; ++r) {
                const uint64_t i = i0 + r;
                if (i >= n) break;
                const uint32_t x = cur[i], p = s ? prv[i] : 0u;
                if (x >= QVZ_A || p >= QVZ_A || (x < AA && p < AA)) continue;
                uint32_t l = 0;
                if (HAVE_LABELS) { l = labels[i] - kb; if (l >= kt) continue; }
                atomicAdd(&out[(uint64_t)l * C * PMF_TILE + p * QVZ_A + x], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kt * ts; i += T) {
        uint32_t v = 0;
        for (uint32_t r = 0; r < R; ++r) v += tile[r * rs + i];
        if (v) {
            const uint32_t l = i / ts, e = i % ts;
            atomicAdd(&out[(uint64_t)l * C * PMF_TILE + (e / AA) * QVZ_A + (e % AA)], v);
        }
    }
}

// ---- column-walking variant: every symbol is loaded exactly once ------------------------------------------------
// k_pmf_hist above reads each column twice (as `cur` of column s and as `prev` of column s+1) and measurement shows
// the second read costs as much as the first: with the LDS work removed it still runs at the HBM ceiling for 2 B per
// symbol. Here a block owns a slab of rows and walks the columns left to right: a thread keeps the previous column's
// 16-byte groups (and the labels) in registers, prefetches the next column while it counts the current one, and G
// consecutive column tiles live in LDS at a time (flushed with global atomics every G columns).
#define PMF_WALK_T 1024
typedef unsigned short pmf_u16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t pmf_lds_u32;

// One (prev, cur) pair that did not pass the packed fast check: full per-symbol logic (cold path).
template <bool HAVE_LABELS>
__device__ __noinline__ void pmf_count_slow(uint32_t x, uint32_t p, uint32_t lraw, uint32_t kb, uint32_t kt, uint32_t AA,
                                            uint32_t *col_tile, uint32_t *gcol, uint32_t C) {
    if (x >= QVZ_A || p >= QVZ_A) return;
    uint32_t l = 0;
    if (HAVE_LABELS) { l = lraw - kb; if (l >= kt) return; }
    if (x < AA && p < AA) atomicAdd(&col_tile[l * AA * AA + p * AA + x], 1u);
    else atomicAdd(&gcol[(uint64_t)l * C * PMF_TILE + p * QVZ_A + x], 1u);        // outside the hinted range
}

// Inner loop budget (k = 1): per 4-symbol word one packed range check (7 VALU) and per symbol v_perm_b32 +
// v_dot2_u32_u16 + ds_add_u32. The kernel is issue-bound, not bandwidth-bound, so instruction count is what matters.
// LMODE: 0 no labels (one cluster); 1 a label byte per read, asked for again in every column when more than two groups per
// thread are in flight (any number of clusters, PMF_KT of them per pass); 2 at most four clusters, all in one pass: a
// thread packs the labels of its groups to 2 bits each ONCE, before the first column - one dword per 16 reads - so that
// they stay in registers over all columns beside prv / cur / nxt (six groups per thread as with one cluster), no label is
// loaded twice, and a symbol's tile is one v_bfe + one v_mad away. A group that holds a label outside the pass's clusters
// (never, for labels that come from the k-means kernels) takes the per-symbol path with its label bytes read again.
template <int LMODE, int Q>
__global__ __launch_bounds__(PMF_WALK_T) void k_pmf_walk(const uint8_t *__restrict__ D, const uint8_t *__restrict__ labels,
                                                          uint64_t n, uint32_t C, uint64_t ld, uint32_t kb, uint32_t kt,
                                                          uint32_t R, uint32_t AA, uint32_t G, uint32_t *__restrict__ cnt) {
    extern __shared__ uint32_t tile[];   // [R][G][kt][AA][AA] (+1 dword skew per replica)
    const uint32_t T = PMF_WALK_T;
    const uint32_t ts = AA * AA, cs = kt * ts;                   // one cluster's tile, one column's tiles
    const uint32_t rs = G * cs + 1;                              // replica stride
    uint32_t *mine = tile + (threadIdx.x % R) * rs;
    const uint64_t slab0 = (uint64_t)blockIdx.x * T * 16 * Q;
    // group q of this thread = the 16 reads from ROW(q); which groups lie wholly / partly inside the matrix is kept as two
    // bit masks instead of Q 64-bit row numbers (registers are what limits the groups per thread)
    const uint64_t base_row = slab0 + (uint64_t)threadIdx.x * 16;
#define ROW(q) (base_row + (uint64_t)(q) * T * 16)
#define LIVE(q) (((full | part) >> (q)) & 1u)
#define FULL(q) ((full >> (q)) & 1u)
    uint32_t full = 0, part = 0;
    constexpr bool HAVE_LABELS = LMODE != 0;
    constexpr bool PACKED = LMODE == 2;
    uint4 prv[Q], cur[Q], lab[PACKED ? 1 : Q];
    uint32_t lab2[PACKED ? Q : 1];                               // PACKED: 16 labels x 2 bits
    uint32_t lbad = 0;                                           // PACKED: bit q = group q holds a label outside [kb, kb + kt)
    uint32_t pbad[Q];                                            // range check of prv[q], made when it was cur[q]
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (ROW(q) + 16 <= n) full |= 1u << q; else if (ROW(q) < n) part |= 1u << q;
        prv[q] = make_uint4(0, 0, 0, 0);                         // column 0 counts under prev = 0 (codebook.c:200)
        pbad[q] = 0;
        if (!PACKED) {
            lab[q] = make_uint4(0, 0, 0, 0);
            if (HAVE_LABELS && Q <= 2 && LIVE(q)) lab[q] = *reinterpret_cast<const uint4 *>(labels + ROW(q));
        } else {
            const uint4 lv = LIVE(q) ? *reinterpret_cast<const uint4 *>(labels + ROW(q)) : make_uint4(0, 0, 0, 0);
            const uint32_t lw4[4] = {lv.x - 0x01010101u * kb, lv.y - 0x01010101u * kb, lv.z - 0x01010101u * kb, lv.w - 0x01010101u * kb};   // kb = 0 in this mode
            const uint32_t kk = 0x01010101u * (0x80u - kt);      // byte + (128 - kt) sets bit 7 iff byte >= kt
            uint32_t packed = 0, bad = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                bad |= ((lw4[w] + kk) | lw4[w]) & 0x80808080u;
                uint32_t x = lw4[w] & 0x03030303u;               // four labels of 2 bits, one per byte -> 8 bits
                x = (x | (x >> 6)) & 0x000F000Fu;
                x = (x | (x >> 12)) & 0xFFu;
                packed |= x << (8 * w);
            }
            lab2[q] = packed;
            if (bad) {                                           // cold: this group is counted after the column walk, symbol by symbol
                lbad |= 1u << q;
                full &= ~(1u << q);                              // the walk skips it
                part &= ~(1u << q);
            }
        }
        cur[q] = LIVE(q) ? *reinterpret_cast<const uint4 *>(D + ROW(q)) : make_uint4(0, 0, 0, 0);
    }
    const uint32_t k80 = 0x01010101u * (0x80u - AA);             // byte + (128 - AA) sets bit 7 iff byte >= AA
    const uint32_t dump = (uint32_t)(uintptr_t)(pmf_lds_u32 *)tile + (uint32_t)(((mine - tile) + rs - 1) * 4);   // the replica's skew dword
    pmf_u16x2 coef;
    coef.x = 4;                                                  // byte address = 4*x + 4*AA*p + tile base
    coef.y = (unsigned short)(4 * AA);
    for (uint32_t g0 = 0; g0 < C; g0 += G) {
        const uint32_t g1 = g0 + G < C ? g0 + G : C;
        for (uint32_t i = threadIdx.x; i < R * rs; i += T) tile[i] = 0;
        __syncthreads();
        for (uint32_t s = g0; s < g1; ++s) {
            uint4 nxt[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q)                          // prefetch column s+1 while column s is counted
                nxt[q] = (s + 1 < C && LIVE(q)) ? *reinterpret_cast<const uint4 *>(D + (uint64_t)(s + 1) * ld + ROW(q))
                                                   : make_uint4(0, 0, 0, 0);
            // With more than two groups per thread the labels do not stay in registers beside prv / cur / nxt (four groups
            // spilled: 16.6 ms instead of 7.1 at 100M x 150, k = 3): they are asked for again in every column - 1 byte per read
            // and column out of L2 / the Infinity Cache (the label vector is 100 MB) - and bigger slabs mean fewer flushes.
            if (HAVE_LABELS && !PACKED && Q > 2) {
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    lab[q] = LIVE(q) ? *reinterpret_cast<const uint4 *>(labels + ROW(q)) : make_uint4(0, 0, 0, 0);
            }
            uint32_t *col_tile = mine + (s - g0) * cs;
            uint32_t *gcol = cnt + ((uint64_t)kb * C + s) * PMF_TILE;
            const uint32_t base = (uint32_t)(uintptr_t)(pmf_lds_u32 *)tile + (uint32_t)((col_tile - tile) * 4);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t cw[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
                const uint32_t pw[4] = {prv[q].x, prv[q].y, prv[q].z, prv[q].w};
                const uint32_t lw[4] = {lab[PACKED ? 0 : q].x, lab[PACKED ? 0 : q].y, lab[PACKED ? 0 : q].z, lab[PACKED ? 0 : q].w};
                uint32_t l2 = lab2[PACKED ? q : 0];
                // the packed labels do not change from column to column: without this the compiler computes the sixteen tile
                // offsets of every group once, in front of the column loop, and spills them (380 VGPRs' worth at six groups)
                asm volatile("" : "+v"(l2));
                // one range check per 16-byte group and column: bit 7 of a byte of cbad is set iff that symbol is outside
                // the tile; the group's check as `prev` is the one it got as `cur` a column earlier
                const uint32_t cbad = (((cw[0] + k80) | cw[0]) | ((cw[1] + k80) | cw[1]) | ((cw[2] + k80) | cw[2]) |
                                       ((cw[3] + k80) | cw[3])) & 0x80808080u;
                const uint32_t gbad = cbad | pbad[q];
                pbad[q] = cbad;
                auto fast_word = [&](int w) {
                    uint32_t addr[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        // [x, 0, p, 0] as two u16 lanes, then 4*x + 4*AA*p + base in one dot product
                        const uint32_t sel = 0x0C000C00u | ((4u + b) << 16) | (uint32_t)b;
                        const uint32_t xp = __builtin_amdgcn_perm(pw[w], cw[w], sel);
                        pmf_u16x2 v;
                        v.x = (unsigned short)(xp & 0xFFFF);
                        v.y = (unsigned short)(xp >> 16);
                        addr[b] = __builtin_amdgcn_udot2(v, coef, base, false);
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (PACKED) {
                            addr[b] = __umul24(__builtin_amdgcn_ubfe(l2, 8 * w + 2 * b, 2), ts * 4) + addr[b];
                        } else if (HAVE_LABELS) {
                            const uint32_t l = ((lw[w] >> (8 * b)) & 0xFF) - kb;
                            // a read of a cluster outside this pass's tiles (or of no cluster at all) counts into the
                            // replica's skew dword, which nobody reads: a select instead of an exec-mask region per symbol
                            addr[b] = l < kt ? addr[b] + __umul24(l, ts * 4) : dump;
                        }
                        // `base` holds the tile's LDS address: the sum is the operand of ds_add_u32 as it stands
                        __hip_atomic_fetch_add((pmf_lds_u32 *)(uintptr_t)addr[b], 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                };
                if (FULL(q) && gbad == 0) {                      // the usual case: sixteen pairs inside the tile
#pragma unroll
                    for (int w = 0; w < 4; ++w) fast_word(w);
                } else if (FULL(q)) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t bad = ((cw[w] + k80) | (pw[w] + k80) | cw[w] | pw[w]) & 0x80808080u;
                        if (bad == 0) {
                            fast_word(w);
                        } else {
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                pmf_count_slow<HAVE_LABELS>((cw[w] >> (8 * b)) & 0xFF, (pw[w] >> (8 * b)) & 0xFF,
                                                            PACKED ? kb + ((l2 >> (8 * w + 2 * b)) & 3u) : (lw[w] >> (8 * b)) & 0xFF,
                                                            kb, kt, AA, col_tile, gcol, C);
                        }
                    }
                } else if (LIVE(q)) {                            // the last, partial group of the matrix
                    for (uint32_t r = 0; r < 16 && ROW(q) + r < n; ++r)
                        pmf_count_slow<HAVE_LABELS>((cw[r >> 2] >> (8 * (r & 3))) & 0xFF, (pw[r >> 2] >> (8 * (r & 3))) & 0xFF,
                                                    PACKED ? kb + ((l2 >> (2 * r)) & 3u) : (lw[r >> 2] >> (8 * (r & 3))) & 0xFF,
                                                    kb, kt, AA, col_tile, gcol, C);
                }
                prv[q] = cur[q];
                cur[q] = nxt[q];
            }
        }
        __syncthreads();
        // Two neighbouring counters (x even, x + 1) leave in one 64-bit atomic: a counter stays below 2^32 (n < 2^32,
        // checked by the launcher), so the low half never carries into the high one. The global atomics of the flush
        // are a visible share of the kernel: 1526 blocks x 150 columns x AA^2 counters at 100M reads.
        for (uint32_t i = 2 * threadIdx.x; i < (g1 - g0) * cs; i += 2 * T) {
            uint32_t v0 = 0, v1 = 0;
            for (uint32_t r = 0; r < R; ++r) { v0 += tile[r * rs + i]; v1 += tile[r * rs + i + 1]; }
            if (v0 | v1) {
                const uint32_t s = g0 + i / cs, l = (i % cs) / ts, e = i % ts;          // AA is even: e and e + 1 share a row
                atomicAdd(reinterpret_cast<unsigned long long *>(
                              &cnt[(((uint64_t)(kb + l) * C + s) * QVZ_A + e / AA) * QVZ_A + e % AA]),
                          (unsigned long long)v0 | ((unsigned long long)v1 << 32));
            }
        }
        __syncthreads();
    }
    if (PACKED && lbad) {
        // Groups that hold a label outside this pass's clusters (a caller's own label vector; the k-means kernels never write
        // one): counted here, symbol by symbol, straight into the global counters.
        for (int q = 0; q < Q; ++q) {
            if (!((lbad >> q) & 1u)) continue;
            const uint64_t r0 = ROW(q);
            for (uint32_t s = 0; s < C; ++s)
                for (uint32_t r = 0; r < 16 && r0 + r < n; ++r) {
                    const uint32_t x = D[(uint64_t)s * ld + r0 + r], p = s ? D[(uint64_t)(s - 1) * ld + r0 + r] : 0u;
                    const uint32_t l = (uint32_t)labels[r0 + r] - kb;
                    if (x >= QVZ_A || p >= QVZ_A || l >= kt) continue;
                    atomicAdd(&cnt[((uint64_t)(kb + l) * C + s) * PMF_TILE + p * QVZ_A + x], 1u);
                }
        }
    }
}
#undef ROW
#undef LIVE
#undef FULL

// largest symbol in an evenly spread sample of every column (64 x 1024 rows per column)
__global__ __launch_bounds__(256) void k_sym_max(const uint8_t *__restrict__ D, uint64_t n, uint64_t ld,
                                                 uint32_t *__restrict__ result) {
    const uint8_t *col = D + (uint64_t)blockIdx.x * ld;
    const uint64_t groups = (n + 1023) / 1024;                   // 1024-row groups in the column
    const uint64_t step = groups > 64 ? groups / 64 : 1;
    uint32_t m = 0;
    for (uint64_t g = 0; g < groups; g += step) {
        const uint64_t i = g * 1024 + (uint64_t)(threadIdx.x & 63) * 16;
        if ((threadIdx.x >> 6) != (g / step) % 4 || i >= n) continue;     // the 4 waves take groups in turn
        const uint4 v = *reinterpret_cast<const uint4 *>(col + i);
        const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (i + 4 * w + b < n) { const uint32_t x = (ws[w] >> (8 * b)) & 0xFF; m = x > m ? x : m; }
    }
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(result, m);
}

// ---- per-device scratch --------------------------------------------------------------------------------------------
// Small device buffers that outlive a call (the symbol-maximum word of the PMF pass, the WELL jump tables and their
// workspaces) belong to the device they were allocated on: a process may encode on several ordinals (qvz_options_t.device,
// QVZ_DEVICES). They are kept per ordinal; a caller holds the ordinal's lock while it enqueues work that uses them, and a
// second stream that takes them over first waits (stream-ordered, hipStreamWaitEvent) for the previous user's last kernel.
#define QVZ_MAX_DEVICES 64
struct DevScratch {
    std::mutex mu;
    uint32_t *d_max = nullptr;      // k_sym_max result
    uint32_t *d_pow = nullptr;      // [WELL_MAXJ][1024*32] : T^(2^j)
    uint32_t *d_states = nullptr;   // chunk start states
    uint64_t states_cap = 0;
    uint32_t *d_tmp = nullptr;      // 2 x 32 words ping-pong for single-state jumps
    int32_t *d_route = nullptr;     // which quantizer kernel serves the current qvz_hip_quantize call (k_quantize_route)
    uint32_t *d_ws = nullptr;       // word stream of one pass of qvz_hip_well_draws
    uint64_t ws_words = 0;
    hipEvent_t busy = nullptr;      // recorded after the last kernel that used the scratch
    hipStream_t last = nullptr;
    bool used = false;
};
static DevScratch g_dev[QVZ_MAX_DEVICES];

struct ScratchLock {                // RAII: lock the current device's scratch for work on `st`
    DevScratch *d = nullptr;
    hipStream_t st;
    int rc = QVZ_OK;
    explicit ScratchLock(hipStream_t stream) : st(stream) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= QVZ_MAX_DEVICES) { rc = qvz_set_error(QVZ_E_HIP, "no current device"); return; }
        d = &g_dev[dev];
        d->mu.lock();
        if (!d->busy && hipEventCreateWithFlags(&d->busy, hipEventDisableTiming) != hipSuccess) { rc = qvz_set_error(QVZ_E_HIP, "event create failed"); return; }
        if (d->used && d->last != st && hipStreamWaitEvent(st, d->busy, 0) != hipSuccess) rc = qvz_set_error(QVZ_E_HIP, "stream wait failed");
    }
    ~ScratchLock() {
        if (!d) return;
        if (d->busy && hipEventRecord(d->busy, st) == hipSuccess) { d->used = true; d->last = st; }
        d->mu.unlock();
    }
};

extern "C" int qvz_hip_pmf_hist(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c, uint64_t ld,
                                uint32_t k, uint32_t *d_cnt, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    if (k == 0 || k > 255) return qvz_set_error(QVZ_E_ARG, "k=%u out of range 1..255", k);
    if (k > 1 && !d_labels) return qvz_set_error(QVZ_E_ARG, "labels required for k > 1");
    if (n >= (1ull << 32)) return qvz_set_error(QVZ_E_ARG, "n >= 2^32 overflows the u32 counters (reference pmf.h:35)");
    hipStream_t st = (hipStream_t)stream;

    // tile side from a device-side sample of the data (one tiny launch + a 4-byte read back)
    // QVZ_PMF (test hook, "key=number,..."): aa = side of the LDS tile instead of the sampled one, algo = 0 the two-read kernel / 1
    // the column-walking one whatever the size, q = its groups per thread - the choices the sizes below make, at sizes a test can afford
    double opt = 0.0;
    unsigned AA = qvz_env_number("QVZ_PMF", "aa", &opt) ? (unsigned)opt : 0u;
    if (AA == 0) {
        ScratchLock lk(st);
        if (lk.rc) return lk.rc;
        if (!lk.d->d_max) QVZ_HIP_CHECK(hipMalloc(&lk.d->d_max, sizeof(uint32_t)));
        uint32_t *d_max = lk.d->d_max;
        uint32_t h_max = 0;
        QVZ_HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_sym_max, dim3(c), dim3(256), 0, st, d_sym, n, ld, d_max);
        QVZ_HIP_CHECK(hipMemcpyAsync(&h_max, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        QVZ_HIP_CHECK(hipStreamSynchronize(st));
        AA = h_max + 1;
    }
    AA = (AA + 1) & ~1u;
    if (AA > QVZ_A) AA = QVZ_A;
    if (AA < 2) AA = 2;

    const unsigned lds_budget = 160 * 1024;
    // The walking kernel needs >= ~1000 slabs of 16384 rows to fill the chip; smaller inputs use the two-read kernel.
    const unsigned algo = qvz_env_number("QVZ_PMF", "algo", &opt) ? (unsigned)opt : (n >= 1000ull * 16384 ? 1u : 0u);
    if (algo >= 1) {
        // column-walking kernel: slabs of 16384*Q rows, G column tiles x R replicas of the counters in LDS
        for (uint32_t kb = 0; kb < k; kb += PMF_KT) {
            const uint32_t kt = (k - kb) < PMF_KT ? (k - kb) : PMF_KT;
            const size_t tile_bytes = (size_t)kt * AA * AA * sizeof(uint32_t);
            // The kernel's ~96 VGPRs leave room for one 1024-thread block per CU, so a block may use the whole LDS.
            // Measured (100M x 150, 42 symbols): G = 3 column tiles x R = 5 replicas is the best (4.30 ms; R = 5, G = 2: 4.43;
            // R = 3, G = 3: 4.59; more than 5 replicas buy nothing; three clusters, two replicas of three tiles: the same within 0.1 ms
            // for any (G, R) the LDS holds - what is left there is the flush, three times the global atomics).
            unsigned G = c < 3 ? c : 3;
            unsigned R = (unsigned)(lds_budget / (G * tile_bytes + 4));
            if (R > 5) R = 5;
            if (R < 1) R = 1;
            while (G > 1 && (size_t)R * (G * tile_bytes + 4) > lds_budget) G -= 1;
            const size_t lds = (size_t)R * (G * tile_bytes + 4);
            // bigger slabs = fewer blocks = fewer flush atomics; six groups per thread is what 128 VGPRs hold (k = 1, and k = 2 .. 4 with
            // the labels packed to 2 bits in registers); more clusters: the label bytes are asked for again in every column (four
            // groups 6.4 ms, two 7.1, six 9.0 at 100M x 150, k = 3 before the packed form)
            const bool packed = k > 1 && k <= 4;
            unsigned Q = qvz_env_number("QVZ_PMF", "q", &opt) ? (unsigned)opt : 0u;
            if (Q == 0) Q = ((k == 1 || packed) && n >= 768ull * 98304) ? 6 : (n >= 768ull * 65536) ? 4 : (n >= 768ull * 32768 ? 2 : 1);
            if (Q != 1 && Q != 2 && Q != 4 && Q != 6) Q = 4;
            const uint64_t slab = (uint64_t)PMF_WALK_T * 16 * Q;
            dim3 grid((unsigned)((n + slab - 1) / slab));
#define LAUNCH_WALK(LBL, QQ)                                                                                           \
    do {                                                                                                               \
        if (lds > 64 * 1024)                                                                                           \
            QVZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pmf_walk<LBL, QQ>),                    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        hipLaunchKernelGGL((k_pmf_walk<LBL, QQ>), grid, dim3(PMF_WALK_T), lds, st, d_sym, d_labels, n, c, ld, kb, kt,  \
                           R, AA, G, d_cnt);                                                                           \
    } while (0)
            if (k == 1) {
                if (Q == 6) LAUNCH_WALK(0, 6); else if (Q >= 4) LAUNCH_WALK(0, 4); else if (Q == 2) LAUNCH_WALK(0, 2); else LAUNCH_WALK(0, 1);
            } else if (packed) {
                if (Q == 6) LAUNCH_WALK(2, 6); else if (Q >= 4) LAUNCH_WALK(2, 4); else if (Q == 2) LAUNCH_WALK(2, 2); else LAUNCH_WALK(2, 1);
            } else {
                if (Q == 6) LAUNCH_WALK(1, 6); else if (Q >= 4) LAUNCH_WALK(1, 4); else if (Q == 2) LAUNCH_WALK(1, 2); else LAUNCH_WALK(1, 1);
            }
#undef LAUNCH_WALK
            QVZ_HIP_CHECK(hipGetLastError());
        }
        return QVZ_OK;
    }
    for (uint32_t kb = 0; kb < k; kb += PMF_KT) {
        const uint32_t kt = (k - kb) < PMF_KT ? (k - kb) : PMF_KT;
        const size_t tile_bytes = ((size_t)kt * AA * AA + 1) * sizeof(uint32_t);
        // replicas: as many as fit in half the LDS (two 1024-thread blocks per CU = full occupancy), at most 16
        unsigned R = (unsigned)((lds_budget / 2) / tile_bytes);
        if (R > 16) R = 16;
        if (R < 1) R = 1;
        while (R > 1 && R * tile_bytes > lds_budget) R -= 1;
        const size_t lds = R * tile_bytes;
        const unsigned T = 1024;
        unsigned per_cu = (unsigned)(lds_budget / lds);
        if (per_cu * T > 2048) per_cu = 2048 / T;
        if (per_cu < 1) per_cu = 1;
        const uint64_t chunks = (n + (uint64_t)T * 16 - 1) / ((uint64_t)T * 16);
        // enough blocks for several rounds per CU so the tail is short, few enough that the flush stays negligible
        unsigned slabs = (256 * per_cu * 4 + c - 1) / c;
        if (slabs < 1) slabs = 1;
        if (slabs > chunks) slabs = (unsigned)chunks;
        dim3 grid(c, slabs);
        if (k == 1) {
            if (lds > 64 * 1024)
                QVZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pmf_hist<false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_pmf_hist<false>, grid, dim3(T), lds, st, d_sym, d_labels, n, c, ld, kb, kt, R, AA, d_cnt);
        } else {
            if (lds > 64 * 1024)
                QVZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pmf_hist<true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_pmf_hist<true>, grid, dim3(T), lds, st, d_sym, d_labels, n, c, ld, kb, kt, R, AA, d_cnt);
        }
        QVZ_HIP_CHECK(hipGetLastError());
    }
    return QVZ_OK;
}

// ------------------------------------------------------------------------------------------------
// WELL-1024a (reference src/well.c:8-24) with GF(2) jump-ahead.
//
// Rotated view v[k] = state[(n+k)&31]: one step is the time-invariant linear map
//     z0=v[31]; z1=v[0]^v[3]^(v[3]>>8); z2=v[24]^(v[24]<<19)^v[10]^(v[10]<<14);
//     v' = [ z0^(z0<<11)^z1^(z1<<7)^z2^(z2<<13),  z1^z2,  v[1], ..., v[30] ]        (v'[0] is the output word)
// so T^(2^j) by repeated squaring of a 1024x1024 bit matrix gives any jump; a jumped state is a valid
// reference state with n = 0. Chunk start states are produced by log2(#chunks) batched mat-vec levels,
// then one thread per chunk runs the plain recurrence (32 steps fully unrolled, all state in registers)
// and writes whole 128-byte lines.
// ------------------------------------------------------------------------------------------------
#define WELL_MAXJ 46
#define WELL_CHUNK_LOG2 13                  // 8192 words (32 768 draws) per generator thread
#define WELL_CHUNK (1u << WELL_CHUNK_LOG2)

static void well_step_host(uint32_t *v) {
    uint32_t z0 = v[31], v1 = v[3], v2 = v[24], v3 = v[10];
    uint32_t z1 = v[0] ^ v1 ^ (v1 >> 8);
    uint32_t z2 = v2 ^ (v2 << 19) ^ v3 ^ (v3 << 14);
    uint32_t out = z0 ^ (z0 << 11) ^ z1 ^ (z1 << 7) ^ z2 ^ (z2 << 13);
    uint32_t y = z1 ^ z2;
    for (int k = 31; k >= 2; --k) v[k] = v[k - 1];
    v[1] = y;
    v[0] = out;
}

// rows[r*32 + w]: bit c of row r set <=> new bit r depends on old bit c
static void well_build_T(uint32_t *rows) {
    memset(rows, 0, 1024 * 32 * sizeof(uint32_t));
    for (int c = 0; c < 1024; ++c) {
        uint32_t v[32];
        memset(v, 0, sizeof(v));
        v[c >> 5] = 1u << (c & 31);
        well_step_host(v);
        for (int r = 0; r < 1024; ++r)
            if ((v[r >> 5] >> (r & 31)) & 1) rows[r * 32 + (c >> 5)] |= 1u << (c & 31);
    }
}

// out = A*A over GF(2): row r of out = XOR of rows c of A for every set bit c of A's row r
__global__ __launch_bounds__(64) void k_gf2_square(const uint32_t *__restrict__ A, uint32_t *__restrict__ out) {
    const uint32_t r = blockIdx.x, w = threadIdx.x & 31;
    uint32_t acc = 0;
    for (uint32_t cw = 0; cw < 32; ++cw) {
        uint32_t bits = A[r * 32 + cw];
        while (bits) {
            const uint32_t b = __builtin_ctz(bits);
            bits &= bits - 1;
            acc ^= A[(cw * 32 + b) * 32 + w];
        }
    }
    if (threadIdx.x < 32) out[r * 32 + w] = acc;
}

// dst[b] = M * src[b] for `count` states; a block handles up to 8 states so each matrix row is fetched once
// per 8 products. Thread r owns output bit r; 64-lane ballots assemble two output words per wave.
#define WELL_SPB 8
__global__ __launch_bounds__(1024) void k_gf2_apply(const uint32_t *__restrict__ M, const uint32_t *__restrict__ src,
                                                     uint32_t *__restrict__ dst, uint64_t count) {
    __shared__ uint32_t sv[WELL_SPB][32];
    const uint64_t b0 = (uint64_t)blockIdx.x * WELL_SPB;
    const uint32_t r = threadIdx.x;
    if (r < WELL_SPB * 32) {
        const uint64_t st = b0 + (r >> 5);
        sv[r >> 5][r & 31] = st < count ? src[st * 32 + (r & 31)] : 0;
    }
    uint32_t row[32];
    const uint4 *mrow = reinterpret_cast<const uint4 *>(M + (uint64_t)r * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint4 t = mrow[q];
        row[4 * q] = t.x; row[4 * q + 1] = t.y; row[4 * q + 2] = t.z; row[4 * q + 3] = t.w;
    }
    __syncthreads();
#pragma unroll
    for (int sidx = 0; sidx < WELL_SPB; ++sidx) {
        if (b0 + sidx >= count) break;
        uint32_t x = 0;
#pragma unroll
        for (int w = 0; w < 32; ++w) x ^= row[w] & sv[sidx][w];
        const unsigned long long bal = __ballot(__popc(x) & 1);
        if ((r & 63) == 0) {
            dst[(b0 + sidx) * 32 + (r >> 5)] = (uint32_t)bal;
            dst[(b0 + sidx) * 32 + (r >> 5) + 1] = (uint32_t)(bal >> 32);
        }
    }
}

// one thread = one chunk of WELL_CHUNK words, reference recurrence with n = 0,31,30,...,1 unrolled.
// `pad` words are left free behind every chunk of the output. With pad = 0 (the linear word stream of qvz_hip_well_words) the 64
// lanes of a store write 64 lines that lie 32 KB apart - one memory channel - and every wave of the launch is at about the same
// place in its chunk: measured 1.06 TB/s of stores. The draw generator, whose word buffer is its own, asks for WELL_PAD = 32
// words (chunks 257 lines apart: consecutive lanes, consecutive channels): 2.09 TB/s (k_draws_transpose reads through well_phys).
#define WELL_PAD 32u
__device__ __host__ __forceinline__ uint64_t well_phys(uint64_t rel, uint32_t pad) { return rel + (rel >> WELL_CHUNK_LOG2) * pad; }
__global__ __launch_bounds__(64) void k_well_gen(const uint32_t *__restrict__ states, uint64_t nchunks, uint64_t nwords,
                                                 uint32_t *__restrict__ words, uint32_t pad) {
    const uint64_t t = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= nchunks) return;
    uint32_t s[32];
    const uint4 *sp = reinterpret_cast<const uint4 *>(states + t * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint4 v = sp[q];
        s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w;
    }
    const uint64_t w0 = t * WELL_CHUNK;
    const uint64_t left = nwords - w0;
    const uint32_t todo = left < WELL_CHUNK ? (uint32_t)left : WELL_CHUNK;
    for (uint32_t base = 0; base < todo; base += 32) {
        uint32_t out[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int n = (32 - i) & 31;                        // well.c:10-23 with compile-time n
            const uint32_t z0 = s[(n + 31) & 31];
            const uint32_t v1 = s[(n + 3) & 31], v2 = s[(n + 24) & 31], v3 = s[(n + 10) & 31];
            const uint32_t z1 = s[n] ^ (v1 ^ (v1 >> 8));
            const uint32_t z2 = (v2 ^ (v2 << 19)) ^ (v3 ^ (v3 << 14));
            s[n] = z1 ^ z2;
            const int nn = (n + 31) & 31;
            s[nn] = (z0 ^ (z0 << 11)) ^ (z1 ^ (z1 << 7)) ^ (z2 ^ (z2 << 13));
            out[i] = s[nn];
        }
        uint32_t *dstw = words + w0 + t * pad + base;
        if (base + 32 <= todo) {
            uint4 *d4 = reinterpret_cast<uint4 *>(dstw);       // w0+base is a multiple of 32 words = 128 B
#pragma unroll
            for (int q = 0; q < 8; ++q) d4[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (base + i < todo) dstw[i] = out[i];
        }
    }
}

// T^(2^j) for the current device (built once per ordinal); the caller holds the ordinal's ScratchLock
static int well_init(DevScratch &W, hipStream_t st) {
    if (W.d_pow) return QVZ_OK;
    const size_t msz = 1024 * 32 * sizeof(uint32_t);
    uint32_t *T = (uint32_t *)malloc(msz);
    if (!T) return qvz_set_error(QVZ_E_NOMEM, "host alloc");
    well_build_T(T);
    QVZ_HIP_CHECK(hipMalloc(&W.d_pow, WELL_MAXJ * msz));
    QVZ_HIP_CHECK(hipMalloc(&W.d_tmp, 2 * 32 * sizeof(uint32_t)));
    QVZ_HIP_CHECK(hipMemcpy(W.d_pow, T, msz, hipMemcpyHostToDevice));
    free(T);
    for (int j = 1; j < WELL_MAXJ; ++j)
        hipLaunchKernelGGL(k_gf2_square, dim3(1024), dim3(64), 0, st, W.d_pow + (size_t)(j - 1) * 1024 * 32,
                           W.d_pow + (size_t)j * 1024 * 32);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// d_out (32 words) = T^words * state ; uses d_tmp ping-pong
static int well_jump_device(DevScratch &W, const uint32_t *h_state, uint64_t words, uint32_t *d_out, hipStream_t st) {
    if (words >> WELL_MAXJ) return qvz_set_error(QVZ_E_ARG, "WELL jump too large");
    uint32_t *a = W.d_tmp, *b = W.d_tmp + 32;
    QVZ_HIP_CHECK(hipMemcpyAsync(a, h_state, 128, hipMemcpyHostToDevice, st));
    for (int j = 0; j < WELL_MAXJ; ++j) {
        if ((words >> j) & 1) {
            hipLaunchKernelGGL(k_gf2_apply, dim3(1), dim3(1024), 0, st, W.d_pow + (size_t)j * 1024 * 32, a, b,
                               (uint64_t)1);
            uint32_t *t = a; a = b; b = t;
        }
    }
    QVZ_HIP_CHECK(hipMemcpyAsync(d_out, a, 128, hipMemcpyDeviceToDevice, st));
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

extern "C" int qvz_hip_well_jump(const uint32_t *h_state, uint64_t words, uint32_t *h_state_out) {
    ScratchLock lk(0);
    if (lk.rc) return lk.rc;
    int rc = well_init(*lk.d, 0);
    if (rc) return rc;
    uint32_t *d = nullptr;
    QVZ_HIP_CHECK(hipMalloc(&d, 128));
    rc = well_jump_device(*lk.d, h_state, words, d, 0);
    if (!rc && hipMemcpy(h_state_out, d, 128, hipMemcpyDeviceToHost) != hipSuccess)
        rc = qvz_set_error(QVZ_E_HIP, "copy back failed");
    (void)hipFree(d);
    return rc;
}

// dst[b] = (jump[b] >> bit) & 1 ? M * src[b] : src[b]: one level of many different jumps at once (thread r owns output bit r)
__global__ __launch_bounds__(1024) void k_gf2_apply_if(const uint32_t *__restrict__ M, const uint32_t *__restrict__ src,
                                                        uint32_t *__restrict__ dst, const uint64_t *__restrict__ jump, int bit,
                                                        uint64_t count) {
    __shared__ uint32_t sv[WELL_SPB][32];
    const uint64_t b0 = (uint64_t)blockIdx.x * WELL_SPB;
    const uint32_t r = threadIdx.x;
    if (r < WELL_SPB * 32) {
        const uint64_t st = b0 + (r >> 5);
        sv[r >> 5][r & 31] = st < count ? src[st * 32 + (r & 31)] : 0;
    }
    uint32_t row[32];
    const uint4 *mrow = reinterpret_cast<const uint4 *>(M + (uint64_t)r * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint4 t = mrow[q];
        row[4 * q] = t.x; row[4 * q + 1] = t.y; row[4 * q + 2] = t.z; row[4 * q + 3] = t.w;
    }
    __syncthreads();
#pragma unroll
    for (int sidx = 0; sidx < WELL_SPB; ++sidx) {
        if (b0 + sidx >= count) break;                            // block-uniform
        const bool on = (jump[b0 + sidx] >> bit) & 1;             // block-uniform
        uint32_t x = 0;
#pragma unroll
        for (int w = 0; w < 32; ++w) x ^= row[w] & sv[sidx][w];
        const unsigned long long bal = __ballot(__popc(x) & 1);
        if ((r & 63) == 0) {
            dst[(b0 + sidx) * 32 + (r >> 5)] = on ? (uint32_t)bal : sv[sidx][r >> 5];
            dst[(b0 + sidx) * 32 + (r >> 5) + 1] = on ? (uint32_t)(bal >> 32) : sv[sidx][(r >> 5) + 1];
        }
    }
}

// Generator states after h_jump[i] outputs each (rotated to n = 0), i = 0 .. count-1: what the .q headers of `count`
// shards store. One masked batched mat-vec per bit of the jumps instead of a chain of launches per shard.
extern "C" int qvz_hip_well_jump_many(const uint32_t *h_state, const uint64_t *h_jump, uint64_t count, uint32_t *h_states_out) {
    if (!h_state || !h_jump || !h_states_out) return qvz_set_error(QVZ_E_ARG, "null argument");
    if (count == 0) return QVZ_OK;
    ScratchLock lk(0);
    if (lk.rc) return lk.rc;
    int rc = well_init(*lk.d, 0);
    if (rc) return rc;
    uint64_t top = 0;
    for (uint64_t i = 0; i < count; ++i) top |= h_jump[i];
    if (top >> WELL_MAXJ) return qvz_set_error(QVZ_E_ARG, "WELL jump too large");
    uint32_t *d_a = nullptr, *d_b = nullptr;
    uint64_t *d_jump = nullptr;
    uint32_t *h_init = (uint32_t *)malloc(count * 128);
    if (!h_init) return qvz_set_error(QVZ_E_NOMEM, "host alloc");
    for (uint64_t i = 0; i < count; ++i) memcpy(h_init + i * 32, h_state, 128);
    hipError_t e = hipMalloc(&d_a, count * 128);
    if (e == hipSuccess) e = hipMalloc(&d_b, count * 128);
    if (e == hipSuccess) e = hipMalloc(&d_jump, count * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(d_a, h_init, count * 128, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_jump, h_jump, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    free(h_init);
    if (e == hipSuccess) {
        for (int j = 0; j < WELL_MAXJ && (top >> j); ++j) {
            if (!((top >> j) & 1)) continue;
            hipLaunchKernelGGL(k_gf2_apply_if, dim3((unsigned)((count + WELL_SPB - 1) / WELL_SPB)), dim3(1024), 0, 0,
                               lk.d->d_pow + (size_t)j * 1024 * 32, d_a, d_b, d_jump, j, count);
            uint32_t *t = d_a; d_a = d_b; d_b = t;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_states_out, d_a, count * 128, hipMemcpyDeviceToHost);
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_jump);
    if (e != hipSuccess) return qvz_set_error(QVZ_E_HIP, "qvz_hip_well_jump_many: %s", hipGetErrorString(e));
    return QVZ_OK;
}

// W.d_states[i] = generator state (n = 0) after first_word + i * WELL_CHUNK words, i < nchunks
static int well_states_locked(DevScratch &W, const uint32_t *h_state, uint64_t first_word, uint64_t nchunks, hipStream_t st) {
    int rc = well_init(W, st);
    if (rc) return rc;
    if (nchunks > W.states_cap) {
        if (W.d_states) { QVZ_HIP_CHECK(hipStreamSynchronize(st)); (void)hipFree(W.d_states); W.d_states = nullptr; W.states_cap = 0; }
        const uint64_t cap = nchunks + nchunks / 4 + 64;
        QVZ_HIP_CHECK(hipMalloc(&W.d_states, cap * 128));
        W.states_cap = cap;
    }
    rc = well_jump_device(W, h_state, first_word, W.d_states, st);
    if (rc) return rc;
    // doubling: states [2^l, 2^(l+1)) = T^(CHUNK*2^l) * states [0, 2^l)
    for (int l = 0; (1ull << l) < nchunks; ++l) {
        const uint64_t have = 1ull << l;
        const uint64_t make = (nchunks - have) < have ? (nchunks - have) : have;
        const int j = WELL_CHUNK_LOG2 + l;
        if (j >= WELL_MAXJ) return qvz_set_error(QVZ_E_ARG, "WELL range too large");
        hipLaunchKernelGGL(k_gf2_apply, dim3((unsigned)((make + WELL_SPB - 1) / WELL_SPB)), dim3(1024), 0, st,
                           W.d_pow + (size_t)j * 1024 * 32, W.d_states, W.d_states + have * 32, make);
    }
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// nwords words from chunk `chunk0` of the states above on (d_words[0] = the first word of that chunk)
static int well_gen_locked(DevScratch &W, uint64_t chunk0, uint64_t nwords, uint32_t *d_words, hipStream_t st, uint32_t pad) {
    if (nwords == 0) return QVZ_OK;
    const uint64_t nchunks = (nwords + WELL_CHUNK - 1) / WELL_CHUNK;
    hipLaunchKernelGGL(k_well_gen, dim3((unsigned)((nchunks + 63) / 64)), dim3(64), 0, st, W.d_states + chunk0 * 32, nchunks,
                       nwords, d_words, pad);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

static int well_words_locked(DevScratch &W, const uint32_t *h_state, uint64_t first_word, uint64_t nwords, uint32_t *d_words,
                             hipStream_t st) {
    if (nwords == 0) return QVZ_OK;
    if ((uintptr_t)d_words & 15) return qvz_set_error(QVZ_E_ARG, "d_words must be 16-byte aligned");
    const int rc = well_states_locked(W, h_state, first_word, (nwords + WELL_CHUNK - 1) / WELL_CHUNK, st);
    return rc ? rc : well_gen_locked(W, 0, nwords, d_words, st, 0);
}

extern "C" int qvz_hip_well_words(const uint32_t *h_state, uint64_t first_word, uint64_t nwords, uint32_t *d_words,
                                  void *stream) {
    ScratchLock lk((hipStream_t)stream);
    if (lk.rc) return lk.rc;
    return well_words_locked(*lk.d, h_state, first_word, nwords, d_words, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Draw matrix: draws[s*ld + i] = 7-bit WELL draw of (read first_read + i, column s), the layout of the symbols.
// The generator emits the stream linearly (draw g = read*C + column lives in word g>>2); the quantizer wants, for a
// fixed column, the draws of consecutive reads next to each other. Gathering them per lane from the linear stream
// fetched 8.5x the algorithmic bytes (rocprofv3 FETCH_SIZE, profiles/): each 128-byte line was re-fetched many
// times. This LDS-tiled transpose reads every word once (coalesced) and writes every draw once (dword per lane).
// ------------------------------------------------------------------------------------------------
#define DT_MAX_WORDS 15360   // 60 KB of LDS per block

__global__ __launch_bounds__(256) void k_draws_transpose(const uint32_t *__restrict__ words, uint64_t word0, uint32_t pad, uint64_t draw0,
                                                         uint64_t n, uint32_t C, uint32_t TR, uint8_t *__restrict__ draws,
                                                         uint64_t ld) {
    extern __shared__ uint32_t w[];                              // the tile's words: TR * C / 4 + 2 (the launcher sizes it)
    const uint64_t i0 = (uint64_t)blockIdx.x * TR;               // first read of this tile (multiple of 4)
    const uint32_t reads = (uint32_t)(n - i0 < TR ? n - i0 : TR);
    const uint64_t g0 = draw0 + i0 * C, g1 = g0 + (uint64_t)reads * C;     // draws [g0, g1) of the stream
    const uint64_t wa = g0 >> 2, wb = (g1 + 3) >> 2;
    for (uint32_t i = threadIdx.x; i < (uint32_t)(wb - wa); i += 256) w[i] = words[well_phys(wa - word0 + i, pad)];   // the generator's chunks are `pad` words apart
    __syncthreads();
    const uint32_t quads = (reads + 3) / 4;                      // a column of the tile = `quads` dwords (TR / 4: a power of two or not)
    const uint32_t base = (uint32_t)(g0 - (wa << 2));             // offset of the tile's first draw inside w[]
    for (uint32_t e = threadIdx.x; e < C * quads; e += 256) {    // consecutive threads: consecutive read quads of a column
        const uint32_t s = e / quads, q = e - s * quads;
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t rd = 4 * q + r;
            if (rd < reads) {
                const uint32_t d = base + rd * C + s;
                packed |= ((w[d >> 2] >> (7 * (d & 3))) & 127u) << (8 * r);   // well_1024a_bits, well.c:33-46
            }
        }
        *reinterpret_cast<uint32_t *>(draws + (uint64_t)s * ld + i0 + 4 * q) = packed;
    }
}

// Generates the draws of reads [first_read, first_read + n) in chunks: words into a reusable workspace
// (qvz_hip_well_words), then the transpose above.
extern "C" int qvz_hip_well_draws(const uint32_t *h_state, uint64_t first_read, uint64_t n, uint32_t c, uint8_t *d_draws,
                                  uint64_t ld, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ScratchLock lk(st);
    if (lk.rc) return lk.rc;
    uint32_t *&d_ws = lk.d->d_ws;
    uint64_t &ws_words = lk.d->ws_words;
    uint32_t TR = (DT_MAX_WORDS * 4 - 8) / c;                    // reads per tile: TR*c draws fit the LDS buffer
    TR &= ~3u;
    if (TR > 256) TR = 256;
    if (TR < 4) return qvz_set_error(QVZ_E_ARG, "c=%u too wide for the draw transposer", c);
    const size_t tile_lds = ((size_t)TR * c / 4 + 4) * sizeof(uint32_t);   // what a tile needs, not the 60 KB the widest may: more tiles per CU
    // Reads per generator pass. A generator thread runs 8192 words (218 reads at C = 150) serially, so a pass of 2^22 reads
    // is only ~300 waves - fewer than the chip has SIMDs - and each of them a dependent chain: the passes took 0.5 ms each
    // for 0.6 GB of words. 2^24 reads per pass (2.5 GB of words at C = 150) fill every SIMD (2^25 .. 2^27: the same 15.3 ms per
    // 100M x 150 draws, for 5 .. 15 GB of buffer).
    // A pass begins on a chunk boundary of the generator (its reads x C draws are a multiple of 4 * WELL_CHUNK) and on a tile
    // boundary of the transposer, so the chunk states of the WHOLE call are made once (one doubling ladder of ~19 launches
    // instead of one of ~17 per pass) and every pass starts from its own.
    uint64_t per_chunk = 4ull * WELL_CHUNK;                          // reads between two chunk boundaries that are read boundaries
    { uint64_t a = c, b = per_chunk; while (b) { const uint64_t t = a % b; a = b; b = t; } per_chunk /= a; }
    uint64_t unit = per_chunk;                                       // lcm(TR, per_chunk)
    { uint64_t a = TR, b = per_chunk; while (b) { const uint64_t t = a % b; a = b; b = t; } unit = TR / a * per_chunk; }
    const uint64_t chunk = 1ull << 24;
    const uint64_t chunk_reads = chunk / unit ? (chunk / unit) * unit : unit;
    const uint64_t need_lin = (n < chunk_reads ? ((n + TR - 1) / TR) * TR : chunk_reads) * c / 4 + 8;
    const uint64_t need = well_phys(need_lin, WELL_PAD) + WELL_PAD + 8;                        // the chunks lie WELL_PAD words apart
    if (need > ws_words) {
        if (d_ws) { QVZ_HIP_CHECK(hipStreamSynchronize(st)); (void)hipFree(d_ws); d_ws = nullptr; ws_words = 0; }
        QVZ_HIP_CHECK(hipMalloc(&d_ws, need * sizeof(uint32_t)));
        ws_words = need;
    }
    const uint64_t W0 = (first_read * c) >> 2, W1 = ((first_read + n) * c + 3) >> 2;
    rc = well_states_locked(*lk.d, h_state, W0, (W1 - W0 + WELL_CHUNK - 1) / WELL_CHUNK, st);
    if (rc) return rc;
    for (uint64_t r0 = 0; r0 < n; r0 += chunk_reads) {
        const uint64_t m = n - r0 < chunk_reads ? n - r0 : chunk_reads;
        const uint64_t g0 = (first_read + r0) * c, g1 = g0 + m * c;
        const uint64_t w0 = g0 >> 2, w1 = (g1 + 3) >> 2;             // w0 - W0 = r0 * c / 4: a multiple of WELL_CHUNK
        rc = well_gen_locked(*lk.d, (w0 - W0) >> WELL_CHUNK_LOG2, w1 - w0, d_ws, st, WELL_PAD);
        if (rc) return rc;
        dim3 grid((unsigned)((m + TR - 1) / TR));
        hipLaunchKernelGGL(k_draws_transpose, grid, dim3(256), tile_lds, st, d_ws, w0, WELL_PAD, g0, m, c, TR, d_draws + r0, ld);
        QVZ_HIP_CHECK(hipGetLastError());
    }
    return QVZ_OK;
}

// ------------------------------------------------------------------------------------------------
// Quantizer (reference src/qv_compressor.c:76-135 + choose_quantizer src/codebook.c:162-171).
//
// One thread owns 4 consecutive reads and walks the columns once; the 4 reads are independent first-order
// chains (previous quantized value selects the context), interleaved for latency hiding. Per column a lane
// loads one dword of symbols and stores one dword each of quantized values / quantizer indices / states.
// The WELL draws arrive in the same column-major layout as the symbols (qvz_hip_well_draws), so they are one more
// coalesced dword per lane and column. Table look-ups (ct, qs) are 2-byte gathers that hit L1/L2: only the live
// contexts of a column are ever touched. The 72x72 distortion table sits in LDS; the per-read error is accumulated in column order so it
// is bit-identical to the reference's `error` (qv_compressor.c:97,118).
// ------------------------------------------------------------------------------------------------
// Which quantizer kernel serves a call is decided on the device: k_quantize_lds wants d(x, y) = f(|x - y|) (all of the
// reference's -d choices; its LDS holds 72 values instead of the matrix), a general matrix (-D FILE) needs k_quantize.
// Looking at the matrix on the host meant a 41 KB copy back and a stream synchronisation in every call. Now a one-block
// kernel looks at it and leaves the route in the second word of the caller's status pair; both quantizer kernels are
// launched and the one that is not meant returns at once.
#define QZ_ROUTE_LDS 1
#define QZ_ROUTE_GENERAL 2
// route[1 + l * C + s] (k_quantize_lds reads it as route[2 + ...]): the quantizers column s of cluster l defines - two per context,
// the contexts of a column are numbered from 0 (`ct`: context index | ratio << 8, QVZ_NOT_FOUND where the previous value is no
// context) - so that a workgroup copies those rows of the column's 144 and no others (a -f codebook has a few dozen).
__global__ __launch_bounds__(256) void k_quantize_route(const double *__restrict__ dist, int32_t *__restrict__ route, int lds_possible,
                                                        const uint16_t *__restrict__ ct, uint32_t K, uint32_t C) {
    __shared__ uint32_t nctx[3 * 1024];                         // (the LDS kernel serves three clusters at most, a read 1022 columns)
    if (lds_possible && K * C <= 3 * 1024) {
        for (uint32_t e = threadIdx.x; e < K * C; e += 256) nctx[e] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < K * C * QVZ_CT_STRIDE; i += 256) {     // consecutive threads, consecutive entries
            const uint32_t v = ct[i] & 0xFFu;
            if (v != QVZ_NOT_FOUND) atomicMax(&nctx[i / QVZ_CT_STRIDE], v + 1);
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < K * C; e += 256) route[1 + e] = (int32_t)(2 * (nctx[e] > QVZ_A ? QVZ_A : nctx[e]));
    }
    __shared__ int general;
    if (threadIdx.x == 0) general = lds_possible ? 0 : 1;
    __syncthreads();
    if (lds_possible)
        for (uint32_t e = threadIdx.x; e < QVZ_A * QVZ_A; e += 256) {
            const uint32_t x = e % QVZ_A, y = e / QVZ_A;
            // bit patterns, as the host compared them: -0.0 / NaN entries of a custom file are "different"
            if (__double_as_longlong(dist[e]) != __double_as_longlong(dist[x > y ? x - y : y - x])) general = 1;
        }
    __syncthreads();
    if (threadIdx.x == 0) *route = general ? QZ_ROUTE_GENERAL : QZ_ROUTE_LDS;
}

__global__ __launch_bounds__(256) void k_quantize(const uint8_t *__restrict__ D, const uint8_t *__restrict__ labels,
                                                  uint64_t n, uint32_t C, uint64_t ld, const uint16_t *__restrict__ ct,
                                                  const uint16_t *__restrict__ qs, const uint8_t *__restrict__ draws,
                                                  uint64_t ldd, const double *__restrict__ dist,
                                                  uint8_t *__restrict__ qv_out, uint8_t *__restrict__ qi_out,
                                                  uint8_t *__restrict__ st_out, uint64_t ldo, double *__restrict__ err_out,
                                                  int32_t *__restrict__ status, const int32_t *__restrict__ route) {
    __shared__ double sdist[QVZ_A * QVZ_A];
    if (route[1] == QZ_ROUTE_LDS) return;                       // the other kernel's job (qvz_hip_quantize: routing without a host round trip)
    for (uint32_t i = threadIdx.x; i < QVZ_A * QVZ_A; i += 256) sdist[i] = dist[i];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t i0 = 4 * t;
    if (i0 >= n) return;
    uint32_t lab4 = 0;
    if (labels) lab4 = *reinterpret_cast<const uint32_t *>(labels + i0);
    const uint16_t *ctp[4], *qsp[4];
    uint32_t prev[4] = {0, 0, 0, 0};
    double err[4] = {0.0, 0.0, 0.0, 0.0};
    bool dead = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t l = (lab4 >> (8 * r)) & 0xFF;
        ctp[r] = ct + (uint64_t)l * C * QVZ_CT_STRIDE;
        qsp[r] = qs + (uint64_t)l * C * QVZ_QS_STRIDE;
    }
    for (uint32_t s = 0; s < C; ++s) {
        const uint32_t x4 = *reinterpret_cast<const uint32_t *>(D + (uint64_t)s * ld + i0);
        const uint32_t d4 = *reinterpret_cast<const uint32_t *>(draws + (uint64_t)s * ldd + i0);   // 7-bit draws, same layout
        uint32_t qv4 = 0, qi4 = 0, st4 = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t draw = (d4 >> (8 * r)) & 127;                                 // well.c:33-46
            uint32_t x = (x4 >> (8 * r)) & 0xFF;
            if (x >= QVZ_A) x = 0;
            const uint32_t cv = ctp[r][(uint64_t)s * QVZ_CT_STRIDE + prev[r]];
            uint32_t ctx = cv & 0xFF;
            if (ctx == QVZ_NOT_FOUND) { dead = dead || (i0 + r < n); ctx = 0; }      // assert, codebook.c:164
            const uint32_t qi = 2 * ctx + (draw >= (cv >> 8) ? 1u : 0u);                // codebook.c:165-170
            const uint32_t qsv = qsp[r][(uint64_t)s * QVZ_QS_STRIDE + qi * QVZ_A + x];
            const uint32_t qv = qsv & 0xFF;
            const double dd = sdist[x + QVZ_A * qv];                                     // distortion.c:151
            err[r] = (s == 0) ? dd : err[r] + dd;
            prev[r] = qv;
            qv4 |= qv << (8 * r);
            qi4 |= qi << (8 * r);
            st4 |= (qsv >> 8) << (8 * r);
        }
        if (qv_out) *reinterpret_cast<uint32_t *>(qv_out + (uint64_t)s * ldo + i0) = qv4;
        if (qi_out) *reinterpret_cast<uint32_t *>(qi_out + (uint64_t)s * ldo + i0) = qi4;
        if (st_out) *reinterpret_cast<uint32_t *>(st_out + (uint64_t)s * ldo + i0) = st4;
    }
    if (err_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i0 + r < n) err_out[i0 + r] = err[r];
    }
    if (dead) atomicOr(status, 1);
}

// The same quantizer with the current column's tables in LDS. k_quantize above is bound by its two 2-byte gathers per
// symbol: 64 lanes, 64 addresses - the texture path takes ~47 cycles per such instruction (measured: 0.68 symbols
// per cycle and CU, 26 % of the HBM peak). Here a 512-thread workgroup (2048 reads) copies column s+1's tables
// (ct 144 B + qs 20 736 B per cluster, 16-byte loads that hit L2) into the other half of a double buffer while it
// quantizes column s out of LDS (ds_read_u16 gathers), one barrier per column; two or three workgroups share a CU (eight or four reads
// per thread) so that one's barrier wait is another's work. The copy costs 10 B per symbol of L2 traffic. It needs a distortion matrix
// that depends on |x - y| only (all of the reference's -d choices, distortion.c:50-95): 72 doubles in LDS instead of
// 72 x 72. With three clusters one buffer and two barriers per column; otherwise (more clusters, or a general
// matrix) k_quantize.
#define QZ_THREADS 512
// symbols and draws are read once here: with the non-temporal hint they leave the caches to the column tables every workgroup
// copies (k = 1: 20.1 -> 19.7 ms, k = 3: 24.9 -> 24.3 ms at 100M x 150; the same hint on the output planes' STORES: 13 % slower)
#define QZ_LD(p) __builtin_nontemporal_load(p)
#define QZ_CLUSTER4 (10 + QVZ_QS_STRIDE / 8)       // uint4 per cluster and column: ct (9) + pad (1) + qs (1296)
template <int NBUF, int NQ, int RW, int T>   // RW dwords = 4 * RW consecutive reads per thread, T threads per workgroup
__global__ __launch_bounds__(T) void k_quantize_lds(const uint8_t *__restrict__ D, const uint8_t *__restrict__ labels,
                                                             uint64_t n, uint32_t C, uint64_t ld, uint32_t K,
                                                             const uint16_t *__restrict__ ct, const uint16_t *__restrict__ qs,
                                                             const uint8_t *__restrict__ draws, uint64_t ldd,
                                                             const double *__restrict__ dist, uint8_t *__restrict__ qv_out,
                                                             uint8_t *__restrict__ qi_out, uint8_t *__restrict__ st_out,
                                                             uint64_t ldo, double *__restrict__ err_out,
                                                             int32_t *__restrict__ status, const int32_t *__restrict__ route) {
    extern __shared__ uint4 qz_smem[];
    if (route[1] != QZ_ROUTE_LDS) return;                       // a general distortion matrix: k_quantize does this call
    double *sdist = reinterpret_cast<double *>(qz_smem);         // d(|x - y|), from row 0 of the matrix
    uint4 *tbuf = qz_smem + QVZ_A / 2;
    const uint32_t per_col4 = K * QZ_CLUSTER4;                   // uint4 per column (all clusters)
    if (threadIdx.x < QVZ_A) sdist[threadIdx.x] = dist[threadIdx.x];
    const uint64_t t = (uint64_t)blockIdx.x * T + threadIdx.x;
    constexpr int R = 4 * RW;
    const uint64_t i0 = (uint64_t)R * t;                         // this thread's R consecutive reads
    const bool active = i0 < n;                                  // idle threads still copy tables and meet the barriers
    uint32_t cbase[R], prev[R];                                  // cluster's table base in the column buffer, in u16
    double err[R];
    bool dead = false;
    // Reads past n (the last thread's, and every read of an idle thread) are quantized as rows of symbol 0 - a walk through
    // contexts that exist whatever the codebook, like any row of real symbols - so that "a context the codebook does not
    // define" (assert, codebook.c:164) needs no test per read and column: one running maximum of the context byte per
    // thread, looked at once at the end. (Per symbol: a compare and two selects less; with |x - y| as one v_sad_u16 and the
    // symbol clamp as one v_min the kernel has 394 instead of 478 VALU instructions: 20.0 -> 19.1 ms with one cluster,
    // 24.7 -> 23.4 with three, at 100M x 150.)
    uint32_t xmask[RW], ctxmax = 0;
#pragma unroll
    for (int w = 0; w < RW; ++w) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) m |= active && i0 + 4 * w + b < n ? 0xFFu << (8 * b) : 0u;
        xmask[w] = m;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t l = 0;
        if (labels && active && i0 + r < ld) l = labels[i0 + r];
        // (NQ == 3 is the one-cluster instantiation: a base the compiler knows to be 0 is one address instruction less per symbol)
        cbase[r] = NQ == 3 ? 0u : (l < K ? l : 0u) * (QZ_CLUSTER4 * 8);
        dead = dead || (active && i0 + r < n && l >= K);
        prev[r] = 0;
        err[r] = 0.0;
    }
    // Column s's tables -> registers -> LDS. The loads are issued by hand (one unconditional 16-byte load per stage register,
    // its address picked by selects) and waited for by hand in park(), a whole column of work later. Left to the compiler the
    // copy was `if (ct part) load .. else if (qs part) load ..`: two exec-mask regions writing the same registers with an
    // s_waitcnt vmcnt(0) between them, so every column paid NQ L2 round trips one after the other (and the wait drained the
    // symbol and draw loads of the next column that had just been issued); as plain C++ loads without the branches it
    // reused the address registers as destinations and waited after each load again. Three clusters (63 KB of tables per
    // column, one workgroup per CU): 26.9 -> 24.7 ms at 100M x 150; one cluster: unchanged (20.0 ms; there the kernel is 70 %
    // VALU-bound: SQ counters in profiles/r03_sq_counters_cfg2.txt).
    typedef uint32_t qz_u32x4 __attribute__((ext_vector_type(4)));
    qz_u32x4 stg[NQ];
    // Of a column's 144 quantizers only those of its contexts are ever looked at (two per context: k_quantize_route counted them):
    // a wave whose part of the tables lies wholly behind them asks for nothing and parks nothing - whole waves, by a scalar branch,
    // so that the loads stay unconditional for the lanes (with a -f codebook's few dozen contexts that is half of a workgroup's
    // copy and more: 10 bytes of L2 traffic per symbol, three loads and three LDS stores per thread and column before).
    // (The 1024-thread instantiation of three clusters copies everything: it has no registers left for this - four spilled -, and
    // with the -r targets it is measured on every context of every column exists: 22.3 -> 24.6 ms with the test in it.)
    constexpr bool PARTIAL = NBUF == 2;
    uint32_t loaded = PARTIAL ? 0u : ~0u;                       // bit q: this wave holds a piece of the column in stg[q] (wave-uniform)
    auto fetch = [&](uint32_t s) {
        uint32_t rows[3] = {~0u, ~0u, ~0u};                     // uint4 of a cluster's tables that are in use: ct, pad, the quantizers' rows
        if (PARTIAL) {
#pragma unroll
            for (uint32_t l = 0; l < 3; ++l) rows[l] = l < K ? 10u + 9u * (uint32_t)route[2 + l * C + s] : 0u;
            loaded = 0;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t i = threadIdx.x + q * T;
            const uint32_t ii = i < per_col4 ? i : 0u;          // threads past the column's tables load entry 0 and park nothing
            const uint32_t l = ii / QZ_CLUSTER4, j = ii % QZ_CLUSTER4;
            if (PARTIAL) {
                const bool need = i < per_col4 && j < (l == 0 ? rows[0] : l == 1 ? rows[1] : rows[2]);
                if (__ballot(need) == 0) continue;              // (the same in every lane of the wave)
                loaded |= 1u << q;
            }
            const uint64_t cs = (uint64_t)l * C + s;
            const uint4 *pct = reinterpret_cast<const uint4 *>(ct + cs * QVZ_CT_STRIDE);
            const uint4 *pqs = reinterpret_cast<const uint4 *>(qs + cs * QVZ_QS_STRIDE);
            const uint4 *src = j < 10 ? pct + (j < 9 ? j : 0u) : pqs + (j - 10);   // (j == 9, the pad between the two tables, gets a copy of entry 0: nobody reads it)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(stg[q]) : "v"(src) : "memory");
        }
    };
    auto park = [&](uint4 *dst) {
        // every load issued so far has landed (the stage registers pass through the wait, so nothing that reads them moves above it)
        if constexpr (NQ == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]) : : "memory");
        else if constexpr (NQ == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]) : : "memory");
        static_assert(NQ == 3 || NQ == 4 || NQ == 6, "stage registers");
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t i = threadIdx.x + q * T;
            if (PARTIAL && !((loaded >> q) & 1u)) continue;     // (wave-uniform)
            if (i < per_col4) dst[i] = make_uint4(stg[q].x, stg[q].y, stg[q].z, stg[q].w);
        }
    };
    fetch(0);
    park(tbuf);
    __syncthreads();
    // the symbols and draws of column s+1 are requested while column s is quantized: the workgroup moves in step
    // (one barrier per column), so an HBM latency left exposed here would be paid 150 times by every wave
    // Addresses: a base that is the same in every lane - the plane's column plus the workgroup's first read, scalar registers and
    // scalar adds per column - and this thread's few KB into the workgroup's reads as a 32-bit offset, instead of a 64-bit vector
    // add per access and column (14 of the loop's 190 vector instructions per eight symbols). Idle threads read the workgroup's
    // first reads (a workgroup exists because its first read does).
    const uint64_t wg0 = (uint64_t)blockIdx.x * T * R;           // first read of this workgroup
    const uint32_t toff = active ? threadIdx.x * R : 0u;
    const uint8_t *const Dw = D + wg0, *const drw = draws + wg0;
    uint32_t x4n[RW], d4n[RW];
#pragma unroll
    for (int w = 0; w < RW; ++w) {
        x4n[w] = QZ_LD(reinterpret_cast<const uint32_t *>(Dw + toff + 4 * w));
        d4n[w] = QZ_LD(reinterpret_cast<const uint32_t *>(drw + toff + 4 * w));      // 7-bit draws, same layout
    }
    for (uint32_t s = 0; s < C; ++s) {
        const uint16_t *cur = reinterpret_cast<const uint16_t *>(tbuf + (NBUF == 2 ? (s & 1) * per_col4 : 0u));
        uint32_t x4[RW], d4[RW], qv4[RW], qi4[RW], st4[RW];
#pragma unroll
        for (int w = 0; w < RW; ++w) { x4[w] = x4n[w] & xmask[w]; d4[w] = d4n[w]; qv4[w] = qi4[w] = st4[w] = 0; }
        if (s + 1 < C) {
            fetch(s + 1);
            // (held in scalar registers by the constraint: left alone the compiler turns every access into a 64-bit vector
            // induction variable of its own)
            // (the OFFSETS pass through the constraint, not the pointers: a pointer out of an asm statement has lost its address
            // space and is loaded from with flat instructions, which count on the LDS counter as well)
            uint64_t doff = wg0 + (uint64_t)(s + 1) * ld, roff = wg0 + (uint64_t)(s + 1) * ldd;
            asm volatile("" : "+s"(doff), "+s"(roff));
            const uint8_t *dcol = D + doff, *rcol = draws + roff;
#pragma unroll
            for (int w = 0; w < RW; ++w) {
                x4n[w] = QZ_LD(reinterpret_cast<const uint32_t *>(dcol + toff + 4 * w));
                d4n[w] = QZ_LD(reinterpret_cast<const uint32_t *>(rcol + toff + 4 * w));
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t draw = (d4[r >> 2] >> (8 * (r & 3))) & 127;                   // well.c:33-46
            uint32_t x = (x4[r >> 2] >> (8 * (r & 3))) & 0xFF;
            x = x < QVZ_A ? x : QVZ_A - 1;                                               // (a byte that is no symbol: the transposer refuses such input; here it must only stay inside the tables)
            const uint16_t *cb = cur + cbase[r];
            const uint32_t cv = cb[prev[r]];
            const uint32_t cb8 = cv & 0xFF;
            ctxmax = cb8 > ctxmax ? cb8 : ctxmax;                                        // QVZ_NOT_FOUND (0xFF) is the largest value there is
            const uint32_t ctx = cv & 0x7F;                                              // (an undefined context indexes inside the column's buffer: 2 * 127 + 1 quantizers at most)
            const uint32_t qi = 2 * ctx + (draw >= (cv >> 8) ? 1u : 0u);                // codebook.c:165-170
            const uint32_t qsv = cb[80 + qi * QVZ_A + x];
            const uint32_t qv = qsv & 0xFF;
            const uint32_t qvc = qv;                                                     // a table entry is a symbol (k_quantize trusts the tables the same way)
            const double dd = sdist[__builtin_amdgcn_sad_u16(x, qvc, 0u)];               // |x - qv|: distortion.c:151
            err[r] += dd;                                                                // from 0.0, as the reference sums a line
            prev[r] = qvc;
            qv4[r >> 2] |= qv << (8 * (r & 3));
            qi4[r >> 2] |= qi << (8 * (r & 3));
            st4[r >> 2] |= (qsv >> 8) << (8 * (r & 3));
        }
        if (active) {
            uint64_t ooff = wg0 + (uint64_t)s * ldo;
            asm volatile("" : "+s"(ooff));
            uint8_t *qv_col = qv_out + ooff, *qi_col = qi_out + ooff, *st_col = st_out + ooff;
            if (RW == 2) {                                       // a thread's eight reads of a column: one 8-byte store per plane
                // (a caller's plane may begin at any multiple of four reads: the type promises dword alignment only)
                typedef uint32_t qz_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
                qz_u32x2 a, b, c2;
                a.x = qv4[0]; a.y = qv4[RW - 1]; b.x = qi4[0]; b.y = qi4[RW - 1]; c2.x = st4[0]; c2.y = st4[RW - 1];
                if (qv_out) *reinterpret_cast<qz_u32x2 *>(qv_col + toff) = a;
                if (qi_out) *reinterpret_cast<qz_u32x2 *>(qi_col + toff) = b;
                if (st_out) *reinterpret_cast<qz_u32x2 *>(st_col + toff) = c2;
            } else {
#pragma unroll
                for (int w = 0; w < RW; ++w) {
                    if (qv_out) *reinterpret_cast<uint32_t *>(qv_col + toff + 4 * w) = qv4[w];
                    if (qi_out) *reinterpret_cast<uint32_t *>(qi_col + toff + 4 * w) = qi4[w];
                    if (st_out) *reinterpret_cast<uint32_t *>(st_col + toff + 4 * w) = st4[w];
                }
            }
        }
        if (s + 1 < C) {
            if (NBUF == 1) __syncthreads();                      // everybody is done with column s's tables
            park(tbuf + (NBUF == 2 ? ((s + 1) & 1) * per_col4 : 0u));
        }
        __syncthreads();
    }
    if (err_out && active) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (i0 + r < n) err_out[i0 + r] = err[r];
    }
    if (dead || ctxmax == QVZ_NOT_FOUND) atomicOr(status, 1);
}

struct QvzQuantArgs {                 // everything qvz_hip_quantize takes
    const uint8_t *d_sym, *d_labels;
    uint64_t n;
    uint32_t c;
    uint64_t ld;
    uint32_t k;
    const uint16_t *d_ct, *d_qs;
    const uint8_t *d_draws;
    uint64_t ld_draws;
    const double *d_dist;
    uint8_t *d_qv, *d_qi, *d_st;
    uint64_t ld_out;
    double *d_err;
    int32_t *d_status;
};
static int quantize_check(const QvzQuantArgs &a) {
    int rc = check_matrix(a.n, a.c, a.ld);
    if (rc) return rc;
    if ((rc = check_matrix(a.n, a.c, a.ld_draws))) return rc;
    if ((a.d_qv || a.d_qi || a.d_st) && (rc = check_matrix(a.n, a.c, a.ld_out))) return rc;
    if (a.k == 0 || a.k > 255) return qvz_set_error(QVZ_E_ARG, "k=%u out of range 1..255", a.k);
    if (a.k > 1 && !a.d_labels) return qvz_set_error(QVZ_E_ARG, "labels required for k > 1");
    if (!a.d_ct || !a.d_qs || !a.d_draws || !a.d_dist || !a.d_status) return qvz_set_error(QVZ_E_ARG, "null table pointer");
    return QVZ_OK;
}

static int quantize_launch(const QvzQuantArgs &a, int32_t *d_route, hipStream_t stream) {
    const uint64_t n = a.n;
    const uint32_t c = a.c, k = a.k;
    const char *force = getenv("QVZ_QUANTIZE_KERNEL");           // test hook: "global" keeps the tables in L1/L2 (the kernel of more than three clusters)
    const bool lds_possible = k <= 3 && !(force && strcmp(force, "global") == 0);
    // [0] mirrors nothing (the kernels write the caller's d_status); [1] = route of this call
    hipLaunchKernelGGL(k_quantize_route, dim3(1), dim3(256), 0, stream, a.d_dist, d_route + 1, lds_possible ? 1 : 0, a.d_ct, a.k, a.c);
    QVZ_HIP_CHECK(hipGetLastError());
    if (lds_possible) {
        // dwords (4 reads each) per thread: 8 reads = 8 independent chains (22.6 -> 20.4 ms at 100M x 150 with one cluster)
        // Three clusters: a column's tables are 63 KB, copied from L2 by every workgroup for every column - with 512
        // threads x 4 reads that was 459 GB of copies per step at 100M x 150. 1024-thread workgroups with eight reads
        // per thread copy a quarter of that: 43.2 -> 27.5 ms (1024 threads x 4 reads: 39.4; eight reads per thread in
        // 512-thread workgroups had been slower than four).
        const uint64_t tpw = k == 3 ? 1024 : QZ_THREADS, nthreads = (n + 7) / 8;
        dim3 grid((unsigned)((nthreads + tpw - 1) / tpw));
        const size_t col = (size_t)k * QZ_CLUSTER4 * 16, lds = (size_t)QVZ_A * 8 + (k <= 2 ? 2 : 1) * col;
#define QZ_LAUNCH(NBUF, NQ, TT)                                                                                                  \
        do {                                                                                                                     \
            QVZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_quantize_lds<NBUF, NQ, 2, TT>),                   \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
            hipLaunchKernelGGL((k_quantize_lds<NBUF, NQ, 2, TT>), grid, dim3(TT), lds, stream, a.d_sym,                          \
                               k > 1 ? a.d_labels : nullptr, n, c, a.ld, k, a.d_ct, a.d_qs, a.d_draws, a.ld_draws, a.d_dist, a.d_qv, a.d_qi, \
                               a.d_st, a.ld_out, a.d_err, a.d_status, d_route);                                                  \
        } while (0)
        static_assert(3 * QZ_THREADS >= QZ_CLUSTER4 && 6 * QZ_THREADS >= 2 * QZ_CLUSTER4 && 4 * 1024 >= 3 * QZ_CLUSTER4, "stage registers");
        if (k == 1) QZ_LAUNCH(2, 3, QZ_THREADS);
        else if (k == 2) QZ_LAUNCH(2, 6, QZ_THREADS);
        else QZ_LAUNCH(1, 4, 1024);
#undef QZ_LAUNCH
    }
    dim3 grid((unsigned)(((n + 3) / 4 + 255) / 256));
    hipLaunchKernelGGL(k_quantize, grid, dim3(256), 0, stream, a.d_sym, k > 1 ? a.d_labels : nullptr, n, c, a.ld,
                       a.d_ct, a.d_qs, a.d_draws, a.ld_draws, a.d_dist, a.d_qv, a.d_qi, a.d_st, a.ld_out, a.d_err, a.d_status, d_route);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

extern "C" int qvz_hip_quantize(const uint8_t *d_sym, const uint8_t *d_labels, uint64_t n, uint32_t c, uint64_t ld,
                                uint32_t k, const uint16_t *d_ct, const uint16_t *d_qs, const uint8_t *d_draws,
                                uint64_t ld_draws, const double *d_dist, uint8_t *d_qv, uint8_t *d_qi, uint8_t *d_st,
                                uint64_t ld_out, double *d_err, int32_t *d_status, void *stream) {
    const QvzQuantArgs a = {d_sym, d_labels, n, c, ld, k, d_ct, d_qs, d_draws, ld_draws, d_dist, d_qv, d_qi, d_st, ld_out, d_err, d_status};
    const int rc = quantize_check(a);
    if (rc) return rc;
    // the route word lives behind the caller's status word in this device's scratch pair (stream-ordered like the rest of it)
    ScratchLock lk((hipStream_t)stream);
    if (lk.rc) return lk.rc;
    if (!lk.d->d_route) QVZ_HIP_CHECK(hipMalloc(&lk.d->d_route, (2 + 3 * 1024) * sizeof(int32_t)));   // route word + the quantizer counts of up to 3 x 1022 columns
    return quantize_launch(a, lk.d->d_route, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Coder feed: (qi, st) column-major -> read-major u16 pairs, through the same 256 x 64 LDS tile as the text
// transposer (two byte planes).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_feed(const uint8_t *__restrict__ QI, const uint8_t *__restrict__ ST,
                                                   uint64_t n, uint32_t C, uint64_t ld, uint16_t *__restrict__ feed) {
    __shared__ uint8_t tq[TT_COLS * TT_PITCH];
    __shared__ uint8_t ts[TT_COLS * TT_PITCH];
    const uint64_t row0 = (uint64_t)blockIdx.x * TT_ROWS;
    const uint32_t col0 = blockIdx.y * TT_COLS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < TT_COLS * (TT_ROWS / 4); i += 256) {
        const uint32_t cc = i >> 6, dw = i & 63;
        uint32_t a = 0, b = 0;
        if (col0 + cc < C && row0 + 4 * dw < ld) {
            a = *reinterpret_cast<const uint32_t *>(QI + (uint64_t)(col0 + cc) * ld + row0 + 4 * dw);
            b = *reinterpret_cast<const uint32_t *>(ST + (uint64_t)(col0 + cc) * ld + row0 + 4 * dw);
        }
        *reinterpret_cast<uint32_t *>(&tq[cc * TT_PITCH + 4 * dw]) = a;
        *reinterpret_cast<uint32_t *>(&ts[cc * TT_PITCH + 4 * dw]) = b;
    }
    __syncthreads();
    const uint32_t col = col0 + lane;
    for (uint32_t rr = wave; rr < TT_ROWS; rr += 4) {
        const uint64_t row = row0 + rr;
        if (row < n && col < C)
            feed[row * C + col] = (uint16_t)(tq[lane * TT_PITCH + rr] | (ts[lane * TT_PITCH + rr] << 8));
    }
}

extern "C" int qvz_hip_pack_feed(const uint8_t *d_qi, const uint8_t *d_st, uint64_t n, uint32_t c, uint64_t ld,
                                 uint16_t *d_feed, void *stream) {
    int rc = check_matrix(n, c, ld);
    if (rc) return rc;
    dim3 grid((unsigned)((n + TT_ROWS - 1) / TT_ROWS), (c + TT_COLS - 1) / TT_COLS);
    hipLaunchKernelGGL(k_pack_feed, grid, dim3(256), 0, (hipStream_t)stream, d_qi, d_st, n, c, ld, d_feed);
    QVZ_HIP_CHECK(hipGetLastError());
    return QVZ_OK;
}

// ------------------------------------------------------------------------------------------------
// memory / stream plumbing for host C callers
// ------------------------------------------------------------------------------------------------
// ---- streams out of the coder's fixed-stride area into one packed run (what goes to the host / into a shard set) ----
// SCATTER: the inverse (packed run -> fixed-stride area), for the decoder
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_gather_streams(const uint8_t *__restrict__ in, uint64_t cap, const uint64_t *__restrict__ sizes,
                                                        const uint64_t *__restrict__ at, uint8_t *__restrict__ out) {
    const uint32_t g = blockIdx.x;
    const uint64_t sz = sizes[g];
    const uint8_t *src = SCATTER ? in + at[g] : in + (uint64_t)g * cap;     // the fixed-stride side is 4-byte aligned (cap is a multiple of 4)
    uint8_t *dst = SCATTER ? out + (uint64_t)g * cap : out + at[g];         // the packed side has any alignment
    for (uint64_t i = (uint64_t)blockIdx.y * 256 + threadIdx.x; i < sz; i += (uint64_t)gridDim.y * 256) dst[i] = src[i];
}

static int move_streams(bool scatter, const uint8_t *d_in, uint64_t cap_per_shard, const uint64_t *h_sizes, uint32_t shards,
                        uint8_t *d_out, void *stream) {
    if (!d_in || !h_sizes || !shards || !d_out) return qvz_set_error(QVZ_E_ARG, "bad gather / scatter arguments");
    hipStream_t st = (hipStream_t)stream;
    uint64_t *h = (uint64_t *)malloc((size_t)shards * 2 * sizeof(uint64_t)), *d = nullptr;
    if (!h) return qvz_set_error(QVZ_E_NOMEM, "host alloc");
    uint64_t at = 0, longest = 0;
    for (uint32_t g = 0; g < shards; ++g) {
        if (h_sizes[g] > cap_per_shard) { free(h); return qvz_set_error(QVZ_E_ARG, "stream %u longer than cap_per_shard", g); }
        h[g] = h_sizes[g]; h[shards + g] = at; at += h_sizes[g];
        if (h_sizes[g] > longest) longest = h_sizes[g];
    }
    hipError_t e = hipMalloc(&d, (size_t)shards * 2 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d, h, (size_t)shards * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && scatter) e = hipMemsetAsync(d_out, 0, (size_t)shards * cap_per_shard, st);   // a decoder looks a few bytes past a stream's end
    if (e == hipSuccess) {
        const unsigned ny = (unsigned)((longest + 16383) / 16384);       // 64 bytes per thread at most
        if (scatter) hipLaunchKernelGGL(k_gather_streams<true>, dim3(shards, ny ? ny : 1), dim3(256), 0, st, d_in, cap_per_shard, d, d + shards, d_out);
        else hipLaunchKernelGGL(k_gather_streams<false>, dim3(shards, ny ? ny : 1), dim3(256), 0, st, d_in, cap_per_shard, d, d + shards, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);            // h and d go away below
    free(h);
    (void)hipFree(d);
    if (e != hipSuccess) return qvz_set_error(QVZ_E_HIP, "gather / scatter of streams: %s", hipGetErrorString(e));
    return QVZ_OK;
}

extern "C" int qvz_hip_gather_streams(const uint8_t *d_coded, uint64_t cap_per_shard, const uint64_t *h_sizes, uint32_t shards,
                                      uint8_t *d_packed, void *stream) {
    return move_streams(false, d_coded, cap_per_shard, h_sizes, shards, d_packed, stream);
}

extern "C" int qvz_hip_scatter_streams(const uint8_t *d_packed, const uint64_t *h_sizes, uint32_t shards, uint8_t *d_coded,
                                       uint64_t cap_per_shard, void *stream) {
    return move_streams(true, d_packed, cap_per_shard, h_sizes, shards, d_coded, stream);
}

extern "C" int qvz_hip_set_device(int device) { QVZ_HIP_CHECK(hipSetDevice(device)); return QVZ_OK; }
extern "C" int qvz_hip_malloc(void **d_ptr, size_t bytes) { QVZ_HIP_CHECK(hipMalloc(d_ptr, bytes ? bytes : 1)); return QVZ_OK; }
extern "C" int qvz_hip_free(void *d_ptr) { if (d_ptr) QVZ_HIP_CHECK(hipFree(d_ptr)); return QVZ_OK; }
extern "C" int qvz_hip_malloc_host(void **h_ptr, size_t bytes) { QVZ_HIP_CHECK(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault)); return QVZ_OK; }
extern "C" int qvz_hip_free_host(void *h_ptr) { if (h_ptr) QVZ_HIP_CHECK(hipHostFree(h_ptr)); return QVZ_OK; }
extern "C" int qvz_hip_memset(void *d_ptr, int value, size_t bytes, void *stream) {
    QVZ_HIP_CHECK(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
    return QVZ_OK;
}
extern "C" int qvz_hip_copy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    QVZ_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return QVZ_OK;
}
extern "C" int qvz_hip_copy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    QVZ_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return QVZ_OK;
}
extern "C" int qvz_hip_stream_create(void **stream) {
    hipStream_t s;
    QVZ_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return QVZ_OK;
}
// Do two streams run their kernels side by side? The runtime multiplexes a process's streams onto a handful of hardware
// queues (four by default; RCCL, torch's stream pool and every library in the process draw from the same few), a queue runs
// its kernels in order, and which queue a new stream gets depends on what was created before it. So a stream that is
// meant to run BESIDE another one is tested for it once, when it is made: a one-wave kernel that spins for 200 us of wall
// clock on each, twice; side by side the pair takes ~0.2 ms, one after the other 0.4.
__global__ __launch_bounds__(64) void k_spin_us(uint32_t us) {
    const uint64_t t0 = wall_clock64();                           // 100 MHz
    while (wall_clock64() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(8);
}

extern "C" int qvz_hip_streams_overlap(void *stream_a, void *stream_b, int *overlap) {
    if (!overlap) return qvz_set_error(QVZ_E_ARG, "null argument");
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    *overlap = 0;
    if (a == b) return QVZ_OK;
    QVZ_HIP_CHECK(hipStreamSynchronize(a));
    QVZ_HIP_CHECK(hipStreamSynchronize(b));
    hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, a, 1u);  // (code object load, first-launch costs)
    hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, b, 1u);
    QVZ_HIP_CHECK(hipStreamSynchronize(a));
    QVZ_HIP_CHECK(hipStreamSynchronize(b));
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, a, 200u);
        hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, b, 200u);
        QVZ_HIP_CHECK(hipStreamSynchronize(a));
        QVZ_HIP_CHECK(hipStreamSynchronize(b));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        if (dt < best) best = dt;
    }
    QVZ_HIP_CHECK(hipGetLastError());
    *overlap = best < 330e-6;
    return QVZ_OK;
}

// HIP events around a kernel, on the stream it is launched on: how a host driver in C times one launch (bench.py's roofline figures)
extern "C" int qvz_hip_event_create(void **event) {
    hipEvent_t e;
    QVZ_HIP_CHECK(hipEventCreate(&e));
    *event = (void *)e;
    return QVZ_OK;
}
extern "C" int qvz_hip_event_record(void *event, void *stream) { QVZ_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return QVZ_OK; }
extern "C" int qvz_hip_event_elapsed_ms(void *first, void *second, double *ms) {   // waits for `second`
    float f = 0.f;
    QVZ_HIP_CHECK(hipEventSynchronize((hipEvent_t)second));
    QVZ_HIP_CHECK(hipEventElapsedTime(&f, (hipEvent_t)first, (hipEvent_t)second));
    *ms = (double)f;
    return QVZ_OK;
}
extern "C" int qvz_hip_event_destroy(void *event) { if (event) QVZ_HIP_CHECK(hipEventDestroy((hipEvent_t)event)); return QVZ_OK; }

extern "C" int qvz_hip_stream_destroy(void *stream) { QVZ_HIP_CHECK(hipStreamDestroy((hipStream_t)stream)); return QVZ_OK; }
extern "C" int qvz_hip_stream_sync(void *stream) { QVZ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); return QVZ_OK; }

extern "C" int qvz_hip_mem_info(uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t f = 0, t = 0;
    QVZ_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return QVZ_OK;
}

extern "C" int qvz_hip_stream_wait(void *stream, void *other) {
    hipEvent_t ev;
    QVZ_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)other);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
    (void)hipEventDestroy(ev);                                   // released once the recorded work has completed
    QVZ_HIP_CHECK(e);
    return QVZ_OK;
}
