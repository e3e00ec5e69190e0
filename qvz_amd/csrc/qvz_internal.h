// qvz_internal.h - shared helpers of the HIP translation units (not part of the C-ABI).
#ifndef QVZ_INTERNAL_H
#define QVZ_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>

#include "qvz_hip.h"

int qvz_set_error(int code, const char *fmt, ...);

#define QVZ_HIP_CHECK(expr)                                                                            \
    do {                                                                                               \
        hipError_t qvz_e_ = (expr);                                                                    \
        if (qvz_e_ != hipSuccess)                                                                      \
            return qvz_set_error(QVZ_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                                 hipGetErrorString(qvz_e_));                                           \
    } while (0)

// ---- the quantizer as the coder's pipeline stage (qvz_kernels.hip; used by qvz_hip_quantize and by qvz_hip_quantize_encode_planes) ----
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
int qvz_quantize_check(const QvzQuantArgs &a);
// Launches the quantizer on `stream`. d_first == nullptr: all n reads. Otherwise one SEGMENT of a sharded encode: the reads
// [first[g] + seg_lo, min(first[g + 1], first[g] + seg_lo + seg_len)) of every stream g < shards (seg_len a multiple of 4; first[g]
// multiples of 4): the coder's model kernel of that segment can start when this launch has ended, whatever the rest of the matrix
// looks like. d_route: two words of device memory of the call's own; decide: look at the distortion matrix and leave the route
// there (the first launch of a call), otherwise the route found by that launch is used.
int qvz_quantize_launch(const QvzQuantArgs &a, const uint64_t *d_first, uint32_t shards, uint64_t seg_lo, uint32_t seg_len,
                        int32_t *d_route, bool decide, hipStream_t stream);

#endif
