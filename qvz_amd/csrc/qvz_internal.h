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

#endif
