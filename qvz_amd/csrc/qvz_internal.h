// qvz_internal.h - shared helpers of the HIP translation units (not part of the C-ABI).
#ifndef QVZ_INTERNAL_H
#define QVZ_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_hip.h"

int qvz_set_error(int code, const char *fmt, ...);

#define QVZ_HIP_CHECK(expr)                                                                            \
    do {                                                                                               \
        hipError_t qvz_e_ = (expr);                                                                    \
        if (qvz_e_ != hipSuccess)                                                                      \
            return qvz_set_error(QVZ_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                                 hipGetErrorString(qvz_e_));                                           \
    } while (0)

// Settings of an environment variable that holds a comma-separated list "flag,key=number,...": the test hooks and diagnostics of
// DESIGN.md section 10 (QVZ_CODER, QVZ_PMF, QVZ_CB_LOG2).
static inline bool qvz_env_flag(const char *var, const char *flag) {
    const char *e = getenv(var);
    const size_t n = strlen(flag);
    while (e && *e) {
        const char *end = strchr(e, ',');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        if (len == n && strncmp(e, flag, n) == 0) return true;
        e = end ? end + 1 : nullptr;
    }
    return false;
}
static inline bool qvz_env_number(const char *var, const char *key, double *out) {
    const char *e = getenv(var);
    const size_t n = strlen(key);
    while (e && *e) {
        const char *end = strchr(e, ',');
        if (strncmp(e, key, n) == 0 && e[n] == '=') { *out = atof(e + n + 1); return true; }
        e = end ? end + 1 : nullptr;
    }
    return false;
}

#endif
