// Shared host-side helpers for the emcid HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <initializer_list>

#include "../../include/emcid_hip.h"

namespace emcid {

extern thread_local char g_last_error[512];

inline int fail(int code, const char* fn, const char* what) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", fn, what);
    return code;
}

inline int check_launch(const char* fn) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_last_error, sizeof(g_last_error), "%s: HIP error: %s", fn, hipGetErrorString(e));
        return EMCID_ERR_HIP;
    }
    return EMCID_OK;
}

#define EMCID_CHECK_ARG(cond)                                                                   \
    do {                                                                                        \
        if (!(cond)) return ::emcid::fail(EMCID_ERR_BAD_ARG, __func__, "bad argument: " #cond); \
    } while (0)

#define EMCID_CHECK_LAUNCH()                          \
    do {                                              \
        int rc_ = ::emcid::check_launch(__func__);    \
        if (rc_) return rc_;                          \
    } while (0)

#define EMCID_TRY(call)            \
    do {                           \
        int rc_ = (call);          \
        if (rc_) return rc_;       \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- optional per-kernel-class HIP-event timing (bench.py's live roofline measurement) ------------------
// A ScopedProf around a launch records an event pair on the launch stream when its class is enabled.
enum KernelClass : int {
    KC_PREP = 0, KC_ASSEMBLE = 1, KC_CHOL_LEAF = 2, KC_CHOL_PANEL = 3, KC_CHOL_TRAIL = 4, KC_TRSM_DIAG = 5,
    KC_TRSM_UPDATE = 6, KC_DELTA_W = 7, KC_GRAM = 8, KC_GATHER = 9, KC_DGEMM = 10, KC_MISC = 11, KC_INV_BUILD = 12, KC_CHOL_INNER = 13, KC_INV_APPLY = 14, KC_INV_BLOCK = 15, KC_LINEAR = 16, KC_COUNT = 17
};
void prof_begin(int cls, hipStream_t st);
void prof_end(int cls, hipStream_t st);
struct ScopedProf {
    int cls; hipStream_t st;
    ScopedProf(int c, hipStream_t s) : cls(c), st(s) { prof_begin(cls, st); }
    ~ScopedProf() { prof_end(cls, st); }
};

constexpr int NB = 128;  // Cholesky block size (diagonal leaf)
constexpr int OB = 512;  // outer block of the triangular solves: the inverse of each OB x OB diagonal block of L is formed
constexpr int TB = 256;  // scratch tile of the block-inverse build
inline int64_t inv_doubles(int64_t dp) { return ((dp + OB - 1) / OB) * ((int64_t)OB * OB + (int64_t)TB * TB); }
constexpr int NPAD = 64; // concept-count padding of the f64 K / X / R stacks

}  // namespace emcid
