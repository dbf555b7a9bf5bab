// Shared helpers for libstem_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/stem_hip.h"

#define STEM_EXPORT extern "C" __attribute__((visibility("default")))

void stem_set_error(const char *fmt, ...);

#define STEM_CHECK_ARG(cond, ...)        \
    do {                                 \
        if (!(cond)) {                   \
            stem_set_error(__VA_ARGS__); \
            return -1;                   \
        }                                \
    } while (0)

// Launch errors are reported without synchronising (hipGetLastError only).
#define STEM_LAUNCH_CHECK(name)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            stem_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- switches ---------------------------------------------------------------------------------------------------------------
// Plan selectors (a tile shape, a split factor: every choice computes the same contraction, differing at most in summation
// order) are integers set through the exported stem_tuning_set() -- tests and sweep tools use them; nothing reads the
// environment per launch.
enum { STEM_TUNE_BX6_TILE = 0, STEM_TUNE_BX6_SPLIT, STEM_TUNE_WG6_SPLIT, STEM_TUNE_ARP_WORKERS, STEM_TUNE_COUNT };
int stem_tuning(int id);
// Switches that CHANGE RESULTS (fewer bf16 products per fp32 product, ablated kernel stages) exist only in a library built
// with -DSTEM_EXPERIMENTS (`make EXPERIMENTS=1` -> libstem_hip_exper.so, for tools/debug): in the shipped library the macro
// below is a constant null pointer and the reduced-precision kernel variants are not even instantiated.
#ifdef STEM_EXPERIMENTS
#include <stdlib.h>
#define STEM_EXPER_ENV(name) getenv(name)
#else
#define STEM_EXPER_ENV(name) (static_cast<const char *>(nullptr))
#endif
