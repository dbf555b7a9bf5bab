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
