// Common helpers for the dreammat HIP library (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DM_OK 0
#define DM_ERR_ARG (-1)
#define DM_ERR_WORKSPACE (-2)
#define DM_ERR_UNSUPPORTED (-3)

#define DM_HD __host__ __device__ __forceinline__

// hipGetLastError() is per-thread and sticky across libraries: a benign error left behind by another
// component of the process (PyTorch probing a feature, ...) must not be reported as ours, so every
// entry point clears it first (DM_ENTER) and checks only its own launches (DM_LAUNCH_CHECK).
#define DM_ENTER() do { (void)hipGetLastError(); } while (0)

// Launch-check: positive return = hipError_t.
#define DM_LAUNCH_CHECK()                          \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

#define DM_HIP(expr)                               \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline int dm_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
