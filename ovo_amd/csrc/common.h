// Internal helpers shared by the HIP translation units of libovo_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ovo_hip.h"

#define OVO_WAVE 64

void ovo_set_error(const char *fmt, ...);

// optional hipEvent profiler (core.hip): kinds 1 = attention (flops), 2 = point-map tracking pass (bytes), 4..7 = MFMA GEMM tiles 128x128 / 128x64 / 64x128 / 64x64,
// 3 / 0 = the 256x256 / 256x128 ping-pong GEMM (flops), 8 = the weights-resident streaming GEMM (flops)
#define OVO_PROF_KINDS 10
bool ovo_prof_enabled();
void ovo_prof_begin(int kind, double work, hipStream_t s);
void ovo_prof_shape(int a, int b, int c);       // optional: shape of the launch just begun (OVO_PROF_DUMP lines)
void ovo_prof_flags(int flags);            // optional: epilogue / operand variant of the launch just begun (gemm_common.h: gemm_flags), last field of an OVO_PROF_DUMP line
void ovo_prof_bytes(double bytes);            // optional: algorithmic HBM bytes of the launch just begun (operands read once, result written once)
void ovo_prof_end(hipStream_t s);
// after ovo_prof_end: work (and algorithmic bytes) of the launch = per_item * (*device_count, read back on the stream now) + fixed
void ovo_prof_count(const int32_t *device_count, double per_item, double fixed, hipStream_t s);

#define OVO_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            ovo_set_error("%s: %s", __func__, msg);              \
            return OVO_E_ARG;                                    \
        }                                                        \
    } while (0)

#define OVO_CHECK_LAUNCH()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            ovo_set_error("%s: %s", __func__, hipGetErrorString(e__));            \
            return OVO_E_LAUNCH;                                                  \
        }                                                                         \
    } while (0)

#define OVO_HIP(call)                                                             \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            ovo_set_error("%s: %s -> %s", __func__, #call, hipGetErrorString(e__)); \
            return OVO_E_LAUNCH;                                                  \
        }                                                                         \
    } while (0)

// Tuning knobs (OVO_GEMM_*, OVO_ATTN_*: tools/ and tests only) are read from the environment ONCE per process -- a launch does not call
// getenv.  Tools and tests that flip them between launches of one process (tools/gemm_bench.py, tools/attn_bench.py, tests/conftest.py) set
// OVO_KNOBS_DYNAMIC=1 before the first launch: every launch then reads them afresh.
static inline bool ovo_knobs_dynamic() {
    static const bool d = getenv("OVO_KNOBS_DYNAMIC") != nullptr;
    return d;
}

static inline int ovo_grid(int64_t work_items, int block, int cap = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
