// Shared helpers for the gfx950 kernels of libqtmpnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/qtmpnn.h"

void qt_set_error(const char* fmt, ...);

#define QT_ARG(cond, msg)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            qt_set_error("%s: %s", __func__, msg);          \
            return QT_E_ARG;                                \
        }                                                   \
    } while (0)

#define QT_LAUNCHED()                                                   \
    do {                                                                \
        hipError_t qt_e_ = hipGetLastError();                           \
        if (qt_e_ != hipSuccess) {                                      \
            qt_set_error("%s: %s", __func__, hipGetErrorString(qt_e_)); \
            return QT_E_LAUNCH;                                         \
        }                                                               \
    } while (0)

// Valid row count: read on the device when n_dev != NULL (hipGraph-capturable, no host sync), else `cap`.
__device__ __forceinline__ int qt_rows(const int32_t* n_dev, int cap) { return n_dev ? min(*n_dev, cap) : cap; }

static inline int qt_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Workgroup barrier for phases that exchange data through LDS ONLY: waits for this wave's LDS operations (lgkmcnt) and not for
// its outstanding global stores / loads (vmcnt).  __syncthreads() drains both: on gfx9 stores count in vmcnt, so a barrier
// behind a burst of row stores waits for their round trip (~1-2 us under load) although nobody reads them in this launch.
// QT_LDS_BARRIER=0 restores __syncthreads() (A/B builds).
#ifndef QT_LDS_BARRIER
#define QT_LDS_BARRIER 1
#endif
__device__ __forceinline__ void qt_lds_barrier() {
#if QT_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// Exclusive scan of one int per thread over a workgroup of up to 1024 threads (blockDim.x a multiple of 64).
// `red` is 16 ints of LDS.  Returns the exclusive prefix; *total gets the block sum.
__device__ __forceinline__ int qt_block_excl_scan(int v, int* red, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int s = red[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// Exclusive scan of one int per thread over a 256-thread workgroup (4 waves of 64).
// `red` is 8 ints of LDS.  Returns the exclusive prefix; *total gets the block sum.
__device__ __forceinline__ int qt_block_excl_scan_256(int v, int* red, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int s = red[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
