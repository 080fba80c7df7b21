// Shared helpers for the gfx950 kernels of libgcnvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gcnvae.h"

namespace gv {

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return GV_OK;
}

#define GV_REQUIRE(cond, code, ...)   \
    do {                              \
        if (!(cond)) {                \
            gv::set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers -------------------------------------------------------------------------
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

// contiguous load/store of N floats; the widest access the element count permits (callers
// guarantee 16-B row bases, so N%4==0 -> 16-B, N%2==0 -> 8-B aligned accesses).
template <int N>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&d)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            float4 t = reinterpret_cast<const float4*>(p)[i];
            d[4 * i] = t.x; d[4 * i + 1] = t.y; d[4 * i + 2] = t.z; d[4 * i + 3] = t.w;
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            float2 t = reinterpret_cast<const float2*>(p)[i];
            d[2 * i] = t.x; d[2 * i + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = p[i];
    }
}

template <int N>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&s)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i)
            reinterpret_cast<float4*>(p)[i] = make_float4(s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) reinterpret_cast<float2*>(p)[i] = make_float2(s[2 * i], s[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = s[i];
    }
}

__device__ __forceinline__ float apply_act(float v, int act) { return act == GV_ACT_RELU ? fmaxf(v, 0.f) : v; }

}  // namespace gv
