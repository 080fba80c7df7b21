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
// Cross-lane reductions on the VALU's DPP path (no LDS-crossbar ds_bpermute round trips): two quad permutes, then
// row_half_mirror and row_mirror leave every lane with the sum of its row of 16; the four row sums are read with
// v_readlane and added in row order.  Fixed order -> bitwise reproducible.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

__device__ __forceinline__ float rl_bcast_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// sum over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_QUAD_1032>(v);
    v += dpp_f<DPP_QUAD_2301>(v);
    v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (rl_bcast_f(v, 0) + rl_bcast_f(v, 16)) + (rl_bcast_f(v, 32) + rl_bcast_f(v, 48));
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_1032>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_2301>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(rl_bcast_f(v, 0), rl_bcast_f(v, 16)), fmaxf(rl_bcast_f(v, 32), rl_bcast_f(v, 48)));
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
