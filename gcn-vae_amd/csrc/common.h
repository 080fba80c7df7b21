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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: a process that drives several devices has to raise it on
// each one.  `done` is the call site's own bit mask (bit d = raised on device d); returns false -- with the library's error text
// set -- when the attribute cannot be raised (the launch that follows would fail with a less telling message).
inline bool raise_dynamic_lds(const void* kernel, int bytes, unsigned long long& done, const char* who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) {
        (void)hipGetLastError();
        dev = 0;
    }
    if (done >> dev & 1ull) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot raise the dynamic LDS limit to %d B on device %d", who, bytes, dev);
        return false;
    }
    done |= 1ull << dev;
    return true;
}

// multiprocessor count of the CURRENT device (cached per device)
inline int current_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) {
        (void)hipGetLastError();
        return 256;
    }
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus[dev] = v;
        else {
            (void)hipGetLastError();
            cus[dev] = 256;
        }
    }
    return cus[dev];
}

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

// Ordered sum over `nsl` row slices of a [nsl][ncols] partial matrix for a 1024-thread block laid out as 16 columns x 64
// slice groups: group g adds its contiguous range of slices in order (4 independent chains, fixed combine), then the 64
// group sums of a column are added in group order.  Returns the column total to the threads of group 0 (others: 0);
// `sm` is a 64 x 16 float scratch.  Many small blocks instead of one wide one: the sum is latency-bound.
__device__ __forceinline__ float sum_slices_16x64(const float* __restrict__ part, int ncols, int nsl, int col, float (*sm)[16]) {
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int per = (nsl + 63) / 64;
    const int s0 = grp * per, s1 = min(nsl, s0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < ncols) {
        int s = s0;
        for (; s + 4 <= s1; s += 4) {
            const float v0 = part[(size_t)s * ncols + col], v1 = part[(size_t)(s + 1) * ncols + col];
            const float v2 = part[(size_t)(s + 2) * ncols + col], v3 = part[(size_t)(s + 3) * ncols + col];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; s < s1; ++s) a0 += part[(size_t)s * ncols + col];
    }
    sm[grp][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float tot = 0.f;
    if (grp == 0) {
#pragma unroll 8
        for (int g = 0; g < 64; ++g) tot += sm[g][cl];
    }
    return tot;
}

// Word fill / copy as ordinary kernels.  hipMemsetAsync / hipMemcpyAsync recorded into a hipGraph become memset / memcpy NODES; on
// this stack a replay that follows other runtime work (a device-to-host copy between two replays is enough) ran such a node with
// stale parameters and faulted (tools/sampler_fault_probe2.py).  Launches of our own kernels are recorded as kernel nodes with
// their arguments by value, which replay correctly, so everything the library clears or copies on the stream goes through these.
__global__ __launch_bounds__(256) inline void k_fill_u32(unsigned* __restrict__ p, unsigned v, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) inline void k_copy_u32(unsigned* __restrict__ d, const unsigned* __restrict__ s, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = s[i];
}
// n_bytes must be a multiple of 4 and p 4-byte aligned (every caller clears int / float arrays)
inline hipError_t fill_words(void* p, unsigned word, size_t n_bytes, hipStream_t st) {
    const long long n = (long long)(n_bytes / 4);
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, st, (unsigned*)p, word, n);
    return hipGetLastError();
}
inline hipError_t copy_words(void* d, const void* s, size_t n_bytes, hipStream_t st) {
    const long long n = (long long)(n_bytes / 4);
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(k_copy_u32, dim3(blocks), dim3(256), 0, st, (unsigned*)d, (const unsigned*)s, n);
    return hipGetLastError();
}

__device__ __forceinline__ float apply_act(float v, int act) { return act == GV_ACT_RELU ? fmaxf(v, 0.f) : v; }

// Philox4x32-10 (Salmon et al., SC'11), the one random generator of the library (gv_rng_fill, the batch sampler)
__device__ __forceinline__ uint4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}


}  // namespace gv
