// Shared by the two relation-phase kernels of K1 (k_phase.hip: one batch of rows per list; k_stream.hip: a ring of rows in flight
// across phase boundaries): launch parameters, LDS-DMA staging, the epilogue, the plan of a block shape.
#pragma once
#include "common.h"

namespace gv {

struct PhaseParams {
    const int* off;          // [(n_tiles * nw * n_phases) + 1] first edge position of every (tile, wave, phase) list
    const int* nbr;          // [E] gathered row of each edge, in (tile, wave, phase, slot) order
    const int* meta;         // [E] ((etype - phase*G) << 4) | item slot k
    const float* coef;       // [E] or NULL
    const int4* titems;      // [n_tiles][nw*K] {row (-1: none), slot (-1: final row), 0, 0}
    const float4* wpk;       // lane-packed weights [parts][R][NQ][L] float4 (+ 64 float4 of slack)
    int n_phases, G, R;
    const float* feat;
    int ld_feat;
    const float* addend;
    int ld_add;
    int act;
    const uint8_t* keep;
    float keep_scale;
    float* out;
    int ld_out;
    float* partial;
    int out_dim;
    int nbp;                 // diagonal blocks per column part
    int L;                   // active lanes per part = nbp / BPL
    int slab;                // float4 per LDS buffer (>= G*NQ*L rounded up to 64)
    int nbuf;                // LDS weight buffers: 2 (phase p+1 lands while p is computed) or 1 (twice the relations per phase)
    int parts, n_tiles, xcd_parts;      // column parts; tiles; 1: the streamed kernel's 1-D grid with part = (id % 8) / 4 (two parts)
    int debug;               // GV_PHASE_DEBUG (probes only; results are wrong): 1 no barriers, 2 no weight staging
};

__device__ __forceinline__ int prl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float prl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One LDS-DMA wave-instruction: 64 x 16 B from per-lane global addresses to lds_byte_addr + lane*16.  Issued from inline
// asm on purpose: hipcc does not count it, so its waits for the feature gathers stay COUNTED (vmcnt(U-1)) instead of
// draining to vmcnt(0) as they do beside a builtin LDS-DMA; the gathers are younger than the phase's DMAs, so every such
// wait still covers them, and the phase ends with an explicit vmcnt(0) before the barrier (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const float4* gbase, unsigned byte_off, unsigned lds_byte_addr) {
    // scalar base + 32-bit per-lane byte offset: no 64-bit per-lane pointer stays live across the launch (it was the one
    // value the 128-register kernels spilled)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(gbase), "s"(lds_byte_addr)
                 : "memory");
}

// A register vector that may be indexed by a wave-uniform run-time position: hipcc lowers that to M0-relative register
// addressing (s_set_gpr_idx), so the vector stays in VGPRs (a K-way branch tree over static arrays instead makes the
// structurizer copy registers around and spill).
template <int N> struct AccVec { typedef float type __attribute__((ext_vector_type(N))); };

// the rows a wave owns leave its registers: partial slot of a split row, or addend + activation + dropout mask + store
template <int PV, int K>
__device__ __forceinline__ void phase_epilogue(const PhaseParams& a, float (&acc)[K][PV], int tile, int nw, int wv, int part, int lane,
                                               bool active) {
    if (!active) return;
    const int col0 = part * (a.out_dim / a.parts) + lane * PV;
    const int4* __restrict__ ti = a.titems + ((size_t)tile * nw + wv) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int4 it = ti[k];
        if (it.x < 0) continue;
        float o[PV];
#pragma unroll
        for (int i = 0; i < PV; ++i) o[i] = acc[k][i];
        if (it.y >= 0) {
            if (a.partial) store_vec<PV>(a.partial + (size_t)it.y * a.out_dim + col0, o);
            continue;
        }
        if (a.addend) {
            float ad[PV];
            load_vec<PV>(a.addend + (size_t)it.x * a.ld_add + col0, ad);
#pragma unroll
            for (int i = 0; i < PV; ++i) o[i] += ad[i];
        }
#pragma unroll
        for (int i = 0; i < PV; ++i) o[i] = apply_act(o[i], a.act);
        if (a.keep) {
            const uint8_t* kp = a.keep + (size_t)it.x * a.out_dim + col0;
#pragma unroll
            for (int i = 0; i < PV; ++i) o[i] = kp[i] ? o[i] * a.keep_scale : 0.f;
        }
        store_vec<PV>(a.out + (size_t)it.x * a.ld_out + col0, o);
    }
}

struct PhasePlan { int bpl, parts, lanes, nq, k, u, qmajor, threads; };
bool phase_plan(int nb, int p, int q, bool trans, int k_req, int threads_req, PhasePlan* out);

// k_stream.hip: launches the streamed kernel for the plan's shape; GV_OK / an error code, or -1000 when no instantiation exists
int launch_phase_stream(const PhaseParams& a, const PhasePlan& pl, int blk_in, int blk_out, bool trans, dim3 grid, dim3 block,
                        size_t lds, hipStream_t st);

}  // namespace gv
