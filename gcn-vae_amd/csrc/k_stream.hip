// K1 by relation phases, STREAMED: see the comment in front of k_agg_stream.  Built with -fno-slp-vectorize (_build.py): hipcc's
// SLP pass pairs the scalar multiply-adds into v_pk_fma_f32, which needs the accumulators and weights in aligned register pairs
// -- ~30 registers of copies in this kernel (128 + scratch with it, 107-117 without at the same ring depth).
#include <stdlib.h>

#include "k_phase.h"

namespace gv {

// ---- the STREAMED form (round 5) ----------------------------------------------------------------------------------------------
// Same lists, same tiles, same LDS weight slabs, same per-row summation order as k_agg_phase (results are bit-identical), but a
// wave treats its (phase-ordered) edge positions as ONE stream with a static ring of D feature rows in flight:
//   * the gather of position j + D is issued while position j is computed, across phase boundaries too -- the rows of the next
//     phase's first edges are already on their way while the workgroup waits at the barrier and restages the weights (the
//     phase kernel starts every (wave, phase) list with an empty pipeline: 2-3 dependent round trips for ~5 edges);
//   * the ring is D register sets addressed STATICALLY (the stream loop is unrolled D times, phase boundaries are taken inside
//     the unrolled body), so hipcc's waits are counted (vmcnt(D-1)) instead of draining to vmcnt(0) as they do when the
//     landing registers are picked by a run-time index;
//   * the accumulator row is picked by wave-uniform branches on the item slot (scalar compares), the weights of wide blocks are
//     fetched and used in NH chunks of output columns (q-major lane lists), a 5-float lane vector is gathered as 16 + 4 B.
template <int N>
__device__ __forceinline__ void gather_vec(const char* __restrict__ row, unsigned lane_off, float (&d)[N]) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    constexpr int N4 = N / 4, R = N - 4 * N4;
    // scalar base + zero-extended 32-bit lane offset (the saddr form); the offset is made opaque at every use, or hipcc
    // re-associates the sum into a loop-invariant 64-bit per-lane pointer (two more live registers, spilled at 128)
    asm volatile("" : "+v"(lane_off));
    const char* __restrict__ p = row + lane_off;
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const f4u t = *reinterpret_cast<const f4u*>(p + 16 * i);
        d[4 * i] = t.x; d[4 * i + 1] = t.y; d[4 * i + 2] = t.z; d[4 * i + 3] = t.w;
    }
    if constexpr (R >= 2) {
        const f2u t = *reinterpret_cast<const f2u*>(p + 16 * N4);
        d[4 * N4] = t.x; d[4 * N4 + 1] = t.y;
    }
    if constexpr (R & 1) d[N - 1] = *reinterpret_cast<const float*>(p + 4 * (N - 1));
}

// a wave-uniform int read as a SCALAR load: the constant address space is the one hipcc always serves from the scalar cache (a
// plain load of a uniform address stays a vector load -- and a vmcnt(0) -- whenever the kernel also stores; the lists are
// read-only for the whole launch)
__device__ __forceinline__ int sload_i(const int* p) {
    typedef const int __attribute__((address_space(4))) cint;
    return *reinterpret_cast<cint*>(reinterpret_cast<uintptr_t>(p));
}

template <int P, int Q, bool TRANS, int BPL, int K, int D, bool QM, int NH, int MAXT = 1024, int LC = 0>
__global__ __launch_bounds__(MAXT) void k_agg_stream(const PhaseParams a) {
    static_assert(NH == 1 || (BPL == 1 && (TRANS || QM)), "column chunks need one q-major block per lane");
    static_assert(D >= 1 && D <= 32, "ring depth");
    extern __shared__ __attribute__((aligned(16))) float4 wlds[];
    constexpr int GV = BPL * P, PV = BPL * Q, WV = BPL * P * Q, NQ = (WV + 3) / 4;
    constexpr bool QMAJ = TRANS || QM;                  // lane list element (q, p) at q * P + p (else p * Q + q)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    // workgroup -> (tile, column part).  a.xcd_parts (two column parts): workgroups are dealt to the 8 XCDs round-robin, so with
    // part = (id % 8) / 4 an XCD's L2 sees ONE part's weight table and feature columns (half the working set) instead of both
    int tile = blockIdx.x, part = blockIdx.y;
    if (a.xcd_parts) {
        const int id = blockIdx.x;
        part = (id & 7) >> 2;
        tile = (id >> 3) * 4 + (id & 3);
        if (tile >= a.n_tiles) return;
    }
    // LC > 0: the part's lane count at compile time (100 blocks: 50 lanes) -- the LDS offsets of a lane's quads are then
    // instruction immediates instead of one v_add per quad and edge
    const int L = LC > 0 ? LC : a.L;
    const bool active = lane < L;
    const int ln = min(lane, L - 1);
    // a feature row = scalar 64-bit row base + one 32-bit per-lane byte offset: no per-lane pointer is live in the loop
    const char* __restrict__ fbase_s = reinterpret_cast<const char*>(a.feat);
    const unsigned lane_off = (unsigned)((part * a.nbp + ln * BPL) * P) * 4u;
    const unsigned row_bytes = (unsigned)a.ld_feat * 4u;
    const float4* __restrict__ wsrc = a.wpk + (size_t)part * a.R * NQ * L;
    const int rel_quads = NQ * L;
    const int n_phases = a.n_phases;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float4*)wlds);

    auto stage = [&](int p, int b) {
        const int r0 = p * a.G;
        const int nrel = min(a.G, a.R - r0);
        const int total = nrel * rel_quads;
        const float4* src = wsrc + (size_t)r0 * rel_quads;
        const unsigned dst = lds_base + (unsigned)(b * a.slab) * 16u;
        for (int i = wv * 64; i < total; i += nw * 64)
            glds16(src, (unsigned)(i + lane) * 16u, __builtin_amdgcn_readfirstlane(dst + (unsigned)i * 16u));
    };

    float acc[K][PV];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < PV; ++i) acc[k][i] = 0.f;

    const int* __restrict__ offp = a.off + ((size_t)tile * nw + wv) * n_phases;
    const int ws = sload_i(offp), we = sload_i(offp + n_phases);
    // edge metadata: positions [cb, cb + 64) in cur_*, [cb + 64, cb + 128) in nxt_*, both LANDED, and [cb + 128, cb + 192) in
    // flight in far_*.  A batch is read 64 positions after its request: far_* is touched (the point where hipcc places its
    // wait) one refill later, behind 2 D younger gathers -- a counted wait that never drains the ring -- and the loop body
    // reads only landed registers (with two batches the body's read of nxt_* was a vmcnt(1) on the last D positions of
    // every batch: hipcc must assume the request was issued at this iteration's refill)
    int cb = ws;
    int cur_n = 0, cur_m = 0, nxt_n = 0, nxt_m = 0, far_n = 0, far_m = 0;
    float cur_c = 1.f, nxt_c = 1.f, far_c = 1.f;
    auto request = [&](int base, int& n_, int& m_, float& c_) {
        n_ = 0; m_ = 0; c_ = 1.f;
        if (base + lane < we) {
            n_ = a.nbr[base + lane];
            m_ = a.meta[base + lane];
            if (a.coef) c_ = a.coef[base + lane];
        }
    };
    request(cb, cur_n, cur_m, cur_c);
    request(cb + 64, nxt_n, nxt_m, nxt_c);
    asm volatile("" : "+v"(cur_n), "+v"(cur_m), "+v"(cur_c), "+v"(nxt_n), "+v"(nxt_m), "+v"(nxt_c));
    request(cb + 128, far_n, far_m, far_c);
    // the ring: position ws + i sits in set i % D
    float xr[D][GV];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int nb_ = ws + u < we ? prl_i(cur_n, u) : 0;
        gather_vec<GV>(fbase_s + (size_t)(unsigned)nb_ * row_bytes, lane_off, xr[u]);
    }

    int j = ws, p = -1, pend = ws;
    int pnext = sload_i(offp + min(1, n_phases));     // end of phase 0's list
    // LDS addresses use the lane itself, not the clamped one: lanes past the part's L slots then read (and ignore) the next
    // quad row instead of all reading lane L-1's quad -- whose banks collide with another lane of their ds_read_b128
    // group (SQ_LDS_BANK_CONFLICT was 30-38 % of SQ_LDS_IDX_ACTIVE with the clamped address)
    const float4* __restrict__ wcur = wlds + lane;

    // the boundary in front of phase pp: its weights are in LDS and nobody reads the buffer that is refilled
    auto boundary = [&](int pp) {
        const bool bar = !(a.debug & 1), stg = !(a.debug & 2);
        if (a.nbuf == 2) {
            if (pp == 0 && stg) stage(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of phase pp's weights has landed
            if (bar) __syncthreads();                              // everybody's has; phase pp - 1's buffer is free
            if (pp + 1 < n_phases && stg) stage(pp + 1, (pp + 1) & 1);
            wcur = wlds + (pp & 1) * a.slab + lane;
        } else {
            if (bar) __syncthreads();
            if (stg) stage(pp, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (bar) __syncthreads();
        }
    };

    bool more = true;
    while (more) {
        // one refill site per D positions (outside the unrolled body: every copy of the body then sees the metadata in the
        // same registers -- with a refill inside each copy hipcc renamed them from copy to copy, with moves and a vmcnt(0)
        // on the common path): positions [cb, cb + 64) sit in cur_*, [cb + 64, cb + 128) in nxt_*, and j - cb < 64 + D
        if (j >= cb + 64) {
            asm volatile("" : "+v"(far_n), "+v"(far_m), "+v"(far_c));       // requested one refill ago: landed long since
            cur_n = nxt_n; cur_m = nxt_m; cur_c = nxt_c;
            nxt_n = far_n; nxt_m = far_m; nxt_c = far_c;
            cb += 64;
            request(cb + 128, far_n, far_m, far_c);
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {
            while (j >= pend && p + 1 < n_phases) {      // phases up to the one that holds position j (all of them at the end)
                ++p;
                boundary(p);
                pend = pnext;
                pnext = sload_i(offp + min(p + 2, n_phases));
            }
            if (j >= we) {
                more = false;
                break;
            }
            const int jl = j - cb;
            const int m = jl < 64 ? prl_i(cur_m, jl) : prl_i(nxt_m, jl - 64);
            const float cf = jl < 64 ? prl_f(cur_c, jl) : prl_f(nxt_c, jl - 64);
            float (&xc)[GV] = xr[u];                     // used in place: the set is re-requested BEHIND its last use (below)
            const float4* wq = wcur + (m >> 4) * rel_quads;
            const int kk = m & 15;
            if constexpr (NH == 1) {
                // (requesting the NEXT position's weights here, under the loop's scalar work, measured 10-14 % slower on every shape)
                float wr[NQ * 4];
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq) {
                    const float4 t = wq[jq * L];
                    wr[4 * jq] = t.x; wr[4 * jq + 1] = t.y; wr[4 * jq + 2] = t.z; wr[4 * jq + 3] = t.w;
                }
                float tv[PV];
#pragma unroll
                for (int b = 0; b < BPL; ++b)
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        float t = 0.f;
#pragma unroll
                        for (int pp = 0; pp < P; ++pp)
                            t = fmaf(xc[b * P + pp], QMAJ ? wr[b * P * Q + q * P + pp] : wr[b * P * Q + pp * Q + q], t);
                        tv[b * Q + q] = t;
                    }
                // the row's accumulators by a chain of wave-uniform compares: in-place v_fmac on static registers (a switch makes
                // hipcc's structurizer copy the accumulators around every case)
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (kk == k) {
#pragma unroll
                        for (int i = 0; i < PV; ++i) acc[k][i] = fmaf(tv[i], cf, acc[k][i]);
                    }
            } else {
                constexpr int QC = (Q + NH - 1) / NH;    // output columns per chunk
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    constexpr int dummy = 0; (void)dummy;
                    const int q0 = h * QC, q1 = min(Q, q0 + QC);
                    const int e0 = q0 * P, e1 = q1 * P;
                    const int qb = e0 / 4, qe = (e1 + 3) / 4;
                    float wr[((QC * P + 3) / 4 + 1) * 4];
#pragma unroll
                    for (int jq = 0; jq < (QC * P + 3) / 4 + 1; ++jq) {
                        if (qb + jq < qe) {
                            const float4 t = wq[(qb + jq) * L];
                            wr[4 * jq] = t.x; wr[4 * jq + 1] = t.y; wr[4 * jq + 2] = t.z; wr[4 * jq + 3] = t.w;
                        }
                    }
                    float tv[QC];
#pragma unroll
                    for (int q = 0; q < QC; ++q) {
                        float t = 0.f;
                        if (q0 + q < q1) {
#pragma unroll
                            for (int pp = 0; pp < P; ++pp) t = fmaf(xc[pp], wr[(q0 + q) * P + pp - 4 * qb], t);
                        }
                        tv[q] = t;
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        if (kk == k) {
#pragma unroll
                            for (int q = 0; q < QC; ++q)
                                if (q0 + q < q1) acc[k][q0 + q] = fmaf(tv[q], cf, acc[k][q0 + q]);
                        }
                    __builtin_amdgcn_sched_barrier(0);   // the next chunk's fetch stays behind this chunk's use (registers)
                }
            }
            // the set's next row is requested only now: requested in front of the arithmetic hipcc lands it in a fresh register
            // set and rotates the ring by moves behind a vmcnt(0) at the loop's end
            // ... and on EVERY path (past the stream's end: row 0, a cache hit nobody reads), so that each copy of the body has
            // exactly one gather per set in flight and hipcc's wait in front of a set is the counted vmcnt(2 (D - 1))
            __builtin_amdgcn_sched_barrier(0);
            {
                const int il = jl + D;
                int nb_ = il < 64 ? prl_i(cur_n, il) : prl_i(nxt_n, il - 64);
                nb_ = j + D < we ? nb_ : 0;
                gather_vec<GV>(fbase_s + (size_t)(unsigned)nb_ * row_bytes, lane_off, xr[u]);
            }
            ++j;
        }
    }
    phase_epilogue<PV, K>(a, acc, tile, nw, wv, part, lane, active);
}

int launch_phase_stream(const PhaseParams& a_in, const PhasePlan& pl, int blk_in, int blk_out, bool trans, dim3 grid, dim3 block,
                        size_t lds, hipStream_t st) {
    int rc = -1000;
    PhaseParams a = a_in;
    {   // two column parts: a 1-D grid whose workgroup id picks the part by XCD (GV_PHASE_XCD=0: the (tile, part) grid)
        const char* e = getenv("GV_PHASE_XCD");
        if (pl.parts == 2 && !(e && e[0] == '0')) {
            a.xcd_parts = 1;
            grid = dim3((unsigned)((a.n_tiles + 3) / 4) * 8, 1);
        }
    }
    // ring depth: the first instantiation listed for a shape, or the one GV_PHASE_STREAM_D names (tuning knob, tools/phase_bench.py)
    const char* e = getenv("GV_PHASE_STREAM_D");
    const int dsel = e ? atoi(e) : 0;
    bool seen = false;      // an instantiation of this shape was skipped for its depth: fall back to the first listed
    for (int pass = 0; pass < 2 && rc == -1000; ++pass) {
#define GV_STREAM_CASE(P_, Q_, T_, B_, K_, D_, QM_, NH_)                                                                 \
    if (rc == -1000 && blk_in == P_ && blk_out == Q_ && trans == T_ && pl.bpl == B_ && pl.k == K_ &&                     \
        (pl.qmajor != 0) == QM_) {                                                                                       \
        if (pass == 0 && dsel != 0 && dsel != D_) seen = true;                                                           \
        else {                                                                                                           \
            auto kern = pl.lanes == 50 ? k_agg_stream<P_, Q_, T_, B_, K_, D_, QM_, NH_, 1024, 50>                        \
                                       : k_agg_stream<P_, Q_, T_, B_, K_, D_, QM_, NH_, 1024, 0>;                        \
            static unsigned long long lds_done = 0;                                                                      \
            if (!raise_dynamic_lds((const void*)kern, 160 * 1024, lds_done, "gv_rgcn_bdd_aggregate_phases")) return GV_ERR_SHAPE; \
            hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                           \
            rc = launch_status("gv_rgcn_bdd_aggregate_phases");                                                         \
        }                                                                                                                \
    }
    GV_STREAM_CASE(5, 10, false, 1, 4, 2, true, 1) GV_STREAM_CASE(5, 10, false, 1, 4, 6, true, 2)
    GV_STREAM_CASE(10, 5, true, 1, 8, 2, false, 1) GV_STREAM_CASE(10, 5, true, 1, 4, 3, false, 3)
    GV_STREAM_CASE(5, 5, false, 1, 8, 6, false, 1) GV_STREAM_CASE(5, 5, false, 1, 4, 6, false, 1)
    GV_STREAM_CASE(5, 5, true, 1, 8, 6, false, 1) GV_STREAM_CASE(5, 5, true, 1, 4, 6, false, 1)
    GV_STREAM_CASE(4, 2, true, 2, 8, 6, false, 1) GV_STREAM_CASE(4, 2, true, 2, 4, 6, false, 1)
    // (2x2 and 2x4 blocks: the batch-per-list kernel stays ahead -- 2x2: 3 % at 1 M nodes / 50 M edges, 25 % at FB15k-237 size; 2x4 on
    // the bench's 50 M-edge graph 11.0 vs 11.7 ms -- no instances here: those shapes fall through to k_agg_phase)
        if (!seen) break;
    }
#undef GV_STREAM_CASE
    return rc;
}

}  // namespace gv
