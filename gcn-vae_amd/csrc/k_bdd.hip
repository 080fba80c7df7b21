// K1: R-GCN block-diagonal relational aggregation for gfx950 (forward, backward-x, backward-W).
//
// One 64-lane wavefront per work item (= one destination row, or one <=chunk-edge slice of a hub
// row).  Lanes split the feature row: lane l owns BPL consecutive diagonal blocks, i.e. BPL*P
// gathered floats (one or two 16-B loads: a 64-lane row read is a coalesced 1 KiB burst), the
// matching BPL*P*Q block weights of the edge's relation and BPL*Q accumulators kept in registers
// across the whole row -> exactly one store per output row, no atomics.  Edge metadata (neighbour,
// relation, coefficient) is read 64 edges at a time, one edge per lane, and broadcast with
// v_readlane so every gather address is scalar-base + lane-offset.  U edges are kept in flight
// per wave to cover L2 / Infinity-Cache latency (the feature table and the relation weights of
// FB15k-237 are cache resident; the kernel is bound by cache bandwidth, not FLOPs: ~1 flop/byte).
#include <stdlib.h>

#include "common.h"

namespace gv {

struct AggParams {
    const int4* items;
    int n_items;
    const int* nbr;
    const int* etype;
    const float* coef;
    const int* coef_idx;
    const float* feat;
    int ld_feat;
    const float* w;
    int w_row;  // floats per relation row = nb * P * Q
    const float* addend;
    int ld_add;
    int act;
    const uint8_t* keep;
    float keep_scale;
    float* out;
    int ld_out;
    float* partial;
    int out_dim;
    int nb;
    int p, q;  // runtime block sizes (generic kernel)
    int nbp;   // blocks per column part (fast kernels: grid.y parts of nbp blocks each, nbp/BPL <= 64 lanes)
};

__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


template <int P, int Q, bool TRANS, int BPL>
__device__ __forceinline__ void block_fma(const float (&x)[BPL * P], const float (&w)[BPL * P * Q], float c,
                                          float (&acc)[BPL * Q]) {
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float wv = TRANS ? w[b * P * Q + q * P + p] : w[b * P * Q + p * Q + q];
                t = fmaf(x[b * P + p], wv, t);
            }
            acc[b * Q + q] = fmaf(t, c, acc[b * Q + q]);
        }
    }
}

template <int P, int Q, bool TRANS, int BPL, int U>
__global__ __launch_bounds__(256) void k_agg_fast(const AggParams a) {
    constexpr int GV = BPL * P, PV = BPL * Q, WV = BPL * P * Q;
    // weight reuse along runs of one relation: where a block's weights outweigh the bookkeeping (1x1 blocks -- the DistMult-shaped
    // launches -- lose: 45 -> 68 us at FB15k-237 size with the conditional loads in their 8-edge batches)
    constexpr bool RUNS = P * Q >= 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;                 // unused tail entry of an upper-bound-sized item list
    if (it.w >= 0 && !a.partial) return;   // split item without a partial buffer: inconsistent handle, never dereference NULL
    const bool active = lane * BPL < a.nbp;
    const int blk0 = blockIdx.y * a.nbp + lane * BPL;       // first diagonal block this lane owns
    const float* __restrict__ fbase = a.feat + blk0 * P;
    const float* __restrict__ wbase = a.w + blk0 * (P * Q);

    float acc[PV];
#pragma unroll
    for (int i = 0; i < PV; ++i) acc[i] = 0.f;
    float wkeep[WV];                  // the weights of relation r_keep (-1: none), kept in registers while its run lasts
#pragma unroll
    for (int i = 0; i < WV; ++i) wkeep[i] = 0.f;
    int r_keep = -1;

    for (int e0 = it.y; e0 < it.z; e0 += 64) {
        const int cnt = min(64, it.z - e0);
        int my_n = 0, my_t = 0;
        float my_c = 1.f;
        if (lane < cnt) {
            my_n = a.nbr[e0 + lane];
            my_t = a.etype[e0 + lane];
            if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
        }
        if constexpr (RUNS) {
            // Consecutive edges of ONE relation share their weights: the registers of the run's first edge serve the whole run (r_keep /
            // wkeep carry an open run across batches and metadata chunks).  All of it is wave-uniform scalar control; per edge the
            // arithmetic is what it was.  Rows whose edges come sorted by relation (static graphs: ops.GraphIndex.rel_sorted) turn every
            // repeated (row, relation) pair into a skipped weight fetch -- the per-edge weight read is what bounds this kernel.
            const bool more_chunks = e0 + 64 < it.z;
            int j = 0;
            for (; j + U <= cnt; j += U) {
                float xv[U][GV], wv[U][WV], cc[U];
                int rr[U];
    #pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s = rl_i(my_n, j + u);
                    rr[u] = rl_i(my_t, j + u);
                    cc[u] = rl_f(my_c, j + u);
                    if (active) {
                        load_vec<GV>(fbase + (size_t)s * a.ld_feat, xv[u]);
                        if (rr[u] != (u ? rr[u - 1] : r_keep)) load_vec<WV>(wbase + (size_t)rr[u] * a.w_row, wv[u]);
                    }
                }
                int src = -1;
    #pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (rr[u] != (u ? rr[u - 1] : r_keep)) src = u;
                    if (active) {
                        if (src < 0) block_fma<P, Q, TRANS, BPL>(xv[u], wkeep, cc[u], acc);
    #pragma unroll
                        for (int v = 0; v <= u; ++v)
                            if (src == v) block_fma<P, Q, TRANS, BPL>(xv[u], wv[v], cc[u], acc);
                    }
                }
                // does the run go on behind this batch?  (at the end of a metadata chunk: assume so if the item has another one)
                const bool cont = j + U < cnt ? rl_i(my_t, j + U) == rr[U - 1] : more_chunks;
                if (src >= 0) {
                    if (cont) {
    #pragma unroll
                        for (int v = 0; v < U; ++v)
                            if (src == v) {
    #pragma unroll
                                for (int i = 0; i < WV; ++i) wkeep[i] = wv[v][i];
                            }
                        r_keep = rr[U - 1];
                    } else {
                        r_keep = -1;
                    }
                }
            }
            for (; j < cnt; ++j) {
                float xv[GV];
                const int s = rl_i(my_n, j);
                const int r = rl_i(my_t, j);
                const float c = rl_f(my_c, j);
                if (active) {
                    load_vec<GV>(fbase + (size_t)s * a.ld_feat, xv);
                    if (r != r_keep) load_vec<WV>(wbase + (size_t)r * a.w_row, wkeep);
                    block_fma<P, Q, TRANS, BPL>(xv, wkeep, c, acc);
                }
                r_keep = r;
            }
    
        } else {
            int j = 0;
            for (; j + U <= cnt; j += U) {
                float xv[U][GV], wv[U][WV], cc[U];
    #pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s = rl_i(my_n, j + u);
                    const int r = rl_i(my_t, j + u);
                    cc[u] = rl_f(my_c, j + u);
                    if (active) {
                        load_vec<GV>(fbase + (size_t)s * a.ld_feat, xv[u]);
                        load_vec<WV>(wbase + (size_t)r * a.w_row, wv[u]);
                    }
                }
                if (active) {
    #pragma unroll
                    for (int u = 0; u < U; ++u) block_fma<P, Q, TRANS, BPL>(xv[u], wv[u], cc[u], acc);
                }
            }
            for (; j < cnt; ++j) {
                float xv[GV], wv[WV];
                const int s = rl_i(my_n, j);
                const int r = rl_i(my_t, j);
                const float c = rl_f(my_c, j);
                if (active) {
                    load_vec<GV>(fbase + (size_t)s * a.ld_feat, xv);
                    load_vec<WV>(wbase + (size_t)r * a.w_row, wv);
                    block_fma<P, Q, TRANS, BPL>(xv, wv, c, acc);
                }
            }
    
        }
    }
    if (!active) return;
    const int col0 = blk0 * Q;
    const int row = it.x;
    if (it.w >= 0) {                             // a slice of a split (hub) row: its partial slot, summed in order by k_agg_fixup
        store_vec<PV>(a.partial + (size_t)it.w * a.out_dim + col0, acc);
        return;
    }
    if (a.addend) {
        float ad[PV];
        load_vec<PV>(a.addend + (size_t)row * a.ld_add + col0, ad);
#pragma unroll
        for (int i = 0; i < PV; ++i) acc[i] += ad[i];
    }
#pragma unroll
    for (int i = 0; i < PV; ++i) acc[i] = apply_act(acc[i], a.act);
    if (a.keep) {
        const uint8_t* kp = a.keep + (size_t)row * a.out_dim + col0;
#pragma unroll
        for (int i = 0; i < PV; ++i) acc[i] = kp[i] ? acc[i] * a.keep_scale : 0.f;
    }
    store_vec<PV>(a.out + (size_t)row * a.ld_out + col0, acc);
}

// ---- few, large diagonal blocks (BASELINE configs[2]: B = 20 blocks of 10x10 / 10x20) -------------------------------------
// One block per lane would leave 44 of 64 lanes idle and put 100-200 weights into a lane's registers.  Here LPB lanes share a
// block: lane l owns QS = Q_out / LPB consecutive OUTPUT columns of block l / LPB, gathers that block's whole input slice (the
// LPB lanes of a block read the same addresses -- one L1 access) and its P_in x QS weight sub-matrix.  P_in / Q_out are the
// gathered / produced block widths (for the transposed product the stored block is Q_out... see the weight index below).
template <int PI, int QO, bool TRANS, int QS, int U>
__global__ __launch_bounds__(256) void k_agg_split(const AggParams a) {
    constexpr int LPB = QO / QS;
    static_assert(QO % QS == 0, "the output columns of a block are dealt evenly to its lanes");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;
    if (it.w >= 0 && !a.partial) return;
    const bool active = lane < a.nbp * LPB;                       // a.nbp blocks per column part
    const int blk = blockIdx.y * a.nbp + lane / LPB, sub = lane % LPB;
    const float* __restrict__ fbase = a.feat + blk * PI;
    // stored block: (blk_in x blk_out) row-major = PI x QO for the plain product, QO x PI (read transposed) for TRANS
    const float* __restrict__ wbase = a.w + blk * (PI * QO);

    float acc[QS];
#pragma unroll
    for (int i = 0; i < QS; ++i) acc[i] = 0.f;

    auto one_edge = [&](const float (&xv)[PI], const float (&wv)[PI * QS], float c) {
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < PI; ++p) t = fmaf(xv[p], wv[TRANS ? q * PI + p : p * QS + q], t);
            acc[q] = fmaf(t, c, acc[q]);
        }
    };
    auto load_w = [&](int r, float (&wv)[PI * QS]) {
        const float* wr = wbase + (size_t)r * a.w_row;
        if constexpr (TRANS) {       // rows sub*QS .. sub*QS+QS-1 of the stored QO x PI block: QS*PI contiguous floats
            load_vec<PI * QS>(wr + sub * QS * PI, wv);
        } else {                     // columns sub*QS .. of every row of the stored PI x QO block: PI pieces of QS floats
#pragma unroll
            for (int p = 0; p < PI; ++p) {
                float t[QS];
                load_vec<QS>(wr + p * QO + sub * QS, t);
#pragma unroll
                for (int q = 0; q < QS; ++q) wv[p * QS + q] = t[q];
            }
        }
    };

    for (int e0 = it.y; e0 < it.z; e0 += 64) {
        const int cnt = min(64, it.z - e0);
        int my_n = 0, my_t = 0;
        float my_c = 1.f;
        if (lane < cnt) {
            my_n = a.nbr[e0 + lane];
            my_t = a.etype[e0 + lane];
            if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
        }
        int j = 0;
        for (; j + U <= cnt; j += U) {
            float xv[U][PI], wv[U][PI * QS], cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = rl_i(my_n, j + u);
                const int r = rl_i(my_t, j + u);
                cc[u] = rl_f(my_c, j + u);
                if (active) {
                    load_vec<PI>(fbase + (size_t)s * a.ld_feat, xv[u]);
                    load_w(r, wv[u]);
                }
            }
            if (active) {
#pragma unroll
                for (int u = 0; u < U; ++u) one_edge(xv[u], wv[u], cc[u]);
            }
        }
        for (; j < cnt; ++j) {
            float xv[PI], wv[PI * QS];
            const int s = rl_i(my_n, j);
            const int r = rl_i(my_t, j);
            const float c = rl_f(my_c, j);
            if (active) {
                load_vec<PI>(fbase + (size_t)s * a.ld_feat, xv);
                load_w(r, wv);
                one_edge(xv, wv, c);
            }
        }
    }
    if (!active) return;
    const int col0 = blk * QO + sub * QS;
    const int row = it.x;
    if (it.w >= 0) {
        store_vec<QS>(a.partial + (size_t)it.w * a.out_dim + col0, acc);
        return;
    }
    if (a.addend) {
        float ad[QS];
        load_vec<QS>(a.addend + (size_t)row * a.ld_add + col0, ad);
#pragma unroll
        for (int i = 0; i < QS; ++i) acc[i] += ad[i];
    }
#pragma unroll
    for (int i = 0; i < QS; ++i) acc[i] = apply_act(acc[i], a.act);
    if (a.keep) {
        const uint8_t* kp = a.keep + (size_t)row * a.out_dim + col0;
#pragma unroll
        for (int i = 0; i < QS; ++i) acc[i] = kp[i] ? acc[i] * a.keep_scale : 0.f;
    }
    store_vec<QS>(a.out + (size_t)row * a.ld_out + col0, acc);
}

// ---- lane-packed weight variant -------------------------------------------------------------------
// The per-edge relation-weight read dominates K1's cache traffic.  With the row layout a lane's weights
// are BPL*P*Q contiguous floats, so one 16-B load instruction touches ~25 cache lines at 25-50 % use.
// Here the weights come in the LANE-PACKED layout written by k_pack_weight,
//     packed[r][jq*L + l][0..3] = quad jq of lane l's weight list        (L = nb/BPL active lanes)
// so every weight instruction reads one contiguous L*16-B burst.  Block ownership is chosen so that the
// FEATURE gather is a contiguous burst too: ADJ (lane owns blocks BPL*l .. BPL*l+BPL-1, one BPL*P-float
// vector load) when P < 4, strided (lane owns blocks l + s*L, one P-float load each) when P >= 4.
// Same arithmetic and per-output operation order as k_agg_fast: results are bit-identical.
template <int BPL, bool ADJ>
__device__ __forceinline__ int owned_block(int lane, int sb, int L) { return ADJ ? lane * BPL + sb : lane + sb * L; }

template <int P, int Q, bool TRANS, int BPL, bool ADJ, int U>
__global__ __launch_bounds__(256) void k_agg_packed(const AggParams a) {
    constexpr int PQ = P * Q, NQ = BPL * PQ / 4;      // weight quads per lane
    static_assert((BPL * PQ) % 4 == 0, "lane-packed layout needs whole float4s per lane");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;                 // unused tail entry of an upper-bound-sized item list
    if (it.w >= 0 && !a.partial) return;   // split item without a partial buffer: inconsistent handle, never dereference NULL
    const int L = a.nb / BPL;
    const bool active = lane < L;
    const float4* __restrict__ wbase = reinterpret_cast<const float4*>(a.w) + lane;

    float acc[BPL * Q];
#pragma unroll
    for (int i = 0; i < BPL * Q; ++i) acc[i] = 0.f;

    auto load_x = [&](int sidx, float (&xv)[BPL * P]) {
        const float* xr = a.feat + (size_t)sidx * a.ld_feat;
        if constexpr (ADJ) {
            load_vec<BPL * P>(xr + lane * BPL * P, xv);
        } else {
#pragma unroll
            for (int sb = 0; sb < BPL; ++sb) {
                float t[P];
                load_vec<P>(xr + (lane + sb * L) * P, t);
#pragma unroll
                for (int i = 0; i < P; ++i) xv[sb * P + i] = t[i];
            }
        }
    };
    auto load_w = [&](int r, float (&wv)[BPL * PQ]) {
        const float4* wr = wbase + (size_t)r * (a.w_row / 4);
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq) {
            const float4 q4 = wr[jq * L];
            wv[4 * jq] = q4.x; wv[4 * jq + 1] = q4.y; wv[4 * jq + 2] = q4.z; wv[4 * jq + 3] = q4.w;
        }
    };
    float wkeep[BPL * PQ];            // the weights of relation r_keep (-1: none): see k_agg_fast
#pragma unroll
    for (int i = 0; i < BPL * PQ; ++i) wkeep[i] = 0.f;
    int r_keep = -1;

    for (int e0 = it.y; e0 < it.z; e0 += 64) {
        const int cnt = min(64, it.z - e0);
        int my_n = 0, my_t = 0;
        float my_c = 1.f;
        if (lane < cnt) {
            my_n = a.nbr[e0 + lane];
            my_t = a.etype[e0 + lane];
            if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
        }
        const bool more_chunks = e0 + 64 < it.z;
        int j0 = 0;
        for (; j0 + U <= cnt; j0 += U) {
            float xv[U][BPL * P], wv[U][BPL * PQ], cc[U];
            int rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sidx = rl_i(my_n, j0 + u);
                rr[u] = rl_i(my_t, j0 + u);
                cc[u] = rl_f(my_c, j0 + u);
                if (active) {
                    load_x(sidx, xv[u]);
                    if (rr[u] != (u ? rr[u - 1] : r_keep)) load_w(rr[u], wv[u]);
                }
            }
            int src = -1;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (rr[u] != (u ? rr[u - 1] : r_keep)) src = u;
                if (active) {
                    if (src < 0) block_fma<P, Q, TRANS, BPL>(xv[u], wkeep, cc[u], acc);
#pragma unroll
                    for (int v = 0; v <= u; ++v)
                        if (src == v) block_fma<P, Q, TRANS, BPL>(xv[u], wv[v], cc[u], acc);
                }
            }
            const bool cont = j0 + U < cnt ? rl_i(my_t, j0 + U) == rr[U - 1] : more_chunks;
            if (src >= 0) {
                if (cont) {
#pragma unroll
                    for (int v = 0; v < U; ++v)
                        if (src == v) {
#pragma unroll
                            for (int i = 0; i < BPL * PQ; ++i) wkeep[i] = wv[v][i];
                        }
                    r_keep = rr[U - 1];
                } else {
                    r_keep = -1;
                }
            }
        }
        for (; j0 < cnt; ++j0) {
            float xv[BPL * P];
            const int sidx = rl_i(my_n, j0);
            const int r = rl_i(my_t, j0);
            const float c = rl_f(my_c, j0);
            if (active) {
                load_x(sidx, xv);
                if (r != r_keep) load_w(r, wkeep);
                block_fma<P, Q, TRANS, BPL>(xv, wkeep, c, acc);
            }
            r_keep = r;
        }
    }
    if (!active) return;
    const int row = it.x;
    if (it.w >= 0) {
#pragma unroll
        for (int sb = 0; sb < BPL; ++sb) {
            const int col = owned_block<BPL, ADJ>(lane, sb, L) * Q;
            float o[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) o[i] = acc[sb * Q + i];
            store_vec<Q>(a.partial + (size_t)it.w * a.out_dim + col, o);
        }
        return;
    }
#pragma unroll
    for (int sb = 0; sb < BPL; ++sb) {
        const int col = owned_block<BPL, ADJ>(lane, sb, L) * Q;
        float o[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) o[i] = acc[sb * Q + i];
        if (a.addend) {
            float ad[Q];
            load_vec<Q>(a.addend + (size_t)row * a.ld_add + col, ad);
#pragma unroll
            for (int i = 0; i < Q; ++i) o[i] += ad[i];
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) o[i] = apply_act(o[i], a.act);
        if (a.keep) {
            const uint8_t* kp = a.keep + (size_t)row * a.out_dim + col;
#pragma unroll
            for (int i = 0; i < Q; ++i) o[i] = kp[i] ? o[i] * a.keep_scale : 0.f;
        }
        store_vec<Q>(a.out + (size_t)row * a.ld_out + col, o);
    }
}

// row layout [R][nb*P*Q] -> lane-packed layout for (BPL, ownership); one thread per float4
__global__ __launch_bounds__(256) void k_pack_weight(const float* w, float* packed, int num_rels, int nb, int pq, int bpl,
                                                     int adj, float* packed2, int bpl2, int adj2) {
    // blockIdx.y = 1: the second layout of the same weights (a layer's forward and backward-x launches pack differently)
    if (blockIdx.y == 1) { packed = packed2; bpl = bpl2; adj = adj2; }
    const int L = nb / bpl, nq = bpl * pq / 4, quads = L * nq;     // float4 per relation row
    const int total = num_rels * quads;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / quads, rem = i - r * quads;
        const int l = rem % L, jq = rem / L;               // quad jq of lane l's weight list
        const int f = 4 * jq, sb = f / pq, within = f - sb * pq;
        const int block = adj ? l * bpl + sb : l + sb * L;
        reinterpret_cast<float4*>(packed)[i] =
            *reinterpret_cast<const float4*>(w + (size_t)r * nb * pq + (size_t)block * pq + within);
    }
}

// Any (P, Q): lane <-> output column, 64 columns per sweep over the item's edges.
template <bool TRANS>
__global__ __launch_bounds__(256) void k_agg_generic(const AggParams a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;                 // unused tail entry of an upper-bound-sized item list
    if (it.w >= 0 && !a.partial) return;   // split item without a partial buffer: inconsistent handle, never dereference NULL
    const int P = a.p, Q = a.q;
    for (int c0 = 0; c0 < a.out_dim; c0 += 64) {
        const int c = c0 + lane;
        const bool active = c < a.out_dim;
        const int b = active ? c / Q : 0;
        const int q = active ? c - b * Q : 0;
        float acc = 0.f;
        for (int e0 = it.y; e0 < it.z; e0 += 64) {
            const int cnt = min(64, it.z - e0);
            int my_n = 0, my_t = 0;
            float my_c = 1.f;
            if (lane < cnt) {
                my_n = a.nbr[e0 + lane];
                my_t = a.etype[e0 + lane];
                if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
            }
            for (int j = 0; j < cnt; ++j) {
                const int s = rl_i(my_n, j);
                const int r = rl_i(my_t, j);
                const float cf = rl_f(my_c, j);
                if (active) {
                    const float* xr = a.feat + (size_t)s * a.ld_feat + b * P;
                    const float* wr = a.w + (size_t)r * a.w_row + b * P * Q;
                    float t = 0.f;
                    for (int p = 0; p < P; ++p) t = fmaf(xr[p], TRANS ? wr[q * P + p] : wr[p * Q + q], t);
                    acc = fmaf(t, cf, acc);
                }
            }
        }
        if (!active) continue;
        if (it.w >= 0) {
            a.partial[(size_t)it.w * a.out_dim + c] = acc;
        } else {
            if (a.addend) acc += a.addend[(size_t)it.x * a.ld_add + c];
            acc = apply_act(acc, a.act);
            if (a.keep) acc = a.keep[(size_t)it.x * a.out_dim + c] ? acc * a.keep_scale : 0.f;
            a.out[(size_t)it.x * a.ld_out + c] = acc;
        }
    }
}

// Sum the partial rows of split segments and apply the epilogue.  One wave per (split segment,
// 64-column tile); the slot sum runs in a fixed order (4 interleaved chains, then a fixed combine)
// so the result is bitwise reproducible, with 8 loads in flight per lane.
__device__ __forceinline__ float ordered_slot_sum(const float* __restrict__ p, int n, size_t stride) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        const float v0 = p[(size_t)(k + 0) * stride], v1 = p[(size_t)(k + 1) * stride];
        const float v2 = p[(size_t)(k + 2) * stride], v3 = p[(size_t)(k + 3) * stride];
        const float v4 = p[(size_t)(k + 4) * stride], v5 = p[(size_t)(k + 5) * stride];
        const float v6 = p[(size_t)(k + 6) * stride], v7 = p[(size_t)(k + 7) * stride];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        a0 += v4; a1 += v5; a2 += v6; a3 += v7;
    }
    for (; k < n; ++k) a0 += p[(size_t)k * stride];
    return (a0 + a1) + (a2 + a3);
}

// One WORKGROUP per (split row, 64-column tile): its waves sum consecutive shares of the row's partial slots, each in slot order,
// and the shares are added in wave order through LDS -- a fixed order, whatever the launch.  (One wave per pair left a hub row of a
// 50 M-edge graph -- thousands of slots -- to a single wave: 630 us per launch, 3 ms per step at BASELINE configs[4].)
__global__ __launch_bounds__(1024) void k_agg_fixup(const int4* fix, int n_fix, const float* partial, int out_dim,
                                                    const float* addend, int ld_add, int act, const uint8_t* keep,
                                                    float keep_scale, float* out, int ld_out) {
    __shared__ float share[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int tiles = (out_dim + 63) >> 6;
    const int4 f = fix[blockIdx.x / tiles];
    if (f.x < 0) return;                  // unused tail entry of an upper-bound-sized fix list (uniform for the workgroup)
    const int c = (blockIdx.x % tiles) * 64 + lane;
    const int per = (f.z + nw - 1) / nw, k0 = min(f.z, w * per), k1 = min(f.z, k0 + per);
    float acc = 0.f;
    if (c < out_dim && k1 > k0) acc = ordered_slot_sum(partial + (size_t)(f.y + k0) * out_dim + c, k1 - k0, out_dim);
    share[w][lane] = acc;
    __syncthreads();
    if (w != 0 || c >= out_dim) return;
    acc = share[0][lane];
    for (int i = 1; i < nw; ++i) acc += share[i][lane];
    if (addend) acc += addend[(size_t)f.x * ld_add + c];
    acc = apply_act(acc, act);
    if (keep) acc = keep[(size_t)f.x * out_dim + c] ? acc * keep_scale : 0.f;
    out[(size_t)f.x * ld_out + c] = acc;
}

// ---------------------------------------------------------------------------------------------
// grad_W: one wave per (relation, <=chunk-edge slice); lane keeps its BPL blocks' PxQ outer-product
// sums in registers over the slice.
struct GradWParams {
    const int4* items;
    int n_items;
    const int* src;
    const int* dst;
    const float* coef;
    const int* coef_idx;
    const float* x;
    int ld_x;
    const float* g;
    int ld_g;
    float* grad_w;
    int w_row;
    float* partial;
    int accumulate;
    int nb;
    int p, q;
    int nbp;
};

template <int P, int Q, int BPL, int U>
__global__ __launch_bounds__(256) void k_gradw_fast(const GradWParams a) {
    constexpr int GV = BPL * P, PV = BPL * Q, WV = BPL * P * Q;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;                 // unused tail entry of an upper-bound-sized item list
    if (it.w >= 0 && !a.partial) return;   // split item without a partial buffer: inconsistent handle, never dereference NULL
    const bool active = lane * BPL < a.nbp;
    const int blk0 = blockIdx.y * a.nbp + lane * BPL;
    const float* __restrict__ xbase = a.x + blk0 * P;
    const float* __restrict__ gbase = a.g + blk0 * Q;
    float acc[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) acc[i] = 0.f;
    for (int e0 = it.y; e0 < it.z; e0 += 64) {
        const int cnt = min(64, it.z - e0);
        int my_s = 0, my_d = 0;
        float my_c = 1.f;
        if (lane < cnt) {
            my_s = a.src[e0 + lane];
            my_d = a.dst[e0 + lane];
            if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
        }
        int j = 0;
        for (; j + U <= cnt; j += U) {
            float xv[U][GV], gvv[U][PV], cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = rl_i(my_s, j + u);
                const int d = rl_i(my_d, j + u);
                cc[u] = rl_f(my_c, j + u);
                if (active) {
                    load_vec<GV>(xbase + (size_t)s * a.ld_x, xv[u]);
                    load_vec<PV>(gbase + (size_t)d * a.ld_g, gvv[u]);
                }
            }
            if (active) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int b = 0; b < BPL; ++b)
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const float xc = xv[u][b * P + p] * cc[u];
#pragma unroll
                            for (int q = 0; q < Q; ++q)
                                acc[b * P * Q + p * Q + q] = fmaf(xc, gvv[u][b * Q + q], acc[b * P * Q + p * Q + q]);
                        }
            }
        }
        for (; j < cnt; ++j) {
            float xv[GV], gvv[PV];
            const int s = rl_i(my_s, j);
            const int d = rl_i(my_d, j);
            const float c = rl_f(my_c, j);
            if (active) {
                load_vec<GV>(xbase + (size_t)s * a.ld_x, xv);
                load_vec<PV>(gbase + (size_t)d * a.ld_g, gvv);
#pragma unroll
                for (int b = 0; b < BPL; ++b)
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float xc = xv[b * P + p] * c;
#pragma unroll
                        for (int q = 0; q < Q; ++q)
                            acc[b * P * Q + p * Q + q] = fmaf(xc, gvv[b * Q + q], acc[b * P * Q + p * Q + q]);
                    }
            }
        }
    }
    if (!active) return;
    if (it.w >= 0) {
        store_vec<WV>(a.partial + (size_t)it.w * a.w_row + blk0 * (P * Q), acc);
    } else {
        float* o = a.grad_w + (size_t)it.x * a.w_row + blk0 * (P * Q);
        if (a.accumulate) {
            float old[WV];
            load_vec<WV>(o, old);
#pragma unroll
            for (int i = 0; i < WV; ++i) acc[i] += old[i];
        }
        store_vec<WV>(o, acc);
    }
}

// grad-W for few, large blocks: lane l owns QS consecutive COLUMNS of block l / LPB of the P x Q stored block -- P x QS
// outer-product sums in registers (instead of P x Q), x's block slice shared by the LPB lanes of a block
template <int P, int Q, int QS, int U>
__global__ __launch_bounds__(256) void k_gradw_split(const GradWParams a) {
    constexpr int LPB = Q / QS;
    static_assert(Q % QS == 0, "the columns of a block are dealt evenly to its lanes");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;
    if (it.w >= 0 && !a.partial) return;
    const bool active = lane < a.nbp * LPB;
    const int blk = blockIdx.y * a.nbp + lane / LPB, sub = lane % LPB;
    const float* __restrict__ xbase = a.x + blk * P;
    const float* __restrict__ gbase = a.g + blk * Q + sub * QS;
    float acc[P * QS];
#pragma unroll
    for (int i = 0; i < P * QS; ++i) acc[i] = 0.f;
    for (int e0 = it.y; e0 < it.z; e0 += 64) {
        const int cnt = min(64, it.z - e0);
        int my_s = 0, my_d = 0;
        float my_c = 1.f;
        if (lane < cnt) {
            my_s = a.src[e0 + lane];
            my_d = a.dst[e0 + lane];
            if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
        }
        int j = 0;
        for (; j + U <= cnt; j += U) {
            float xv[U][P], gvv[U][QS], cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sidx = rl_i(my_s, j + u);
                const int d = rl_i(my_d, j + u);
                cc[u] = rl_f(my_c, j + u);
                if (active) {
                    load_vec<P>(xbase + (size_t)sidx * a.ld_x, xv[u]);
                    load_vec<QS>(gbase + (size_t)d * a.ld_g, gvv[u]);
                }
            }
            if (active) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float xc = xv[u][p] * cc[u];
#pragma unroll
                        for (int q = 0; q < QS; ++q) acc[p * QS + q] = fmaf(xc, gvv[u][q], acc[p * QS + q]);
                    }
            }
        }
        for (; j < cnt; ++j) {
            float xv[P], gvv[QS];
            const int sidx = rl_i(my_s, j);
            const int d = rl_i(my_d, j);
            const float c = rl_f(my_c, j);
            if (active) {
                load_vec<P>(xbase + (size_t)sidx * a.ld_x, xv);
                load_vec<QS>(gbase + (size_t)d * a.ld_g, gvv);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float xc = xv[p] * c;
#pragma unroll
                    for (int q = 0; q < QS; ++q) acc[p * QS + q] = fmaf(xc, gvv[q], acc[p * QS + q]);
                }
            }
        }
    }
    if (!active) return;
    float* o = (it.w >= 0 ? a.partial + (size_t)it.w * a.w_row : a.grad_w + (size_t)it.x * a.w_row) + blk * (P * Q) + sub * QS;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float t[QS];
#pragma unroll
        for (int q = 0; q < QS; ++q) t[q] = acc[p * QS + q];
        if (it.w < 0 && a.accumulate) {
            float old[QS];
            load_vec<QS>(o + p * Q, old);
#pragma unroll
            for (int q = 0; q < QS; ++q) t[q] += old[q];
        }
        store_vec<QS>(o + p * Q, t);
    }
}


__global__ __launch_bounds__(256) void k_gradw_generic(const GradWParams a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= a.n_items) return;
    const int4 it = a.items[wave];
    if (it.x < 0) return;                 // unused tail entry of an upper-bound-sized item list
    if (it.w >= 0 && !a.partial) return;   // split item without a partial buffer: inconsistent handle, never dereference NULL
    const int P = a.p, Q = a.q;
    for (int c0 = 0; c0 < a.w_row; c0 += 64) {
        const int c = c0 + lane;
        const bool active = c < a.w_row;
        const int b = active ? c / (P * Q) : 0;
        const int rem = c - b * P * Q;
        const int p = rem / Q, q = rem - p * Q;
        float acc = 0.f;
        for (int e0 = it.y; e0 < it.z; e0 += 64) {
            const int cnt = min(64, it.z - e0);
            int my_s = 0, my_d = 0;
            float my_c = 1.f;
            if (lane < cnt) {
                my_s = a.src[e0 + lane];
                my_d = a.dst[e0 + lane];
                if (a.coef) my_c = a.coef_idx ? a.coef[a.coef_idx[e0 + lane]] : a.coef[e0 + lane];
            }
            for (int j = 0; j < cnt; ++j) {
                const int s = rl_i(my_s, j);
                const int d = rl_i(my_d, j);
                const float cf = rl_f(my_c, j);
                if (active)
                    acc = fmaf(a.x[(size_t)s * a.ld_x + b * P + p] * cf, a.g[(size_t)d * a.ld_g + b * Q + q], acc);
            }
        }
        if (!active) continue;
        if (it.w >= 0) {
            a.partial[(size_t)it.w * a.w_row + c] = acc;
        } else {
            float* o = a.grad_w + (size_t)it.x * a.w_row + c;
            *o = a.accumulate ? *o + acc : acc;
        }
    }
}

__global__ __launch_bounds__(256) void k_gradw_fixup(const int4* fix, int n_fix, const float* partial, int w_row,
                                                     float* grad_w, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int tiles = (w_row + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= n_fix * tiles) return;
    const int4 f = fix[wave / tiles];
    if (f.x < 0) return;                  // unused tail entry of an upper-bound-sized fix list
    const int c = (wave % tiles) * 64 + lane;
    if (c >= w_row) return;
    const float acc = ordered_slot_sum(partial + (size_t)f.y * w_row + c, f.z, w_row);
    float* o = grad_w + (size_t)f.x * w_row + c;
    *o = accumulate ? *o + acc : acc;
}

// the same sums with a lane on FOUR consecutive columns (w_row % 4 == 0: 16-B loads, a quarter of the waves; at h = 500 a
// relation's row is 10-20 kB and its ~9 slices were summed by 40-80 waves of 256-B loads: 85 us per launch, three per step).
// Element by element ordered_slot_sum's four chains and combine: bit-identical.
__global__ __launch_bounds__(256) void k_gradw_fixup4(const int4* fix, int n_fix, const float* partial, int w_row,
                                                      float* grad_w, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int quads = w_row >> 2, tiles = (quads + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= n_fix * tiles) return;
    const int4 f = fix[wave / tiles];
    if (f.x < 0) return;
    const int c4 = (wave % tiles) * 64 + lane;
    if (c4 >= quads) return;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(partial + (size_t)f.y * w_row) + c4;
    const size_t stride = (size_t)quads;
    const int n = f.z;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    auto add = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        const float4 v0 = p[(size_t)(k + 0) * stride], v1 = p[(size_t)(k + 1) * stride];
        const float4 v2 = p[(size_t)(k + 2) * stride], v3 = p[(size_t)(k + 3) * stride];
        const float4 v4 = p[(size_t)(k + 4) * stride], v5 = p[(size_t)(k + 5) * stride];
        const float4 v6 = p[(size_t)(k + 6) * stride], v7 = p[(size_t)(k + 7) * stride];
        add(a0, v0); add(a1, v1); add(a2, v2); add(a3, v3);
        add(a0, v4); add(a1, v5); add(a2, v6); add(a3, v7);
    }
    for (; k < n; ++k) add(a0, p[(size_t)k * stride]);
    float4 acc = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                             (a0.w + a1.w) + (a2.w + a3.w));
    float4* o = reinterpret_cast<float4*>(grad_w + (size_t)f.x * w_row) + c4;
    if (accumulate) {
        const float4 t = *o;
        acc.x = t.x + acc.x; acc.y = t.y + acc.y; acc.z = t.z + acc.z; acc.w = t.w + acc.w;
    }
    *o = acc;
}

// ---------------------------------------------------------------------------------------------
__global__ void k_items_count(const int* rowptr, int n_seg, int chunk, int* n_chunks, int* n_slots, int* is_split) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const int deg = rowptr[s + 1] - rowptr[s];
    const int nch = max(1, (deg + chunk - 1) / chunk);
    n_chunks[s] = nch;
    n_slots[s] = nch > 1 ? nch : 0;
    is_split[s] = nch > 1 ? 1 : 0;
}

__global__ void k_items_fill(const int* rowptr, int n_seg, int chunk, const int* item_off, const int* slot_off,
                             const int* fix_off, int4* items, int4* fix) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const int beg = rowptr[s], end = rowptr[s + 1];
    const int nch = max(1, (end - beg + chunk - 1) / chunk);
    const int ib = item_off[s];
    for (int k = 0; k < nch; ++k) {
        const int b = beg + k * chunk;
        items[ib + k] = make_int4(s, b, min(end, b + chunk), nch > 1 ? slot_off[s] + k : -1);
    }
    if (nch > 1) fix[fix_off[s]] = make_int4(s, slot_off[s], nch, 0);
}

template <typename K, typename Pm>
static int launch_items(K kern, const Pm& p, int n_items, hipStream_t st, const char* what, int parts = 1) {
    if (n_items <= 0) return GV_OK;
    const int waves_per_block = 4;
    dim3 grid((n_items + waves_per_block - 1) / waves_per_block, parts), block(64 * waves_per_block);
    hipLaunchKernelGGL(kern, grid, block, 0, st, p);
    return launch_status(what);
}

}  // namespace gv

using namespace gv;

extern "C" int gv_segment_items_count(const int32_t* rowptr, int n_seg, int chunk, int32_t* n_chunks,
                                      int32_t* n_slots, int32_t* is_split, void* stream) {
    GV_REQUIRE(rowptr && n_chunks && n_slots && is_split, GV_ERR_NULL, "gv_segment_items_count: NULL pointer");
    GV_REQUIRE(n_seg >= 0 && chunk > 0, GV_ERR_SHAPE, "gv_segment_items_count: n_seg=%d chunk=%d", n_seg, chunk);
    if (n_seg == 0) return GV_OK;
    hipLaunchKernelGGL(k_items_count, dim3((n_seg + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, n_seg,
                       chunk, n_chunks, n_slots, is_split);
    return launch_status("gv_segment_items_count");
}

extern "C" int gv_segment_items_fill(const int32_t* rowptr, int n_seg, int chunk, const int32_t* item_off,
                                     const int32_t* slot_off, const int32_t* fix_off, int32_t* items, int32_t* fix,
                                     void* stream) {
    GV_REQUIRE(rowptr && item_off && slot_off && fix_off && items, GV_ERR_NULL, "gv_segment_items_fill: NULL pointer");
    GV_REQUIRE(n_seg >= 0 && chunk > 0, GV_ERR_SHAPE, "gv_segment_items_fill: n_seg=%d chunk=%d", n_seg, chunk);
    if (n_seg == 0) return GV_OK;
    hipLaunchKernelGGL(k_items_fill, dim3((n_seg + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, n_seg,
                       chunk, item_off, slot_off, fix_off, (int4*)items, (int4*)fix);
    return launch_status("gv_segment_items_fill");
}

namespace {
// Lane mapping of the register kernels: BPL adjacent blocks per lane (so that a lane gathers 16 B where the block
// size allows: P=1 -> 4 blocks, P=2 -> 2, P=4 -> 1..2; odd sizes P=5 -> 2 blocks = 40 B as five 8-B loads,
// P=10 -> 1 block = 40 B) and `parts` column parts (grid.y) when the row needs more than 64 lanes.
// Returns false when no such mapping exists -> generic kernel.
struct LanePlan { int bpl; int parts; };
bool lane_plan(int nb, int p, LanePlan* out, bool aggregate = false) {
    int cands[3] = {0, 0, 0};
    switch (p) {
        case 1: cands[0] = 4; cands[1] = 2; break;
        case 2: cands[0] = 2; cands[1] = 4; cands[2] = 1; break;
        case 4: cands[0] = 1; cands[1] = 2; break;
        case 8: cands[0] = 1; break;
        case 5: {
            // aggregation: ONE 5-wide block per lane and two column parts (grid.y) instead of two blocks per lane: half the
            // weight registers (50 instead of 100 floats per edge), 3-4 waves per SIMD instead of 1-2 with two edges in flight
            // -- measured at h = 500: 5x10 forward 1201 -> 891 us, 5x5 forward / backward-x 529 / 523 -> 436 us, step
            // 5.44 -> 4.91 ms, bit-identical results (GV_K1_BPL1=0 restores two blocks per lane)
            static const int bpl1 = getenv("GV_K1_BPL1") ? atoi(getenv("GV_K1_BPL1")) : 1;
            if (bpl1 && aggregate) { cands[0] = 1; cands[1] = 2; } else cands[0] = 2;
            break;
        }
        case 10: cands[0] = 1; break;
        default: return false;
    }
    for (int c = 0; c < 3 && cands[c]; ++c) {
        const int bpl = cands[c];
        if (nb % bpl) continue;
        const int lanes = nb / bpl;
        for (int parts = 1; parts <= 16; ++parts) {
            if (lanes % parts) continue;
            const int per = lanes / parts;
            if (per <= 64 && (per >= 16 || parts == 1)) { out->bpl = bpl; out->parts = parts; return true; }
        }
    }
    return false;
}
}  // namespace

namespace {
// lane-packed kernels: BPL = fewest blocks per lane that fit the row into one wave; ownership by the
// gathered block size (adjacent for P < 4 so that the feature gather stays one 16-B vector per lane)
struct PackPlan { int bpl; int adj; };
bool pack_plan(int nb, int p_gather, int q_out, bool trans, PackPlan* plan) {
    int bpl = 0;
    if (p_gather < 4) {                       // adjacent blocks must make up >= 4 gathered floats per lane
        bpl = 4 / p_gather;
        if (nb % bpl != 0 || nb / bpl > 64) return false;
    } else {
        for (int b : {1, 2})
            if (!bpl && nb % b == 0 && nb / b <= 64) bpl = b;
        if (!bpl) return false;
    }
    if ((bpl * p_gather * q_out) % 4 != 0) return false;
    const bool ok = !trans ? ((p_gather == 2 && (q_out == 2 || q_out == 4)) || (p_gather == 4 && (q_out == 4 || q_out == 8)))
                           : ((p_gather == 2 && q_out == 2) || (p_gather == 4 && (q_out == 2 || q_out == 4)) ||
                              (p_gather == 8 && q_out == 4));
    if (!ok) return false;
    plan->bpl = bpl;
    plan->adj = p_gather < 4 ? 1 : 0;
    return true;
}
}  // namespace

extern "C" int gv_rgcn_bdd_pack_supported(int num_bases, int blk_in, int blk_out, int transpose_w) {
    PackPlan pl;
    return pack_plan(num_bases, blk_in, blk_out, transpose_w != 0, &pl) ? 1 : 0;
}

extern "C" int gv_rgcn_bdd_pack_weight(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out,
                                       int transpose_w, float* packed, void* stream) {
    GV_REQUIRE(weight && packed, GV_ERR_NULL, "gv_rgcn_bdd_pack_weight: NULL pointer");
    PackPlan pl;
    GV_REQUIRE(pack_plan(num_bases, blk_in, blk_out, transpose_w != 0, &pl), GV_ERR_SHAPE,
               "gv_rgcn_bdd_pack_weight: no lane-packed kernel for num_bases=%d blocks %dx%d trans=%d", num_bases, blk_in,
               blk_out, transpose_w);
    GV_REQUIRE(aligned16(weight) && aligned16(packed), GV_ERR_ALIGN, "gv_rgcn_bdd_pack_weight: 16-B alignment required");
    const int total = num_rels * num_bases * blk_in * blk_out / 4;
    hipLaunchKernelGGL(k_pack_weight, dim3(min(2048, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       packed, num_rels, num_bases, blk_in * blk_out, pl.bpl, pl.adj, (float*)nullptr, 0, 0);
    return launch_status("gv_rgcn_bdd_pack_weight");
}

extern "C" int gv_rgcn_bdd_pack_weight_pair(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out,
                                            float* packed_fwd, float* packed_bwd, void* stream) {
    GV_REQUIRE(weight && packed_fwd && packed_bwd, GV_ERR_NULL, "gv_rgcn_bdd_pack_weight_pair: NULL pointer");
    PackPlan pf, pb;
    GV_REQUIRE(pack_plan(num_bases, blk_in, blk_out, false, &pf) && pack_plan(num_bases, blk_out, blk_in, true, &pb),
               GV_ERR_SHAPE, "gv_rgcn_bdd_pack_weight_pair: no lane-packed kernels for num_bases=%d blocks %dx%d", num_bases,
               blk_in, blk_out);
    GV_REQUIRE(aligned16(weight) && aligned16(packed_fwd) && aligned16(packed_bwd), GV_ERR_ALIGN,
               "gv_rgcn_bdd_pack_weight_pair: 16-B alignment required");
    const int total = num_rels * num_bases * blk_in * blk_out / 4;
    hipLaunchKernelGGL(k_pack_weight, dim3(min(2048, (total + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, weight,
                       packed_fwd, num_rels, num_bases, blk_in * blk_out, pf.bpl, pf.adj, packed_bwd, pb.bpl, pb.adj);
    return launch_status("gv_rgcn_bdd_pack_weight_pair");
}

extern "C" int gv_rgcn_bdd_aggregate(const int32_t* items, int n_items, const int32_t* fix, int n_fix,
                                     const int32_t* nbr, const int32_t* etype, const float* coef,
                                     const int32_t* coef_idx, const float* feat, int ld_feat, const float* weight,
                                     int num_rels, int num_bases, int blk_in, int blk_out, int transpose_w,
                                     int weight_packed, const float* addend, int ld_addend, int act,
                                     const uint8_t* keep, float keep_scale, float* out, int ld_out, float* partial,
                                     void* stream) {
    GV_REQUIRE(n_items >= 0 && n_fix >= 0, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate: negative item count");
    if (n_items == 0) return GV_OK;
    // nbr / etype may be NULL for an edge-less graph (every item then has begin == end and never reads them)
    GV_REQUIRE(items && feat && weight && out, GV_ERR_NULL, "gv_rgcn_bdd_aggregate: NULL pointer");
    GV_REQUIRE(num_bases > 0 && blk_in > 0 && blk_out > 0 && num_rels > 0, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate: num_bases=%d blk_in=%d blk_out=%d num_rels=%d", num_bases, blk_in, blk_out,
               num_rels);
    GV_REQUIRE(n_fix == 0 || (fix && partial), GV_ERR_NULL, "gv_rgcn_bdd_aggregate: split segments need fix+partial");
    GV_REQUIRE(ld_feat >= num_bases * blk_in && ld_out >= num_bases * blk_out, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate: leading dimension smaller than the row");
    GV_REQUIRE(act == GV_ACT_NONE || act == GV_ACT_RELU, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate: unknown act %d", act);
    AggParams a;
    a.items = (const int4*)items; a.n_items = n_items; a.nbr = nbr; a.etype = etype; a.coef = coef;
    a.coef_idx = coef_idx; a.feat = feat; a.ld_feat = ld_feat; a.w = weight;
    a.w_row = num_bases * blk_in * blk_out; a.addend = addend; a.ld_add = ld_addend; a.act = act; a.keep = keep;
    a.keep_scale = keep_scale; a.out = out; a.ld_out = ld_out; a.partial = partial;
    a.out_dim = num_bases * blk_out; a.nb = num_bases; a.p = blk_in; a.q = blk_out;
    hipStream_t st = (hipStream_t)stream;
    // vector paths: 16-B accesses need every row base 16-B aligned (ld % 4); the odd block sizes (5, 10) only
    // issue 8-B accesses, for which ld % 2 suffices -- h = 500, 1000 satisfy both
    const bool vec_ok = aligned16(feat) && aligned16(weight) && aligned16(out) && (ld_feat % 4 == 0) &&
                        (ld_out % 4 == 0) && (!addend || (aligned16(addend) && ld_addend % 4 == 0)) &&
                        (!partial || aligned16(partial)) && (a.out_dim % 4 == 0) && (a.w_row % 4 == 0);
    int rc = -1000;
    static const int u_env = getenv("GV_K1_U") ? atoi(getenv("GV_K1_U")) : 0;     // tuning knob (tools/microbench.py)
    if (weight_packed) {
        PackPlan pl;
        GV_REQUIRE(vec_ok && pack_plan(num_bases, blk_in, blk_out, transpose_w != 0, &pl), GV_ERR_SHAPE,
                   "gv_rgcn_bdd_aggregate: lane-packed weights unsupported for num_bases=%d blocks %dx%d (or unaligned)",
                   num_bases, blk_in, blk_out);
#define GV_PK_CASE(P_, Q_, T_, B_, A_, U_)                                                                   \
    if (rc == -1000 && blk_in == P_ && blk_out == Q_ && (transpose_w != 0) == T_ && pl.bpl == B_ && pl.adj == A_) \
        rc = launch_items(k_agg_packed<P_, Q_, T_, B_, (A_ != 0), U_>, a, n_items, st, "gv_rgcn_bdd_aggregate(packed)");
#define GV_PK_U(P_, Q_, T_, B_, A_)                                                                            \
    if (rc == -1000 && u_env && blk_in == P_ && blk_out == Q_ && (transpose_w != 0) == T_ && pl.bpl == B_ && pl.adj == A_) { \
        if (u_env == 2) rc = launch_items(k_agg_packed<P_, Q_, T_, B_, (A_ != 0), 2>, a, n_items, st, "agg(packed,U2)");    \
        if (u_env == 4) rc = launch_items(k_agg_packed<P_, Q_, T_, B_, (A_ != 0), 4>, a, n_items, st, "agg(packed,U4)");    \
        if (u_env == 8) rc = launch_items(k_agg_packed<P_, Q_, T_, B_, (A_ != 0), 8>, a, n_items, st, "agg(packed,U8)");    \
    }
        GV_PK_U(2, 2, false, 2, 1) GV_PK_U(2, 4, false, 2, 1) GV_PK_U(2, 2, true, 2, 1) GV_PK_U(4, 2, true, 2, 0)
#undef GV_PK_U
        GV_PK_CASE(2, 2, false, 2, 1, 4)
        GV_PK_CASE(2, 4, false, 2, 1, 4)
        GV_PK_CASE(4, 4, false, 1, 0, 4) GV_PK_CASE(4, 4, false, 2, 0, 2)
        GV_PK_CASE(4, 8, false, 1, 0, 2) GV_PK_CASE(4, 8, false, 2, 0, 2)
        GV_PK_CASE(2, 2, true, 2, 1, 2)
        GV_PK_CASE(4, 2, true, 1, 0, 8) GV_PK_CASE(4, 2, true, 2, 0, 4)      // (U = 4 since the weight-run reuse: 126 -> 111 us at FB15k-237 size)
        GV_PK_CASE(4, 4, true, 1, 0, 4) GV_PK_CASE(4, 4, true, 2, 0, 2)
        GV_PK_CASE(8, 4, true, 1, 0, 2) GV_PK_CASE(8, 4, true, 2, 0, 2)
#undef GV_PK_CASE
    }
    // few, large blocks (B <= 32 of width >= 10): output columns of a block split over lanes (k_agg_split)
#define GV_SPLIT_CASE(PI_, QO_, T_, QS_, U_, PARTS_)                                                                       \
    if (rc == -1000 && vec_ok && blk_in == PI_ && blk_out == QO_ && (transpose_w != 0) == T_ && num_bases % PARTS_ == 0 &&  \
        (num_bases / PARTS_) * (QO_ / QS_) <= 64 && num_bases <= 32) {                                                      \
        a.nbp = num_bases / PARTS_;                                                                                         \
        rc = launch_items(k_agg_split<PI_, QO_, T_, QS_, U_>, a, n_items, st, "gv_rgcn_bdd_aggregate(split)", PARTS_);      \
    }
    GV_SPLIT_CASE(10, 10, false, 5, 2, 1) GV_SPLIT_CASE(10, 10, true, 5, 2, 1)
    GV_SPLIT_CASE(10, 20, false, 4, 2, 2) GV_SPLIT_CASE(20, 10, true, 2, 2, 2)
#undef GV_SPLIT_CASE
    LanePlan lp{0, 1};
    const bool has_plan = lane_plan(num_bases, blk_in, &lp, true);
    const int bpl = has_plan ? lp.bpl : 0;
    if (rc == -1000) a.nbp = has_plan ? num_bases / lp.parts : num_bases;
#define GV_AGG_CASE(P_, Q_, T_, B_, U_)                                                               \
    if (rc == -1000 && vec_ok && blk_in == P_ && blk_out == Q_ && (transpose_w != 0) == T_ && bpl == B_) \
        rc = launch_items(k_agg_fast<P_, Q_, T_, B_, U_>, a, n_items, st, "gv_rgcn_bdd_aggregate", lp.parts);
#define GV_AGG_U(P_, Q_, T_, B_)                                                                              \
    if (rc == -1000 && u_env && vec_ok && blk_in == P_ && blk_out == Q_ && (transpose_w != 0) == T_ && bpl == B_) { \
        if (u_env == 2) rc = launch_items(k_agg_fast<P_, Q_, T_, B_, 2>, a, n_items, st, "agg(U2)", lp.parts);          \
        if (u_env == 4) rc = launch_items(k_agg_fast<P_, Q_, T_, B_, 4>, a, n_items, st, "agg(U4)", lp.parts);          \
        if (u_env == 8) rc = launch_items(k_agg_fast<P_, Q_, T_, B_, 8>, a, n_items, st, "agg(U8)", lp.parts);          \
    }
    GV_AGG_U(2, 2, false, 2) GV_AGG_U(2, 4, false, 2) GV_AGG_U(2, 2, true, 2) GV_AGG_U(4, 2, true, 2)
#undef GV_AGG_U
    GV_AGG_CASE(1, 1, false, 4, 8)
    GV_AGG_CASE(1, 2, false, 4, 4)
    GV_AGG_CASE(2, 2, false, 2, 8)
    GV_AGG_CASE(2, 4, false, 2, 4)
    GV_AGG_CASE(4, 4, false, 1, 4)
    GV_AGG_CASE(4, 4, false, 2, 2)
    GV_AGG_CASE(4, 8, false, 1, 2)
    GV_AGG_CASE(4, 8, false, 2, 2)
    GV_AGG_CASE(1, 1, true, 4, 8)
    GV_AGG_CASE(2, 1, true, 2, 8)
    GV_AGG_CASE(2, 2, true, 2, 8)
    GV_AGG_CASE(4, 2, true, 1, 8)
    GV_AGG_CASE(4, 2, true, 2, 4)
    GV_AGG_CASE(4, 4, true, 1, 4)
    GV_AGG_CASE(4, 4, true, 2, 2)
    GV_AGG_CASE(8, 4, true, 1, 2)
    GV_AGG_CASE(5, 5, false, 1, 2)
    GV_AGG_CASE(5, 10, false, 1, 2)
    GV_AGG_CASE(5, 5, true, 1, 2)
    GV_AGG_CASE(5, 5, false, 2, 2)
    GV_AGG_CASE(5, 10, false, 2, 1)
    GV_AGG_CASE(5, 5, true, 2, 2)
    GV_AGG_CASE(10, 5, true, 1, 1)      // one edge in flight at 4 waves / SIMD beats two at 2 (978 vs 1022 us at h = 500; U = 4: 1234)
    GV_AGG_CASE(10, 10, false, 1, 1)
    GV_AGG_CASE(10, 10, true, 1, 1)
#undef GV_AGG_CASE
    if (rc == -1000) {
        if (transpose_w)
            rc = launch_items(k_agg_generic<true>, a, n_items, st, "gv_rgcn_bdd_aggregate(generic)");
        else
            rc = launch_items(k_agg_generic<false>, a, n_items, st, "gv_rgcn_bdd_aggregate(generic)");
    }
    if (rc != GV_OK) return rc;
    if (n_fix > 0) {         // split (hub) rows: their slots summed in order, then the epilogue
        const int pairs = n_fix * ((a.out_dim + 63) / 64);
        hipLaunchKernelGGL(k_agg_fixup, dim3(pairs), dim3(pairs >= 65536 ? 1024 : 256), 0, st, (const int4*)fix, n_fix, partial,
                           a.out_dim, addend, ld_addend, act, keep, keep_scale, out, ld_out);
        return launch_status("gv_rgcn_bdd_aggregate(fixup)");
    }
    return GV_OK;
}

extern "C" int gv_rgcn_bdd_fixup(const int32_t* fix, int n_fix, const float* partial, int out_dim, const float* addend,
                                 int ld_addend, int act, const uint8_t* keep, float keep_scale, float* out, int ld_out,
                                 void* stream) {
    GV_REQUIRE(n_fix >= 0, GV_ERR_SHAPE, "gv_rgcn_bdd_fixup: negative count");
    if (n_fix == 0) return GV_OK;
    GV_REQUIRE(fix && partial && out, GV_ERR_NULL, "gv_rgcn_bdd_fixup: NULL pointer");
    const int pairs = n_fix * ((out_dim + 63) / 64);
    hipLaunchKernelGGL(k_agg_fixup, dim3(pairs), dim3(pairs >= 65536 ? 1024 : 256), 0, (hipStream_t)stream, (const int4*)fix, n_fix,
                       partial, out_dim, addend, ld_addend, act, keep, keep_scale, out, ld_out);
    return launch_status("gv_rgcn_bdd_fixup");
}

extern "C" int gv_rgcn_bdd_grad_weight(const int32_t* items, int n_items, const int32_t* fix, int n_fix,
                                       const int32_t* src, const int32_t* dst, const float* coef,
                                       const int32_t* coef_idx, const float* x, int ld_x, const float* g, int ld_g,
                                       int num_bases, int blk_in, int blk_out, float* grad_w, float* partial,
                                       int accumulate, void* stream) {
    GV_REQUIRE(n_items >= 0 && n_fix >= 0, GV_ERR_SHAPE, "gv_rgcn_bdd_grad_weight: negative item count");
    if (n_items == 0) return GV_OK;
    GV_REQUIRE(items && x && g && grad_w, GV_ERR_NULL, "gv_rgcn_bdd_grad_weight: NULL pointer");   // src/dst: see above
    GV_REQUIRE(num_bases > 0 && blk_in > 0 && blk_out > 0, GV_ERR_SHAPE, "gv_rgcn_bdd_grad_weight: bad block sizes");
    GV_REQUIRE(n_fix == 0 || (fix && partial), GV_ERR_NULL, "gv_rgcn_bdd_grad_weight: split segments need fix+partial");
    GV_REQUIRE(ld_x >= num_bases * blk_in && ld_g >= num_bases * blk_out, GV_ERR_SHAPE,
               "gv_rgcn_bdd_grad_weight: leading dimension smaller than the row");
    GradWParams a;
    a.items = (const int4*)items; a.n_items = n_items; a.src = src; a.dst = dst; a.coef = coef; a.coef_idx = coef_idx;
    a.x = x; a.ld_x = ld_x; a.g = g; a.ld_g = ld_g; a.grad_w = grad_w; a.w_row = num_bases * blk_in * blk_out;
    a.partial = partial; a.accumulate = accumulate; a.nb = num_bases; a.p = blk_in; a.q = blk_out;
    hipStream_t st = (hipStream_t)stream;
    const bool vec_ok = aligned16(x) && aligned16(g) && aligned16(grad_w) && (ld_x % 4 == 0) && (ld_g % 4 == 0) &&
                        (!partial || aligned16(partial)) && (a.w_row % 4 == 0);
    int rc = -1000;
#define GV_GW_SPLIT(P_, Q_, QS_, U_, PARTS_)                                                                              \
    if (rc == -1000 && vec_ok && blk_in == P_ && blk_out == Q_ && num_bases % PARTS_ == 0 && num_bases <= 32 &&          \
        (num_bases / PARTS_) * (Q_ / QS_) <= 64) {                                                                        \
        a.nbp = num_bases / PARTS_;                                                                                       \
        rc = launch_items(k_gradw_split<P_, Q_, QS_, U_>, a, n_items, st, "gv_rgcn_bdd_grad_weight(split)", PARTS_);      \
    }
    GV_GW_SPLIT(10, 10, 5, 2, 1) GV_GW_SPLIT(10, 20, 4, 2, 2)
#undef GV_GW_SPLIT
    LanePlan lp{0, 1};
    // 5x10 blocks: one block per lane and two column parts (50 accumulators, two edges in flight) instead of two blocks per
    // lane with one edge in flight: 340 -> 303 us at h = 500; 5x5 is indifferent (205 us either way) and keeps two per lane
    const bool has_plan = lane_plan(num_bases, blk_in, &lp, blk_in == 5 && blk_out >= 10);
    const int bpl = has_plan ? lp.bpl : 0;
    if (rc == -1000) a.nbp = has_plan ? num_bases / lp.parts : num_bases;
#define GV_GW_CASE(P_, Q_, B_, U_)                                                  \
    if (rc == -1000 && vec_ok && blk_in == P_ && blk_out == Q_ && bpl == B_)        \
        rc = launch_items(k_gradw_fast<P_, Q_, B_, U_>, a, n_items, st, "gv_rgcn_bdd_grad_weight", lp.parts);
    GV_GW_CASE(1, 1, 4, 8)
    GV_GW_CASE(1, 2, 4, 4)
    GV_GW_CASE(2, 2, 2, 8)
    GV_GW_CASE(2, 4, 2, 4)
    GV_GW_CASE(4, 4, 1, 4)
    GV_GW_CASE(4, 4, 2, 2)
    GV_GW_CASE(4, 8, 1, 2)
    GV_GW_CASE(4, 8, 2, 2)
    GV_GW_CASE(5, 5, 2, 2)
    GV_GW_CASE(5, 10, 2, 1)
    GV_GW_CASE(5, 10, 1, 2)
    GV_GW_CASE(10, 10, 1, 1)
#undef GV_GW_CASE
    if (rc == -1000) rc = launch_items(k_gradw_generic, a, n_items, st, "gv_rgcn_bdd_grad_weight(generic)");
    if (rc != GV_OK) return rc;
    if (n_fix > 0) {
        if (a.w_row % 4 == 0 && aligned16(partial) && aligned16(grad_w)) {
            const int waves = n_fix * ((a.w_row / 4 + 63) / 64);
            hipLaunchKernelGGL(k_gradw_fixup4, dim3((waves + 3) / 4), dim3(256), 0, st, (const int4*)fix, n_fix, partial,
                               a.w_row, grad_w, accumulate);
        } else {
            const int waves = n_fix * ((a.w_row + 63) / 64);
            hipLaunchKernelGGL(k_gradw_fixup, dim3((waves + 3) / 4), dim3(256), 0, st, (const int4*)fix, n_fix, partial,
                               a.w_row, grad_w, accumulate);
        }
        return launch_status("gv_rgcn_bdd_grad_weight(fixup)");
    }
    return GV_OK;
}
