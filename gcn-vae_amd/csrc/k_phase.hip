// K1 by RELATION PHASES: the relation weights reach the lanes through LDS instead of the L1 -> VGPR path.
//
// k_agg_fast / k_agg_packed (k_bdd.hip) read, per EDGE, the edge's feature row (0.8-2 kB) and its relation's block weights
// (1.6-3.2 kB at h = 200, 10-20 kB at h = 500) through the vector-memory return path, which is what bounds them (~40 B/clk per
// CU).  Here a workgroup owns a TILE of nw x K work items (destination rows, or <= chunk-edge slices of
// hub rows): wave w keeps the K output rows of its items in registers for the whole launch, and the workgroup walks the
// relation types in PHASES of G consecutive relations whose block weights are staged ONCE per tile into LDS (lane-packed,
// LDS-DMA, double buffered: phase p+1 lands while phase p is computed, one barrier per phase).  In phase p a wave processes
// the edges of its K rows whose relation lies in that phase (index built once per static graph: ops.PhaseOrder): per edge
// one feature-row gather from global memory, the block weights by ds_read_b128 (conflict-free: lane-consecutive quads),
// FMA into registers, one add into the row's accumulators (wave-uniform switch on the item slot).  Weight bytes through
// the vector-memory path drop from E x w_row to tiles x table size; the per-edge weights come over the 256 B/clk LDS read path.
// No atomics: every row is summed by one wave in (phase, original edge) order -> bitwise reproducible.
#include <stdlib.h>

#include "k_phase.h"

namespace gv {

// LEAN (wide blocks, where registers decide how many rows a wave can own): one block per lane whose weights are q-major in the
// lane's list (the transposed product's natural order); the weights are fetched and used in two halves of the output columns,
// and with M <= 2 rows in flight the row is picked by a select instead of through the indexed register vector -- ~45 registers
// less, which buys 8 rows per wave (one round of workgroups at FB15k-237 size) for the 10x5 blocks.
template <int P, int Q, bool TRANS, int BPL, int K, int M, bool LEAN = false>
__global__ __launch_bounds__(1024) void k_agg_phase(const PhaseParams a) {
    static_assert(!LEAN || (BPL == 1 && M <= 2), "lean mode: one q-major block per lane, at most two rows in flight");
    extern __shared__ __attribute__((aligned(16))) float4 wlds[];
    constexpr int GV = BPL * P, PV = BPL * Q, WV = BPL * P * Q, NQ = (WV + 3) / 4;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    const int tile = blockIdx.x, part = blockIdx.y;
    const int L = a.L;
    // lanes beyond the part's L lane slots shadow lane L-1 (same addresses: no extra traffic) so that the whole loop nest
    // is wave-uniform control flow; only their final stores are masked
    const bool active = lane < L;
    const int ln = min(lane, L - 1);
    const char* __restrict__ fbase_b = reinterpret_cast<const char*>(a.feat + (part * a.nbp + ln * BPL) * P);
    const unsigned row_bytes = (unsigned)a.ld_feat * 4u;
    const float4* __restrict__ wsrc = a.wpk + (size_t)part * a.R * NQ * L;
    const int rel_quads = NQ * L;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float4*)wlds);

    // stage the weights of phase p into buffer b: a straight copy of G*NQ*L (rounded up to whole wave-instructions)
    // lane-packed quads, 1 KiB per LDS-DMA instruction; the slack behind the table and behind each buffer absorbs the tail
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

    // this wave's lists: contiguous positions [off[base], off[base + n_phases]), one list per phase.  64 list offsets are
    // held one per lane (refilled every 63 phases); the edge metadata streams through two 64-edge register batches, the
    // next one always in flight, so that no phase waits on an offset -> metadata -> gather chain
    const int* __restrict__ offp = a.off + ((size_t)tile * nw + wv) * a.n_phases;
    int ob = 0;
    int ov = offp[min(lane, a.n_phases)];
    const int we = offp[a.n_phases];
    int cb = prl_i(ov, 0);
    int cur_n = 0, cur_m = 0, nxt_n = 0, nxt_m = 0;
    float cur_c = 1.f, nxt_c = 1.f;
    if (cb + lane < we) {
        cur_n = a.nbr[cb + lane];
        cur_m = a.meta[cb + lane];
        if (a.coef) cur_c = a.coef[cb + lane];
    }
    if (cb + 64 + lane < we) {
        nxt_n = a.nbr[cb + 64 + lane];
        nxt_m = a.meta[cb + 64 + lane];
        if (a.coef) nxt_c = a.coef[cb + 64 + lane];
    }
    if (a.nbuf == 2) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    for (int p = 0; p < a.n_phases; ++p) {
        if (a.nbuf == 2) {
            if (p + 1 < a.n_phases) stage(p + 1, (p + 1) & 1);
        } else {                                   // one buffer: refill it between two barriers
            __syncthreads();
            stage(p, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (p + 1 - ob >= 64) {                    // the next 64 list offsets
            ob = p;
            ov = offp[min(ob + lane, a.n_phases)];
        }
        const int ee = prl_i(ov, p + 1 - ob);
        int j = prl_i(ov, p - ob);
        const float4* __restrict__ wcur = wlds + (a.nbuf == 2 ? (p & 1) * a.slab : 0) + ln;
        while (j < ee) {
            if (j >= cb + 64) {                    // next metadata batch; request the one after it
                cur_n = nxt_n; cur_m = nxt_m; cur_c = nxt_c;
                cb += 64;
                nxt_n = 0; nxt_m = 0; nxt_c = 1.f;
                if (cb + 64 + lane < we) {
                    nxt_n = a.nbr[cb + 64 + lane];
                    nxt_m = a.meta[cb + 64 + lane];
                    if (a.coef) nxt_c = a.coef[cb + 64 + lane];
                }
            }
            // A batch of <= M edges (lists are sorted by item slot): all feature rows are requested into ONE register
            // vector, then the slots are walked with STATIC accumulators -- slot kk's edges are a contiguous run of the
            // batch -- and the run's feature rows are picked out of the vector by their (wave-uniform) position: hipcc
            // lowers that to M0-relative register reads (s_set_gpr_idx), GV per edge, cheaper than indexing the accumulators.
            const int j0 = j - cb;                 // lane of the batch's first edge
            const int nbt = min(min(ee - j, M), 64 - j0);
            float xt[M][GV];
#pragma unroll
            for (int u = 0; u < M; ++u) {
#pragma unroll
                for (int i = 0; i < GV; ++i) xt[u][i] = 0.f;
                if (u < nbt)
                    load_vec<GV>(reinterpret_cast<const float*>(fbase_b + (size_t)(unsigned)prl_i(cur_n, j0 + u) * row_bytes),
                                 xt[u]);
            }
            constexpr int NX = M * GV <= 16 ? 16 : 32;
            static_assert(M * GV <= 32, "the feature rows in flight must fit one 32-register vector");
            typename AccVec<NX>::type xb;
            if constexpr (!LEAN) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xb[i] = i < M * GV ? xt[i / GV][i % GV] : 0.f;
            }
            int jj = j0;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const unsigned long long mine = __ballot(lane >= j0 && lane < j0 + nbt && (cur_m & 15) == kk);
                const int j1 = jj + __popcll(mine);
                for (; jj < j1; ++jj) {
                    const int m = prl_i(cur_m, jj);
                    const float cf = prl_f(cur_c, jj);
                    const float4* wq = wcur + (m >> 4) * rel_quads;
                    if constexpr (LEAN) {
                        const bool first = (M == 1) || (jj == j0);
                        float xc[GV];
#pragma unroll
                        for (int i = 0; i < GV; ++i) xc[i] = first ? xt[0][i] : xt[M - 1][i];
                        constexpr int QH = (Q + 1) / 2;                     // output columns of the first half
                        constexpr int Q0 = (QH * P + 3) / 4;                // its quads: floats [0, QH*P)
                        constexpr int QB = (QH * P) / 4, OFF = QH * P - 4 * QB;      // second half starts inside quad QB
                        {
                            float w0[Q0 * 4];
#pragma unroll
                            for (int jq = 0; jq < Q0; ++jq) {
                                const float4 t = wq[jq * L];
                                w0[4 * jq] = t.x; w0[4 * jq + 1] = t.y; w0[4 * jq + 2] = t.z; w0[4 * jq + 3] = t.w;
                            }
#pragma unroll
                            for (int q = 0; q < QH; ++q) {
                                float t = 0.f;
#pragma unroll
                                for (int pp = 0; pp < P; ++pp) t = fmaf(xc[pp], w0[q * P + pp], t);
                                acc[kk][q] = fmaf(t, cf, acc[kk][q]);
                            }
                        }
                        asm volatile("" ::: "memory");                      // the second half's fetch stays behind the first half's use
                        {
                            float w1[(NQ - QB) * 4];
#pragma unroll
                            for (int jq = QB; jq < NQ; ++jq) {
                                const float4 t = wq[jq * L];
                                w1[4 * (jq - QB)] = t.x; w1[4 * (jq - QB) + 1] = t.y; w1[4 * (jq - QB) + 2] = t.z; w1[4 * (jq - QB) + 3] = t.w;
                            }
#pragma unroll
                            for (int q = QH; q < Q; ++q) {
                                float t = 0.f;
#pragma unroll
                                for (int pp = 0; pp < P; ++pp) t = fmaf(xc[pp], w1[(q - QH) * P + pp + OFF], t);
                                acc[kk][q] = fmaf(t, cf, acc[kk][q]);
                            }
                        }
                        continue;
                    }
                    float wr[NQ * 4];
#pragma unroll
                    for (int jq = 0; jq < NQ; ++jq) {
                        const float4 t = wq[jq * L];
                        wr[4 * jq] = t.x; wr[4 * jq + 1] = t.y; wr[4 * jq + 2] = t.z; wr[4 * jq + 3] = t.w;
                    }
                    const int xo = (jj - j0) * GV;
                    float xc[GV];
#pragma unroll
                    for (int i = 0; i < GV; ++i) xc[i] = xb[xo + i];
#pragma unroll
                    for (int b = 0; b < BPL; ++b) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            float t = 0.f;
#pragma unroll
                            for (int pp = 0; pp < P; ++pp) {
                                const float wvv = TRANS ? wr[b * P * Q + q * P + pp] : wr[b * P * Q + pp * Q + q];
                                t = fmaf(xc[b * P + pp], wvv, t);
                            }
                            acc[kk][b * Q + q] = fmaf(t, cf, acc[kk][b * Q + q]);
                        }
                    }
                }
            }
            j += nbt;
        }
        if (a.nbuf == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of phase p+1's weights has landed
            __syncthreads();        // ... everybody's has, and phase p's buffer is free
        }
    }

    phase_epilogue<PV, K>(a, acc, tile, nw, wv, part, lane, active);
}

// row layout [R][nb*P*Q] -> [parts][R][NQ][L] float4: quad jq of the WV = BPL*P*Q weights of lane l of a column part
// (zero padded to whole quads); one thread per output quad
// qmajor_q > 0 (one block per lane): element e of the lane's list is stored block element (e % p, e / p), i.e. the p x q block is
// transposed on the way -- the order the lean kernel walks (two halves of the output columns)
__global__ __launch_bounds__(256) void k_pack_weight_phase(const float* __restrict__ w, float4* __restrict__ out, int num_rels,
                                                           int nb, int pq, int bpl, int parts, int qmajor_q) {
    const int wv = bpl * pq, nq = (wv + 3) / 4;
    const int L = nb / (bpl * parts);
    const size_t total = (size_t)parts * num_rels * nq * L;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int l = (int)(i % L);
        size_t t = i / L;
        const int jq = (int)(t % nq);
        t /= nq;
        const int r = (int)(t % num_rels), part = (int)(t / num_rels);
        const float* blk = w + (size_t)r * nb * pq + ((size_t)part * L + l) * wv;
        const float* src = blk + 4 * jq;
        float4 v;
        if (qmajor_q > 0) {
            const int pw = pq / qmajor_q;                       // block is pw x qmajor_q, row-major
            float e4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * jq + c;                       // list element e = (q, p) with q = e / pw
                e4[c] = e < wv ? blk[(e % pw) * qmajor_q + e / pw] : 0.f;
            }
            v = make_float4(e4[0], e4[1], e4[2], e4[3]);
        } else {
            v.x = 4 * jq + 0 < wv ? src[0] : 0.f;
            v.y = 4 * jq + 1 < wv ? src[1] : 0.f;
            v.z = 4 * jq + 2 < wv ? src[2] : 0.f;
            v.w = 4 * jq + 3 < wv ? src[3] : 0.f;
        }
        out[i] = v;
    }
}

// instantiated shapes: blocks per lane as in k_agg_fast where the lane's gather then is 16 B (2-wide blocks: two per lane),
// one 5- or 10-wide block per lane otherwise; the fewest column parts that fit a part's lanes into one wave
bool phase_plan(int nb, int p, int q, bool trans, int k_req, int threads_req, PhasePlan* out) {
    // k = rows per wave (K * PV accumulator registers); M feature rows in flight (<= 32 registers), per instantiation below
    int bpl = 0, u = 8, k = 0, qmajor = 0;
    if (!trans) {
        if (p == 2 && q == 2) { bpl = 2; k = 8; }
        else if (p == 2 && q == 4) { bpl = 2; u = 8; k = 8; }   // two blocks per lane (one column part): 8 rows x 8 accumulators, 8 rows in flight (6 until the LDS-DMA source became a scalar base: 128 registers, no scratch; configs[4] scale 11.2 -> 11.0 ms)
        else if (p == 5 && q == 5) { bpl = 1; u = 6; k = 8; }     // one block per lane, two column parts: 8 rows x 5 accumulators, 6 rows in flight (4 until round 3: 272 -> 257 us at h = 500); 228 tile-parts = one round of workgroups at FB15k-237 size
        else if (p == 5 && q == 10) { bpl = 1; u = 2; k = 4; qmajor = 1; }     // lean kernel on a q-major packed block (5, 6 rows: spills)     // lean kernel on a q-major packed block
    } else {
        if (p == 2 && q == 2) { bpl = 2; k = 8; }
        else if (p == 4 && q == 2) { bpl = 2; u = 4; k = 8; }
        else if (p == 5 && q == 5) { bpl = 1; u = 6; k = 8; }
        else if (p == 10 && q == 5) { bpl = 1; u = 2; k = 8; }      // lean kernel: 8 rows x 5 accumulators, weights in two halves
    }
    if (!bpl || nb % bpl) return false;
    // workgroup size: 1 024 threads (16 waves of 128 registers) unless the caller names one.  (Measured and not kept for the 5x10
    // blocks, whose 10 accumulators per row leave such a wave 4 rows: 12 waves of 168 registers owning 11 rows each -- one round of
    // workgroups instead of two, 13 instead of 5 edges per wave and phase -- 493 us against 401: three waves per SIMD hide less
    // than four; 8 waves of 16 rows 641 us.)
    const int threads = threads_req > 0 ? threads_req : 1024;
    if (k_req && k_req != k) {
        if (trans && p == 10 && q == 5 && k_req == 3) { k = 3; u = 3; }      // the plain kernel, 3 rows per wave
        else if (k_req != 4 || k < 4) return false;       // the 8-row shapes also come with 4 rows per wave
        else {
            k = k_req;
            if (p == 5 && q == 5) u = 6;
        }
    }
    const int slots = nb / bpl;
    int parts = 0;
    for (int c = 1; c <= 16 && !parts; ++c)
        if (slots % c == 0 && slots / c <= 64) parts = c;
    if (!parts) return false;
    out->bpl = bpl; out->parts = parts; out->lanes = slots / parts; out->nq = (bpl * p * q + 3) / 4; out->k = k; out->u = u;
    out->qmajor = qmajor; out->threads = threads;
    return true;
}

}  // namespace gv

using namespace gv;

extern "C" int gv_rgcn_bdd_phase_plan(int num_bases, int blk_in, int blk_out, int transpose_w, int num_rels, int lds_bytes,
                                      int num_buffers, int rows_per_wave, int block_threads, int32_t* plan_host /*[7]*/) {
    PhasePlan pl;
    if (num_bases <= 0 || blk_in <= 0 || blk_out <= 0 || num_rels <= 0 || !plan_host) return 0;
    if (num_buffers != 1 && num_buffers != 2) return 0;
    if (block_threads < 0 || block_threads > 1024 || block_threads % 64) return 0;
    if (!phase_plan(num_bases, blk_in, blk_out, transpose_w != 0, rows_per_wave, block_threads, &pl)) return 0;
    const int rel_bytes = pl.nq * pl.lanes * 16;
    int g = (lds_bytes / num_buffers - 1024) / rel_bytes;
    if (g > num_rels) g = num_rels;
    if (g > 4096) g = 4096;
    if (g < 1) return 0;
    plan_host[0] = pl.bpl; plan_host[1] = pl.parts; plan_host[2] = pl.k; plan_host[3] = g;
    plan_host[4] = (num_rels + g - 1) / g;                       /* phases */
    plan_host[5] = pl.parts * num_rels * pl.nq * pl.lanes * 4 + 64 * 4;      /* floats of the packed weight buffer */
    plan_host[6] = pl.threads;
    return 1;
}

extern "C" int gv_rgcn_bdd_pack_weight_phase(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out,
                                             int transpose_w, float* packed, void* stream) {
    GV_REQUIRE(weight && packed, GV_ERR_NULL, "gv_rgcn_bdd_pack_weight_phase: NULL pointer");
    PhasePlan pl;
    GV_REQUIRE(phase_plan(num_bases, blk_in, blk_out, transpose_w != 0, 0, 0, &pl), GV_ERR_SHAPE,
               "gv_rgcn_bdd_pack_weight_phase: no phase kernel for num_bases=%d blocks %dx%d trans=%d", num_bases, blk_in,
               blk_out, transpose_w);
    GV_REQUIRE(aligned16(packed), GV_ERR_ALIGN, "gv_rgcn_bdd_pack_weight_phase: 16-B alignment required");
    const size_t total = (size_t)pl.parts * num_rels * pl.nq * pl.lanes;
    hipLaunchKernelGGL(k_pack_weight_phase, dim3((unsigned)min((size_t)2048, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, (float4*)packed, num_rels, num_bases, blk_in * blk_out, pl.bpl, pl.parts,
                       pl.qmajor ? blk_out : 0);
    return launch_status("gv_rgcn_bdd_pack_weight_phase");
}

extern "C" int gv_rgcn_bdd_aggregate_phases(const int32_t* off, const int32_t* nbr, const int32_t* meta, const float* coef,
                                            const int32_t* tile_items, int n_tiles, const int32_t* fix, int n_fix,
                                            const float* feat, int ld_feat, const float* weight_packed, int num_rels,
                                            int num_bases, int blk_in, int blk_out, int transpose_w, int rows_per_wave,
                                            int rels_per_phase, int num_buffers, int block_threads, const float* addend,
                                            int ld_addend, int act,
                                            const uint8_t* keep, float keep_scale, float* out, int ld_out, float* partial,
                                            void* stream) {
    GV_REQUIRE(n_tiles >= 0 && n_fix >= 0, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_phases: negative count");
    if (n_tiles == 0) return GV_OK;
    GV_REQUIRE(off && tile_items && feat && weight_packed && out, GV_ERR_NULL, "gv_rgcn_bdd_aggregate_phases: NULL pointer");
    GV_REQUIRE(num_bases > 0 && blk_in > 0 && blk_out > 0 && num_rels > 0, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_phases: num_bases=%d blk_in=%d blk_out=%d num_rels=%d", num_bases, blk_in, blk_out,
               num_rels);
    GV_REQUIRE(n_fix == 0 || (fix && partial), GV_ERR_NULL, "gv_rgcn_bdd_aggregate_phases: split rows need fix+partial");
    GV_REQUIRE(ld_feat >= num_bases * blk_in && ld_out >= num_bases * blk_out, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_phases: leading dimension smaller than the row");
    GV_REQUIRE(act == GV_ACT_NONE || act == GV_ACT_RELU, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_phases: unknown act %d", act);
    GV_REQUIRE(block_threads >= 64 && block_threads <= 1024 && block_threads % 64 == 0, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_phases: block_threads=%d", block_threads);
    PhasePlan pl;
    GV_REQUIRE(phase_plan(num_bases, blk_in, blk_out, transpose_w != 0, rows_per_wave, block_threads, &pl), GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_phases: no phase kernel for blocks %dx%d trans=%d num_bases=%d with %d rows per wave",
               blk_in, blk_out, transpose_w, num_bases, rows_per_wave);
    GV_REQUIRE(rels_per_phase >= 1 && rels_per_phase <= 4096, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_phases: rels_per_phase=%d",
               rels_per_phase);
    const int rel_quads = pl.nq * pl.lanes;
    const int slab = ((rels_per_phase * rel_quads + 63) / 64) * 64;
    GV_REQUIRE(num_buffers == 1 || num_buffers == 2, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_phases: num_buffers=%d", num_buffers);
    const size_t lds = (size_t)num_buffers * slab * 16 + 256;      // + the quads lanes past the part's last slot read (and ignore)
    GV_REQUIRE(lds <= 160 * 1024, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_phases: %d relations per phase need %zu B of LDS",
               rels_per_phase, lds);
    // 16-B accesses where a lane's gathered / stored vector is a multiple of 4 floats, 8-B where it is even, 4-B otherwise
    const int gv_f = pl.bpl * blk_in, pv_f = pl.bpl * blk_out;
    const bool al_ok = aligned16(feat) && aligned16(weight_packed) && aligned16(out) && (!addend || aligned16(addend)) &&
                       (!partial || aligned16(partial)) && (ld_feat % (gv_f % 4 == 0 ? 4 : gv_f % 2 == 0 ? 2 : 1) == 0) &&
                       (ld_out % (pv_f % 4 == 0 ? 4 : pv_f % 2 == 0 ? 2 : 1) == 0) &&
                       (!addend || ld_addend % (pv_f % 4 == 0 ? 4 : pv_f % 2 == 0 ? 2 : 1) == 0) &&
                       ((num_bases * blk_out) % (pv_f % 4 == 0 ? 4 : pv_f % 2 == 0 ? 2 : 1) == 0);
    GV_REQUIRE(al_ok, GV_ERR_ALIGN, "gv_rgcn_bdd_aggregate_phases: rows must be aligned to the lane's vector width");
    PhaseParams a;
    a.off = off; a.nbr = nbr; a.meta = meta; a.coef = coef; a.titems = (const int4*)tile_items;
    a.wpk = (const float4*)weight_packed; a.G = rels_per_phase; a.R = num_rels;
    a.n_phases = (num_rels + rels_per_phase - 1) / rels_per_phase;
    a.feat = feat; a.ld_feat = ld_feat; a.addend = addend; a.ld_add = ld_addend; a.act = act; a.keep = keep;
    a.keep_scale = keep_scale; a.out = out; a.ld_out = ld_out; a.partial = partial; a.out_dim = num_bases * blk_out;
    a.nbp = num_bases / pl.parts; a.L = pl.lanes; a.slab = slab; a.nbuf = num_buffers;
    { const char* e = getenv("GV_PHASE_DEBUG"); a.debug = e ? atoi(e) : 0; }
    a.parts = pl.parts; a.n_tiles = n_tiles; a.xcd_parts = 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(n_tiles, pl.parts), block(block_threads);
    int rc = -1000;
    // the streamed form (k_stream.hip; GV_PHASE_STREAM=0 keeps the batch-per-list kernel below): same lists, bit-identical sums
    {
        const char* e = getenv("GV_PHASE_STREAM");
        if (!(e && e[0] == '0')) rc = launch_phase_stream(a, pl, blk_in, blk_out, transpose_w != 0, grid, block, lds, st);
    }
#define GV_PHASE_CASE(P_, Q_, T_, B_, K_, U_) GV_PHASE_CASE_L(P_, Q_, T_, B_, K_, U_, false)
#define GV_PHASE_CASE_L(P_, Q_, T_, B_, K_, U_, LEAN_)  /* U_ = M: feature rows in flight */                                                                        \
    if (rc == -1000 && blk_in == P_ && blk_out == Q_ && (transpose_w != 0) == T_ && pl.bpl == B_ && pl.k == K_) {    \
        auto kern = k_agg_phase<P_, Q_, T_, B_, K_, U_, LEAN_>;                                                             \
        static unsigned long long lds_done = 0;                                                                      \
        if (!raise_dynamic_lds((const void*)kern, 160 * 1024, lds_done, "gv_rgcn_bdd_aggregate_phases")) return GV_ERR_SHAPE; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                           \
        rc = launch_status("gv_rgcn_bdd_aggregate_phases");                                                         \
    }
    GV_PHASE_CASE(2, 2, false, 2, 8, 8) GV_PHASE_CASE(2, 2, false, 2, 4, 8)
    GV_PHASE_CASE(2, 4, false, 2, 8, 8) GV_PHASE_CASE(2, 4, false, 2, 4, 6)
    GV_PHASE_CASE(2, 2, true, 2, 8, 8) GV_PHASE_CASE(2, 2, true, 2, 4, 8)
    GV_PHASE_CASE(4, 2, true, 2, 8, 4) GV_PHASE_CASE(4, 2, true, 2, 4, 4)
    GV_PHASE_CASE(5, 5, false, 1, 8, 6) GV_PHASE_CASE(5, 5, false, 1, 4, 6)
    GV_PHASE_CASE_L(5, 10, false, 1, 4, 2, true)
    GV_PHASE_CASE(5, 5, true, 1, 8, 6) GV_PHASE_CASE(5, 5, true, 1, 4, 6)
    GV_PHASE_CASE_L(10, 5, true, 1, 8, 2, true) GV_PHASE_CASE_L(10, 5, true, 1, 4, 2, true) GV_PHASE_CASE(10, 5, true, 1, 3, 3)
#undef GV_PHASE_CASE
#undef GV_PHASE_CASE_L
    GV_REQUIRE(rc != -1000, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_phases: no instantiation for blocks %dx%d trans=%d", blk_in,
               blk_out, transpose_w);
    if (rc != GV_OK) return rc;
    if (n_fix > 0)      // split (hub) rows: the ordered slot sum + epilogue of k_bdd.hip
        return gv_rgcn_bdd_fixup(fix, n_fix, partial, a.out_dim, addend, ld_addend, act, keep, keep_scale, out, ld_out, stream);
    return GV_OK;
}
