// K1 with ALL relation weights RESIDENT IN LDS: graphs with few relation types and few, wide diagonal blocks
// (BASELINE configs[2]: WN18RR, R = 22 directed types, num_bases = 20 -> 10x10 / 10x20 / 20x10 blocks).
//
// The per-row kernels of k_bdd.hip read, per EDGE, the whole block-weight row of the edge's relation (8-16 kB at this
// shape) through the L1 -> VGPR path for 812 B of algorithmic bytes; with R = 22 the whole table is 176 / 352 kB, so a
// column part of it (88 kB) stays in a CU's LDS for the whole launch:
//   * grid = (workgroups, column parts); one 1 024-thread workgroup per CU copies its part of the lane-packed table
//     [R][NQ][LANES] float4 into LDS once (LDS-DMA, 1 KiB per wave-instruction) and then walks work items
//     (destination rows / <= chunk-edge slices of hub rows, the same int4 lists as gv_rgcn_bdd_aggregate) in a
//     wave-strided order, so hub slices spread over all waves;
//   * a block-diagonal product only needs the part's own input columns: per edge and part ONE coalesced piece of the
//     feature row (PF = blocks-per-part x P floats: 400 or 200 B) is loaded once (16 / 8 B per lane, no duplicates),
//     parked in a wave-private LDS ring and read back block-wise (lanes of a block broadcast-read its P inputs);
//   * lane = (block, QS output columns): P x QS weights per edge by NQ conflict-free ds_read_b128 (lane-consecutive
//     quads), QS register accumulators for the whole row, epilogue (+ self-loop addend, ReLU, dropout mask) fused
//     into the single store of the row.  No atomics; same fma chains as k_agg_split -> bit-identical results for the
//     same work-item lists.
// Item descriptors are fetched 64 per wave-instruction (lane k = the wave's k-th item) and an item's edge metadata
// one item ahead, so a wave's dependent chain per item is the feature gather alone.
#include <stdlib.h>

#include "common.h"

namespace gv {

struct LdsAggParams {
    const int4* items;
    int n_items;
    const int* nbr;
    const int* etype;
    const float* coef;
    const int* coef_idx;
    const float* feat;
    int ld_feat;
    const float4* wpk;       // [parts][R][NQ][LANES] float4 (+ 64 float4 of slack behind the table)
    int R;
    const float* addend;
    int ld_add;
    int act;
    const uint8_t* keep;
    float keep_scale;
    float* out;
    int ld_out;
    float* partial;
    int out_dim;
    int debug;               // ablation switches (GV_K1_LDS_DEBUG; 0 in production): 1 = no LDS reads / fmas, 2 = no ring writes
};

__device__ __forceinline__ int lrl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lrl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ void rot_i(int& d, int s) { asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(s) : "memory"); }
__device__ __forceinline__ void rot_f(float& d, float s) { asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(s) : "memory"); }

// One LDS-DMA wave-instruction: 64 x 16 B from per-lane global addresses to lds_byte_addr + lane*16 (see k_phase.hip)
__device__ __forceinline__ void lds_dma16(const float4* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}

// P = gathered block width, Q = output block width, QS = output columns per lane, BPP = diagonal blocks per column
// part, U = edges per step (ring slots per wave).  Work items must not be longer than 64 edges (one metadata fetch each).
//
// A wave's work is a sequence of STEPS: a step = up to U consecutive edges of one item (an item without edges is one empty
// step).  Software pipeline, everything one stage ahead of its use:
//   descriptors   64 items per fetch (lane k = the wave's k-th item)
//   edge metadata of item k+2          requested when item k starts
//   epilogue operands (addend, keep)   of item k+1, requested when item k starts
//   feature pieces of step s+1         requested (two register sets used in turn: no copies) before step s is computed
// so the only wait in front of a step is for loads that had a whole step of LDS reads and fmas to land, and the three
// other waves of the SIMD fill what is left.
template <int P, int Q, int QS, int BPP, int U>
__global__ __launch_bounds__(1024) void k_agg_lds(const LdsAggParams a) {
    constexpr int LPB = Q / QS, LANES = BPP * LPB, NW = P * QS, NQ = NW / 4;
    constexpr int PF = BPP * P;                            // floats of a feature row this part reads
    constexpr int VW = (PF % 100 == 0) ? 2 : 1;            // floats per staging lane: 50 lanes x 8 B / 50 lanes x 4 B
    constexpr int XL = PF / VW;                            // staging lanes
    static_assert(Q % QS == 0 && NW % 4 == 0 && LANES <= 64 && PF % VW == 0 && XL <= 64, "lane mapping");
    static_assert(P % 2 == 0, "block inputs are read back from LDS in 8- or 16-B pieces");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    const int part = blockIdx.y;
    const int tq = a.R * NQ * LANES;                       // quads of this part's table
    const int tq_pad = (tq + 63) & ~63;

    {   // the part's weights: a straight copy, 1 KiB per wave-instruction; the slack behind the table absorbs the tail
        const float4* src = a.wpk + (size_t)part * tq;
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem);
        for (int i = wv * 64; i < tq; i += nw * 64)
            lds_dma16(src + i + lane, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)i * 16u));
    }

    const bool active = lane < LANES;
    const int ln = min(lane, LANES - 1);                   // lanes beyond the part shadow the last one (uniform control flow)
    const int blk_l = ln / LPB, sub = ln % LPB;
    float* const ring = reinterpret_cast<float*>(smem + tq_pad) + wv * (U * PF);
    const float4* const wl = smem + ln;
    const float* const xr_base = ring + blk_l * P;
    float* const xw_base = ring + min(lane, XL - 1) * VW;
    const bool xl_on = lane < XL;
    const float* const fsrc = a.feat + part * PF + min(lane, XL - 1) * VW;
    const size_t ld = (size_t)a.ld_feat;
    const int col0 = (part * BPP + blk_l) * Q + sub * QS;
    const int nwaves = gridDim.x * nw;
    const int gw = blockIdx.x * nw + wv;
    const bool has_add = a.addend != nullptr, has_keep = a.keep != nullptr;

    struct Meta { int n, t; float c; };
    struct Epi { float ad[QS]; unsigned kp; };
    auto load_meta = [&](int pos, int cnt) {
        Meta m{0, 0, 1.f};
        if (lane < cnt) {
            m.n = a.nbr[pos + lane];
            m.t = a.etype[pos + lane];
            if (a.coef) m.c = a.coef_idx ? a.coef[a.coef_idx[pos + lane]] : a.coef[pos + lane];
        }
        return m;
    };
    // Epilogue operands of a row, requested one item ahead.  The loads are UNCONDITIONAL (a conditional load into a
    // pre-set register makes hipcc copy it behind a full wait): rows that need none read row 0, an absent operand reads the
    // output buffer instead; what was read is selected away where it is used (finish).
    const float* const add_base = (has_add ? a.addend : a.out) + col0;
    const size_t add_ld = has_add ? (size_t)a.ld_add : (size_t)a.ld_out;
    const uint8_t* const keep_base = has_keep ? a.keep + col0 : reinterpret_cast<const uint8_t*>(a.out + col0);
    const size_t keep_ld = has_keep ? (size_t)a.out_dim : (size_t)a.ld_out * 4;
    auto load_epi = [&](int row, int slot) {
        Epi e;
        const size_t r = (size_t)(unsigned)max(row, 0);
        load_vec<QS>(add_base + r * add_ld, e.ad);
        const uint8_t* kp = keep_base + r * keep_ld;
        if constexpr (QS == 2) e.kp = *reinterpret_cast<const uint16_t*>(kp);
        else if constexpr (QS == 4) e.kp = *reinterpret_cast<const uint32_t*>(kp);
        else {
            e.kp = 0;
#pragma unroll
            for (int i = 0; i < QS; ++i) e.kp |= (unsigned)kp[i] << (8 * i);
        }
        return e;
    };
    // the feature pieces of one step: U loads back to back, no branch between them (edges beyond the step re-read its last
    // one, a step without edges reads row 0)
    auto issue_x = [&](const Meta& m, int j, int cnt, float (&xs)[U][VW]) {
        const int last = max(cnt, 1) - 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = lrl_i(m.n, min(j + u, last));
            load_vec<VW>(fsrc + (size_t)(unsigned)s * ld, xs[u]);
        }
    };
    auto edge_fma = [&](int u, int r, float c, float (&acc)[QS]) {
        float xv[P], wr[NW];
        load_vec<P>(xr_base + u * PF, xv);
        const float4* wq = wl + (size_t)r * (NQ * LANES);
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            const float4 t = wq[q4 * LANES];
            wr[4 * q4] = t.x; wr[4 * q4 + 1] = t.y; wr[4 * q4 + 2] = t.z; wr[4 * q4 + 3] = t.w;
        }
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p) t = fmaf(xv[p], wr[q * P + p], t);
            acc[q] = fmaf(t, c, acc[q]);
        }
    };
    // park the step's pieces in the wave's ring and run its edges
    auto consume = [&](const Meta& m, int j, int nb, const float (&xs)[U][VW], float (&acc)[QS]) {
        if (xl_on && !(a.debug & 2)) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < nb) store_vec<VW>(xw_base + u * PF, xs[u]);
        }
        if (a.debug & 1) nb = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // other lanes of this wave read the pieces
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int u = 0;
        for (; u + 2 <= nb; u += 2) {                             // pairs: the second edge's LDS reads under the first one's fmas
            edge_fma(u, lrl_i(m.t, j + u), lrl_f(m.c, j + u), acc);
            edge_fma(u + 1, lrl_i(m.t, j + u + 1), lrl_f(m.c, j + u + 1), acc);
        }
        if (u < nb) edge_fma(u, lrl_i(m.t, j + u), lrl_f(m.c, j + u), acc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next step overwrites the ring
        __builtin_amdgcn_wave_barrier();
    };
    auto finish = [&](int row, int slot, const Epi& e, float (&acc)[QS]) {
        if (!active || row < 0) return;
        const bool fin = slot < 0;                                 // else: a slice of a hub row, summed by the fix-up pass
        float* dstp = fin ? a.out + (size_t)row * a.ld_out + col0 : a.partial + (size_t)slot * a.out_dim + col0;
#pragma unroll
        for (int i = 0; i < QS; ++i) {
            float v = apply_act(has_add ? acc[i] + e.ad[i] : acc[i], a.act);
            if (has_keep) v = ((e.kp >> (8 * i)) & 0xffu) ? v * a.keep_scale : 0.f;
            acc[i] = fin ? v : acc[i];
        }
        store_vec<QS>(dstp, acc);
    };

    bool first = true;
    for (long long base = gw; base < a.n_items; base += 64ll * nwaves) {
        // this wave's next 64 items: base, base + nwaves, ...
        int4 itv = make_int4(-1, 0, 0, -1);
        {
            const long long i0 = base + (long long)lane * nwaves;
            if (i0 < a.n_items) itv = a.items[i0];
        }
        const int nk = (int)min(64ll, (a.n_items - base + nwaves - 1) / nwaves);
        auto desc = [&](int k, int& row, int& slot, int& beg, int& cnt) {      // k >= nk: none
            const int kk = min(k, 63);
            row = lrl_i(itv.x, kk); slot = lrl_i(itv.w, kk); beg = lrl_i(itv.y, kk);
            cnt = min(64, lrl_i(itv.z, kk) - beg);
            if (k >= nk || row < 0) { row = -1; cnt = 0; }
            if (slot >= 0 && !a.partial) { row = -1; cnt = 0; }
        };
        int rowA, slotA, begA, cntA, rowB, slotB, begB, cntB, rowC, slotC, begC, cntC;
        desc(0, rowA, slotA, begA, cntA);
        desc(1, rowB, slotB, begB, cntB);
        desc(2, rowC, slotC, begC, cntC);
        Meta mA = load_meta(begA, cntA), mB = load_meta(begB, cntB), mC = load_meta(begC, cntC);
        Epi eA = load_epi(rowA, slotA), eB = load_epi(rowB, slotB);
        float x0[U][VW], x1[U][VW];
        issue_x(mA, 0, cntA, x0);
        if (first) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the weight copy (the compiler does not count the DMAs)
            __syncthreads();
            first = false;
        }
        float acc[QS];
#pragma unroll
        for (int i = 0; i < QS; ++i) acc[i] = 0.f;
        int k = 0, j = 0;
        // one step: request the NEXT step's pieces into xn, run this step out of xc; the two register sets swap roles by
        // the call sequence below (static names: no copies, so the wait in front of a step never covers the loads just issued)
        auto step = [&](const float (&xc)[U][VW], float (&xn)[U][VW]) -> bool {
            const int nb = min(U, cntA - j);
            const bool same = j + U < cntA;                        // the next step is in this item, else the next item's first
            Meta mN;
            mN.n = same ? mA.n : mB.n; mN.t = 0; mN.c = 0.f;
            issue_x(mN, same ? j + U : 0, same ? cntA : cntB, xn);
            consume(mA, j, nb, xc, acc);
            j += U;
            if (j < cntA) return false;
            finish(rowA, slotA, eA, acc);
#pragma unroll
            for (int i = 0; i < QS; ++i) acc[i] = 0.f;
            if (++k >= nk) return true;
            rowA = rowB; slotA = slotB; begA = begB; cntA = cntB;
            rowB = rowC; slotB = slotC; begB = begC; cntB = cntC;
            // The register rotation is done by OPAQUE moves, before the next requests are issued: left to hipcc, the loop-carried
            // values get their copies at the end of the block, i.e. behind the new loads, which then land in temporaries and are
            // copied over behind a full wait.
            rot_i(mA.n, mB.n); rot_i(mA.t, mB.t); rot_f(mA.c, mB.c);
            rot_i(mB.n, mC.n); rot_i(mB.t, mC.t); rot_f(mB.c, mC.c);
#pragma unroll
            for (int i = 0; i < QS; ++i) rot_f(eA.ad[i], eB.ad[i]);
            { int t; rot_i(t, (int)eB.kp); eA.kp = (unsigned)t; }
            desc(k + 2, rowC, slotC, begC, cntC);
            mC = load_meta(begC, cntC);
            eB = load_epi(rowB, slotB);
            j = 0;
            return false;
        };
        for (;;) {
            if (step(x0, x1)) break;
            if (step(x1, x0)) break;
        }
    }
    if (first) {                                                   // a wave without items still joins the staging barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// row layout [R][nb * bi * bo] -> [parts][R][NQ][LANES] float4.  Lane l of a part owns block part*BPP + l / LPB and its
// output columns sub*QS .. (sub = l % LPB); its list is q-major: element q*P + p multiplies input p into output q.
// Stored block: P x Q row-major for the plain product, Q x P (read transposed) for transpose_w.
__global__ __launch_bounds__(256) void k_pack_weight_lds(const float* __restrict__ w, float4* __restrict__ out, int num_rels,
                                                         int nb, int P, int Q, int QS, int BPP, int trans) {
    const int LPB = Q / QS, LANES = BPP * LPB, NQ = P * QS / 4, parts = nb / BPP;
    const size_t total = (size_t)parts * num_rels * NQ * LANES;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int l = (int)(i % LANES);
        size_t t = i / LANES;
        const int jq = (int)(t % NQ);
        t /= NQ;
        const int r = (int)(t % num_rels), part = (int)(t / num_rels);
        const int blk = part * BPP + l / LPB, sub = l % LPB;
        const float* wb = w + ((size_t)r * nb + blk) * (P * Q);
        float e4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 4 * jq + c, q = e / P, p = e % P, col = sub * QS + q;
            e4[c] = trans ? wb[col * P + p] : wb[p * Q + col];
        }
        out[i] = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
}

namespace {
struct LdsPlan { int qs, bpp, parts, lanes, nq, u, pf; };
constexpr int LDS_WAVES = 16;
constexpr int LDS_BUDGET = 160 * 1024;

// instantiated shapes (gathered block width, output block width); transpose_w only changes the packing
bool lds_plan(int nb, int p, int q, int num_rels, LdsPlan* out) {
    int qs = 0, bpp = 0, u = 4;
    if (p == 10 && q == 10) { qs = 2; bpp = 10; }
    else if (p == 10 && q == 20) { qs = 2; bpp = 5; }
    else if (p == 20 && q == 10) { qs = 1; bpp = 5; }
    else return false;
    if (nb % bpp) return false;
    const int lanes = bpp * (q / qs), nq = p * qs / 4, pf = bpp * p;
    const size_t tq = (size_t)num_rels * nq * lanes;
    const size_t lds = ((tq + 63) & ~(size_t)63) * 16 + (size_t)LDS_WAVES * u * pf * 4;
    if (lds > (size_t)LDS_BUDGET) return false;
    out->qs = qs; out->bpp = bpp; out->parts = nb / bpp; out->lanes = lanes; out->nq = nq; out->u = u; out->pf = pf;
    return true;
}
}  // namespace

}  // namespace gv

using namespace gv;

extern "C" int gv_rgcn_bdd_lds_plan(int num_bases, int blk_in, int blk_out, int num_rels, int32_t* plan_host /*[3]*/) {
    LdsPlan pl;
    if (num_bases <= 0 || blk_in <= 0 || blk_out <= 0 || num_rels <= 0) return 0;
    if (!lds_plan(num_bases, blk_in, blk_out, num_rels, &pl)) return 0;
    if (plan_host) {
        plan_host[0] = pl.parts;
        plan_host[1] = pl.parts * num_rels * pl.nq * pl.lanes * 4 + 64 * 4;      /* floats of the packed weight buffer */
        plan_host[2] = 64;                                                        /* preferred work-item chunk (edges) */
    }
    return 1;
}

extern "C" int gv_rgcn_bdd_pack_weight_lds(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out,
                                           int transpose_w, float* packed, void* stream) {
    GV_REQUIRE(weight && packed, GV_ERR_NULL, "gv_rgcn_bdd_pack_weight_lds: NULL pointer");
    LdsPlan pl;
    GV_REQUIRE(num_bases > 0 && num_rels > 0 && lds_plan(num_bases, blk_in, blk_out, num_rels, &pl), GV_ERR_SHAPE,
               "gv_rgcn_bdd_pack_weight_lds: no LDS-resident kernel for num_bases=%d blocks %dx%d with %d relations", num_bases,
               blk_in, blk_out, num_rels);
    GV_REQUIRE(aligned16(packed), GV_ERR_ALIGN, "gv_rgcn_bdd_pack_weight_lds: 16-B alignment required");
    const size_t total = (size_t)pl.parts * num_rels * pl.nq * pl.lanes;
    hipLaunchKernelGGL(k_pack_weight_lds, dim3((unsigned)min((size_t)2048, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, (float4*)packed, num_rels, num_bases, blk_in, blk_out, pl.qs, pl.bpp,
                       transpose_w ? 1 : 0);
    return launch_status("gv_rgcn_bdd_pack_weight_lds");
}

extern "C" int gv_rgcn_bdd_aggregate_lds(const int32_t* items, int n_items, const int32_t* fix, int n_fix,
                                         const int32_t* nbr, const int32_t* etype, const float* coef,
                                         const int32_t* coef_idx, const float* feat, int ld_feat, const float* weight_packed,
                                         int num_rels, int num_bases, int blk_in, int blk_out, const float* addend,
                                         int ld_addend, int act, const uint8_t* keep, float keep_scale, float* out, int ld_out,
                                         float* partial, int max_workgroups, void* stream) {
    GV_REQUIRE(n_items >= 0 && n_fix >= 0, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_lds: negative item count");
    if (n_items == 0) return GV_OK;
    GV_REQUIRE(items && feat && weight_packed && out, GV_ERR_NULL, "gv_rgcn_bdd_aggregate_lds: NULL pointer");
    GV_REQUIRE(n_fix == 0 || (fix && partial), GV_ERR_NULL, "gv_rgcn_bdd_aggregate_lds: split segments need fix+partial");
    GV_REQUIRE(act == GV_ACT_NONE || act == GV_ACT_RELU, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_lds: unknown act %d", act);
    LdsPlan pl;
    GV_REQUIRE(num_bases > 0 && num_rels > 0 && lds_plan(num_bases, blk_in, blk_out, num_rels, &pl), GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_lds: no LDS-resident kernel for num_bases=%d blocks %dx%d with %d relations", num_bases,
               blk_in, blk_out, num_rels);
    GV_REQUIRE(ld_feat >= num_bases * blk_in && ld_out >= num_bases * blk_out, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_lds: leading dimension smaller than the row");
    const int out_dim = num_bases * blk_out;
    // 16-B pieces of the feature rows, 8-B (QS = 2) or 4-B stores: row bases 16-B aligned, even leading dimensions
    const bool al_ok = aligned16(feat) && aligned16(weight_packed) && aligned16(out) && ld_feat % 4 == 0 && ld_out % 2 == 0 &&
                       (!addend || (aligned16(addend) && ld_addend % 2 == 0)) && (!partial || aligned16(partial)) &&
                       out_dim % 2 == 0;
    GV_REQUIRE(al_ok, GV_ERR_ALIGN, "gv_rgcn_bdd_aggregate_lds: rows must be 16-B aligned");
    LdsAggParams a;
    a.items = (const int4*)items; a.n_items = n_items; a.nbr = nbr; a.etype = etype; a.coef = coef; a.coef_idx = coef_idx;
    a.feat = feat; a.ld_feat = ld_feat; a.wpk = (const float4*)weight_packed; a.R = num_rels; a.addend = addend;
    a.ld_add = ld_addend; a.act = act; a.keep = keep; a.keep_scale = keep_scale; a.out = out; a.ld_out = ld_out;
    a.partial = partial; a.out_dim = out_dim;
    static const int dbg = getenv("GV_K1_LDS_DEBUG") ? atoi(getenv("GV_K1_LDS_DEBUG")) : 0;
    a.debug = dbg;
    hipStream_t st = (hipStream_t)stream;
    const size_t tq = (size_t)num_rels * pl.nq * pl.lanes;
    const size_t lds = ((tq + 63) & ~(size_t)63) * 16 + (size_t)LDS_WAVES * pl.u * pl.pf * 4;
    // one workgroup per CU over all column parts; never more workgroups than 16-item shares of the list
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n_cu = v;
        else
            n_cu = 256;
        (void)hipGetLastError();
    }
    int wgs = (max_workgroups > 0 ? max_workgroups : n_cu) / pl.parts;
    const int need = (n_items + LDS_WAVES - 1) / LDS_WAVES;
    if (wgs > need) wgs = need;
    if (wgs < 1) wgs = 1;
    const dim3 grid(wgs, pl.parts), block(64 * LDS_WAVES);
    int rc = -1000;
#define GV_LDS_CASE(P_, Q_, QS_, BPP_, U_)                                                                              \
    if (rc == -1000 && blk_in == P_ && blk_out == Q_ && pl.qs == QS_ && pl.bpp == BPP_ && pl.u == U_) {                 \
        auto kern = k_agg_lds<P_, Q_, QS_, BPP_, U_>;                                                                   \
        static bool attr_done = false;                                                                                  \
        if (!attr_done) {                                                                                               \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET) != hipSuccess) \
                (void)hipGetLastError();                                                                                \
            attr_done = true;                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                              \
        rc = launch_status("gv_rgcn_bdd_aggregate_lds");                                                               \
    }
    GV_LDS_CASE(10, 10, 2, 10, 4)
    GV_LDS_CASE(10, 20, 2, 5, 4)
    GV_LDS_CASE(20, 10, 1, 5, 4)
#undef GV_LDS_CASE
    GV_REQUIRE(rc != -1000, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_lds: no instantiation for blocks %dx%d", blk_in, blk_out);
    if (rc != GV_OK) return rc;
    if (n_fix > 0)
        return gv_rgcn_bdd_fixup(fix, n_fix, partial, out_dim, addend, ld_addend, act, keep, keep_scale, out, ld_out, stream);
    return GV_OK;
}
