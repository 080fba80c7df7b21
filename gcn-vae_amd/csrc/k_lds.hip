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
};

__device__ __forceinline__ int lrl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lrl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One LDS-DMA wave-instruction: 64 x 16 B from per-lane global addresses to lds_byte_addr + lane*16 (see k_phase.hip)
__device__ __forceinline__ void lds_dma16(const float4* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}

// P = gathered block width, Q = output block width, QS = output columns per lane, BPP = diagonal blocks per column
// part, U = edges staged per batch (ring slots per wave)
template <int P, int Q, int QS, int BPP, int U>
__global__ __launch_bounds__(1024) void k_agg_lds(const LdsAggParams a) {
    constexpr int LPB = Q / QS, LANES = BPP * LPB, NW = P * QS, NQ = NW / 4;
    constexpr int PF = BPP * P;                            // floats of a feature row this part reads
    constexpr int VW = (PF % 4 == 0) ? 4 : 2;              // floats per staging lane
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
    const bool xl_on = lane < XL;
    const float* const fsrc = a.feat + part * PF + min(lane, XL - 1) * VW;
    const size_t ld = (size_t)a.ld_feat;
    const int col0 = (part * BPP + blk_l) * Q + sub * QS;
    const int nwaves = gridDim.x * nw;
    const int gw = blockIdx.x * nw + wv;

    auto load_meta = [&](int pos, int cnt, int& n_, int& t_, float& c_) {
        n_ = 0; t_ = 0; c_ = 1.f;
        if (lane < cnt) {
            n_ = a.nbr[pos + lane];
            t_ = a.etype[pos + lane];
            if (a.coef) c_ = a.coef_idx ? a.coef[a.coef_idx[pos + lane]] : a.coef[pos + lane];
        }
    };

    // this wave's items: gw, gw + nwaves, ... ; 64 descriptors per fetch (lane k = k-th of them)
    int4 itv = make_int4(-1, 0, 0, -1);
    {
        const long long i0 = (long long)gw + (long long)lane * nwaves;
        if (i0 < a.n_items) itv = a.items[i0];
    }
    int nx_n, nx_t;
    float nx_c;
    {
        const int y = lrl_i(itv.y, 0), z = lrl_i(itv.z, 0);
        load_meta(y, min(64, z - y), nx_n, nx_t, nx_c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the weight copy (the compiler does not count the DMAs)
    __syncthreads();

    for (long long ib = gw, k = 0; ib < a.n_items; ib += nwaves, ++k) {
        if (k == 64) {                                     // next 64 descriptors
            k = 0;
            itv = make_int4(-1, 0, 0, -1);
            const long long i0 = ib + (long long)lane * nwaves;
            if (i0 < a.n_items) itv = a.items[i0];
            const int y = lrl_i(itv.y, 0), z = lrl_i(itv.z, 0);
            load_meta(y, min(64, z - y), nx_n, nx_t, nx_c);
        }
        const int kk = (int)k;
        const int row = lrl_i(itv.x, kk), e_beg = lrl_i(itv.y, kk), e_end = lrl_i(itv.z, kk), slot = lrl_i(itv.w, kk);
        int my_n = nx_n, my_t = nx_t;
        float my_c = nx_c;
        if (kk + 1 < 64) {                                 // metadata of the wave's next item, one item ahead
            const int y = lrl_i(itv.y, kk + 1), z = lrl_i(itv.z, kk + 1);
            load_meta(y, min(64, z - y), nx_n, nx_t, nx_c);      // descriptor -1 (none): y = z = 0, nothing is read
        }
        if (row < 0 || (slot >= 0 && !a.partial)) continue;

        float acc[QS];
#pragma unroll
        for (int i = 0; i < QS; ++i) acc[i] = 0.f;

        for (int e0 = e_beg; e0 < e_end; e0 += 64) {
            const int cnt = min(64, e_end - e0);
            if (e0 != e_beg) load_meta(e0, cnt, my_n, my_t, my_c);      // slices longer than 64 edges: fetched in place
            for (int j = 0; j < cnt; j += U) {
                const int nb = min(U, cnt - j);
                // stage: U pieces requested together (edges beyond the batch re-read its last one: no branch between loads)
                float xs[U][VW];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s = lrl_i(my_n, min(j + u, cnt - 1));
                    load_vec<VW>(fsrc + (size_t)(unsigned)s * ld, xs[u]);
                }
                if (xl_on) {
#pragma unroll
                    for (int u = 0; u < U; ++u) store_vec<VW>(ring + u * PF + lane * VW, xs[u]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // other lanes of this wave read the pieces
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u < nb) {
                        const int r = lrl_i(my_t, j + u);
                        const float c = lrl_f(my_c, j + u);
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
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next batch overwrites the ring
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!active) continue;
        if (slot >= 0) {
            store_vec<QS>(a.partial + (size_t)slot * a.out_dim + col0, acc);
            continue;
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
