// K1 with ALL relation weights RESIDENT IN LDS: graphs with few relation types and few, wide diagonal blocks
// (BASELINE configs[2]: WN18RR, R = 22 directed types, num_bases = 20 -> 10x10 / 10x20 / 20x10 blocks).
//
// The per-row kernels of k_bdd.hip read, per EDGE, the whole block-weight row of the edge's relation (8-16 kB at this
// shape) through the L1 -> VGPR path for 812 B of algorithmic bytes; with R = 22 the whole table is 176 / 352 kB, so a
// column part of it (88 kB) stays in a CU's LDS for the whole launch.  At ~4 edges per row the other cost is per-ROW
// overhead, so the unit of work here is not a row but a SUPER-ITEM: a run of consecutive rows with <= 64 edges in all (or a
// <= 64-edge slice of a hub row), whose edge metadata is ONE coalesced fetch and whose rows cost an epilogue each, no more.
//   * grid = (workgroups, column parts); one 1 024-thread workgroup per CU copies its part of the lane-packed table
//     [R][NQ][CL] float4 into LDS once (LDS-DMA, 1 KiB per wave-instruction); waves take super-items in a strided order;
//   * lane = (diagonal block, group of IPL INPUT columns, half of the output columns): per edge it loads just its own IPL
//     inputs of the neighbour's row (8 / 16 B, the lanes of a block cover its inputs once -- nothing is exchanged between
//     lanes per edge), reads its IPL x OPL weights by NQ conflict-free ds_read_b128 and adds into OPL partial sums; the five
//     input groups of a block sit in adjacent lanes of one 16-lane DPP row and are summed ONCE PER ROW (3 row_shr adds per
//     output), the row's part then goes through 400 B of wave-private LDS into column order for a coalesced epilogue
//     (+ self-loop addend, ReLU, dropout mask) and store;
//   * finished rows wait in a wave-private LDS buffer (KB rows) and get their epilogue in BATCHES: the operands (addend,
//     keep) of all buffered rows are requested together, so the edge loop itself contains no load but the feature pieces
//     and no wait but theirs;
//   * software pipeline: the feature pieces of step s+1 (U edges) are requested before step s is computed (two register
//     sets with static names), the next super-item's metadata one super-item ahead.
// No atomics; every row is summed by one wave in a fixed order -> bitwise reproducible (the order differs from the per-row
// kernels': inputs are grouped before edges are summed).  Rows without edges are written by a second loop of the same launch.
#include <stdlib.h>

#include "common.h"

namespace gv {

struct LdsAggParams {
    const int4* sitems;      // {first edge position, end position, partial slot (-1: whole rows), 0}
    int n_sitems;
    const int* erow;         // [E] row of every edge position
    const int* empty;        // rows without edges
    int n_empty;
    const int* nbr;
    const int* etype;
    const float* coef;
    const int* coef_idx;
    const float* feat;
    int ld_feat;
    const float4* wpk;       // [parts][R][NQ][CL] float4 (+ 64 float4 of slack behind the table)
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
    int debug;               // ablation switch (GV_K1_LDS_DEBUG; 0 in production): 1 = no LDS weight reads / fmas
};

__device__ __forceinline__ int lrl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lrl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Register rotation by OPAQUE moves: left to hipcc, loop-carried values get their copies at the end of the block, i.e. behind
// the new loads, which then land in temporaries and are copied over behind a full wait.
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

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114;

template <int N> struct XVec { typedef float type __attribute__((ext_vector_type(N))); };

// P = gathered block width, Q = output block width, IPL = inputs per lane (P / IPL = 5 input groups per block), OH = lanes
// sharing a block's outputs (each Q / OH of them), BPP = diagonal blocks per column part, U = edges per step.
// Lane layout: slot = block * OH + half (<= 12 slots), three slots per 16-lane DPP row, lane = 16 * (slot / 3) + 5 * (slot % 3) + g.
template <int P, int Q, int IPL, int OH, int BPP, int U, int KB, bool BF>
__global__ __launch_bounds__(1024) void k_agg_lds(const LdsAggParams a) {
    // BF: bf16 operands, fp32 accumulate (BASELINE configs[2]'s precision): the table holds bf16 PAIRS (the weights of two
    // adjacent inputs into one output), an edge's inputs are scaled by its coefficient in fp32, rounded to bf16 pairs
    // (v_cvt_pk_bf16_f32, nearest even) and fed to v_dot2c_f32_bf16 -- half the LDS bytes per product, so twice the output
    // columns per part; feature rows and sums stay fp32 in memory.
    constexpr int NIG = P / IPL, OPL = Q / OH, NW = BF ? (IPL / 2) * OPL : IPL * OPL, NQ = (NW + 3) / 4, SLOTS = BPP * OH;
    // the weight table is indexed by LANE (lane-consecutive 16-B quads: conflict-free ds_read_b128), CL = lanes up to the last slot's
    constexpr int CL = 16 * ((SLOTS - 1) / 3) + 5 * ((SLOTS - 1) % 3) + NIG;
    constexpr int PO = BPP * Q;                            // output columns of a part
    constexpr int EW = PO / 50, EL = PO / EW;              // epilogue: 50 lanes x EW columns (1, 2 or 4)
    constexpr int XN = U * IPL;                            // KB: finished rows parked per wave before their epilogue
    static_assert(NIG == 5 && P % IPL == 0 && Q % OH == 0 && (BF || NW % 4 == 0) && SLOTS <= 12 && PO % 50 == 0 && (EW == 1 || EW == 2 || EW == 4), "lane mapping");
    static_assert(KB >= U + 2, "room for the rows one step can complete");
    static_assert(IPL == 2 || IPL == 4, "a lane's inputs are one 8- or 16-B load");
    typedef typename XVec<XN>::type xvec;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    const int part = blockIdx.y;
    const int tq = a.R * NQ * CL;                          // quads of this part's table
    const int tq_pad = (tq + 63) & ~63;

    {   // the part's weights: a straight copy, 1 KiB per wave-instruction; the slack behind the table absorbs the tail
        const float4* src = a.wpk + (size_t)part * tq;
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem);
        for (int i = wv * 64; i < tq; i += nw * 64)
            lds_dma16(src + i + lane, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)i * 16u));
    }

    // lanes without a slot (position 15 of a row, slots beyond the part's) shadow a real lane: same addresses, uniform control
    // flow; nothing they compute is read
    const int pos = min(lane & 15, 14), slot_raw = (lane >> 4) * 3 + pos / 5, g = pos % 5;
    const bool valid = (lane & 15) < 15 && slot_raw < SLOTS;
    const int slot_l = min(slot_raw, SLOTS - 1);
    const int blk_l = slot_l / OH, half = slot_l % OH;
    const int cl = valid ? lane : 16 * (slot_l / 3) + 5 * (slot_l % 3) + g;      // table column of this lane (shadow lanes: their twin's)
    const float4* const wl = smem + cl;
    float* const rbuf = reinterpret_cast<float*>(smem + tq_pad) + wv * (KB * PO);     // wave-private: KB finished rows (this part's columns)
    const float* const fsrc = a.feat + (part * BPP + blk_l) * P + g * IPL;
    const size_t ld = (size_t)a.ld_feat;
    const int el = min(lane, EL - 1);
    const int colE = part * PO + el * EW;
    const bool e_on = lane < EL;
    const int nwaves = gridDim.x * nw;
    const int gw = blockIdx.x * nw + wv;
    const bool has_add = a.addend != nullptr, has_keep = a.keep != nullptr;

    struct Meta { int n, t, r; float c; };
    struct Epi { float ad[EW]; unsigned kp; };
    auto load_meta = [&](int pos0, int cnt) {
        Meta m{0, 0, 0, 1.f};
        if (lane < cnt) {
            m.n = a.nbr[pos0 + lane];
            m.t = a.etype[pos0 + lane];
            m.r = a.erow[pos0 + lane];
            if (a.coef) m.c = a.coef_idx ? a.coef[a.coef_idx[pos0 + lane]] : a.coef[pos0 + lane];
        }
        return m;
    };
    // Epilogue operands of a row.  The loads are UNCONDITIONAL: an absent operand reads the output buffer instead (what was
    // read is selected away where it is used), a row id < 0 (the slice of a hub row: no epilogue) reads row 0.
    const float* const add_base = (has_add ? a.addend : a.out) + colE;
    const size_t add_ld = has_add ? (size_t)a.ld_add : (size_t)a.ld_out;
    const uint8_t* const keep_base = has_keep ? a.keep + colE : reinterpret_cast<const uint8_t*>(a.out + colE);
    const size_t keep_ld = has_keep ? (size_t)a.out_dim : (size_t)a.ld_out * 4;
    auto load_epi = [&](int row) {
        Epi e;
        const size_t r = (size_t)(unsigned)max(row, 0);
        load_vec<EW>(add_base + r * add_ld, e.ad);
        const uint8_t* kp = keep_base + r * keep_ld;
        if constexpr (EW == 4) e.kp = *reinterpret_cast<const uint32_t*>(kp);
        else if constexpr (EW == 2) e.kp = *reinterpret_cast<const uint16_t*>(kp);
        else e.kp = *kp;
        return e;
    };
    auto store_row = [&](int row, const Epi& e, float (&v)[EW]) {      // epilogue + store of the part's columns of one final row
#pragma unroll
        for (int i = 0; i < EW; ++i) {
            float t = apply_act(has_add ? v[i] + e.ad[i] : v[i], a.act);
            if (has_keep) t = ((e.kp >> (8 * i)) & 0xffu) ? t * a.keep_scale : 0.f;
            v[i] = t;
        }
        if (e_on) store_vec<EW>(a.out + (size_t)row * a.ld_out + colE, v);
    };
    // Finished rows: ids in `rid` (lane i = buffer slot i; id >= 0: a final row, -(slot + 1): the slice of a hub row that goes
    // raw to partial[slot]), their sums in rbuf.
    int rid = 0, nbuf = 0;
    // a row is complete: the block's five input groups summed (row_shr adds inside the 16-lane DPP row: valid at g = 4) and
    // the part of the row parked in column order
    auto park = [&](int id, const float (&acc)[OPL]) {
        float t[OPL];
#pragma unroll
        for (int o = 0; o < OPL; ++o) {
            // one instruction per add (hipcc pairs the adds into v_pk_add_f32 otherwise, which cannot take the DPP operand)
            float s1, s2;
            asm("v_add_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(s1) : "v"(acc[o]));
            asm("v_add_f32_dpp %0, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(s2) : "v"(s1));
            asm("v_add_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(t[o]) : "v"(acc[o]), "v"(s2));
        }
        if (valid && g == NIG - 1) store_vec<OPL>(rbuf + nbuf * PO + slot_l * OPL, t);
        rid = lane == nbuf ? id : rid;
        ++nbuf;
    };
    // epilogue of all parked rows: every row's operands requested first, then read back / finished / stored row by row
    auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the parked sums are read by other lanes of this wave
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Epi e[KB];
        int ids[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            ids[i] = lrl_i(rid, i);
            e[i] = load_epi(i < nbuf ? ids[i] : 0);
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if (i < nbuf) {
                float v[EW];
                load_vec<EW>(rbuf + i * PO + el * EW, v);
                if (ids[i] >= 0) store_row(ids[i], e[i], v);
                else if (e_on) store_vec<EW>(a.partial + (size_t)(-ids[i] - 1) * a.out_dim + colE, v);
            }
        }
        nbuf = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the buffer is free again
        __builtin_amdgcn_wave_barrier();
    };

    // Super-items are dealt to the workgroups in a strided order (workgroup b takes b, b + gridDim.x, ...: hub slices and
    // short runs spread evenly) and, inside a workgroup, to whichever wave asks next: the wave's first two are fixed (wave w:
    // shares w and nw + w, requested while the weights are still landing), from share 2 nw on a counter in LDS hands them out.
    // A wave always knows three: A (running), B (metadata in registers), C (descriptor requested).
    int* const ctr = reinterpret_cast<int*>(reinterpret_cast<float*>(smem + tq_pad) + nw * (KB * PO));
    if (threadIdx.x == 0) *ctr = 2 * nw;
    const long long wg0 = blockIdx.x, wgs = gridDim.x;
    auto fetch = [&](int share, int& slot, int& beg, int& cnt) {      // share = this workgroup's share-th super-item (uniform)
        const long long id = wg0 + (long long)share * wgs;
        beg = 0; cnt = 0; slot = -1;
        if (id < a.n_sitems) {
            const int4 d = a.sitems[id];                               // a uniform address: a scalar load
            beg = d.x; slot = d.z;
            cnt = max(0, min(64, d.y - d.x));
            if (slot >= 0 && !a.partial) cnt = 0;
        }
        return id < a.n_sitems;
    };
    auto grab = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(ctr, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    int slotA, begA, cntA, slotB, begB, cntB, slotC, begC, cntC;
    bool moreA = fetch(wv, slotA, begA, cntA);
    bool moreB = fetch(nw + wv, slotB, begB, cntB);
    Meta mA = load_meta(begA, cntA), mB = load_meta(begB, cntB);
    auto issue_x = [&](int nsrc, int j, int cnt, xvec& xs) {       // U loads back to back, no branch between them
        const int last = max(cnt, 1) - 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = lrl_i(nsrc, min(j + u, last));
            float tmp[IPL];
            load_vec<IPL>(fsrc + (size_t)(unsigned)s * ld, tmp);
#pragma unroll
            for (int i = 0; i < IPL; ++i) xs[u * IPL + i] = tmp[i];
        }
    };
    xvec x0, x1;
    issue_x(mA.n, 0, cntA, x0);
    // rows without edges: epilogue of a zero aggregate, four rows' operands in flight -- done FIRST, while the weight copy
    // lands (they need no weights)
    for (long long base = gw; base < a.n_empty; base += 64ll * nwaves) {
        int myrow = -1;
        {
            const long long i0 = base + (long long)lane * nwaves;
            if (i0 < a.n_empty) myrow = a.empty[i0];
        }
        const int nk = (int)min(64ll, (a.n_empty - base + nwaves - 1) / nwaves);
        for (int k = 0; k < nk; k += 4) {
            int rows[4];
            Epi e[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rows[t] = (k + t < nk) ? lrl_i(myrow, min(k + t, 63)) : -1;
                e[t] = load_epi(rows[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (rows[t] >= 0) {
                    float v[EW];
#pragma unroll
                    for (int i = 0; i < EW; ++i) v[i] = 0.f;
                    store_row(rows[t], e[t], v);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the weight copy (the compiler does not count the DMAs)
    __syncthreads();
    bool moreC = fetch(grab(), slotC, begC, cntC);
    float acc[OPL];
#pragma unroll
    for (int o = 0; o < OPL; ++o) acc[o] = 0.f;
    int j = 0, cur_row = -1;
    // one step: request the NEXT step's pieces into xn, run this step out of xc; the two register sets swap roles by the call
    // sequence below (static names: no copies, so the wait in front of a step never covers the loads just issued)
    auto step = [&](const xvec& xc, xvec& xn) -> bool {
        const int nb = (a.debug & 1) ? 0 : min(U, cntA - j);
        const bool same = j + U < cntA;                            // the next step is in this super-item, else the next one's first
        issue_x(same ? mA.n : mB.n, same ? j + U : 0, same ? cntA : cntB, xn);
#pragma unroll
        for (int u = 0; u < U; ++u) {                              // unrolled: the pieces are picked by static register names
            if (u >= nb) break;
            const int p = j + u;
            const int r = lrl_i(mA.r, p);
            if (r != cur_row) {                                    // first edge of a row: the previous one is complete
                if (cur_row >= 0) park(cur_row, acc);
#pragma unroll
                for (int o = 0; o < OPL; ++o) acc[o] = 0.f;
                cur_row = r;
            }
            const int rel = lrl_i(mA.t, p);
            const float c = lrl_f(mA.c, p);
            const float4* wq = wl + (size_t)rel * (NQ * CL);
            float wr[NQ * 4];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4) {
                const float4 t = wq[q4 * CL];
                wr[4 * q4] = t.x; wr[4 * q4 + 1] = t.y; wr[4 * q4 + 2] = t.z; wr[4 * q4 + 3] = t.w;
            }
            if constexpr (BF) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int k2 = 0; k2 < IPL / 2; ++k2) {
                    const f32x2 xs = {xc[u * IPL + 2 * k2] * c, xc[u * IPL + 2 * k2 + 1] * c};
                    const bf16x2 xp = __builtin_convertvector(xs, bf16x2);
#pragma unroll
                    for (int o = 0; o < OPL; ++o) {
                        bf16x2 wp;
                        __builtin_memcpy(&wp, &wr[k2 * OPL + o], 4);
                        acc[o] = __builtin_amdgcn_fdot2_f32_bf16(xp, wp, acc[o], false);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < IPL; ++i) {
                    const float xs = xc[u * IPL + i] * c;
#pragma unroll
                    for (int o = 0; o < OPL; ++o) acc[o] = fmaf(xs, wr[i * OPL + o], acc[o]);
                }
            }
        }
        j += U;
        bool done = false;
        if (j >= cntA) {                                           // the super-item's last row (or the slice of a hub row)
            if (cur_row >= 0) park(slotA >= 0 ? -(slotA + 1) : cur_row, acc);
#pragma unroll
            for (int o = 0; o < OPL; ++o) acc[o] = 0.f;
            cur_row = -1;
            if (!moreB) done = true;
            else {
                slotA = slotB; begA = begB; cntA = cntB; moreA = moreB;
                rot_i(mA.n, mB.n); rot_i(mA.t, mB.t); rot_i(mA.r, mB.r); rot_f(mA.c, mB.c);
                slotB = slotC; begB = begC; cntB = cntC; moreB = moreC;
                mB = load_meta(begB, cntB);
                moreC = moreC && fetch(grab(), slotC, begC, cntC);
                if (!moreC) cntC = 0;
                j = 0;
            }
        }
        if (nbuf + U + 1 > KB || done) drain();                    // room for every row the next step can complete
        return done;
    };
    if (moreA) {
        for (;;) {
            if (step(x0, x1)) break;
            if (step(x1, x0)) break;
        }
    }
}

// row layout [R][nb * bi * bo] -> [parts][R][NQ][CL] float4, CL = the kernel's lanes up to the last used one.  Lane (slot, g) of a part owns block
// part*BPP + slot / OH, its inputs g*IPL .. and its outputs half*OPL .. (half = slot % OH); its list is input-major: element
// i*OPL + o multiplies input g*IPL + i into output half*OPL + o.  Stored block: P x Q row-major for the plain product,
// Q x P (read transposed) for transpose_w.
__device__ __forceinline__ unsigned bf16_bits_rne(float f) {       // round to nearest even, as torch's .to(bfloat16) and v_cvt_pk_bf16_f32
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {f, 0.f};
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &b, 4);
    return u & 0xffffu;
}

__global__ __launch_bounds__(256) void k_pack_weight_lds(const float* __restrict__ w, float4* __restrict__ out, int num_rels,
                                                         int nb, int P, int Q, int IPL, int OH, int BPP, int CL, int trans, int bf) {
    const int OPL = Q / OH, NW = bf ? (IPL / 2) * OPL : IPL * OPL, NQ = (NW + 3) / 4, NIG = P / IPL, SLOTS = BPP * OH, parts = nb / BPP;
    const size_t total = (size_t)parts * num_rels * NQ * CL;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx % CL);
        size_t t = idx / CL;
        const int jq = (int)(t % NQ);
        t /= NQ;
        const int r = (int)(t % num_rels), part = (int)(t / num_rels);
        const int pos = lane & 15, slot = (lane >> 4) * 3 + pos / NIG, g = pos % NIG;
        float e4[4] = {0.f, 0.f, 0.f, 0.f};
        if (pos < 15 && slot < SLOTS) {                 // lanes without a slot: zeros
            const int blk = part * BPP + slot / OH, half = slot % OH;
            const float* wb = w + ((size_t)r * nb + blk) * (P * Q);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * jq + c;
                if (e >= NW) continue;
                if (bf) {                               // dword e = (input pair k2, output o): inputs g*IPL + 2 k2 (low half), + 1
                    const int k2 = e / OPL, o = e % OPL, in = g * IPL + 2 * k2, col = half * OPL + o;
                    const float lo = trans ? wb[col * P + in] : wb[in * Q + col];
                    const float hi = trans ? wb[col * P + in + 1] : wb[(in + 1) * Q + col];
                    e4[c] = __uint_as_float(bf16_bits_rne(lo) | (bf16_bits_rne(hi) << 16));
                } else {
                    const int i = e / OPL, o = e % OPL, in = g * IPL + i, col = half * OPL + o;
                    e4[c] = trans ? wb[col * P + in] : wb[in * Q + col];
                }
            }
        }
        out[idx] = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
}

namespace {
struct LdsPlan { int ipl, oh, bpp, parts, cl, nq, u, kb, po; };
constexpr int LDS_WAVES = 16;
constexpr int LDS_BUDGET = 160 * 1024;
constexpr int LDS_SITEM_EDGES = 64;

// instantiated shapes (gathered block width, output block width, bf16 operands); transpose_w only changes the packing
bool lds_plan(int nb, int p, int q, int num_rels, bool bf, LdsPlan* out) {
    int ipl = 0, oh = 0, bpp = 0, u = 4, kb = 8;
    if (!bf) {
        if (p == 10 && q == 10) { ipl = 2; oh = 1; bpp = 10; }
        else if (p == 10 && q == 20) { ipl = 2; oh = 2; bpp = 5; }
        else if (p == 20 && q == 10) { ipl = 4; oh = 2; bpp = 5; }
        else return false;
    } else {      // half the LDS bytes per product: twice the output columns per part where the lanes allow
        if (p == 10 && q == 10) { ipl = 2; oh = 1; bpp = 10; }
        else if (p == 10 && q == 20) { ipl = 2; oh = 1; bpp = 10; u = 3; kb = 5; }
        else if (p == 20 && q == 10) { ipl = 4; oh = 1; bpp = 10; }
        else return false;
    }
    static const int u_env = getenv("GV_K1_LDS_U") ? atoi(getenv("GV_K1_LDS_U")) : 0;      // tuning knob: 6 = longer steps
    if (u_env == 6 && bpp * q == 100) { u = 6; kb = 10; }
    if (nb % bpp) return false;
    const int slots = bpp * oh, cl = 16 * ((slots - 1) / 3) + 5 * ((slots - 1) % 3) + p / ipl;      // table columns = lanes
    const int opl = q / oh, nw = bf ? (ipl / 2) * opl : ipl * opl, nq = (nw + 3) / 4, po = bpp * q;
    const size_t tq = (size_t)num_rels * nq * cl;
    const size_t lds = ((tq + 63) & ~(size_t)63) * 16 + (size_t)LDS_WAVES * kb * po * 4 + 16;
    if (lds > (size_t)LDS_BUDGET) return false;
    out->ipl = ipl; out->oh = oh; out->bpp = bpp; out->parts = nb / bpp; out->cl = cl; out->nq = nq; out->u = u; out->kb = kb;
    out->po = po;
    return true;
}
}  // namespace

}  // namespace gv

using namespace gv;

extern "C" int gv_rgcn_bdd_lds_plan(int num_bases, int blk_in, int blk_out, int num_rels, int bf16_operands,
                                    int32_t* plan_host /*[3]*/) {
    LdsPlan pl;
    if (num_bases <= 0 || blk_in <= 0 || blk_out <= 0 || num_rels <= 0) return 0;
    if (!lds_plan(num_bases, blk_in, blk_out, num_rels, bf16_operands != 0, &pl)) return 0;
    if (plan_host) {
        plan_host[0] = pl.parts;
        plan_host[1] = pl.parts * num_rels * pl.nq * pl.cl * 4 + 64 * 4;      /* floats of the packed weight buffer */
        plan_host[2] = LDS_SITEM_EDGES;                                        /* most edges a super-item may hold */
    }
    return 1;
}

extern "C" int gv_rgcn_bdd_pack_weight_lds(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out,
                                           int transpose_w, int bf16_operands, float* packed, void* stream) {
    GV_REQUIRE(weight && packed, GV_ERR_NULL, "gv_rgcn_bdd_pack_weight_lds: NULL pointer");
    LdsPlan pl;
    GV_REQUIRE(num_bases > 0 && num_rels > 0 && lds_plan(num_bases, blk_in, blk_out, num_rels, bf16_operands != 0, &pl),
               GV_ERR_SHAPE, "gv_rgcn_bdd_pack_weight_lds: no LDS-resident kernel for num_bases=%d blocks %dx%d with %d relations",
               num_bases, blk_in, blk_out, num_rels);
    GV_REQUIRE(aligned16(packed), GV_ERR_ALIGN, "gv_rgcn_bdd_pack_weight_lds: 16-B alignment required");
    const size_t total = (size_t)pl.parts * num_rels * pl.nq * pl.cl;
    hipLaunchKernelGGL(k_pack_weight_lds, dim3((unsigned)min((size_t)2048, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, (float4*)packed, num_rels, num_bases, blk_in, blk_out, pl.ipl, pl.oh, pl.bpp,
                       pl.cl, transpose_w ? 1 : 0, bf16_operands ? 1 : 0);
    return launch_status("gv_rgcn_bdd_pack_weight_lds");
}

extern "C" int gv_rgcn_bdd_aggregate_lds(const int32_t* sitems, int n_sitems, const int32_t* erow, const int32_t* empty_rows,
                                         int n_empty, const int32_t* fix, int n_fix, const int32_t* nbr, const int32_t* etype,
                                         const float* coef, const int32_t* coef_idx, const float* feat, int ld_feat,
                                         const float* weight_packed, int num_rels, int num_bases, int blk_in, int blk_out,
                                         int bf16_operands, const float* addend, int ld_addend, int act, const uint8_t* keep,
                                         float keep_scale, float* out, int ld_out, float* partial, int max_workgroups,
                                         void* stream) {
    GV_REQUIRE(n_sitems >= 0 && n_fix >= 0 && n_empty >= 0, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_lds: negative count");
    if (n_sitems == 0 && n_empty == 0) return GV_OK;
    GV_REQUIRE(feat && weight_packed && out && (n_sitems == 0 || (sitems && erow && nbr && etype)) && (n_empty == 0 || empty_rows),
               GV_ERR_NULL, "gv_rgcn_bdd_aggregate_lds: NULL pointer");
    GV_REQUIRE(n_fix == 0 || (fix && partial), GV_ERR_NULL, "gv_rgcn_bdd_aggregate_lds: split rows need fix+partial");
    GV_REQUIRE(act == GV_ACT_NONE || act == GV_ACT_RELU, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_lds: unknown act %d", act);
    const bool bf = bf16_operands != 0;
    LdsPlan pl;
    GV_REQUIRE(num_bases > 0 && num_rels > 0 && lds_plan(num_bases, blk_in, blk_out, num_rels, bf, &pl), GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_lds: no LDS-resident kernel for num_bases=%d blocks %dx%d with %d relations", num_bases,
               blk_in, blk_out, num_rels);
    GV_REQUIRE(ld_feat >= num_bases * blk_in && ld_out >= num_bases * blk_out, GV_ERR_SHAPE,
               "gv_rgcn_bdd_aggregate_lds: leading dimension smaller than the row");
    const int out_dim = num_bases * blk_out;
    // 8- / 16-B pieces of the feature rows, 8- or 4-B stores: row bases 16-B aligned, leading dimensions multiples of 4 / 2
    const int ew = pl.po / 50 >= 2 ? pl.po / 50 : 2;      // widest epilogue access in floats
    const bool al_ok = aligned16(feat) && aligned16(weight_packed) && aligned16(out) && ld_feat % 4 == 0 && ld_out % ew == 0 &&
                       (!addend || (aligned16(addend) && ld_addend % ew == 0)) && (!partial || aligned16(partial)) &&
                       out_dim % ew == 0;
    GV_REQUIRE(al_ok, GV_ERR_ALIGN, "gv_rgcn_bdd_aggregate_lds: rows must be 16-B aligned");
    LdsAggParams a;
    a.sitems = (const int4*)sitems; a.n_sitems = n_sitems; a.erow = erow; a.empty = empty_rows; a.n_empty = n_empty;
    a.nbr = nbr; a.etype = etype; a.coef = coef; a.coef_idx = coef_idx;
    a.feat = feat; a.ld_feat = ld_feat; a.wpk = (const float4*)weight_packed; a.R = num_rels; a.addend = addend;
    a.ld_add = ld_addend; a.act = act; a.keep = keep; a.keep_scale = keep_scale; a.out = out; a.ld_out = ld_out;
    a.partial = partial; a.out_dim = out_dim;
    static const int dbg = getenv("GV_K1_LDS_DEBUG") ? atoi(getenv("GV_K1_LDS_DEBUG")) : 0;
    a.debug = dbg;
    hipStream_t st = (hipStream_t)stream;
    const size_t tq = (size_t)num_rels * pl.nq * pl.cl;
    const size_t lds = ((tq + 63) & ~(size_t)63) * 16 + (size_t)LDS_WAVES * pl.kb * pl.po * 4 + 16;
    // one workgroup per CU over all column parts (its waves share a counter that deals out the workgroup's super-items)
    const int n_cu = current_device_cus();
    int wgs = (max_workgroups > 0 ? max_workgroups : n_cu) / pl.parts;
    const int longest = n_sitems > n_empty ? n_sitems : n_empty;
    const int need = (longest + LDS_WAVES - 1) / LDS_WAVES;
    if (wgs > need) wgs = need;
    if (wgs < 1) wgs = 1;
    const dim3 grid(wgs, pl.parts), block(64 * LDS_WAVES);
    int rc = -1000;
#define GV_LDS_CASE(P_, Q_, IPL_, OH_, BPP_, U_, KB_, BF_)                                                              \
    if (rc == -1000 && blk_in == P_ && blk_out == Q_ && bf == BF_ && pl.ipl == IPL_ && pl.oh == OH_ && pl.bpp == BPP_ && \
        pl.u == U_ && pl.kb == KB_) {                                                                                   \
        auto kern = k_agg_lds<P_, Q_, IPL_, OH_, BPP_, U_, KB_, BF_>;                                                   \
        static unsigned long long lds_done = 0;                                                                         \
        if (!raise_dynamic_lds((const void*)kern, LDS_BUDGET, lds_done, "gv_rgcn_bdd_aggregate_lds")) return GV_ERR_SHAPE; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                              \
        rc = launch_status("gv_rgcn_bdd_aggregate_lds");                                                               \
    }
    GV_LDS_CASE(10, 10, 2, 1, 10, 4, 8, false)
    GV_LDS_CASE(10, 10, 2, 1, 10, 6, 10, false)
    GV_LDS_CASE(10, 20, 2, 2, 5, 6, 10, false)
    GV_LDS_CASE(10, 10, 2, 1, 10, 6, 10, true)
    GV_LDS_CASE(20, 10, 4, 1, 10, 6, 10, true)
    GV_LDS_CASE(10, 20, 2, 2, 5, 4, 8, false)
    GV_LDS_CASE(20, 10, 4, 2, 5, 4, 8, false)
    GV_LDS_CASE(10, 10, 2, 1, 10, 4, 8, true)
    GV_LDS_CASE(10, 20, 2, 1, 10, 3, 5, true)
    GV_LDS_CASE(20, 10, 4, 1, 10, 4, 8, true)
#undef GV_LDS_CASE
    GV_REQUIRE(rc != -1000, GV_ERR_SHAPE, "gv_rgcn_bdd_aggregate_lds: no instantiation for blocks %dx%d", blk_in, blk_out);
    if (rc != GV_OK) return rc;
    if (n_fix > 0)
        return gv_rgcn_bdd_fixup(fix, n_fix, partial, out_dim, addend, ld_addend, act, keep, keep_scale, out, ld_out, stream);
    return GV_OK;
}
