// K4 fused, fp32: the CHAIN of masked-MLP products of one MADE pass (kgvae/flow_network.py:85-98) -- or its backward-x chain -- in
// ONE launch on the fp32 MFMA, walking only the parts of the masked weights that hold a non-zero.
//
// Why: the reference's own regime (kgvae/README.md:4-7: fp32, 3 IAF blocks, mini-batches of ~10 k nodes) runs 15 + 15 such chains
// of five 200-wide products per step; as a launch per product (gv_gemm_f32) that is ~300 launches of 19-25 us at 0.3 of the fp32
// MFMA peak, each re-reading its activations from memory.  Here a workgroup owns 64 rows for the whole chain and the activations
// stay in LDS between the layers.
//
// Unlike the bf16 chain (k_chain.hip), which is bound by latency and by the bytes it writes, this one is MFMA-bound
// (v_mfma_f32_32x32x2_f32: 64 cycles per issue, 1/16 of the bf16 rate), so what pays is (1) not multiplying zeros and (2) an even
// load on the four SIMDs:
//   * create_masks (kgvae/flow_network.py:65-83) makes every masked weight (block) lower-triangular.  The unit of skipping is a
//     GROUP: 8 consecutive k of one 32-column tile (= one 16-B fragment per lane = four MFMA steps).  gv_made_chain_f32_plan reads
//     the 0/1 masks once and writes, per (layer, column tile), the bit set of groups that hold a non-zero; the kernel walks the
//     set bits in ascending order -- the same fp32 fma chain as the full walk minus terms that are exactly +-0 (x + 0 == x for
//     every finite x once the sum is non-zero, and a sum that starts at +0 stays +0 under +-0 terms): bit-identical to
//     gv_gemm_f32 on finite inputs.
//   * a column tile of a layer is one unit of work (64 rows x 32 columns, two accumulators sharing the weight fragment); units
//     cost between 4 and 25 groups, so the plan also deals each layer's units to the workgroup's four waves (one per SIMD),
//     longest first to the least loaded wave.  FB15k-237 width 200: 28-30 group times per hidden layer instead of 50.
//
// Operands.  A (activations): LDS, row-major with the k of each group stored even-first ([k0 k2 k4 k6 | k1 k3 k5 k7]) so that the
// lane of k-half h reads the four values it feeds to four consecutive MFMA steps (k = 2 s + h) with ONE ds_read_b128; row pitch
// 8 j + 4 floats keeps the 16 lanes of a read group on 16 different bank quads.  B (weights): FRAGMENT-PACKED in global memory
// (gv_made_pack_weight_f32: [tile][group][lane] float4), loaded straight to registers four groups ahead of their use, across
// unit and layer boundaries (weights do not depend on the activations).  MFMA operand order (A, B): a lane owns one output COLUMN
// and 16 rows per accumulator, so every store instruction of the epilogue writes two 128-B row pieces.
#include <stdlib.h>

#include "common.h"

namespace gv {

typedef float f32x16c __attribute__((ext_vector_type(16)));

constexpr int C32_LISTS = 4;        // unit lists of a plan: one per SIMD
#ifndef GV_C32_FINE
#define GV_C32_FINE 0           /* 0 (default): units of 64 rows x 32 columns (two accumulators sharing the weight fragment), 8 waves =
                                 * 2 per SIMD, 25-group fragment sets.  1: units of 32 x 32 (ONE accumulator, the unit carries its row
                                 * half), 16 waves = 4 per SIMD, 13-group sets -- finer balance and shorter epilogues, but every
                                 * weight fragment is fetched twice and a unit is two chunks: measured 85 / 85 us per pass against
                                 * 76 / 83 (forward / backward-x), the step 5.31 vs 5.34 ms: not worth its 40 B of scratch.
                                 * (A third form -- the two waves of a SIMD sharing every 64 x 32 unit by row half, i.e. two
                                 * single-accumulator chains per SIMD -- ran 92 us: ~780 cycles per pair of groups instead of 512.) */
#endif
constexpr int C32_BM = GV_C32_FINE == 2 ? 32 : 64;        // rows per workgroup
constexpr bool C32_SINGLE = GV_C32_FINE != 0;             // one accumulator per unit
constexpr bool C32_HALVES = GV_C32_FINE == 1;             // ... the unit carries its row half (two units per tile)
constexpr int C32_WAVES = GV_C32_FINE == 1 ? 16 : GV_C32_FINE == 2 ? 4 : 8;   // waves w, w + 4, ... share a SIMD (a workgroup's waves go
                                                          // to the SIMDs cyclically) and take every (WAVES / 4)-th entry of list w & 3
constexpr int C32_SLOTS = C32_WAVES / 4;
constexpr int C32_THREADS = C32_WAVES * 64;
constexpr int C32_L = GV_CHAIN_MAX_LAYERS;
constexpr int C32_MAXT = GV_CHAIN32_MAX_TILES;            // column tiles per layer (n <= 512)
constexpr int C32_MAXU = 2 * C32_L * C32_MAXT;            // units per list in the plan buffer (global memory)
constexpr int C32_MAXU_LDS = 96;                          // ... of a chain that is launched (checked on the host): the LDS copy
constexpr int C32_CHG = GV_C32_FINE == 1 ? 13 : 25;            // groups per register set of weight fragments
// plan words: [0, 4) units per list; then 4 lists of C32_MAXU (layer << 8 | half << 7 | tile); then [layer][tile] group sets (lo, hi)
constexpr int C32_PLAN_LISTS = 4, C32_PLAN_SETS = C32_PLAN_LISTS + C32_LISTS * C32_MAXU;
constexpr int C32_PLAN_WORDS = C32_PLAN_SETS + 2 * C32_L * C32_MAXT;
static_assert(C32_PLAN_WORDS == GV_CHAIN32_PLAN_WORDS, "include/gcnvae.h states the plan size");
// the LDS copy is compact: counts, 4 lists of C32_MAXU_LDS, the group sets
constexpr int C32_LPLAN_SETS = C32_PLAN_LISTS + C32_LISTS * C32_MAXU_LDS, C32_LPLAN_WORDS = C32_LPLAN_SETS + 2 * C32_L * C32_MAXT;

struct Chain32Args {
    float* x;                    // [m][ldx]: input of layer 0 (passes: the first pass's slice of the stacked inputs)
    const int32_t* plan;         // gv_made_chain_f32_plan
    const int32_t* rows_dev;     // optional device scalar: rows [*rows_dev, m) are padding
    int ldx, m, n_layers, ld0, ld1, debug;
    gv_chain32_iaf iaf;          // mode 0: one pass, no update (gv_made_chain_f32)
    gv_chain32_layer L[C32_L];
};

// position of column c inside its row of an LDS tile (even k first inside every group of 8)
__device__ __forceinline__ int c32_pos(int c) { return (c & ~7) | ((c & 1) << 2) | ((c & 7) >> 1); }

// the plan is copied into LDS once and read there (wave-uniform reads through the LDS counter: a global read per unit would make
// every unit boundary wait for ALL weight fragments in flight -- loads share one in-order counter)
__device__ __forceinline__ int sload(const int32_t* p) { return __builtin_amdgcn_readfirstlane(*p); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. waits for every
// global store of the epilogue to be acknowledged and for the next unit's 25 weight fragments -- a microsecond per layer.  The
// LDS tile a layer writes is complete when every wave's LDS counter is zero; global outputs are not read again in this launch.
__device__ __forceinline__ void c32_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct C32Unit {
    int ui;                      // index into the list
    int layer, tile, half;       // half: rows 32 half .. 32 half + 31 of the tile (single-accumulator units)
    unsigned long long set;      // groups of the unit
};

// load entry `ui` of a list (layer = n_layers when the list is exhausted); plan = the compact LDS copy
__device__ __forceinline__ void c32_open(const int32_t* plan, int n_layers, const int32_t* list, int nu, C32Unit& c) {
    if (c.ui < nu) {
        const int e = sload(list + c.ui);
        c.layer = e >> 8;
        c.tile = e & 0x7f;
        c.half = (e >> 7) & 1;
        const int32_t* s = plan + C32_LPLAN_SETS + 2 * (c.layer * C32_MAXT + c.tile);
        c.set = (unsigned long long)(unsigned)sload(s) | ((unsigned long long)(unsigned)sload(s + 1) << 32);
    } else {
        c.layer = n_layers;
        c.tile = 0;
        c.half = 0;
        c.set = 0ull;
    }
}

// The fragment set is fenced at the start of a chunk (its loads were issued a whole unit earlier); the groups then run without a
// vector-memory wait.  All C32_CHG loads are issued unconditionally (groups past the chunk's last re-read its last fragment): a
// conditional load leaves a register half-defined, and the compiler answers a set of those with copies.
__device__ __forceinline__ void c32_landed(float4 (&q)[C32_CHG], unsigned& keep) {
#pragma unroll
    for (int i = 0; i < C32_CHG; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w));
    asm volatile("" : "+v"(keep));
}

// base: the tile's first fragment + lane; left: the groups still to do (the lowest C32_CHG of them are loaded)
__device__ __forceinline__ void c32_issue(float4 (&q)[C32_CHG], const float4* base, unsigned off, unsigned long long left) {
    int g = 0;
#pragma unroll
    for (int i = 0; i < C32_CHG; ++i) {
        if (left != 0ull) {
            g = __builtin_ctzll(left);
            left &= left - 1;
        }
        q[i] = base[off + g * 64];
    }
}

// up to C32_CHG groups (ascending) of `left` on the fragment set: per group one ds_read_b128 per accumulator (requested a group
// ahead) and four MFMA steps on two accumulators sharing the weight fragment.  Returns the groups not done.
__device__ __forceinline__ unsigned long long c32_mma(f32x16c& acc0, f32x16c& acc1, const float4 (&q)[C32_CHG], const float* a_lo,
                                                      const float* a_hi, unsigned long long left) {
    int g = __builtin_ctzll(left);
    float4 a0 = *reinterpret_cast<const float4*>(a_lo + 8 * g), a1 = a0;
    if (!C32_SINGLE) a1 = *reinterpret_cast<const float4*>(a_hi + 8 * g);
#pragma unroll
    for (int i = 0; i < C32_CHG; ++i)
        if (left != 0ull) {
            left &= left - 1;
            float4 n0 = a0, n1 = a1;
            if (i + 1 < C32_CHG && left != 0ull) {
                g = __builtin_ctzll(left);
                n0 = *reinterpret_cast<const float4*>(a_lo + 8 * g);
                if (!C32_SINGLE) n1 = *reinterpret_cast<const float4*>(a_hi + 8 * g);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, q[i].x, acc0, 0, 0, 0);
            if (!C32_SINGLE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, q[i].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, q[i].y, acc0, 0, 0, 0);
            if (!C32_SINGLE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, q[i].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, q[i].z, acc0, 0, 0, 0);
            if (!C32_SINGLE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, q[i].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, q[i].w, acc0, 0, 0, 0);
            if (!C32_SINGLE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, q[i].w, acc1, 0, 0, 0);
            a0 = n0;
            a1 = n1;
            __builtin_amdgcn_sched_barrier(0);
        }
    return left;
}

// A layer's descriptor held in SGPRs for the length of a unit.  The descriptors are kernel arguments indexed by a run-time layer
// number; left to itself the compiler re-reads a field from the kernarg buffer wherever it is used (45 scalar loads, each with its
// own wait, in one epilogue: 7 000 cycles).  v_readfirstlane makes each value opaque: loaded once per unit.
struct C32Layer {
    const float* mask;
    float* out_f32;
    int n, relu, accumulate, ldmask, ldc, has_bias;
};
__device__ __forceinline__ int c32_pin(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* c32_pin_ptr(T* q) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    // (through the global address space: a pointer rebuilt from integers is a generic one, and generic accesses are FLAT
    // instructions -- slower, and counted against the LDS counter too, i.e. waited for at every layer barrier)
    typedef __attribute__((address_space(1))) T GT;
    return (T*)(GT*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ C32Layer c32_layer(const gv_chain32_layer& L) {
    C32Layer r;
    r.mask = c32_pin_ptr(L.mask);
    r.out_f32 = c32_pin_ptr(L.out_f32);
    r.n = c32_pin(L.n); r.relu = c32_pin(L.relu); r.accumulate = c32_pin(L.accumulate); r.ldmask = c32_pin(L.ldmask); r.ldc = c32_pin(L.ldc);
    r.has_bias = c32_pin(L.bias != nullptr ? 1 : 0);
    return r;
}

// The epilogue of one unit.  LOADS: the layer reads in its epilogue (a backward layer's ReLU mask, an accumulating output); the
// other instance holds NO load and therefore no vector-memory wait at all -- as one body, the wait in front of the first use of a
// (possibly never loaded) mask value is executed on every path, and with the next unit's 25 fragments and the previous batch's
// stores in flight it costs a store round trip per batch of eight rows (measured: 10 000 cycles per epilogue instead of ~1 000).
template <bool LOADS>
__device__ __forceinline__ void c32_epilogue(const f32x16c& acc0, const f32x16c& acc1, const C32Layer& Ly, int tile, int m0, int m,
                                             float* An, int ldn, const float* bias_l, int l31, int lhi, int debug, int half) {
    // opaque copies: without them the compiler hoists the per-row 64-bit offsets out of the unit loop and spills them
    int le = l31, he = lhi;
    asm volatile("" : "+v"(le), "+v"(he));
    const int col = tile * 32 + le;
    const bool cv = col < Ly.n;
    const int colc = min(col, Ly.n - 1);
    const float bv = Ly.has_bias ? bias_l[colc] : 0.f;
    const bool to_lds = An != nullptr && cv && !(debug & 2);
    const int pos = c32_pos(colc);
    const bool acc_old = LOADS && Ly.out_f32 && Ly.accumulate;
    const bool masked = LOADS && Ly.mask;
    // Addresses: row r of the lane = (wave-uniform row of register r) + 4 (lane >> 5).  The uniform part goes into a SCALAR base
    // per register, the lane part is ONE 32-bit offset for the whole unit -- the stores take the saddr + voffset form.  (Per-lane
    // 64-bit row arithmetic for each of the 32 values made the epilogue ALU-bound: 5 400 cycles with every memory operation removed.)
    const unsigned lane_c = (unsigned)(4 * he * Ly.ldc + col);            // elements from the row base of register r in out_f32
    const unsigned lane_m = (unsigned)(4 * he * Ly.ldmask + colc);
    const unsigned lane_l = (unsigned)(4 * he * ldn + pos);
    const bool full = m0 + C32_BM <= m && tile * 32 + 32 <= Ly.n;         // (wave-uniform) every element of the unit exists
    // eight rows at a time: the next unit's fragments (100 registers) may be in flight through all of this
#pragma unroll
    for (int mh = 0; mh < (C32_SINGLE ? 2 : 4); ++mh) {
        // (single-accumulator units: the accumulator holds rows 32 half .. 32 half + 31 of the tile)
        const int mt = C32_SINGLE ? half : mh >> 1, r0 = (mh & 1) * 8;
        float mk[8], old[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mk[j] = 1.f, old[j] = 0.f;
        if constexpr (LOADS) {
            // the mask values and the previous contents of an accumulating output are requested TOGETHER (one round trip), at
            // clamped addresses -- no per-lane branch, no wait between them
            if (masked && full) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j, ru = mt * 32 + (r & 3) + 8 * (r >> 2);
                    mk[j] = (Ly.mask + (size_t)(m0 + ru) * Ly.ldmask)[lane_m];
                }
            } else if (masked) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j, ru = mt * 32 + (r & 3) + 8 * (r >> 2);
                    mk[j] = Ly.mask[(size_t)min(m0 + ru + 4 * he, m - 1) * Ly.ldmask + colc];
                }
            }
            if (acc_old && full) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j, ru = mt * 32 + (r & 3) + 8 * (r >> 2);
                    old[j] = (Ly.out_f32 + (size_t)(m0 + ru) * Ly.ldc)[lane_c];
                }
            } else if (acc_old) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j, ru = mt * 32 + (r & 3) + 8 * (r >> 2);
                    old[j] = Ly.out_f32[(size_t)min(m0 + ru + 4 * he, m - 1) * Ly.ldc + colc];
                }
            }
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = ((!C32_SINGLE && mt) ? acc1[r0 + j] : acc0[r0 + j]) + bv;
            if (Ly.relu) t = fmaxf(t, 0.f);
            if (masked) t = mk[j] > 0.f ? t : 0.f;
            v[j] = t;
        }
        if (to_lds) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0 + j, ru = mt * 32 + (r & 3) + 8 * (r >> 2);
                An[ru * ldn + lane_l] = v[j];
            }
        }
        if (Ly.out_f32 && !(debug & 1)) {
            if (full) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j, ru = mt * 32 + (r & 3) + 8 * (r >> 2);
                    float* rowp = Ly.out_f32 + (size_t)(m0 + ru) * Ly.ldc;
                    rowp[lane_c] = acc_old ? old[j] + v[j] : v[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j, row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * he;
                    if (cv && m0 + row < m) Ly.out_f32[(size_t)(m0 + row) * Ly.ldc + col] = acc_old ? old[j] + v[j] : v[j];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- the IAF update around the chain (gv_made_passes_f32), on the workgroup's own 64 rows, four columns per thread ----------
// forward: x_new = count > 0 ? z * expf(alpha + mu) : x_old   (k_iaf_fwd's arithmetic)
__device__ __forceinline__ void c32_update_fwd(const float* z, const float* net, int ld_net, const float* x_old, int ldx, const int* cnt,
                                               float* x_new, int ld_new, int m0, int m, int d, bool reversed = false) {
    const int q = d >> 2;
    for (int i = threadIdx.x; i < C32_BM * q; i += C32_THREADS) {
        const int row = m0 + i / q, c = (i % q) << 2;
        if (row >= m) continue;
        const int4 cn = *reinterpret_cast<const int4*>(cnt + c);
        const bool any = cn.x > 0 || cn.y > 0 || cn.z > 0 || cn.w > 0, all = cn.x > 0 && cn.y > 0 && cn.z > 0 && cn.w > 0;
        float4 zz = make_float4(0.f, 0.f, 0.f, 0.f), mu = zz, al = zz, xo = zz;
        if (any) {
            zz = *reinterpret_cast<const float4*>(z + (size_t)row * d + c);
            mu = *reinterpret_cast<const float4*>(net + (size_t)row * ld_net + c);
            al = *reinterpret_cast<const float4*>(net + (size_t)row * ld_net + d + c);
        }
        if (!all) xo = *reinterpret_cast<const float4*>(x_old + (size_t)row * ldx + c);
        float4 o;
        o.x = cn.x > 0 ? zz.x * expf(al.x + mu.x) : xo.x;
        o.y = cn.y > 0 ? zz.y * expf(al.y + mu.y) : xo.y;
        o.z = cn.z > 0 ? zz.z * expf(al.z + mu.z) : xo.z;
        o.w = cn.w > 0 ? zz.w * expf(al.w + mu.w) : xo.w;
        if (reversed) *reinterpret_cast<float4*>(x_new + (size_t)row * ld_new + (d - 4 - c)) = make_float4(o.w, o.z, o.y, o.x);
        else *reinterpret_cast<float4*>(x_new + (size_t)row * ld_new + c) = o;
    }
}

// log_det[row] = sum of the row's alpha columns (k_rowsum's order: lane l adds columns l, l + 64, ...; the wave's lanes by wave_sum)
__device__ __forceinline__ void c32_logdet(const float* net, int ld_net, float* log_det, int m0, int m, int d, int wave, int lane) {
    for (int r = wave; r < C32_BM; r += C32_WAVES) {
        const int row = m0 + r;
        if (row >= m) break;
        const float* src = net + (size_t)row * ld_net + d;
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc += src[c];
        acc = wave_sum(acc);
        if (lane == 0) log_det[row] = acc;
    }
}

// backward (k_iaf_bwd_v4's arithmetic): g_z (+)= g count e, g_mu = g count z e, g_alpha = g_logdet + g_mu, e = expf(alpha + mu);
// the handed-through gradient (count == 0: g, else 0) goes to g_old, where the chain's last layer adds its own
__device__ __forceinline__ void c32_update_bwd(const float* z, const float* net, int ld_net, const int* cnt, const float* g_in, int ld_gin,
                                               const float* gld, float* g_z, bool gz_write, float* g_net, int ld_gnet, float* g_old,
                                               int ld_gold, int m0, int m, int d, bool reversed = false, int live = 0x7fffffff) {
    const int q = d >> 2;
    for (int i = threadIdx.x; i < C32_BM * q; i += C32_THREADS) {
        const int row = m0 + i / q, c = (i % q) << 2;
        if (row >= m) continue;
        if (row >= live) {
            // a padding row of a static-shape batch (rows [*rows_dev, m)): whatever the caller left in its dL/dx and dL/dlogdet, its
            // [g_mu | g_alpha] is ZERO -- the weight / bias gradient products reduce over all m stacked rows, padding included
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz_write) *reinterpret_cast<float4*>(g_z + (size_t)row * d + c) = zero;
            *reinterpret_cast<float4*>(g_net + (size_t)row * ld_gnet + c) = zero;
            *reinterpret_cast<float4*>(g_net + (size_t)row * ld_gnet + d + c) = zero;
            *reinterpret_cast<float4*>(g_old + (size_t)row * ld_gold + c) = zero;
            continue;
        }
        const int4 cnt4 = *reinterpret_cast<const int4*>(cnt + c);
        float4 g = *reinterpret_cast<const float4*>(g_in + (size_t)row * ld_gin + (reversed ? d - 4 - c : c));
        if (reversed) g = make_float4(g.w, g.z, g.y, g.x);
        const float4 zz = *reinterpret_cast<const float4*>(z + (size_t)row * d + c);
        const float4 mu = *reinterpret_cast<const float4*>(net + (size_t)row * ld_net + c), al = *reinterpret_cast<const float4*>(net + (size_t)row * ld_net + d + c);
        float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!gz_write) old = *reinterpret_cast<const float4*>(g_z + (size_t)row * d + c);
        const float gl = gld ? gld[row] : 0.f;
        const int cn[4] = {cnt4.x, cnt4.y, cnt4.z, cnt4.w};
        const float gv[4] = {g.x, g.y, g.z, g.w}, zv[4] = {zz.x, zz.y, zz.z, zz.w}, mv[4] = {mu.x, mu.y, mu.z, mu.w}, av[4] = {al.x, al.y, al.z, al.w};
        float o_z[4], o_mu[4], o_al[4], o_old[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g_mu = 0.f, g_al = gl, gz = 0.f, go = gv[e];
            if (cn[e] > 0) {
                const float ex = expf(av[e] + mv[e]);
                const float gc = gv[e] * (float)cn[e];
                gz = gc * ex;
                g_mu = gc * zv[e] * ex;
                g_al += g_mu;
                go = 0.f;
            }
            o_z[e] = gz; o_mu[e] = g_mu; o_al[e] = g_al; o_old[e] = go;
        }
        *reinterpret_cast<float4*>(g_z + (size_t)row * d + c) = make_float4(old.x + o_z[0], old.y + o_z[1], old.z + o_z[2], old.w + o_z[3]);
        *reinterpret_cast<float4*>(g_net + (size_t)row * ld_gnet + c) = make_float4(o_mu[0], o_mu[1], o_mu[2], o_mu[3]);
        *reinterpret_cast<float4*>(g_net + (size_t)row * ld_gnet + d + c) = make_float4(o_al[0], o_al[1], o_al[2], o_al[3]);
        *reinterpret_cast<float4*>(g_old + (size_t)row * ld_gold + c) = make_float4(o_old[0], o_old[1], o_old[2], o_old[3]);
    }
}

__global__ __launch_bounds__(C32_THREADS) void k_made_chain_f32(const Chain32Args p) {
    extern __shared__ __attribute__((aligned(16))) float c32_lds[];
    const int nl = p.n_layers, m0 = blockIdx.x * C32_BM;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    float* const buf0 = c32_lds;
    float* const buf1 = c32_lds + C32_BM * p.ld0;

    // a static-shape batch pads its node arrays: a workgroup whose rows are all padding stores zeros (nothing where it would
    // accumulate) instead of running the layers, as gv_gemm_f32_live_rows does for its padding tiles
    const bool padding = p.rows_dev && m0 >= *p.rows_dev;
    const int n_pass = p.iaf.mode ? p.iaf.passes : 1;
    const long long step = p.iaf.mode ? (long long)p.iaf.step : 0ll;
    const gv_chain32_layer& Llast = p.L[nl - 1];

    // The layer descriptors are kernel arguments, indexed by a run-time layer number below: scalar loads from the kernarg buffer, one
    // cold miss (a microsecond) per 64-B line at the FIRST use -- i.e. in every layer's first unit, in series.  Touch every line
    // now, together (measured on a 3-layer chain with nothing but control flow left in it: 3.1 us per layer before).
    {
        const int* raw = reinterpret_cast<const int*>(&p);
        int touch = 0;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Chain32Args) + 63) / 64; ++i) touch += raw[min(16 * i, (int)sizeof(Chain32Args) / 4 - 1)];
        asm volatile("" ::"s"(touch));
    }
    int32_t* const plan = reinterpret_cast<int32_t*>(c32_lds + C32_BM * (p.ld0 + p.ld1));
    for (int i = threadIdx.x; i < C32_LPLAN_WORDS; i += C32_THREADS) {         // counts, the used head of every list, the group sets
        int src = i;
        if (i >= C32_LPLAN_SETS) src = C32_PLAN_SETS + (i - C32_LPLAN_SETS);
        else if (i >= C32_PLAN_LISTS) src = C32_PLAN_LISTS + ((i - C32_PLAN_LISTS) / C32_MAXU_LDS) * C32_MAXU + (i - C32_PLAN_LISTS) % C32_MAXU_LDS;
        plan[i] = p.plan[src];
    }
    // every layer's bias behind it (no global round trip in an epilogue); bias_at[l] = where layer l's starts
    float* const bias_lds = reinterpret_cast<float*>(plan + C32_LPLAN_WORDS);
    {
        int at = 0;
        for (int l = 0; l < nl; ++l) {
            if (p.L[l].bias)
                for (int i = threadIdx.x; i < p.L[l].n; i += C32_THREADS) bias_lds[at + i] = p.L[l].bias[i];
            if (p.L[l].bias) at += p.L[l].n;          // (a layer without a bias takes no room: the backward chains have none)
        }
    }
    __syncthreads();

    int ts_n = 0;
    auto stamp = [&]() {
        if ((p.debug & 16) && blockIdx.x == 0 && ts_n < 64) {
            const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
            if (lane == 0) const_cast<int32_t*>(p.plan)[C32_PLAN_WORDS + wave * 64 + ts_n] = (int32_t)t;
            ++ts_n;
        }
    };
    stamp();
    const int nu = sload(plan + (wave & 3));
    const int32_t* const list = plan + C32_PLAN_LISTS + (wave & 3) * C32_MAXU_LDS;
    const float4* const any_b = reinterpret_cast<const float4*>(p.L[0].w_packed);      // a readable address for idle loads

    // where a unit's fragments start (its tile's group 0, this lane's slot)
    auto b_of = [&](const C32Unit& u, const float4*& base, unsigned& off) {
        base = any_b;
        off = lane;
        if (u.layer < nl) {
            base = reinterpret_cast<const float4*>(c32_pin_ptr(p.L[u.layer].w_packed));
            off = (unsigned)(u.tile * ((c32_pin(p.L[u.layer].k) + 7) >> 3) * 64 + lane);
        }
    };

    const int slot = wave >> 2;          // this wave takes entries slot, slot + C32_SLOTS, ... of its SIMD's list
    float4 q[C32_CHG];
  for (int s = 0; s < n_pass; ++s) {
    const long long so = (long long)s * step;            // rows from the first pass's slices to this pass's
    float* const xs = p.x + so * p.ldx;
    // the first unit's weight fragments are requested before the pass's element-wise front
    C32Unit cu = {slot, 0, 0, 0, 0ull};
    c32_open(plan, nl, list, padding ? 0 : nu, cu);
    {
        const float4* b0;
        unsigned off;
        b_of(cu, b0, off);
        c32_issue(q, b0, off, cu.set);
    }
    if (p.iaf.mode == 2) {
        // the update's backward for this pass: [g_mu | g_alpha] into the chain's input slice, dL/dz, the handed-through gradient
        const int d = p.iaf.d;
        const float* g_in = s == 0 ? p.iaf.g_in : Llast.out_f32 + (so - step) * Llast.ldc;
        c32_update_bwd(p.iaf.z, p.iaf.net + so * p.iaf.ld_net, p.iaf.ld_net, p.iaf.colcount + (step > 0 ? s : -s) * d, g_in,
                       s == 0 ? d : Llast.ldc, (s == 0 && (p.iaf.flags & 1)) ? p.iaf.g_logdet : nullptr, p.iaf.g_z,
                       s == 0 && (p.iaf.flags & 2), xs, p.ldx, Llast.out_f32 + so * Llast.ldc, Llast.ldc, m0, p.m, d,
                       s == 0 && (p.iaf.flags & 4), p.rows_dev ? *p.rows_dev : 0x7fffffff);
        __syncthreads();         // (drains the stores: the staging below and the last layer's epilogue read them back)
    }
    if (p.iaf.mode == 1 && s == 0 && (p.iaf.flags & 4)) {
        // the block's pass 0 fed the MLP an all-zero input: its [mu | alpha] is one row; its update gives the first stacked pass's input
        c32_update_fwd(p.iaf.z, p.iaf.net0, 0, p.iaf.z, p.iaf.d, p.iaf.cnt0, xs, p.ldx, m0, p.m, p.iaf.d);
        __syncthreads();         // (drains the stores: the staging below reads them back)
    }
    if (padding) {
        for (int l = 0; l < nl; ++l) {
            const gv_chain32_layer& Lz = p.L[l];
            if (!Lz.out_f32 || Lz.accumulate) continue;
            const int q4 = Lz.n >> 2;
            float* o = Lz.out_f32 + so * Lz.ldc;
            for (int i = threadIdx.x; i < C32_BM * q4; i += C32_THREADS) {
                const int row = m0 + i / q4, c = (i % q4) << 2;
                if (row < p.m) *reinterpret_cast<float4*>(o + (size_t)row * Lz.ldc + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    } else {
        // ---- stage x: 64 rows x k0 floats, 16-B pieces along the rows, zero outside [0, m) ----------------------------------
        const int k0 = p.L[0].k, q4 = k0 >> 2;       // k0 % 8 == 0
        for (int i = threadIdx.x; i < C32_BM * q4; i += C32_THREADS) {
            const int row = i / q4, c = (i - row * q4) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + row < p.m) v = *reinterpret_cast<const float4*>(xs + (size_t)(m0 + row) * p.ldx + c);
            // columns c .. c + 3 of a group of 8: (c, c + 2) are neighbours in the even-k half, (c + 1, c + 3) in the odd-k half
            float* o = buf0 + row * p.ld0 + (c & ~7) + ((c & 4) >> 1);
            *reinterpret_cast<float2*>(o) = make_float2(v.x, v.z);
            *reinterpret_cast<float2*>(o + 4) = make_float2(v.y, v.w);
        }
    }
    __syncthreads();
    int layer = 0, bias_at = 0, bias_layer = 0;

    while (cu.layer < nl) {
        while (layer < cu.layer) {       // cross layer boundaries: the previous layer's tile is complete
            c32_barrier();
            ++layer;
        }
        stamp();
        C32Layer Ly = c32_layer(p.L[cu.layer]);
        if (Ly.out_f32) Ly.out_f32 += so * Ly.ldc;
        if (Ly.mask) Ly.mask += so * Ly.ldmask;
        while (bias_layer < cu.layer) { if (p.L[bias_layer].bias) bias_at += p.L[bias_layer].n; ++bias_layer; }
        const float* A = (cu.layer & 1) ? buf1 : buf0;
        const int lda = (cu.layer & 1) ? p.ld1 : p.ld0;
        const float4* b0;
        unsigned off;
        b_of(cu, b0, off);
        f32x16c acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        unsigned long long left = cu.set;
        int ra = l31, ha = lhi;        // opaque: a hoisted fragment address is one more register held across the whole loop
        asm volatile("" : "+v"(ra), "+v"(ha));
        stamp();
        for (;;) {
            c32_landed(q, off);
            if (p.debug & 4) left = 0ull;
            else left = c32_mma(acc0, acc1, q, A + ((C32_SINGLE ? 32 * cu.half : 0) + ra) * lda + 4 * ha, A + (32 + ra) * lda + 4 * ha, left);
            if (left == 0ull) break;
            c32_issue(q, b0, off, left);          // a unit of more than C32_CHG groups: its next chunk
        }
        // the wave's next unit: its fragments are requested now and land during the epilogue, the barrier and -- mostly -- the
        // OTHER wave of this SIMD's unit
        stamp();
        C32Unit nx = {cu.ui + C32_SLOTS, 0, 0, 0, 0ull};
        c32_open(plan, nl, list, nu, nx);
        const float4* nb0;
        unsigned noff;
        b_of(nx, nb0, noff);
        // (an epilogue that LOADS -- a backward layer's mask, an accumulating output -- comes first: loads return in order, its
        // waits would be waits for the 25 fragments too)
        const bool loads_in_epilogue = Ly.mask || (Ly.out_f32 && Ly.accumulate);
        if (!loads_in_epilogue && !(p.debug & 8)) c32_issue(q, nb0, noff, nx.set);
        stamp();
        // ---- epilogue: + bias, ReLU, the ReLU mask of a backward layer, the fp32 store (+ accumulate), the next layer's LDS tile ----
        {
            float* const An = ((cu.layer + 1) & 1) ? buf1 : buf0;
            const int ldn = ((cu.layer + 1) & 1) ? p.ld1 : p.ld0;
            if (loads_in_epilogue) c32_epilogue<true>(acc0, acc1, Ly, cu.tile, m0, p.m, cu.layer + 1 < nl ? An : nullptr, ldn, bias_lds + bias_at, l31, lhi, p.debug, cu.half);
            else c32_epilogue<false>(acc0, acc1, Ly, cu.tile, m0, p.m, cu.layer + 1 < nl ? An : nullptr, ldn, bias_lds + bias_at, l31, lhi, p.debug, cu.half);
        }
        if (loads_in_epilogue && !(p.debug & 8)) c32_issue(q, nb0, noff, nx.set);
        stamp();
        cu = nx;
    }
    while (layer < nl - 1) {         // every wave passes every layer boundary
        c32_barrier();
        ++layer;
    }
    if (p.iaf.mode == 1) {
        // the IAF update behind the pass: the workgroup's [mu | alpha] rows are read back (drained stores), x_new is the next pass's
        // input slice -- or x_out behind the launch's last pass
        __syncthreads();
        const int d = p.iaf.d;
        const bool to_out = s + 1 == n_pass && (p.iaf.flags & 1);
        c32_update_fwd(p.iaf.z, Llast.out_f32 + so * Llast.ldc, Llast.ldc, xs, p.ldx, p.iaf.colcount + (step > 0 ? s : -s) * d,
                       to_out ? p.iaf.x_out : xs + step * p.ldx, to_out ? d : p.ldx, m0, p.m, d, to_out && (p.iaf.flags & 16));
        if (s + 1 == n_pass && (p.iaf.flags & 8)) c32_logdet(Llast.out_f32 + so * Llast.ldc, Llast.ldc, p.iaf.log_det, m0, p.m, d, wave, lane);
    }
    if (s + 1 < n_pass) __syncthreads();          // the next pass reads what this one stored, and reuses the LDS tiles
  }
    stamp();
}

// ---- packing: packed[((t * KG + g) * 64 + lane)] = float4{ B[8 g + 2 i + (lane >> 5)][32 t + (lane & 31)], i = 0..3 },
// zero outside B.  forward layer: B[k][n] = W[n][k] (W [n][k], the product x W^T); backward-x: B[k][n] = W[k][n].
struct Pack32One { const float* w; int ld, n, k; float4* fwd; float4* bwd; };
struct Pack32Multi { Pack32One e[C32_L]; };

__global__ __launch_bounds__(256) void k_pack32(const Pack32Multi p) {
    const Pack32One& e = p.e[blockIdx.y];
    const int kg_f = (e.k + 7) >> 3, nt_f = (e.n + 31) >> 5, tot_f = e.fwd ? nt_f * kg_f * 64 : 0;
    const int kg_b = (e.n + 7) >> 3, nt_b = (e.k + 31) >> 5, tot_b = e.bwd ? nt_b * kg_b * 64 : 0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < tot_f + tot_b; idx += gridDim.x * 256) {
        const bool is_b = idx >= tot_f;
        const int j = is_b ? idx - tot_f : idx, kg = is_b ? kg_b : kg_f;
        const int lane = j & 63, tg = j >> 6, g = tg % kg, t = tg / kg;
        const int col = t * 32 + (lane & 31), k0 = 8 * g + (lane >> 5);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k0 + 2 * i;
            v[i] = 0.f;
            if (!is_b) { if (col < e.n && kk < e.k) v[i] = e.w[(size_t)col * e.ld + kk]; }        // B[kk][col] = W[col][kk]
            else { if (col < e.k && kk < e.n) v[i] = e.w[(size_t)kk * e.ld + col]; }             // B[kk][col] = W[kk][col]
        }
        (is_b ? e.bwd : e.fwd)[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- the plan: group sets from the 0/1 masks, units dealt to the four waves -----------------------------------------------
struct Plan32One { const float* mask; int ld, n, k, transposed; };       // mask [rows][ld] of the layer's W (n x k as the layer sees B)
struct Plan32Args { Plan32One e[C32_L]; int n_layers; int32_t* plan; };

__global__ __launch_bounds__(256) void k_chain32_plan(const Plan32Args p) {
    __shared__ unsigned long long sets[C32_L * C32_MAXT];
    for (int i = threadIdx.x; i < C32_L * C32_MAXT; i += 256) sets[i] = 0ull;
    __syncthreads();
    for (int l = 0; l < p.n_layers; ++l) {
        const Plan32One& e = p.e[l];
        const int nt = (e.n + 31) >> 5, kg = (e.k + 7) >> 3;
        if (!e.mask) {          // no mask: every group that exists
            for (int t = threadIdx.x; t < nt; t += 256) sets[l * C32_MAXT + t] = kg >= 64 ? ~0ull : ((1ull << kg) - 1ull);
            continue;
        }
        // one thread per (column, group): any non-zero among the 8 k of the group
        for (int i = threadIdx.x; i < e.n * kg; i += 256) {
            const int col = i / kg, g = i - col * kg;
            bool any = false;
            for (int kk = 8 * g; kk < min(e.k, 8 * g + 8); ++kk) {
                // layer's B[kk][col]: forward B = W^T (mask [n][k]: entry [col][kk]); backward-x B = W (mask [k_rows][n_cols]: entry [kk][col])
                const float mv = e.transposed ? e.mask[(size_t)kk * e.ld + col] : e.mask[(size_t)col * e.ld + kk];
                any = any || mv != 0.f;
            }
            if (any) atomicOr(&sets[l * C32_MAXT + (col >> 5)], 1ull << g);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cnt[C32_LISTS] = {0, 0, 0, 0};
        int32_t* lists = p.plan + C32_PLAN_LISTS;
        for (int l = 0; l < p.n_layers; ++l) {
            const int nt = (p.e[l].n + 31) >> 5;
            int order[C32_MAXT], cost[C32_MAXT];
            for (int t = 0; t < nt; ++t) {
                // a tile whose set is empty (an output no input reaches: the first column of MADE's last layer) still needs its
                // epilogue (bias, stores): it keeps one group, whose weights are all zero
                if (sets[l * C32_MAXT + t] == 0ull) sets[l * C32_MAXT + t] = 1ull;
                order[t] = t;
                cost[t] = __popcll(sets[l * C32_MAXT + t]);
            }
            for (int i = 1; i < nt; ++i)          // insertion sort, longest first (stable: ties keep the tile order)
                for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) {
                    const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
                }
            int load[C32_LISTS] = {0, 0, 0, 0};
            for (int i = 0; i < nt; ++i) {
                int w = 0;
                for (int c = 1; c < C32_LISTS; ++c) if (load[c] < load[w]) w = c;
                load[w] += cost[order[i]] + 1;          // (+ 1: a unit's epilogue is worth about a group)
                lists[w * C32_MAXU + cnt[w]++] = (l << 8) | order[i];
                if (C32_HALVES) {                       // the tile's second row half: its own unit, to the list that is least loaded now
                    int w2 = 0;
                    for (int c = 1; c < C32_LISTS; ++c) if (load[c] < load[w2]) w2 = c;
                    load[w2] += cost[order[i]] + 1;
                    lists[w2 * C32_MAXU + cnt[w2]++] = (l << 8) | 0x80 | order[i];
                }
            }
        }
        for (int w = 0; w < C32_LISTS; ++w) p.plan[w] = cnt[w];
        for (int i = 0; i < C32_L * C32_MAXT; ++i) {
            p.plan[C32_PLAN_SETS + 2 * i] = (int32_t)(unsigned)(sets[i] & 0xffffffffull);
            p.plan[C32_PLAN_SETS + 2 * i + 1] = (int32_t)(unsigned)(sets[i] >> 32);
        }
    }
}

}  // namespace gv

using namespace gv;

/* floats of one packed copy of a B operand with n columns over a reduction of k */
extern "C" int64_t gv_made_pack_weight_f32_elems(int n, int k) {
    if (n <= 0 || k <= 0) return 0;
    return (int64_t)((n + 31) / 32) * ((k + 7) / 8) * 64 * 4;
}

extern "C" int gv_made_pack_weight_f32_multi(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                                             float* const* packed_fwd, float* const* packed_bwd, void* stream) {
    GV_REQUIRE(count >= 1 && count <= C32_L, GV_ERR_SHAPE, "gv_made_pack_weight_f32_multi: count=%d", count);
    GV_REQUIRE(w && ld && n && k && packed_fwd && packed_bwd, GV_ERR_NULL, "gv_made_pack_weight_f32_multi: NULL table");
    Pack32Multi p;
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        GV_REQUIRE(n[i] > 0 && k[i] > 0 && ld[i] >= k[i] && w[i] && (packed_fwd[i] || packed_bwd[i]), GV_ERR_SHAPE,
                   "gv_made_pack_weight_f32_multi: entry %d: n=%d k=%d ld=%d", i, n[i], k[i], ld[i]);
        GV_REQUIRE((!packed_fwd[i] || aligned16(packed_fwd[i])) && (!packed_bwd[i] || aligned16(packed_bwd[i])), GV_ERR_ALIGN,
                   "gv_made_pack_weight_f32_multi: packed buffers must be 16-B aligned");
        p.e[i].w = w[i]; p.e[i].ld = ld[i]; p.e[i].n = n[i]; p.e[i].k = k[i];
        p.e[i].fwd = (float4*)packed_fwd[i]; p.e[i].bwd = (float4*)packed_bwd[i];
        most = max(most, (gv_made_pack_weight_f32_elems(n[i], k[i]) + gv_made_pack_weight_f32_elems(k[i], n[i])) / 4);
    }
    hipLaunchKernelGGL(k_pack32, dim3((unsigned)((most + 255) / 256), count), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("gv_made_pack_weight_f32_multi");
}

static bool c32_pitches(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer, int* ld0, int* ld1) {
    int w0 = 0, w1 = 0;          // widest input staged in buffer 0 (even layers) / buffer 1 (odd layers)
    for (int i = 0; i < n_layers; ++i) {
        if (n_of_layer[i] <= 0 || k_of_layer[i] <= 0 || n_of_layer[i] % 8 || k_of_layer[i] % 8) return false;
        if (n_of_layer[i] > 32 * C32_MAXT || k_of_layer[i] > 512) return false;
        if (i > 0 && k_of_layer[i] != n_of_layer[i - 1]) return false;
        ((i & 1) ? w1 : w0) = max((i & 1) ? w1 : w0, k_of_layer[i]);
    }
    *ld0 = w0 + 4;              // 8 j + 4: the 16 lanes of a ds_read_b128 group fall on 16 different bank quads
    *ld1 = w1 ? w1 + 4 : 0;
    size_t widths = 0;
    for (int i = 0; i < n_layers; ++i) widths += (size_t)n_of_layer[i];       // the biases
    size_t units = 0;
    for (int i = 0; i < n_layers; ++i) units += (size_t)((n_of_layer[i] + 31) / 32) * (C32_HALVES ? 2 : 1);
    if (units > (size_t)C32_MAXU_LDS) return false;          // (every list of the plan then fits its LDS copy)
    return (size_t)C32_BM * (*ld0 + *ld1) * sizeof(float) + C32_LPLAN_WORDS * sizeof(int32_t) + widths * sizeof(float) <= 160 * 1024;
}

/* 1 when gv_made_chain_f32 can run this chain (widths are multiples of 8, n <= 32 GV_CHAIN32_MAX_TILES, k <= 512; both LDS tiles fit) */
extern "C" int gv_made_chain_f32_fits(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer) {
    if (n_layers < 1 || n_layers > C32_L || !n_of_layer || !k_of_layer) return 0;
    int ld0, ld1;
    return c32_pitches(n_layers, n_of_layer, k_of_layer, &ld0, &ld1) ? 1 : 0;
}

extern "C" int gv_made_chain_f32_plan(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer, const float* const* masks,
                                      const int32_t* ldmask, const int32_t* transposed, int32_t* plan, void* stream) {
    GV_REQUIRE(n_layers >= 1 && n_layers <= C32_L && n_of_layer && k_of_layer && plan, GV_ERR_SHAPE, "gv_made_chain_f32_plan: n_layers=%d", n_layers);
    int ld0, ld1;
    GV_REQUIRE(c32_pitches(n_layers, n_of_layer, k_of_layer, &ld0, &ld1), GV_ERR_SHAPE, "gv_made_chain_f32_plan: the chain does not fit gv_made_chain_f32");
    Plan32Args p;
    p.n_layers = n_layers;
    p.plan = plan;
    for (int i = 0; i < n_layers; ++i) {
        p.e[i].mask = masks ? masks[i] : nullptr;
        p.e[i].n = n_of_layer[i];
        p.e[i].k = k_of_layer[i];
        p.e[i].transposed = transposed ? transposed[i] : 0;
        p.e[i].ld = (masks && masks[i]) ? ldmask[i] : 0;
        GV_REQUIRE(!p.e[i].mask || p.e[i].ld >= (p.e[i].transposed ? p.e[i].n : p.e[i].k), GV_ERR_SHAPE,
                   "gv_made_chain_f32_plan: layer %d: mask pitch %d", i, p.e[i].ld);
    }
    hipLaunchKernelGGL(k_chain32_plan, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("gv_made_chain_f32_plan");
}

static int made_chain_f32(float* x, int ldx, int m, int n_layers, const gv_chain32_layer* layers, const int32_t* plan,
                          const int32_t* rows_dev, const gv_chain32_iaf* iaf, void* stream);

extern "C" int gv_made_chain_f32(const float* x, int ldx, int m, int n_layers, const gv_chain32_layer* layers, const int32_t* plan,
                                 const int32_t* rows_dev, void* stream) {
    return made_chain_f32(const_cast<float*>(x), ldx, m, n_layers, layers, plan, rows_dev, nullptr, stream);
}

extern "C" int gv_made_passes_f32(float* x, int ldx, int m, int n_layers, const gv_chain32_layer* layers, const int32_t* plan,
                                  const int32_t* rows_dev, const gv_chain32_iaf* iaf, void* stream) {
    GV_REQUIRE(iaf && (iaf->mode == 1 || iaf->mode == 2) && iaf->passes >= 1 && iaf->d > 0 && iaf->d % 4 == 0, GV_ERR_SHAPE,
               "gv_made_passes_f32: mode 1 / 2, passes >= 1, d %% 4 == 0");
    GV_REQUIRE(iaf->z && iaf->colcount && aligned16(iaf->z) && aligned16(iaf->colcount), GV_ERR_NULL, "gv_made_passes_f32: z / colcount");
    GV_REQUIRE(n_layers >= 1 && n_layers <= C32_L && layers, GV_ERR_SHAPE, "gv_made_passes_f32: n_layers=%d", n_layers);
    const gv_chain32_layer& last = layers[n_layers - 1];
    if (iaf->mode == 1) {
        GV_REQUIRE(layers[0].k == iaf->d && last.n == 2 * iaf->d && last.out_f32 && !last.accumulate && ldx >= iaf->d, GV_ERR_SHAPE,
                   "gv_made_passes_f32 (forward): the chain maps d -> [mu | alpha] (2 d, stored), ldx >= d");
        GV_REQUIRE(!(iaf->flags & 1) || (iaf->x_out && aligned16(iaf->x_out)), GV_ERR_NULL, "gv_made_passes_f32 (forward): x_out");
        GV_REQUIRE(!(iaf->flags & 4) || (iaf->net0 && iaf->cnt0 && aligned16(iaf->net0) && aligned16(iaf->cnt0)), GV_ERR_NULL,
                   "gv_made_passes_f32 (forward): pass 0's row / counts");
        GV_REQUIRE(!(iaf->flags & 8) || iaf->log_det, GV_ERR_NULL, "gv_made_passes_f32 (forward): log_det");
    } else {
        GV_REQUIRE(layers[0].k == 2 * iaf->d && last.n == iaf->d && last.out_f32 && last.accumulate && ldx >= 2 * iaf->d, GV_ERR_SHAPE,
                   "gv_made_passes_f32 (backward): the chain maps [g_mu | g_alpha] (2 d) -> d, the last layer accumulates, ldx >= 2 d");
        GV_REQUIRE(iaf->net && iaf->g_in && iaf->g_z && iaf->ld_net >= 2 * iaf->d && iaf->ld_net % 4 == 0 && aligned16(iaf->net) &&
                   aligned16(iaf->g_in) && aligned16(iaf->g_z) && (!(iaf->flags & 1) || iaf->g_logdet), GV_ERR_NULL,
                   "gv_made_passes_f32 (backward): net / g_in / g_z / g_logdet");
    }
    return made_chain_f32(x, ldx, m, n_layers, layers, plan, rows_dev, iaf, stream);
}

static int made_chain_f32(float* x, int ldx, int m, int n_layers, const gv_chain32_layer* layers, const int32_t* plan,
                          const int32_t* rows_dev, const gv_chain32_iaf* iaf, void* stream) {
    GV_REQUIRE(m >= 0 && n_layers >= 1 && n_layers <= C32_L, GV_ERR_SHAPE, "gv_made_chain_f32: m=%d n_layers=%d", m, n_layers);
    if (m == 0) return GV_OK;
    GV_REQUIRE(x && layers && plan, GV_ERR_NULL, "gv_made_chain_f32: NULL pointer");
    GV_REQUIRE(ldx % 4 == 0 && aligned16(x) && ldx >= layers[0].k, GV_ERR_ALIGN, "gv_made_chain_f32: x rows are 16-B aligned (ldx=%d)", ldx);
    Chain32Args p;
    int32_t ns[C32_L], ks[C32_L];
    for (int i = 0; i < n_layers; ++i) {
        const gv_chain32_layer& L = layers[i];
        ns[i] = L.n;
        ks[i] = L.k;
        GV_REQUIRE(L.w_packed && aligned16(L.w_packed), GV_ERR_NULL, "gv_made_chain_f32: layer %d has no packed weight", i);
        GV_REQUIRE(L.out_f32 || i + 1 < n_layers, GV_ERR_NULL, "gv_made_chain_f32: the last layer stores nothing");
        GV_REQUIRE((!L.out_f32 || (L.ldc >= L.n && L.ldc % 4 == 0 && aligned16(L.out_f32))) && (!L.mask || L.ldmask >= L.n), GV_ERR_ALIGN,
                   "gv_made_chain_f32: layer %d: leading dimension / alignment", i);
        p.L[i] = L;
    }
    GV_REQUIRE(c32_pitches(n_layers, ns, ks, &p.ld0, &p.ld1), GV_ERR_SHAPE,
               "gv_made_chain_f32: widths are multiples of 8 (n <= %d, k <= 512), k = the previous layer's n, two LDS tiles <= 160 KB",
               32 * C32_MAXT);
    p.x = x; p.ldx = ldx; p.m = m; p.n_layers = n_layers; p.plan = plan; p.rows_dev = rows_dev;
    if (iaf) p.iaf = *iaf;
    else { p.iaf = gv_chain32_iaf(); p.iaf.mode = 0; p.iaf.passes = 1; }
    { const char* e = getenv("GV_C32_DEBUG"); p.debug = e ? atoi(e) : 0; }
    size_t widths = 0;
    for (int i = 0; i < n_layers; ++i) widths += layers[i].bias ? (size_t)ns[i] : 0;
    const size_t lds = (size_t)C32_BM * (p.ld0 + p.ld1) * sizeof(float) + C32_LPLAN_WORDS * sizeof(int32_t) + widths * sizeof(float);
    static unsigned long long lds_done = 0;
    if (!raise_dynamic_lds((const void*)k_made_chain_f32, 160 * 1024, lds_done, "gv_made_chain_f32")) return GV_ERR_SHAPE;
    hipLaunchKernelGGL(k_made_chain_f32, dim3((unsigned)((m + C32_BM - 1) / C32_BM)), dim3(C32_THREADS), lds, (hipStream_t)stream, p);
    return launch_status("gv_made_chain_f32");
}
