// K4 fused: the CHAIN of masked-MLP products of one MADE pass (kgvae/flow_network.py:85-98: x -> relu(W1 x + b1) -> ... ->
// [mu | alpha]; and the backward-x chain g_L -> (g_L W_L) * [a_{L-1} > 0] -> ... -> g_x) in ONE launch.
//
// Why: at FB15k-237 size one product is 1.2 GFLOP over 14541 rows -- 12-20 us as its own launch, of which 6 us are fixed cost
// (profiles/round2/made_gemm_ablation.txt) -- and a step with 3 IAF blocks holds ~140 of them in one dependency chain.  Here a
// workgroup owns 64 rows for the WHOLE chain: the activations of a layer never leave the CU (two ping-pong LDS tiles
// of 64 x K bf16), only the copies the backward pass needs are stored (bf16 row-major from the LDS tile, 16-B pieces; the
// transposed bf16 copy straight from the accumulators: 32 lanes = 32 consecutive rows of one column = 64 contiguous bytes per
// store).  Seven MMA waves own the column tiles; an eighth wave copies the row-major tile out while they are in the next layer.
//
// Operands: A fragments from LDS (conflict-free ds_read_b128: rows of LDK = 16 j + 8 elements); B fragments (the weights)
// from a FRAGMENT-PACKED copy in global memory (gv_made_pack_weight: tile of 32 output columns x 16-deep step = 64 lanes x 16 B,
// contiguous), loaded by the wave that owns the columns straight into registers -- no LDS traffic, no barrier for B.  A wave owns
// 64 rows x 32 columns (2 accumulator tiles of v_mfma_f32_32x32x16_bf16, eight waves per workgroup: two per SIMD hide each other's waits).
// The B fragments of the wave's NEXT unit (next 13 steps: next column tile, reduction chunk or layer) are requested right after
// the current unit's last MFMA, into the same registers: in flight during the epilogue and the barrier.
//
// Same arithmetic as gv_gemm_bf16_nt per product (operands rounded to bf16, k accumulated in 16-deep steps in order, fp32
// accumulators, epilogue order bias -> ReLU -> mask -> stores): results are bit-identical to the launch-per-product path.
#include <stdlib.h>

#include "common.h"

namespace gv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#ifndef GV_CHAIN_ABL
#define GV_CHAIN_ABL 0          /* compile-time ablation bits (probes): 256 no transposed stores, 512 no LDS tile, 1024 no sign words, 2048 no fragment loads */
#endif
#ifndef GV_CHAIN_FAST_EPILOGUE
#define GV_CHAIN_FAST_EPILOGUE 1
#endif
#ifndef GV_CHAIN_LDS_BARRIER
#define GV_CHAIN_LDS_BARRIER 0      /* measured: no gain at WN18RR size (5.57 vs 5.59 ms), the stand-alone fused chain 115 -> 123 us */
#endif
constexpr int CH_BM = 64;        // rows per workgroup
constexpr int CH_KS = 13;        // 16-deep steps per register set of B fragments (208 of k)
constexpr int CH_MMA_WAVES = 7, CH_STORE_WAVES = 1;      // 8 waves: two per SIMD, 256 VGPRs each
constexpr int CH_MMA_THREADS = CH_MMA_WAVES * 64, CH_STORE_THREADS = CH_STORE_WAVES * 64;
constexpr int CH_THREADS = CH_MMA_THREADS + CH_STORE_THREADS;
constexpr int CH_L = GV_CHAIN_MAX_LAYERS;
constexpr int CH_MAX_PASSES = 6;       // passes of one launch of the IAF-backward instance (the reference's MADE makes six; the first is a row kernel)

struct ChainArgs {
    const uint16_t* x;           // [m][ldx] bf16: input of layer 0
    int ldx, m, n_layers, ldk, has_mask;
    int32_t* stamps;             // gv_made_chain_debug_stamps: s_memtime stamps of workgroup 0 (probes only), NULL otherwise
    gv_chain_iafb ib;            // ib.gx != NULL: layer 0's input is made in the prologue (the IAF update's backward), x unused
    int ib_lds_off;              // floats from the bias block to the stage's 2 x IB_COLS x IB_LD transposition block
    int ld0;                     // IB: row stride of layer 0's input tile (== ldk, or wider: the tile then spans both activation buffers)
    gv_chain_fwd_row0 row0;      // FW, row0.net_row != NULL: pass 0's update makes layer 0's input of the first pass (x unused)
    int n_passes;                // IB: passes of a MADE's backward in this launch (pass[q]: what differs from pass to pass)
    gv_chain_layer L[CH_L];
    struct Pass {
        const float* ex; const float* gx; const float* gld; const int32_t* cc; uint16_t* gnt; float* of; const float* add;
        int32_t flags, pad;
        const uint32_t* bits[CH_L];
        uint16_t* out_t[CH_L];
        // the FORWARD passes launch (gv_made_chain_fwd) reads the same table: gx = x_old, of = x_new, add = alpha, cc = the pass's column
        // counts, gnt = x_new's tiled transposed copy, bits / out_t = the hidden layers' sign words and tiled copies (outputs), and
        const int32_t* keep;     // ... the next pass's column counts (x_new is stored where one of them is 0)
        uint16_t* ob;            // ... x_new as row-major bf16 in memory (NULL: it only stays in LDS, as the next pass's input)
    } pass[CH_MAX_PASSES];
};

__device__ __forceinline__ uint16_t bf_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

struct ChainUnit { int l, tile, ch; };

// a kernel-argument field as an opaque SGPR value: read once where it is pinned, not re-read at every later use
__device__ __forceinline__ int chain_pin(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* chain_pin_ptr(T* q) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    typedef __attribute__((address_space(1))) T GT;         // (a pointer rebuilt from integers is generic: FLAT accesses otherwise)
    return (T*)(GT*)(((unsigned long long)hi << 32) | lo);
}

// the same as an opaque scalar made HERE (volatile: not hoisted out of an enclosing loop, where it would be held -- or spilled -- across it)
template <typename T>
__device__ __forceinline__ T* chain_pin_here(T* q) {
    asm volatile("" : "+s"(q));
    typedef __attribute__((address_space(1))) T GT;
    return (T*)(GT*)q;
}
__device__ __forceinline__ int chain_pin_here(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

__device__ __forceinline__ void chain_barrier() {
    if (GV_CHAIN_LDS_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}

// column tiles of a layer: 32 output columns each; an IAF layer's tile is 16 mu columns + their 16 alpha columns
__device__ __forceinline__ int chain_tiles(const gv_chain_layer& L) { return L.iaf_z ? ((L.n >> 1) + 15) >> 4 : (L.n + 31) >> 5; }
__device__ __forceinline__ int chain_first_tile(const gv_chain_layer&, int wave) { return wave; }
__device__ __forceinline__ int chain_next_tile(const gv_chain_layer&, int tile) { return tile + CH_MMA_WAVES; }
// (giving a wave PAIRS of neighbouring IAF tiles -- the two 64-B halves of a row's cache line -- measured 5 us slower)

// successor of unit u in this wave's order (l == n_layers: none)
__device__ __forceinline__ ChainUnit chain_next(const ChainArgs& p, int nl, ChainUnit u, int wave) {
    const int ks = (p.L[u.l].k + 15) >> 4, nch = (ks + CH_KS - 1) / CH_KS;
    if (u.ch + 1 < nch) return {u.l, u.tile, u.ch + 1};
    if (chain_next_tile(p.L[u.l], u.tile) < chain_tiles(p.L[u.l])) return {u.l, chain_next_tile(p.L[u.l], u.tile), 0};
    int l = u.l + 1;
    while (l < nl && chain_first_tile(p.L[l], wave) >= chain_tiles(p.L[l])) ++l;
    return {l, l < nl ? chain_first_tile(p.L[l], wave) : wave, 0};
}

// where the B fragments of unit u start (packed weight of its layer + this lane's 16-B slot of tile u.tile, step 13 u.ch)
// and how many steps it has
__device__ __forceinline__ void chain_unit_b(const ChainArgs& p, ChainUnit u, int lane, const uint4*& base, unsigned& off, int& ksc) {
    const int ks = (p.L[u.l].k + 15) >> 4;
    ksc = min(CH_KS, ks - u.ch * CH_KS);
    base = reinterpret_cast<const uint4*>(p.L[u.l].w_packed);
    off = (unsigned)((u.tile * ks + u.ch * CH_KS) * 64 + lane);
}

// One unit: up to 13 16-deep steps of a 64 x 32 accumulator block (two MFMAs per step share the B fragment).  At the start of a
// unit its fragment set is fenced (the loads were issued after the previous unit's last MFMA); the steps then run without a
// single vector-memory wait, and the set is reloaded as a whole for the next unit.  (Reloading a fragment's registers right after
// ITS step looks better, but the compiler cannot count the loads in flight across the unit loop and puts s_waitcnt vmcnt(0) -- a
// full L2 round trip -- in front of every step: 6 us per unit instead of 0.4.)
// (the lane offset of the next loads passes through the fence too: otherwise the scheduler may hoist them above it; the offset,
// not the pointer -- a laundered pointer loses its address space and turns the loads into flat ones, which count against the LDS
// counter too)
__device__ __forceinline__ void chain_landed(uint4 (&q)[CH_KS], unsigned& next_b) {
    asm volatile("" : "+v"(q[0].x), "+v"(q[1].x), "+v"(q[2].x), "+v"(q[3].x), "+v"(q[4].x), "+v"(q[5].x), "+v"(q[6].x),
                 "+v"(q[7].x), "+v"(q[8].x), "+v"(q[9].x), "+v"(q[10].x), "+v"(q[11].x), "+v"(q[12].x), "+v"(next_b));
}
static_assert(CH_KS == 13, "chain_landed names every fragment");

// all 13 loads, unconditionally (steps past the unit's last re-read its last fragment): a conditional load leaves a register
// half-defined, and the compiler answers a set of those with copies and spills
__device__ __forceinline__ void chain_issue(uint4 (&q)[CH_KS], const uint4* base, unsigned off, int ksc) {
#pragma unroll
    for (int s = 0; s < CH_KS; ++s) q[s] = base[off + min(s, ksc - 1) * 64];
}

// The A fragments are read from LDS TWO steps ahead of the MFMAs that use them (an LDS round trip is longer than one step's
// two MFMAs); a scheduling barrier per step keeps the compiler from hoisting all 26 fragment reads (104 registers) to the top.
__device__ __forceinline__ void chain_mma(f32x16_t (&acc)[2], const uint4 (&q)[CH_KS], const uint16_t* a0, int ldk32, int ksc) {
    bf16x8 c0 = *reinterpret_cast<const bf16x8*>(a0), c1 = *reinterpret_cast<const bf16x8*>(a0 + ldk32);
    bf16x8 d0 = *reinterpret_cast<const bf16x8*>(a0 + 16), d1 = *reinterpret_cast<const bf16x8*>(a0 + ldk32 + 16);
#pragma unroll
    for (int s = 0; s < CH_KS; ++s)
        if (s < ksc) {
            bf16x8 n0 = d0, n1 = d1;
            if (s + 2 < CH_KS) {
                n0 = *reinterpret_cast<const bf16x8*>(a0 + (s + 2) * 16);
                n1 = *reinterpret_cast<const bf16x8*>(a0 + ldk32 + (s + 2) * 16);
            }
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, q[s]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, c0, acc[0], 0, 0, 0);      // W fragment first: lane <-> row
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, c1, acc[1], 0, 0, 0);
            c0 = d0;
            c1 = d1;
            d0 = n0;
            d1 = n1;
            __builtin_amdgcn_sched_barrier(0);
        }
}

// 64 rows x (cols / 8) 16-B pieces between global rows and an LDS tile; zero outside [0, m) x [0, cols).  (row, piece) advance
// incrementally -- no integer division per piece.
__device__ __forceinline__ void chain_stage(uint16_t* tile, int ldk, const uint16_t* src, int lds_, int m0, int m, int cols,
                                            int cols_padded) {
    const int ppr = cols_padded >> 3, total = CH_BM * ppr;
    const int drow = CH_THREADS / ppr, dpc = CH_THREADS - drow * ppr;
    int row = (int)threadIdx.x / ppr, pc = (int)threadIdx.x - row * ppr;
    for (int base = 0; base < total; base += CH_THREADS * 4) {
        uint4 v[4];
        int rw[4], kk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rw[j] = row;
            kk[j] = pc << 3;
            v[j] = make_uint4(0, 0, 0, 0);
            if (row < CH_BM && m0 + row < m && kk[j] < cols) v[j] = *reinterpret_cast<const uint4*>(src + (size_t)(m0 + row) * lds_ + kk[j]);
            row += drow;
            pc += dpc;
            if (pc >= ppr) { pc -= ppr; ++row; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rw[j] < CH_BM) *reinterpret_cast<uint4*>(tile + rw[j] * ldk + kk[j]) = v[j];
    }
}

// a layer's ReLU mask into its LDS tile, from a TRANSPOSED source [cols][ld] (a forward chain's out_bf16_t: what the MADE
// backward uses) or a row-major one [m][ld].  Every extra address register at a layer boundary is a spill (the next unit's 52
// fragment registers are in flight there), hence the rolled loops.  Rows past m are zero.
__device__ __forceinline__ void chain_stage_mask(uint16_t* tile, int ldk, const gv_chain_layer& L, int m0, int m) {
    if (L.mask_t) {
        // 16-B pieces = 8 consecutive rows of one column, four in flight; each lands through a ROLLED loop of 2-B LDS writes (one
        // address register walking down the rows, the piece shifted along in two 64-bit halves)
        const int total = L.n * (CH_BM / 8);
        for (int base = (int)threadIdx.x; base < total; base += CH_THREADS * 4) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(base + j * CH_THREADS, total - 1), r8 = (i & 7) << 3;
                v[j] = *reinterpret_cast<const uint4*>(L.mask_t + (size_t)(i >> 3) * L.ldmask_t + min(m0 + r8, (m - 1) & ~7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + j * CH_THREADS, r8 = (i & 7) << 3;
                if (i >= total) break;
                uint16_t* o = tile + r8 * ldk + (i >> 3);
                unsigned long long lo = ((unsigned long long)v[j].y << 32) | v[j].x, hi = ((unsigned long long)v[j].w << 32) | v[j].z;
                int left = m - m0 - r8;       // rows of the piece that exist (the others are zero)
#pragma unroll 1
                for (int e = 0; e < 8; ++e) {
                    *o = left > 0 ? (uint16_t)(lo & 0xffffu) : (uint16_t)0;
                    lo = (lo >> 16) | (hi << 48);
                    hi >>= 16;
                    o += ldk;
                    --left;
                }
            }
        }
        return;
    }
    // row-major [m][ld]: element by element, consecutive lanes along a row, eight 2-B loads in flight per lane
    const int cols = L.n, total = cols * CH_BM;
    const int dslow = CH_THREADS / cols, dfast = CH_THREADS - dslow * cols;
    int slow = (int)threadIdx.x / cols, fast = (int)threadIdx.x - slow * cols;
    for (int base = (int)threadIdx.x; base < total; base += CH_THREADS * 8) {
        uint16_t v[8];
        int at[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            at[j] = base + j * CH_THREADS < total ? slow * ldk + fast : -1;
            v[j] = (at[j] >= 0 && m0 + slow < m) ? L.mask[(size_t)(m0 + slow) * L.ldmask + fast] : (uint16_t)0;
            slow += dslow;
            fast += dfast;
            if (fast >= cols) { fast -= cols; ++slow; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (at[j] >= 0) tile[at[j]] = v[j];
    }
}

// MMA waves.  The weight fragment is the FIRST MFMA operand, so a lane owns ONE ROW of the tile (lane & 31, two accumulator
// tiles = rows r and 32 + r) and its 16 registers are 4 groups of 4 consecutive columns (8 g + 4 (lane >> 5) + 0..3):
//   next layer's LDS tile: one ds_write_b64 per group;  ReLU mask: one ds_read_b64 per group;  fp32 output: one 16-B store per
//   group;  transposed bf16 copy: per register a 2-B store whose 32 lanes are 32 consecutive rows = 64 contiguous bytes.
// (biases and the accumulate operand are fetched group by group, not up front: the kernel has to fit 128 registers)
constexpr bool LEAN = true;
template <bool FULL>
__device__ __forceinline__ void chain_epilogue(const f32x16_t (&acc)[2], const gv_chain_layer& Ly, int tile, int m0, int m,
                                               uint16_t* An, int ldk, int kp_next, const uint16_t* mbuf, const float* bias_l,
                                               const int* cnt_lds, const uint32_t* bits_l, int r, int h) {
    // opaque copies: without them the compiler hoists per-row predicates and 64-bit offsets out of the unit loop and spills
    asm volatile("" : "+v"(r), "+v"(h));
    float4 old[LEAN ? 1 : 2][LEAN ? 1 : 4];       // accumulate: all previous values requested at once (one round trip)
    const bool acc_old = FULL && Ly.out_f32 && Ly.accumulate;
    float4 bias4[LEAN ? 1 : 4];        // all four groups' biases requested before the first group's arithmetic (one LDS round trip)
    if constexpr (!LEAN) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = tile * 32 + 8 * g + 4 * h, row = mt * 32 + r;
                old[mt][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (acc_old && c0 < Ly.n && m0 + row < m)
                    old[mt][g] = *reinterpret_cast<const float4*>(Ly.out_f32 + (size_t)(m0 + row) * Ly.ldc + c0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = tile * 32 + 8 * g + 4 * h;
            bias4[g] = (Ly.bias && c0 < Ly.n) ? *reinterpret_cast<const float4*>(bias_l + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = tile * 32 + 8 * g + 4 * h;       // widths are multiples of 8: a group is inside or outside as a whole
        const bool cv = c0 < Ly.n;
        float4 bv;
        if constexpr (LEAN) bv = (Ly.bias && cv) ? *reinterpret_cast<const float4*>(bias_l + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        else bv = bias4[g];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 32 + r;
            float v[4] = {acc[mt][4 * g] + bv.x, acc[mt][4 * g + 1] + bv.y, acc[mt][4 * g + 2] + bv.z, acc[mt][4 * g + 3] + bv.w};
            if (Ly.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (FULL && (Ly.mask || Ly.mask_t) && cv) {
                const uint2 mv = *reinterpret_cast<const uint2*>(mbuf + row * ldk + c0);
                if ((int16_t)(mv.x & 0xffff) <= 0) v[0] = 0.f;
                if ((int16_t)(mv.x >> 16) <= 0) v[1] = 0.f;
                if ((int16_t)(mv.y & 0xffff) <= 0) v[2] = 0.f;
                if ((int16_t)(mv.y >> 16) <= 0) v[3] = 0.f;
            }
            if (!FULL && Ly.mask_bits && cv) {      // the row's word of this column tile (staged in LDS at the start of the kernel)
                const uint32_t wbits = bits_l[row * ((Ly.n + 31) >> 5) + tile] >> (8 * g + 4 * h);
                if (!(wbits & 1u)) v[0] = 0.f;
                if (!(wbits & 2u)) v[1] = 0.f;
                if (!(wbits & 4u)) v[2] = 0.f;
                if (!(wbits & 8u)) v[3] = 0.f;
            }
            if (Ly.out_f32 && cv && m0 + row < m) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (acc_old) {
                    float4 pv;
                    if constexpr (LEAN) pv = *reinterpret_cast<const float4*>(Ly.out_f32 + (size_t)(m0 + row) * Ly.ldc + c0);
                    else pv = old[mt][g];
                    o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
                }
                if (Ly.add_src) {      // the gradient the IAF update hands through: only where a column's count is 0 (counts in LDS)
                    const int4 ac = *reinterpret_cast<const int4*>(cnt_lds + c0);
                    if (ac.x <= 0 || ac.y <= 0 || ac.z <= 0 || ac.w <= 0) {
                        const float4 av = *reinterpret_cast<const float4*>(Ly.add_src + (size_t)(m0 + row) * Ly.ldc + c0);
                        if (ac.x <= 0) o.x += av.x;
                        if (ac.y <= 0) o.y += av.y;
                        if (ac.z <= 0) o.z += av.z;
                        if (ac.w <= 0) o.w += av.w;
                    }
                }
                *reinterpret_cast<float4*>(Ly.out_f32 + (size_t)(m0 + row) * Ly.ldc + c0) = o;
            }
            const uint16_t b0 = bf_bits(v[0]), b1 = bf_bits(v[1]), b2 = bf_bits(v[2]), b3 = bf_bits(v[3]);
            if (c0 < kp_next)
                *reinterpret_cast<uint2*>(An + row * ldk + c0) =
                    cv ? make_uint2(b0 | ((uint32_t)b1 << 16), b2 | ((uint32_t)b3 << 16)) : make_uint2(0, 0);
            // (tiled copies: the rows of the last tile past m are written as zeros -- they take part in the weight-gradient reduction,
            // and the caller's buffer then needs no fill)
            if (Ly.out_bf16_t && cv && (m0 + row < m || (!FULL && Ly.t_tile))) {      // (the lean instance: every MADE pass)
                const bool real = m0 + row < m;
                uint16_t* o = Ly.out_bf16_t + (size_t)c0 * Ly.ldt + (Ly.t_tile ? (int)blockIdx.x * Ly.t_tile : m0) + row;
                o[0] = real ? b0 : (uint16_t)0;
                o[Ly.ldt] = real ? b1 : (uint16_t)0;
                o[2 * (size_t)Ly.ldt] = real ? b2 : (uint16_t)0;
                o[3 * (size_t)Ly.ldt] = real ? b3 : (uint16_t)0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one group at a time keeps the address registers few
    }
}

// The LAST layer of a backward chain with the IAF-backward stage: fp32 output (dL/dx_old of the pass) + the gradient the update
// hands through where a column's count is 0 (counts in LDS) -- what chain_epilogue does for such a layer, with the pass's pointers
// given explicitly (the pass loop: pass q's output is pass q + 1's dL/dx_new).  Same arithmetic, element by element.
__device__ __forceinline__ void chain_epilogue_last(const f32x16_t (&acc)[2], int n, int tile, int m0, int m, const float* bias_l,
                                                    float* of, int ldc, const float* add, const int* cnt_lds, int r, int h, int add_rev = 0) {
    asm volatile("" : "+v"(r), "+v"(h));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = tile * 32 + 8 * g + 4 * h;
        if (c0 >= n) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias_l) bv = *reinterpret_cast<const float4*>(bias_l + c0);
        const int4 ac = *reinterpret_cast<const int4*>(cnt_lds + c0);
        const bool any0 = add && (ac.x <= 0 || ac.y <= 0 || ac.z <= 0 || ac.w <= 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 32 + r;
            if (m0 + row >= m) continue;
            float4 o = make_float4(acc[mt][4 * g] + bv.x, acc[mt][4 * g + 1] + bv.y, acc[mt][4 * g + 2] + bv.z, acc[mt][4 * g + 3] + bv.w);
            if (any0) {
                float4 av;
                if (add_rev) {      // (the handed-through gradient is dL/dx_new: reversed columns as in the stage)
                    const float4 t4 = *reinterpret_cast<const float4*>(add + (size_t)(m0 + row) * ldc + (n - 4 - c0));
                    av = make_float4(t4.w, t4.z, t4.y, t4.x);
                } else {
                    av = *reinterpret_cast<const float4*>(add + (size_t)(m0 + row) * ldc + c0);
                }
                if (ac.x <= 0) o.x += av.x;
                if (ac.y <= 0) o.y += av.y;
                if (ac.z <= 0) o.z += av.z;
                if (ac.w <= 0) o.w += av.w;
            }
            *reinterpret_cast<float4*>(of + (size_t)(m0 + row) * ldc + c0) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The epilogue of a HIDDEN layer of the fused MADE chains, in its common case: every row of the workgroup exists, the
// transposed copy goes out in 64-row tiles (t_tile), no fp32 output.  In-kernel stamps (tools/probes/chain_stamps.py) showed the
// general epilogue above at 5 000-7 000 cycles per unit against 1 600 for the unit's MFMAs -- ~900 instructions, most of them
// 64-bit address arithmetic, per-store predicates with a branch each and kernel-argument reloads between the stores.  Here the
// lane's address into the tile is ONE 32-bit offset computed once per kernel (voff = 4 h * 64 + r), every one of the 32
// transposed stores is base (scalar) + voff + an immediate, nothing is predicated per lane, and the layer's fields arrive as
// values.  bits_row: this lane's row of ReLU mask words (backward layers), or nullptr.
// PRED (the forward passes launch, whose last workgroup may hold fewer than 64 rows): rows past m leave as zeros -- LDS tile, tiled
// copy, sign bits -- and their sign words are not stored
template <bool PRED = false>
__device__ __forceinline__ void chain_epilogue_fast(const f32x16_t (&acc)[2], int n, int tile, int relu, uint16_t* An, int ldk, int kp_next,
                                                    const float* bias_l, const uint32_t* bits_l, int nt_bits, uint16_t* tbase,
                                                    uint32_t* obits, int ldbits, int r, int h, int last_row = CH_BM - 1) {
    asm volatile("" : "+v"(r), "+v"(h));
    uint32_t sb[2] = {0u, 0u};          // the signs of the lane's 16 columns of rows r / 32 + r, at their bit positions in the tile's word
    const unsigned voff = (unsigned)(4 * h * 64 + r);
    uint16_t* const tb = tbase + tile * 32 * 64;                 // column 32 tile of this workgroup's 64-row tile
    uint32_t w0 = 0xffffffffu, w1 = 0xffffffffu;
    if (bits_l) {
        w0 = bits_l[r * nt_bits + tile] >> (4 * h);
        w1 = bits_l[(32 + r) * nt_bits + tile] >> (4 * h);
    }
    if (PRED) {
        if (r > last_row) w0 = 0u;
        if (32 + r > last_row) w1 = 0u;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0u = tile * 32 + 8 * g;                       // widths are multiples of 8: both halves of a group are inside or outside
        if (c0u >= n) {
            if (c0u < kp_next) {                                  // the padding columns of the next layer's tile
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<uint2*>(An + (mt * 32 + r) * ldk + c0u + 4 * h) = make_uint2(0, 0);
            }
            continue;
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias_l) bv = *reinterpret_cast<const float4*>(bias_l + c0u + 4 * h);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float v[4] = {acc[mt][4 * g] + bv.x, acc[mt][4 * g + 1] + bv.y, acc[mt][4 * g + 2] + bv.z, acc[mt][4 * g + 3] + bv.w};
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const uint32_t wb = (mt ? w1 : w0) >> (8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(wb & (1u << e))) v[e] = 0.f;
            const uint16_t b0 = bf_bits(v[0]), b1 = bf_bits(v[1]), b2 = bf_bits(v[2]), b3 = bf_bits(v[3]);
            if (!(GV_CHAIN_ABL & 512)) *reinterpret_cast<uint2*>(An + (mt * 32 + r) * ldk + c0u + 4 * h) = make_uint2(b0 | ((uint32_t)b1 << 16), b2 | ((uint32_t)b3 << 16));
            uint16_t* o = tb + voff + g * 8 * 64 + mt * 32;
            if (!(GV_CHAIN_ABL & 256)) {
                o[0] = b0;
                o[64] = b1;
                o[128] = b2;
                o[192] = b3;
            }
            sb[mt] |= (((int16_t)b0 > 0 ? 1u : 0u) | ((int16_t)b1 > 0 ? 2u : 0u) | ((int16_t)b2 > 0 ? 4u : 0u) | ((int16_t)b3 > 0 ? 8u : 0u)) << (8 * g + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (obits && !(GV_CHAIN_ABL & 1024)) {
        // ReLU sign words (gv_chain_layer.out_bits) straight from the registers: a row's word is the OR of its two lanes (h = 0, 1);
        // the store wave -- which used to rebuild the words from the LDS tile, 7 000 cycles per layer once the epilogues above had
        // shrunk to 4 000 -- skips the layer
        const int partner = (int)(((threadIdx.x & 63) ^ 32) << 2);
        const uint32_t o0 = (uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)sb[0]), o1 = (uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)sb[1]);
        if (h == 0) {
            if (!PRED || r <= last_row) obits[(size_t)r * ldbits + tile] = sb[0] | o0;
            if (!PRED || 32 + r <= last_row) obits[(size_t)(32 + r) * ldbits + tile] = sb[1] | o1;
        }
    }
}

// The IAF update in the last layer's epilogue (kgvae/flow_network.py:92-96): the layer's weight is packed so that a tile holds
// 16 mu columns (accumulator groups 0, 1) and the SAME columns' alpha (groups 2, 3), so a lane has both halves of
//   x_new[r][c] = colcount[c] > 0 ? z[r][c] * expf(alpha + mu) : x_old[r][c]
// in registers.  x_new leaves as fp32 (iaf_x_new), as the bf16 tile the store wave copies out (out_bf16) and transposed
// (out_bf16_t); expf(alpha + mu) (iaf_ex: all the backward needs), alpha (iaf_alpha: the log-det row sums of the last pass) and
// [mu | alpha] (out_f32, natural column order) are optional.  Same arithmetic, element by element, as gv_iaf_update_fwd_bf16
// on the stored [mu | alpha].
__device__ __forceinline__ void chain_epilogue_iaf(const f32x16_t (&acc)[2], const gv_chain_layer& Ly, int tile,
                                                   int m0, int m, uint16_t* An, int ldk, int kp_next, const float* bias_l,
                                                   const int* cnt_lds, int r, int h) {
    asm volatile("" : "+v"(r), "+v"(h));
    const int d = Ly.n >> 1;
    // z of the lane's four (column group, row) pairs FIRST (clamped, unconditional): loads and stores share one in-order counter
    // on this ISA, so a load issued behind the epilogue's stores waits for every one of them to be acknowledged (measured:
    // 131 us per chain with the loads interleaved, ~75 with them in front).  16 registers: the caller requests the next unit's
    // weight fragments AFTER this epilogue, not before it.
    float4 zp[4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            zp[2 * g + mt] = (Ly.iaf_reserved & 1) ? make_float4(1.f, 1.f, 1.f, 1.f)
                                                   : *reinterpret_cast<const float4*>(Ly.iaf_z + (size_t)min(m0 + mt * 32 + r, m - 1) * Ly.iaf_ld +
                                                                                      min(tile * 16 + 8 * g + 4 * h, d - 4));
    int4 cn[2];
    float4 xo[4];
    bool keep[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {      // ... and x_old where a column of the group is passed through (the middle passes' last column)
        const int c0 = tile * 16 + 8 * g + 4 * h;
        cn[g] = c0 < d ? *reinterpret_cast<const int4*>(cnt_lds + c0) : make_int4(1, 1, 1, 1);
        keep[g] = true;
        if (Ly.iaf_keep_colcount && c0 < d) {
            const int4 kc = *reinterpret_cast<const int4*>(Ly.iaf_keep_colcount + c0);
            keep[g] = kc.x <= 0 || kc.y <= 0 || kc.z <= 0 || kc.w <= 0;
        }
        const bool pass = cn[g].x <= 0 || cn[g].y <= 0 || cn[g].z <= 0 || cn[g].w <= 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xo[2 * g + mt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pass && m0 + mt * 32 + r < m)
                xo[2 * g + mt] = *reinterpret_cast<const float4*>(Ly.iaf_x_old + (size_t)(m0 + mt * 32 + r) * Ly.iaf_ld + c0);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int c0 = tile * 16 + 8 * g + 4 * h;
            const bool cv = c0 < d;
            const int row = mt * 32 + r;
            const bool live = cv && m0 + row < m;
            const size_t e = (size_t)(m0 + row) * Ly.iaf_ld + c0;
            float4 v = make_float4(acc[mt][4 * g + 8], acc[mt][4 * g + 9], acc[mt][4 * g + 10], acc[mt][4 * g + 11]);      // alpha
            float4 s = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);            // mu
            if (Ly.bias && cv) {
                const float4 ba = *reinterpret_cast<const float4*>(bias_l + d + c0), bm = *reinterpret_cast<const float4*>(bias_l + c0);
                v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
                s.x += bm.x; s.y += bm.y; s.z += bm.z; s.w += bm.w;
            }
            const int dbg = Ly.iaf_reserved;      // ablation switches of tools/probes/chain_iaf_probe.py (0 in production)
            if (live && Ly.out_f32 && !(dbg & 2)) {
                float* o = Ly.out_f32 + (size_t)(m0 + row) * Ly.ldc + c0;
                *reinterpret_cast<float4*>(o) = s;
                *reinterpret_cast<float4*>(o + d) = v;
            }
            if (live && Ly.iaf_alpha && !(dbg & 2)) *reinterpret_cast<float4*>(Ly.iaf_alpha + e) = v;
            if (!(dbg & 16)) { v.x = expf(v.x + s.x); v.y = expf(v.y + s.y); v.z = expf(v.z + s.z); v.w = expf(v.w + s.w); }
            if (live && Ly.iaf_ex && !(dbg & 2)) *reinterpret_cast<float4*>(Ly.iaf_ex + e) = v;
            s = zp[2 * g + mt];
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
            s = xo[2 * g + mt];
            if (cn[g].x <= 0) v.x = s.x;
            if (cn[g].y <= 0) v.y = s.y;
            if (cn[g].z <= 0) v.z = s.z;
            if (cn[g].w <= 0) v.w = s.w;
            if (live && Ly.iaf_x_new && keep[g] && !(dbg & 2)) *reinterpret_cast<float4*>(Ly.iaf_x_new + e) = v;
            const uint16_t b0 = bf_bits(v.x), b1 = bf_bits(v.y), b2 = bf_bits(v.z), b3 = bf_bits(v.w);
            if (c0 < kp_next && !(dbg & 8))
                *reinterpret_cast<uint2*>(An + row * ldk + c0) =
                    cv ? make_uint2(b0 | ((uint32_t)b1 << 16), b2 | ((uint32_t)b3 << 16)) : make_uint2(0, 0);
            if (Ly.out_bf16_t && cv && (live || Ly.t_tile) && !(dbg & 4)) {      // (tiled copies: zeros in the rows past m, as above)
                uint16_t* o = Ly.out_bf16_t + (size_t)c0 * Ly.ldt + (Ly.t_tile ? (int)blockIdx.x * Ly.t_tile : m0) + row;
                o[0] = live ? b0 : (uint16_t)0;
                o[Ly.ldt] = live ? b1 : (uint16_t)0;
                o[2 * (size_t)Ly.ldt] = live ? b2 : (uint16_t)0;
                o[3 * (size_t)Ly.ldt] = live ? b3 : (uint16_t)0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same update in its common case (every row of the workgroup exists, no [mu | alpha] output, transposed copy in 64-row tiles
// or none): as chain_epilogue_fast, the lane's place in the [m][ld] fp32 operands is two 32-bit offsets computed once per unit
// (rows r and 32 + r), every access is a pinned scalar base + that offset + an immediate, nothing is predicated per lane and the
// layer's fields arrive as values.  Stamps: 12 000-18 000 cycles per unit with the general form above.  Same arithmetic, element
// by element.
struct ChainIafPins {
    const float* z; const float* x_old; float* x_new; float* ex; float* alpha; const int* keep; uint16_t* tbase;
    int ld, d, has_bias, identity;
    int rev;      // x_new leaves with its columns reversed (the PermuteLayer behind an IAF block, kgvae/model.py:60-66, folded in)
};
// PRED: the workgroup may hold fewer than 64 rows (last_row = its last one): loads of the rows past it read row last_row, their
// stores are skipped, their bf16 copies (LDS tile, tiled transposed copy) are zeros
template <bool PRED = false>
__device__ __forceinline__ void chain_epilogue_iaf_fast(const f32x16_t (&acc)[2], const ChainIafPins& P, int tile, uint16_t* An, int ldk,
                                                        int kp_next, const float* bias_l, const int* cnt_lds, int r, int h, int last_row = CH_BM - 1) {
    asm volatile("" : "+v"(r), "+v"(h));
    const int d = P.d, cu = tile * 16;                       // first mu column of the tile
    const bool live[2] = {!PRED || r <= last_row, !PRED || 32 + r <= last_row};
    const unsigned vo[2] = {(unsigned)((PRED ? min(r, last_row) : r) * P.ld + 4 * h),
                            (unsigned)((PRED ? min(32 + r, last_row) : 32 + r) * P.ld + 4 * h)};
    const unsigned voff_t = (unsigned)(4 * h * 64 + r);
    // z of the lane's four (group, row) pairs first, then x_old where a column is handed through: loads and stores share one
    // in-order counter, a load issued behind the epilogue's stores waits for every one of them
    float4 zp[4], xo[4];
    int4 cn[2];
    bool keep[2], pass[2], gv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        gv[g] = cu + 8 * g < d;                              // (d % 8 == 0: both halves of a group are inside or outside)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            zp[2 * g + mt] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (gv[g] && !P.identity) zp[2 * g + mt] = *reinterpret_cast<const float4*>(P.z + cu + 8 * g + vo[mt]);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        cn[g] = gv[g] ? *reinterpret_cast<const int4*>(cnt_lds + cu + 8 * g + 4 * h) : make_int4(1, 1, 1, 1);
        keep[g] = true;
        if (P.keep && gv[g]) {
            const int4 kc = *reinterpret_cast<const int4*>(P.keep + cu + 8 * g + 4 * h);
            keep[g] = kc.x <= 0 || kc.y <= 0 || kc.z <= 0 || kc.w <= 0;
        }
        pass[g] = cn[g].x <= 0 || cn[g].y <= 0 || cn[g].z <= 0 || cn[g].w <= 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xo[2 * g + mt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pass[g]) xo[2 * g + mt] = *reinterpret_cast<const float4*>(P.x_old + cu + 8 * g + vo[mt]);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!gv[g]) {
            if (cu + 8 * g < kp_next) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<uint2*>(An + (mt * 32 + r) * ldk + cu + 8 * g + 4 * h) = make_uint2(0, 0);
            }
            continue;
        }
        float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bm = ba;
        if (P.has_bias) {
            ba = *reinterpret_cast<const float4*>(bias_l + d + cu + 8 * g + 4 * h);
            bm = *reinterpret_cast<const float4*>(bias_l + cu + 8 * g + 4 * h);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float4 v = make_float4(acc[mt][4 * g + 8], acc[mt][4 * g + 9], acc[mt][4 * g + 10], acc[mt][4 * g + 11]);      // alpha
            float4 s = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);            // mu
            v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
            s.x += bm.x; s.y += bm.y; s.z += bm.z; s.w += bm.w;
            const unsigned e = vo[mt] + (unsigned)(cu + 8 * g);
            if (P.alpha && live[mt]) *reinterpret_cast<float4*>(P.alpha + e) = v;
            v.x = expf(v.x + s.x); v.y = expf(v.y + s.y); v.z = expf(v.z + s.z); v.w = expf(v.w + s.w);
            if (P.ex && live[mt]) *reinterpret_cast<float4*>(P.ex + e) = v;
            s = zp[2 * g + mt];
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
            s = xo[2 * g + mt];
            if (cn[g].x <= 0) v.x = s.x;
            if (cn[g].y <= 0) v.y = s.y;
            if (cn[g].z <= 0) v.z = s.z;
            if (cn[g].w <= 0) v.w = s.w;
            if (P.x_new && keep[g] && live[mt]) {
                if (P.rev) *reinterpret_cast<float4*>(P.x_new + (vo[mt] + (unsigned)(d - 4 - cu - 8 * g - 8 * h))) = make_float4(v.w, v.z, v.y, v.x);
                else *reinterpret_cast<float4*>(P.x_new + e) = v;
            }
            uint16_t b0 = bf_bits(v.x), b1 = bf_bits(v.y), b2 = bf_bits(v.z), b3 = bf_bits(v.w);
            if (PRED && !live[mt]) b0 = b1 = b2 = b3 = 0;
            if (cu + 8 * g < kp_next)
                *reinterpret_cast<uint2*>(An + (mt * 32 + r) * ldk + cu + 8 * g + 4 * h) = make_uint2(b0 | ((uint32_t)b1 << 16), b2 | ((uint32_t)b3 << 16));
            if (P.tbase) {
                uint16_t* o = P.tbase + (cu + 8 * g) * 64 + voff_t + mt * 32;
                o[0] = b0;
                o[64] = b1;
                o[128] = b2;
                o[192] = b3;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// store wave: the row-major bf16 copy of a layer's result out of its LDS tile, 16-B pieces, reads issued in batches of 8.
// (row, piece) advance incrementally: an integer division per piece would cost this single wave more than the copy itself
__device__ __forceinline__ void chain_store_rows(uint16_t* out, int ldb, int ppr, const uint16_t* An, int ldk, int m0, int m, int ts);
template <bool FULL>
__device__ __forceinline__ void chain_store(const gv_chain_layer& Ly, const uint16_t* An, int ldk, int m0, int m, int ts, bool bits_done) {
    if (!FULL && Ly.out_bits && !bits_done) {       // sign bits of the layer's (bf16-rounded) result, word [row][tile of 32 columns], out of the finished tile
        const int nt = (Ly.n + 31) >> 5;
        for (int i = ts; i < CH_BM * nt; i += CH_STORE_THREADS) {
            const int row = i / nt, t = i - row * nt;
            uint32_t word = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = t * 32 + q * 8;
                if (c < Ly.n) {      // widths are multiples of 8: a piece is inside or outside as a whole
                    const uint4 v = *reinterpret_cast<const uint4*>(An + row * ldk + c);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        word |= (((int16_t)(w[e] & 0xffffu) > 0 ? 1u : 0u) | ((int16_t)(w[e] >> 16) > 0 ? 2u : 0u)) << (q * 8 + 2 * e);
                }
            }
            if (m0 + row < m) Ly.out_bits[(size_t)(m0 + row) * Ly.ldbits + t] = word;
        }
    }
    if (!Ly.out_bf16) return;
    chain_store_rows(Ly.out_bf16, Ly.ldb, (Ly.iaf_z ? Ly.n >> 1 : Ly.n) >> 3, An, ldk, m0, m, ts);      // an IAF layer's tile holds x_new: d columns
}
__device__ __forceinline__ void chain_store_rows(uint16_t* out, int ldb, int ppr, const uint16_t* An, int ldk, int m0, int m, int ts) {
    const int total = CH_BM * ppr;
    const int drow = CH_STORE_THREADS / ppr, dpc = CH_STORE_THREADS - drow * ppr;      // one step of 64 pieces
    int row = ts / ppr, pc = ts - row * ppr;
    for (int base = 0; base < total; base += CH_STORE_THREADS * 8) {
        uint4 v[8];
        int rw[8], kk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rw[j] = min(row, CH_BM - 1);                  // unconditional reads: no half-defined registers
            kk[j] = pc << 3;
            v[j] = *reinterpret_cast<const uint4*>(An + rw[j] * ldk + kk[j]);
            rw[j] = row;
            row += drow;
            pc += dpc;
            if (pc >= ppr) { pc -= ppr; ++row; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (rw[j] < CH_BM && m0 + rw[j] < m) *reinterpret_cast<uint4*>(out + (size_t)(m0 + rw[j]) * ldb + kk[j]) = v[j];
    }
}

// The IAF update's BACKWARD as the first stage of a backward chain (gv_made_chain_iafb; kgvae/flow_network.py:92-96 differentiated):
// per element, from dL/dx_new (gx), ex = exp(alpha + mu) (what the forward chain kept), z and the pass's column counts,
//   count > 0:  g_z = gx count ex,  g_mu = gx count z ex,  g_alpha = g_logdet + g_mu;   count == 0:  g_mu = 0, g_alpha = g_logdet
// -- the arithmetic of gv_iaf_update_bwd_bf16_ex, element by element.  [g_mu | g_alpha] goes as bf16 straight into layer 0's LDS
// tile (the separate launch wrote it row-major to memory and the chain read it back), its transposed copy (the operand of the last
// layer's weight gradient) leaves through a [column][row] LDS block of 64 columns at a time, g_z is added in place.  Four column
// blocks per 200-wide row; the next block's operands are requested before the current one leaves.
constexpr int IB_LD = 68;      // rows of the transposition block, padded (8-B reads of four rows, conflict-free 2-B writes)
constexpr int IB_COLS = 32;    // its columns (g_mu and g_alpha each: 2 x 32 x 68 x 2 B = 8.5 KB -- with 64 the backward chain of a 200-wide MADE
                               // would not fit twice into a CU's LDS)
__device__ __forceinline__ void chain_stage_iafb(uint16_t* tile, int ldk, uint16_t* tmx, const gv_chain_iafb& ib, int m0, int m,
                                                 const float* ex_q, const float* gx_q, const float* gld_q, const int32_t* cc_q,
                                                 uint16_t* gnt_q, int flags_q) {
    const int d = ib.d, nblk = (d + IB_COLS - 1) / IB_COLS;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));      // opaque: inside the pass loop everything derived from it would be hoisted and held across the unit loop
    uint16_t (*tm)[IB_LD] = reinterpret_cast<uint16_t (*)[IB_LD]>(tmx);                      // [32 columns][rows]: g_mu
    uint16_t (*ta)[IB_LD] = reinterpret_cast<uint16_t (*)[IB_LD]>(tmx + IB_COLS * IB_LD);    // g_alpha
    // a block is 64 rows x 8 pieces of four columns = 512 pieces: one per thread
    const int cq = (t & 7) << 2, rr = t >> 3, r = m0 + rr;
    struct Ops { float4 g, e, z, a; int4 cn; };
    // pinned (scalar) bases + ONE 32-bit byte offset: 64-bit per-lane addresses of four arrays were what spilled
    const float* const gx_p = chain_pin_here(gx_q);
    const float* const ex_p = chain_pin_here(ex_q);
    const float* const z_p = chain_pin_here(ib.z);
    float* const gz_p = chain_pin_here(ib.gz);
    const int ld = chain_pin_here(ib.ld), fl = chain_pin_here(flags_q);
    const unsigned off0 = (unsigned)(r * ld + cq) * 4u;
    auto at = [](const float* base, unsigned byte_off) { return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off); };
    auto request = [&](int blk, Ops& o) {
        const int c = blk * IB_COLS + cq;
        o.cn = make_int4(0, 0, 0, 0);
        if (c < d) o.cn = *reinterpret_cast<const int4*>(cc_q + c);
        const bool any = o.cn.x > 0 || o.cn.y > 0 || o.cn.z > 0 || o.cn.w > 0;
        const unsigned e = off0 + (unsigned)blk * (IB_COLS * 4u);
        o.g = o.e = o.z = o.a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < m && c < d) {
            if (fl & 2) {       // dL/dx_new arrives with its columns reversed (the PermuteLayer behind the block, folded in)
                const float4 t4 = *at(gx_p, off0 - (unsigned)cq * 4u + (unsigned)(d - 4 - c) * 4u);
                o.g = make_float4(t4.w, t4.z, t4.y, t4.x);
            } else {
                o.g = *at(gx_p, e);
            }
            if (!(fl & 1)) o.a = *at(gz_p, e);
            if (any) {
                o.z = *at(z_p, e);
                o.e = *at(ex_p, e);
            }
        }
    };
    Ops cur, nxt;
    request(0, cur);
    for (int blk = 0; blk < nblk; ++blk) {
        if (blk + 1 < nblk) request(blk + 1, nxt);      // the next block's operands fly under this block's arithmetic and way out
        const int c = blk * IB_COLS + cq;
        const int cn[4] = {cur.cn.x, cur.cn.y, cur.cn.z, cur.cn.w};
        {
            const float gv[4] = {cur.g.x, cur.g.y, cur.g.z, cur.g.w}, zv[4] = {cur.z.x, cur.z.y, cur.z.z, cur.z.w},
                        ev[4] = {cur.e.x, cur.e.y, cur.e.z, cur.e.w};
            uint16_t bm[4] = {0, 0, 0, 0}, ba[4] = {0, 0, 0, 0};
            if (r < m && c < d) {
                const float gl = gld_q ? gld_q[r] : 0.f;
                float gz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float g_mu = 0.f, g_al = gl, g_z = 0.f;
                    if (cn[q] > 0) {
                        const float gc = gv[q] * (float)cn[q];
                        g_z = gc * ev[q];
                        g_mu = gc * zv[q] * ev[q];
                        g_al += g_mu;
                    }
                    gz[q] = g_z;
                    bm[q] = bf_bits(g_mu);
                    ba[q] = bf_bits(g_al);
                }
                float4 acc4 = cur.a;
                acc4.x += gz[0]; acc4.y += gz[1]; acc4.z += gz[2]; acc4.w += gz[3];
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(gz_p) + off0 + (unsigned)blk * (IB_COLS * 4u)) = acc4;
            }
            if (c < d) {      // layer 0's input row: [g_mu | g_alpha] (rows past m: zeros)
                *reinterpret_cast<uint2*>(tile + rr * ldk + c) = make_uint2(bm[0] | ((uint32_t)bm[1] << 16), bm[2] | ((uint32_t)bm[3] << 16));
                *reinterpret_cast<uint2*>(tile + rr * ldk + d + c) = make_uint2(ba[0] | ((uint32_t)ba[1] << 16), ba[2] | ((uint32_t)ba[3] << 16));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tm[cq + q][rr] = bm[q];
                ta[cq + q][rr] = ba[q];
            }
        }
        __syncthreads();
        // the block's transposed copies: four consecutive rows of a column per 8-B store (16 lanes = 128 contiguous bytes); the whole
        // 64-row tile is written (zeros in the rows past m: they take part in the weight-gradient reduction)
        uint16_t* const gt = gnt_q + (size_t)blockIdx.x * ib.t_tile;
        for (int i = t; i < 2 * IB_COLS * 16; i += CH_THREADS) {
            const int half = i / (IB_COLS * 16), cc = (i >> 4) & (IB_COLS - 1), rq = (i & 15) << 2;
            if (blk * IB_COLS + cc >= d) continue;
            const uint2 v = *reinterpret_cast<const uint2*>(half ? &ta[cc][rq] : &tm[cc][rq]);
            *reinterpret_cast<uint2*>(gt + (size_t)(half * d + blk * IB_COLS + cc) * 64 + rq) = v;
        }
        __syncthreads();
        if (blk + 1 < nblk) cur = nxt;
    }
    // the padding columns [2 d, ldk) of the tile: zeros (the last 16-deep step of layer 0 may reach into them)
    for (int i = t; i < CH_BM * (((2 * d + 15) & ~15) - 2 * d); i += CH_THREADS) {
        const int pw = ((2 * d + 15) & ~15) - 2 * d;
        tile[(i / pw) * ldk + 2 * d + i % pw] = 0;
    }
}

// Pass 0's IAF update as the first stage of the forward passes launch (gv_made_chain_fwd_row0): the net's output of pass 0 is ONE row
// ([mu | alpha], the MADE of a zero input: kgvae/flow_network.py:88-97 with x = 0), every column's count is positive, so
//   x[r][c] = z[r][c] * expf(alpha[c] + mu[c])
// -- the arithmetic of gv_iaf_update_fwd_bf16_tiles on a broadcast row.  x leaves as fp32 (the passes' x_old), as bf16 into layer 0's
// LDS tile and, through a [column][row] block of 32 columns, as the tiled transposed copy; rows past m: zeros in both bf16 forms.
__device__ __forceinline__ void chain_stage_row0(uint16_t* tile, int ldk, uint16_t* tmx, const gv_chain_fwd_row0& f, const float* z, int ld,
                                                 int d, int t_tile, int m0, int m) {
    const int nblk = (d + IB_COLS - 1) / IB_COLS;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    uint16_t (*tm)[IB_LD] = reinterpret_cast<uint16_t (*)[IB_LD]>(tmx);
    const int cq = (t & 7) << 2, rr = t >> 3, r = m0 + rr;      // a block is 64 rows x 8 pieces of four columns: one per thread
    const float* const z_p = chain_pin_here(z);
    float* const x_p = chain_pin_here(f.x_f32);
    const float* const n_p = chain_pin_here(f.net_row);
    const unsigned off0 = (unsigned)(r * ld + cq) * 4u;
    uint16_t* const gt = f.x_t + (size_t)blockIdx.x * t_tile;
    for (int blk = 0; blk < nblk; ++blk) {
        const int c = blk * IB_COLS + cq;
        uint16_t b[4] = {0, 0, 0, 0};
        if (r < m && c < d) {
            const unsigned e = off0 + (unsigned)blk * (IB_COLS * 4u);
            const float4 zv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(z_p) + e);
            const float4 mu = *reinterpret_cast<const float4*>(n_p + c), al = *reinterpret_cast<const float4*>(n_p + d + c);
            float4 x;
            x.x = zv.x * expf(al.x + mu.x);
            x.y = zv.y * expf(al.y + mu.y);
            x.z = zv.z * expf(al.z + mu.z);
            x.w = zv.w * expf(al.w + mu.w);
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(x_p) + e) = x;
            b[0] = bf_bits(x.x); b[1] = bf_bits(x.y); b[2] = bf_bits(x.z); b[3] = bf_bits(x.w);
        }
        if (c < d) *reinterpret_cast<uint2*>(tile + rr * ldk + c) = make_uint2(b[0] | ((uint32_t)b[1] << 16), b[2] | ((uint32_t)b[3] << 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) tm[cq + q][rr] = b[q];
        __syncthreads();
        {       // the block's transposed copy: four consecutive rows of a column per 8-B store; 32 x 16 pieces = one per thread
            const int cc = (t >> 4) & (IB_COLS - 1), rq = (t & 15) << 2;
            if (blk * IB_COLS + cc < d)
                *reinterpret_cast<uint2*>(gt + (size_t)(blk * IB_COLS + cc) * 64 + rq) = *reinterpret_cast<const uint2*>(&tm[cc][rq]);
        }
        __syncthreads();
    }
    for (int i = t; i < CH_BM * (((d + 15) & ~15) - d); i += CH_THREADS) {      // the tile's padding columns [d, 16 ceil(d / 16))
        const int pw = ((d + 15) & ~15) - d;
        tile[(i / pw) * ldk + d + i % pw] = 0;
    }
}

// Every wave walks its own list of units (layer, column tile, 13-step chunk) in ONE flat loop.  Between units a wave crosses layer
// boundaries; crossing layer l -> l + 1 is the same for every wave of the workgroup:
//   barrier (layer l's tile complete)  ->  [mask of layer l + 1 staged by everyone, barrier]  ->  store wave: layer l's copies.
// ONE fragment register set, reloaded for the wave's next unit right after the unit's last MFMA: the loads land during the
// epilogue and the barrier (2-3 us, several L2 round trips).  With it the kernel fits 128 registers per lane = four waves per
// SIMD = TWO workgroups per CU where the LDS tiles allow (the forward chain: 60 KB), which is what counts once there are more row
// tiles than CUs (WN18RR: 640 tiles, forward chain 93 -> 72 us).  A second set requested a whole unit ahead (the first design,
// two unit bodies with fixed set roles) measured the same at 228 tiles and cost 52 registers.
// FULL = false: the instance for chains without tile masks (mask / mask_t) and without an accumulating fp32 output -- every
// MADE pass of the fused path; without that code it keeps clear of the 128-register limit (the full instance spills ~30 B)
// FW: the forward passes launch (gv_made_chain_fwd): ALL passes of a MADE's forward; p.pass[q] holds pass q's outputs and IAF operands,
// pass q's x_new stays in LDS as pass q + 1's input (the buffer roles swap from pass to pass when the layer count is odd)
template <bool FULL, bool IB, bool FW>
__device__ __forceinline__ void chain_body(const ChainArgs& p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t chain_lds[];
    const int ldk = p.ldk, nl = p.n_layers, m0 = blockIdx.x * CH_BM;
    uint16_t* mbuf = chain_lds + 2 * CH_BM * ldk;
    float* bias_lds = reinterpret_cast<float*>(chain_lds + (p.has_mask ? 3 : 2) * CH_BM * ldk);
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int r = lane & 31, h = lane >> 5, tid = threadIdx.x;
    const bool mma_wave = wave < CH_MMA_WAVES;
    // IB, layer 0's 2 d-wide input tile at its own row stride, laid over BOTH activation buffers (the others are sized for the
    // hidden widths: 134 -> 76 KB of LDS for a 200-wide MADE, two workgroups per CU): layer 0's output then lands on columns its
    // input still holds -- one more barrier, between the layer's last MFMA and its epilogues (every wave has at most one tile of
    // it: host-checked)
    const int ld0 = IB ? p.ld0 : ldk;
    const bool wide0 = IB && p.ld0 != p.ldk;
    const bool l0_unit = mma_wave && wave < chain_tiles(p.L[0]);
    int ts_n = 0;
    auto stamp = [&]() {
        if (p.stamps && blockIdx.x == 0 && ts_n < 64) {
            const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
            if (lane == 0) p.stamps[wave * 64 + ts_n] = (int32_t)t;
            ++ts_n;
        }
    };
    stamp();

    // the first B fragments of an MMA wave are requested before anything else
    uint4 qa[CH_KS];
    ChainUnit u = {nl, wave, 0};
    const uint4* const any_b = reinterpret_cast<const uint4*>(p.L[0].w_packed);      // a readable address for idle loads
    const uint4* first_b = any_b;
    unsigned first_off = 0;
    int first_ksc = 1;
    {
        const uint4* b0 = any_b;
        unsigned off = lane;
        int ksc = 1;
        if (mma_wave) {
            u.l = 0;
            while (u.l < nl && chain_first_tile(p.L[u.l], wave) >= chain_tiles(p.L[u.l])) ++u.l;
            if (u.l < nl) u.tile = chain_first_tile(p.L[u.l], wave);
            if (u.l < nl) chain_unit_b(p, u, lane, b0, off, ksc);
        }
        first_b = b0; first_off = off; first_ksc = ksc;
        // (the IAF-backward instance requests them behind its stage: the stage's two operand sets need the registers)
        // (... and the forward passes launch at the head of every pass)
        if constexpr (!IB && !FW) chain_issue(qa, b0, off, ksc);
    }
    int bias_total = 0, bits_base = 0;
    {       // every layer's bias into LDS (all loads in flight together): no global round trip in an epilogue
        float bv[CH_L];
        int off = 0;
#pragma unroll
        for (int l = 0; l < CH_L; ++l) {
            bv[l] = 0.f;
            if (l < nl && p.L[l].bias && (int)threadIdx.x < p.L[l].n) bv[l] = p.L[l].bias[threadIdx.x];
        }
#pragma unroll
        for (int l = 0; l < CH_L; ++l)
            if (l < nl) {
                if ((int)threadIdx.x < p.L[l].n) bias_lds[off + threadIdx.x] = bv[l];
                off += p.L[l].n;
            }
        bias_total = off;
        const gv_chain_layer& Ll = p.L[nl - 1];      // an IAF layer's column counts behind the biases
        if (Ll.iaf_z && (int)threadIdx.x < (Ll.n >> 1))
            reinterpret_cast<int*>(bias_lds + off)[threadIdx.x] = (FW ? p.pass[0].cc : Ll.iaf_colcount)[threadIdx.x];
        else if (Ll.add_src && (int)threadIdx.x < Ll.n)
            reinterpret_cast<int*>(bias_lds + off)[threadIdx.x] = Ll.add_colcount[threadIdx.x];
        off += Ll.iaf_z ? Ll.n >> 1 : (Ll.add_src ? Ll.n : 0);
        bits_base = off;
    }
    if constexpr (!FULL) {       // every layer's mask bits of this workgroup's 64 rows behind them ([row][tile] words), requested together as well
        uint32_t bw[CH_L];
#pragma unroll
        for (int l = 0; l < CH_L; ++l) {
            bw[l] = 0u;
            if (l < nl && p.L[l].mask_bits) {
                const int nt = (p.L[l].n + 31) >> 5, row = (int)threadIdx.x / nt, t = (int)threadIdx.x - row * nt;
                if (row < CH_BM && m0 + row < p.m) bw[l] = p.L[l].mask_bits[(size_t)(m0 + row) * p.L[l].ldbits + t];
            }
        }
        uint32_t* bl = reinterpret_cast<uint32_t*>(bias_lds + bits_base);
#pragma unroll
        for (int l = 0; l < CH_L; ++l)
            if (l < nl && p.L[l].mask_bits) {
                const int nt = (p.L[l].n + 31) >> 5;
                if ((int)threadIdx.x < CH_BM * nt) bl[threadIdx.x] = bw[l];
                for (int i = (int)threadIdx.x + CH_THREADS; i < CH_BM * nt; i += CH_THREADS) {      // layers wider than 256 columns
                    const int row = i / nt, t = i - row * nt;
                    bl[i] = m0 + row < p.m ? p.L[l].mask_bits[(size_t)(m0 + row) * p.L[l].ldbits + t] : 0u;
                }
                bl += CH_BM * nt;
            }
    }
    // IB: ALL passes of a MADE's backward in this launch (p.n_passes; p.pass[q] holds what differs from pass to pass: pass q's
    // dL/dx_new is pass q - 1's fp32 output).  A workgroup keeps its 64 rows through the passes: every step of a pass is row-local.
    int q = 0, par = 0;          // par (FW): the buffer that holds layer 0's input of this pass
    const int last_row = min(CH_BM, p.m - m0) - 1;
  next_pass:
    if constexpr (FW) {
        asm volatile("" : "+v"(lane));      // (per pass from an opaque copy, as below)
        tid = lane | (wave << 6); r = lane & 31; h = lane >> 5;
        if (q > 0) {              // x_new of the pass before is in LDS already; this wave's first unit again
            u = {nl, wave, 0};
            first_b = any_b; first_off = lane; first_ksc = 1;
            if (mma_wave) {
                u.l = 0;
                while (u.l < nl && chain_first_tile(p.L[u.l], wave) >= chain_tiles(p.L[u.l])) ++u.l;
                if (u.l < nl) u.tile = chain_first_tile(p.L[u.l], wave);
                if (u.l < nl) chain_unit_b(p, u, lane, first_b, first_off, first_ksc);
            }
        } else if (p.row0.net_row) {
            const gv_chain_layer& Ll = p.L[nl - 1];
            chain_stage_row0(chain_lds, ldk, reinterpret_cast<uint16_t*>(bias_lds + p.ib_lds_off), p.row0, Ll.iaf_z, Ll.iaf_ld, Ll.n >> 1, Ll.t_tile,
                             m0, p.m);
        } else {
            chain_stage(chain_lds, ldk, p.x, p.ldx, m0, p.m, p.L[0].k, (p.L[0].k + 15) & ~15);
        }
        chain_issue(qa, first_b, first_off, first_ksc);
    } else if constexpr (IB) {           // the IAF update's backward makes layer 0's input here (its block buffer sits behind the bit tiles)
        {
            // per pass again, from an opaque copy: what derives from the thread index is otherwise computed once in front of the
            // pass loop and held across it, on top of what the unit loop holds
            asm volatile("" : "+v"(lane));
            tid = lane | (wave << 6); r = lane & 31; h = lane >> 5;
            const ChainArgs::Pass& pq = p.pass[q];
            chain_stage_iafb(chain_lds, ld0, reinterpret_cast<uint16_t*>(bias_lds + p.ib_lds_off), p.ib, m0, p.m, pq.ex, pq.gx, pq.gld, pq.cc,
                             pq.gnt, pq.flags);
        }
        if (q > 0) {              // this wave's first unit again
            u = {nl, wave, 0};
            first_b = any_b; first_off = lane; first_ksc = 1;
            if (mma_wave) {
                u.l = 0;
                while (u.l < nl && chain_first_tile(p.L[u.l], wave) >= chain_tiles(p.L[u.l])) ++u.l;
                if (u.l < nl) u.tile = chain_first_tile(p.L[u.l], wave);
                if (u.l < nl) chain_unit_b(p, u, lane, first_b, first_off, first_ksc);
            }
        }
        chain_issue(qa, first_b, first_off, first_ksc);
    } else if (p.L[0].x_dup_half) {      // x holds one half of the columns, the other half repeats it ([g_mu | g_alpha] with g_alpha == g_mu)
        const int half = p.L[0].k >> 1;
        chain_stage(chain_lds, ldk, p.x, p.ldx, m0, p.m, half, half);
        chain_stage(chain_lds + half, ldk, p.x, p.ldx, m0, p.m, half, half);
    } else {
        chain_stage(chain_lds, ldk, p.x, p.ldx, m0, p.m, p.L[0].k, (p.L[0].k + 15) & ~15);
    }
    if (FULL && (p.L[0].mask || p.L[0].mask_t)) chain_stage_mask(mbuf, ldk, p.L[0], m0, p.m);
    __syncthreads();

    f32x16_t acc[2];
    int layer = 0, bias_off = 0, bits_off = 0;
    if (IB || FW) { // defined here: nothing of the previous pass's accumulators is carried around the pass loop and through the stage
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
    }

    // cross layer boundaries until this wave stands in layer `target` (nl: past the last layer).  The boundary orders LDS traffic
    // only (chain_barrier): __syncthreads() also drains the vector-memory counter -- every transposed 2-B store, sign-bit word and
    // fp32 output of the epilogue acknowledged, the next unit's 13 weight fragments landed -- a microsecond per layer and tile
#define CHAIN_CROSS(target)                                                                                              \
    while (layer < (target)) {                                                                                          \
        if (IB && wide0 && layer == 0 && !l0_unit) chain_barrier();      /* (the barrier in front of layer 0's epilogues) */ \
        chain_barrier();                                                                                                \
        if (FULL && layer + 1 < nl && (p.L[layer + 1].mask || p.L[layer + 1].mask_t)) {                                 \
            chain_stage_mask(mbuf, ldk, p.L[layer + 1], m0, p.m);                                                       \
            __syncthreads();                                                                                            \
        }                                                                                                               \
        if (FW) {        /* hidden layers leave through their epilogues alone; x_new's row-major copy where the pass wants one */ \
            if (!mma_wave && layer + 1 == nl && p.pass[q].ob)                                                            \
                chain_store_rows(p.pass[q].ob, p.L[layer].ldb, p.L[layer].n >> 4, chain_lds + ((layer + 1 + par) & 1) * CH_BM * ldk, ldk, \
                                 m0, p.m, tid - CH_MMA_THREADS);                                                        \
        } else                                                                                                          \
        if (!mma_wave && (layer + 1 < nl || p.L[layer].iaf_z))                                                          \
            chain_store<FULL>(p.L[layer], chain_lds + ((layer + 1) & 1) * CH_BM * ldk, ldk, m0, p.m, tid - CH_MMA_THREADS, \
                              !FULL && GV_CHAIN_FAST_EPILOGUE && layer + 1 < nl && m0 + CH_BM <= p.m && p.L[layer].out_bf16_t && \
                              p.L[layer].t_tile > 0 && !p.L[layer].out_f32 && !p.L[layer].add_src && !p.L[layer].iaf_z);  \
        bias_off += p.L[layer].n;                                                                                       \
        bits_off += p.L[layer].mask_bits ? CH_BM * ((p.L[layer].n + 31) >> 5) : 0;                                      \
        ++layer;                                                                                                        \
        if (IB || FW) { /* a unit behind a boundary starts at chunk 0: said here, the accumulators are dead across the boundary */ \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;                                 \
        }                                                                                                               \
    }

    // one unit: fence (its fragments were requested during the previous unit's epilogue), MFMAs, reload for the next unit, epilogue
#define CHAIN_UNIT1(Q)                                                                                                   \
    {                                                                                                                   \
        const gv_chain_layer& Ly = p.L[u.l];                                                                            \
        const int ks = (Ly.k + 15) >> 4, nch = (ks + CH_KS - 1) / CH_KS;                                                \
        const ChainUnit nu = chain_next(p, nl, u, wave);                                                                \
        const uint4* nb0 = any_b;                                                                                       \
        unsigned noff = lane;                                                                                           \
        int nksc = 1;                                                                                                   \
        if (nu.l < nl) chain_unit_b(p, nu, lane, nb0, noff, nksc);                                                      \
        stamp();                                                                                                        \
        chain_landed(Q, noff);                                                                                          \
        stamp();                                                                                                        \
        if (u.ch == 0) {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;                                 \
        }                                                                                                               \
        const uint16_t* A = chain_lds + ((u.l + (FW ? par : 0)) & 1) * CH_BM * ldk;                                     \
        uint16_t* const An = chain_lds + ((u.l + 1 + (FW ? par : 0)) & 1) * CH_BM * ldk;                                \
        int ra = r, ha = h;        /* opaque: a hoisted fragment address is one more register held across the whole loop */ \
        asm volatile("" : "+v"(ra), "+v"(ha));                                                                          \
        const int lda = (IB && u.l == 0) ? ld0 : ldk;                                                                   \
        chain_mma(acc, Q, A + ra * lda + 8 * ha + u.ch * CH_KS * 16, 32 * lda, min(CH_KS, ks - u.ch * CH_KS));          \
        stamp();                                                                                                        \
        const bool iaf_unit = !FW && Ly.iaf_z && u.ch + 1 == nch;                                                       \
        if (FW && u.ch + 1 == nch && u.l + 1 == nl) {      /* the IAF update with this pass's operands */               \
            const ChainArgs::Pass& pq = p.pass[q];                                                                      \
            ChainIafPins P;                                                                                             \
            P.ld = chain_pin(Ly.iaf_ld); P.d = chain_pin(Ly.n) >> 1; P.has_bias = Ly.bias ? 1 : 0; P.identity = 0;      \
            P.rev = (chain_pin(pq.flags) >> 1) & 1;                                                                     \
            const size_t row0 = (size_t)m0 * P.ld;                                                                      \
            P.z = chain_pin_ptr(Ly.iaf_z) + row0; P.x_old = chain_pin_ptr(pq.gx) + row0;                                \
            P.x_new = pq.of ? chain_pin_ptr(pq.of) + row0 : nullptr;                                                    \
            P.ex = pq.ex ? chain_pin_ptr(const_cast<float*>(pq.ex)) + row0 : nullptr;                                   \
            P.alpha = pq.add ? chain_pin_ptr(const_cast<float*>(pq.add)) + row0 : nullptr;                              \
            P.keep = pq.keep ? chain_pin_ptr(pq.keep) : nullptr;                                                        \
            P.tbase = pq.gnt ? chain_pin_ptr(pq.gnt) + (size_t)blockIdx.x * chain_pin(Ly.t_tile) : nullptr;            \
            chain_epilogue_iaf_fast<true>(acc, P, u.tile, An, ldk, (pq.flags & 1) ? (P.d + 15) & ~15 : 0, bias_lds + bias_off, \
                                          reinterpret_cast<const int*>(bias_lds + bias_total), r, h, last_row);         \
        }                                                                                                               \
        if (iaf_unit) {                                                                                                 \
            const bool fast_iaf = GV_CHAIN_FAST_EPILOGUE && m0 + CH_BM <= p.m && !Ly.out_f32 && !(Ly.iaf_reserved & ~1) &&  \
                                  (!Ly.out_bf16_t || Ly.t_tile > 0);                                                    \
            if (fast_iaf) {                                                                                             \
                ChainIafPins P;                                                                                         \
                P.ld = chain_pin(Ly.iaf_ld); P.d = chain_pin(Ly.n) >> 1; P.has_bias = Ly.bias ? 1 : 0; P.identity = chain_pin(Ly.iaf_reserved) & 1; \
                P.rev = 0;                                                                                              \
                const size_t row0 = (size_t)m0 * P.ld;                                                                  \
                P.z = chain_pin_ptr(Ly.iaf_z) + row0; P.x_old = chain_pin_ptr(Ly.iaf_x_old) + row0;                     \
                P.x_new = Ly.iaf_x_new ? chain_pin_ptr(Ly.iaf_x_new) + row0 : nullptr;                                  \
                P.ex = Ly.iaf_ex ? chain_pin_ptr(Ly.iaf_ex) + row0 : nullptr;                                           \
                P.alpha = Ly.iaf_alpha ? chain_pin_ptr(Ly.iaf_alpha) + row0 : nullptr;                                  \
                P.keep = Ly.iaf_keep_colcount ? chain_pin_ptr(Ly.iaf_keep_colcount) : nullptr;                          \
                P.tbase = Ly.out_bf16_t ? chain_pin_ptr(Ly.out_bf16_t) + (size_t)blockIdx.x * chain_pin(Ly.t_tile) : nullptr; \
                chain_epilogue_iaf_fast(acc, P, u.tile, chain_lds + ((u.l + 1) & 1) * CH_BM * ldk, ldk,                 \
                                        Ly.out_bf16 ? (P.d + 15) & ~15 : 0, bias_lds + bias_off,                        \
                                        reinterpret_cast<const int*>(bias_lds + bias_total), r, h);                     \
            } else                                                                                                      \
            chain_epilogue_iaf(acc, Ly, u.tile, m0, p.m, chain_lds + ((u.l + 1) & 1) * CH_BM * ldk, ldk,                \
                               Ly.out_bf16 ? ((Ly.n >> 1) + 15) & ~15 : 0, bias_lds + bias_off,                         \
                               reinterpret_cast<const int*>(bias_lds + bias_total), r, h);                              \
        }                                                                                                               \
        if (!(GV_CHAIN_ABL & 2048)) chain_issue(Q, nb0, noff, nksc);                                                    \
        if (IB && wide0 && u.l == 0 && u.ch + 1 == nch) chain_barrier();                                                \
        if (FW) {                                                                                                       \
            if (u.ch + 1 == nch && u.l + 1 < nl)                                                                        \
                chain_epilogue_fast<true>(acc, chain_pin(Ly.n), u.tile, chain_pin(Ly.relu), An, ldk, (chain_pin(Ly.n) + 15) & ~15, \
                                          Ly.bias ? bias_lds + bias_off : nullptr, nullptr, 0,                          \
                                          chain_pin_ptr(p.pass[q].out_t[u.l]) + (size_t)blockIdx.x * chain_pin(Ly.t_tile), \
                                          chain_pin_ptr(const_cast<uint32_t*>(p.pass[q].bits[u.l])) + (size_t)m0 * chain_pin(Ly.ldbits), \
                                          chain_pin(Ly.ldbits), r, h, last_row);                                        \
        } else                                                                                                          \
        if (IB && u.ch + 1 == nch) {                                                                                    \
            /* backward chain behind the IAF-backward stage: hidden layers on the fast epilogue also in the last, partial tile (its  \
               rows past m hold zeros from the stage on: no bias, the mask bits of such rows are 0 -- zeros go into the tile's pad    \
               rows, which is what they must hold); the last layer's fp32 output with this pass's pointers */           \
            if (u.l + 1 < nl)                                                                                           \
                chain_epilogue_fast(acc, chain_pin(Ly.n), u.tile, chain_pin(Ly.relu), chain_lds + ((u.l + 1) & 1) * CH_BM * ldk, ldk, \
                                    (chain_pin(Ly.n) + 15) & ~15, Ly.bias ? bias_lds + bias_off : nullptr,              \
                                    Ly.mask_bits ? reinterpret_cast<const uint32_t*>(bias_lds + bits_base) + bits_off : nullptr, \
                                    (chain_pin(Ly.n) + 31) >> 5, chain_pin_ptr(p.pass[q].out_t[u.l]) + (size_t)blockIdx.x * chain_pin(Ly.t_tile), \
                                    nullptr, 0, r, h);                                                                  \
            else                                                                                                        \
                chain_epilogue_last(acc, Ly.n, u.tile, m0, p.m, Ly.bias ? bias_lds + bias_off : nullptr, p.pass[q].of, Ly.ldc, \
                                    p.pass[q].add, reinterpret_cast<const int*>(bias_lds + bias_total), r, h, p.pass[q].flags & 2); \
        } else                                                                                                          \
        if (u.ch + 1 == nch && !iaf_unit) {                                                                             \
            /* the common hidden layer of the fused chains (FULL == false): every row exists, transposed copy in 64-row tiles */ \
            const bool fast = !FULL && GV_CHAIN_FAST_EPILOGUE && u.l + 1 < nl && m0 + CH_BM <= p.m && Ly.out_bf16_t && Ly.t_tile > 0 && \
                              !Ly.out_f32 && !Ly.add_src;                                                               \
            if (fast)                                                                                                   \
                chain_epilogue_fast(acc, chain_pin(Ly.n), u.tile, chain_pin(Ly.relu), chain_lds + ((u.l + 1) & 1) * CH_BM * ldk, ldk, \
                                    (chain_pin(Ly.n) + 15) & ~15, Ly.bias ? bias_lds + bias_off : nullptr,              \
                                    Ly.mask_bits ? reinterpret_cast<const uint32_t*>(bias_lds + bits_base) + bits_off : nullptr, \
                                    (chain_pin(Ly.n) + 31) >> 5, chain_pin_ptr(Ly.out_bf16_t) + (size_t)blockIdx.x * chain_pin(Ly.t_tile), \
                                    Ly.out_bits ? chain_pin_ptr(Ly.out_bits) + (size_t)m0 * chain_pin(Ly.ldbits) : nullptr, chain_pin(Ly.ldbits), r, h); \
            else                                                                                                        \
            chain_epilogue<FULL>(acc, Ly, u.tile, m0, p.m, chain_lds + ((u.l + 1) & 1) * CH_BM * ldk, ldk,              \
                           u.l + 1 < nl ? (Ly.n + 15) & ~15 : 0, mbuf, bias_lds + bias_off,                             \
                           reinterpret_cast<const int*>(bias_lds + bias_total),                                         \
                           reinterpret_cast<const uint32_t*>(bias_lds + bits_base) + bits_off, r, h);                   \
        }                                                                                                               \
        stamp();                                                                                                        \
        u = nu;                                                                                                         \
    }

    stamp();
    for (;;) {
        CHAIN_CROSS(u.l)
        if (u.l >= nl) break;
        CHAIN_UNIT1(qa)
    }
    stamp();
    if constexpr (FW) {
        if (++q < p.n_passes) {
            __syncthreads();      // (x_new of this pass is the next one's x_old: with the vector-memory counter drained)
            const gv_chain_layer& Ll = p.L[nl - 1];
            if (tid < (Ll.n >> 1)) reinterpret_cast<int*>(bias_lds + bias_total)[tid] = p.pass[q].cc[tid];
            par ^= nl & 1;
            layer = 0; bias_off = 0; bits_off = 0;
            goto next_pass;
        }
    }
    if constexpr (IB) {
        if (++q < p.n_passes) {
            // every wave is behind the last layer's barrier: the next pass's column counts and mask bits replace this pass's in LDS (the
            // stage's barriers order them in front of their first readers)
            const ChainArgs::Pass& pq = p.pass[q];
            const gv_chain_layer& Ll = p.L[nl - 1];
            __syncthreads();      // (with the vector-memory counter drained: this pass's fp32 output, every wave's part of it, is the next stage's gx)
            const int tx = tid;
            if (tx < Ll.n) reinterpret_cast<int*>(bias_lds + bias_total)[tx] = pq.cc[tx];
            uint32_t* bl = reinterpret_cast<uint32_t*>(bias_lds + bits_base);
            for (int l = 0; l < nl; ++l)
                if (p.L[l].mask_bits) {
                    const int nt = (p.L[l].n + 31) >> 5;
                    const uint32_t* src = pq.bits[l];
                    for (int i = tx; i < CH_BM * nt; i += CH_THREADS) {
                        const int row = i / nt, t = i - row * nt;
                        bl[i] = m0 + row < p.m ? src[(size_t)(m0 + row) * p.L[l].ldbits + t] : 0u;
                    }
                    bl += CH_BM * nt;
                }
            layer = 0; bias_off = 0; bits_off = 0;
            goto next_pass;
        }
    }
#undef CHAIN_CROSS
#undef CHAIN_UNIT1
}

template <bool FULL, bool IB = false>
__global__ __launch_bounds__(CH_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_made_chain(const ChainArgs p) {
    chain_body<FULL, IB, false>(p);
}
__global__ __launch_bounds__(CH_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_made_chain_fwd(const ChainArgs p) {
    chain_body<false, false, true>(p);
}

// packed[(t * ks + s) * 64 + lane][e] = bf16(B[32 t + (lane & 31)][16 s + 8 (lane >> 5) + e]), zero outside B;
// forward: B = W [n][k]; backward: B = W^T [k][n] (its tiles run over k, its steps over n)
struct PackOne { const float* w; int ld, n, k; uint4* fwd; uint4* bwd; int iaf; };      // iaf: fwd tiles = 16 mu + 16 alpha rows
struct PackMulti { PackOne e[GV_CHAIN_MAX_LAYERS]; };

__device__ __forceinline__ void pack_b_frag(const float* __restrict__ w, int ld, int n, int k, uint4* fwd, uint4* bwd, int iaf);

__global__ __launch_bounds__(256) void k_pack_b_frag(const float* __restrict__ w, int ld, int n, int k, uint4* fwd, uint4* bwd, int iaf) {
    pack_b_frag(w, ld, n, k, fwd, bwd, iaf);
}
// every layer of a MADE in one launch (blockIdx.y = layer)
__global__ __launch_bounds__(256) void k_pack_b_frag_multi(const PackMulti p) {
    const PackOne& e = p.e[blockIdx.y];
    pack_b_frag(e.w, e.ld, e.n, e.k, e.fwd, e.bwd, e.iaf);
}

__device__ __forceinline__ void pack_b_frag(const float* __restrict__ w, int ld, int n, int k, uint4* fwd, uint4* bwd, int iaf) {
    const int half = n >> 1;
    const int ks_f = (k + 15) >> 4, nt_f = iaf ? (half + 15) >> 4 : (n + 31) >> 5, tot_f = nt_f * ks_f * 64;
    const int ks_b = (n + 15) >> 4, nt_b = (k + 31) >> 5, tot_b = nt_b * ks_b * 64;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < tot_f + tot_b; idx += gridDim.x * 256) {
        const bool is_b = idx >= tot_f;
        const int j = is_b ? idx - tot_f : idx, ks = is_b ? ks_b : ks_f;
        const int lane = j & 63, ts = j >> 6, s = ts % ks, t = ts / ks;
        int row = t * 32 + (lane & 31);
        const int c0 = s * 16 + 8 * (lane >> 5);
        bool row_ok = true;
        if (iaf && !is_b) {      // B-row j of tile t: mu column 16 t + j (j < 16), alpha column 16 t + j - 16 = W row half + that
            const int j = lane & 31, c = t * 16 + (j & 15);
            row = j < 16 ? c : half + c;
            row_ok = c < half;
        }
        uint16_t b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float v = 0.f;
            if (!is_b) { if (row_ok && row < n && c < k) v = w[(size_t)row * ld + c]; }
            else { if (row < k && c < n) v = w[(size_t)c * ld + row]; }
            b[e] = bf_bits(v);
        }
        const uint4 o = make_uint4(b[0] | ((uint32_t)b[1] << 16), b[2] | ((uint32_t)b[3] << 16), b[4] | ((uint32_t)b[5] << 16),
                                   b[6] | ((uint32_t)b[7] << 16));
        if (is_b) { if (bwd) bwd[j] = o; }
        else if (fwd) fwd[j] = o;
    }
}

}  // namespace gv

using namespace gv;

/* bf16 elements of one packed copy of a [n][k] operand */
extern "C" int64_t gv_made_pack_weight_elems(int n, int k) {
    if (n <= 0 || k <= 0) return 0;
    return (int64_t)((n + 31) / 32) * ((k + 15) / 16) * 64 * 8;
}

extern "C" int gv_made_pack_weight(const float* w, int ld, int n, int k, uint16_t* packed_fwd, uint16_t* packed_bwd, void* stream) {
    GV_REQUIRE(n > 0 && k > 0 && ld >= k, GV_ERR_SHAPE, "gv_made_pack_weight: n=%d k=%d ld=%d", n, k, ld);
    GV_REQUIRE(w && (packed_fwd || packed_bwd), GV_ERR_NULL, "gv_made_pack_weight: NULL pointer");
    GV_REQUIRE((!packed_fwd || aligned16(packed_fwd)) && (!packed_bwd || aligned16(packed_bwd)), GV_ERR_ALIGN,
               "gv_made_pack_weight: packed buffers must be 16-B aligned");
    const int64_t total = (gv_made_pack_weight_elems(n, k) + gv_made_pack_weight_elems(k, n)) / 8;
    hipLaunchKernelGGL(k_pack_b_frag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ld, n, k,
                       (uint4*)packed_fwd, (uint4*)packed_bwd, 0);
    return launch_status("gv_made_pack_weight");
}

extern "C" int gv_made_pack_weight_iaf(const float* w, int ld, int n, int k, uint16_t* packed_fwd, void* stream) {
    GV_REQUIRE(n > 0 && k > 0 && ld >= k && n % 16 == 0, GV_ERR_SHAPE, "gv_made_pack_weight_iaf: n=%d (= 2 d, d %% 8 == 0) k=%d ld=%d", n, k, ld);
    GV_REQUIRE(w && packed_fwd, GV_ERR_NULL, "gv_made_pack_weight_iaf: NULL pointer");
    GV_REQUIRE(aligned16(packed_fwd), GV_ERR_ALIGN, "gv_made_pack_weight_iaf: packed buffer must be 16-B aligned");
    const int64_t total = gv_made_pack_weight_elems(n, k) / 8;
    hipLaunchKernelGGL(k_pack_b_frag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ld, n, k,
                       (uint4*)packed_fwd, (uint4*)nullptr, 1);
    return launch_status("gv_made_pack_weight_iaf");
}

static int pack_weight_multi(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                             uint16_t* const* packed_fwd, uint16_t* const* packed_bwd, int iaf_last, void* stream);

extern "C" int gv_made_pack_weight_multi(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                                         uint16_t* const* packed_fwd, uint16_t* const* packed_bwd, void* stream) {
    return pack_weight_multi(count, w, ld, n, k, packed_fwd, packed_bwd, 0, stream);
}

extern "C" int gv_made_pack_weight_multi_iaf(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                                             uint16_t* const* packed_fwd, uint16_t* const* packed_bwd, void* stream) {
    return pack_weight_multi(count, w, ld, n, k, packed_fwd, packed_bwd, 1, stream);
}

static int pack_weight_multi(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                             uint16_t* const* packed_fwd, uint16_t* const* packed_bwd, int iaf_last, void* stream) {
    GV_REQUIRE(count >= 1 && count <= GV_CHAIN_MAX_LAYERS, GV_ERR_SHAPE, "gv_made_pack_weight_multi: count=%d", count);
    GV_REQUIRE(w && ld && n && k && packed_fwd && packed_bwd, GV_ERR_NULL, "gv_made_pack_weight_multi: NULL table");
    PackMulti p;
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        GV_REQUIRE(n[i] > 0 && k[i] > 0 && ld[i] >= k[i] && w[i] && (packed_fwd[i] || packed_bwd[i]), GV_ERR_SHAPE,
                   "gv_made_pack_weight_multi: entry %d: n=%d k=%d ld=%d", i, n[i], k[i], ld[i]);
        GV_REQUIRE((!packed_fwd[i] || aligned16(packed_fwd[i])) && (!packed_bwd[i] || aligned16(packed_bwd[i])), GV_ERR_ALIGN,
                   "gv_made_pack_weight_multi: packed buffers must be 16-B aligned");
        p.e[i].w = w[i]; p.e[i].ld = ld[i]; p.e[i].n = n[i]; p.e[i].k = k[i];
        p.e[i].fwd = (uint4*)packed_fwd[i]; p.e[i].bwd = (uint4*)packed_bwd[i];
        p.e[i].iaf = (iaf_last && i + 1 == count) ? 1 : 0;
        GV_REQUIRE(!p.e[i].iaf || n[i] % 16 == 0, GV_ERR_SHAPE, "gv_made_pack_weight_multi_iaf: the last layer is [mu | alpha], n = 2 d with d %% 8 == 0 (n=%d)", n[i]);
        most = max(most, (gv_made_pack_weight_elems(n[i], k[i]) + gv_made_pack_weight_elems(k[i], n[i])) / 8);
    }
    hipLaunchKernelGGL(k_pack_b_frag_multi, dim3((unsigned)((most + 255) / 256), count), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("gv_made_pack_weight_multi");
}

/* bytes of LDS a chain needs (0: the chain does not fit this kernel) */
static int chain_ldk(int n_layers, const gv_chain_layer* layers, bool* has_mask) {
    int kmax = 0;
    *has_mask = false;
    for (int i = 0; i < n_layers; ++i) {
        kmax = max(kmax, (layers[i].k + 15) & ~15);
        if (layers[i].mask || layers[i].mask_t) { *has_mask = true; kmax = max(kmax, layers[i].n); }
    }
    int ldk = (kmax + 15) / 16 * 16 + 8;          // 16 j + 8 elements: 16-B fragment reads of 8 rows hit 32 distinct banks
    return ldk;
}

static int32_t* g_chain_stamps = nullptr;
/* probes only: 8 x 64 int32 that workgroup 0's waves fill with s_memtime stamps (start; per unit: before / after the fragment
 * fence, after the MFMAs, after the epilogue; end); NULL switches it off */
extern "C" int gv_made_chain_debug_stamps(int32_t* buffer) {
    g_chain_stamps = buffer;
    return GV_OK;
}

static int made_chain_launch(const uint16_t* x, int ldx, int m, int n_layers, const gv_chain_layer* layers, const gv_chain_iafb* stage,
                             void* stream);

extern "C" int gv_made_chain(const uint16_t* x, int ldx, int m, int n_layers, const gv_chain_layer* layers, void* stream) {
    return made_chain_launch(x, ldx, m, n_layers, layers, nullptr, stream);
}

extern "C" int gv_made_chain_iafb(const gv_chain_iafb* stage, int m, int n_layers, const gv_chain_layer* layers, void* stream) {
    GV_REQUIRE(stage, GV_ERR_NULL, "gv_made_chain_iafb: NULL stage");
    return made_chain_launch(nullptr, 0, m, n_layers, layers, stage, stream);
}

static int made_chain_launch(const uint16_t* x, int ldx, int m, int n_layers, const gv_chain_layer* layers, const gv_chain_iafb* stage,
                             void* stream) {
    GV_REQUIRE(m >= 0 && n_layers >= 1 && n_layers <= GV_CHAIN_MAX_LAYERS, GV_ERR_SHAPE, "gv_made_chain: m=%d n_layers=%d", m, n_layers);
    if (m == 0) return GV_OK;
    GV_REQUIRE((x || stage) && layers, GV_ERR_NULL, "gv_made_chain: NULL pointer");
    if (stage) {
        const gv_chain_iafb& ib = *stage;
        GV_REQUIRE(ib.z && ib.ex && ib.gx && ib.gz && ib.colcount && ib.gnt, GV_ERR_NULL, "gv_made_chain_iafb: NULL operand");
        GV_REQUIRE(ib.d > 0 && ib.d % 4 == 0 && layers[0].k == 2 * ib.d && ib.ld >= ib.d && ib.ld % 4 == 0 && ib.t_tile >= 128 * ib.d &&
                       ib.t_tile % 4 == 0 && (int64_t)ib.t_tile * ((m + CH_BM - 1) / CH_BM) <= INT32_MAX,
                   GV_ERR_SHAPE, "gv_made_chain_iafb: d=%d (d %% 4 == 0, layer 0 takes 2 d = %d columns), ld=%d, t_tile=%d", ib.d,
                   layers[0].k, ib.ld, ib.t_tile);
        GV_REQUIRE(aligned16(ib.z) && aligned16(ib.ex) && aligned16(ib.gx) && aligned16(ib.gz) && aligned16(ib.colcount) &&
                       (reinterpret_cast<uintptr_t>(ib.gnt) & 7u) == 0, GV_ERR_ALIGN, "gv_made_chain_iafb: fp32 rows must be 16-B aligned");
        GV_REQUIRE(!layers[0].x_dup_half, GV_ERR_SHAPE, "gv_made_chain_iafb: the stage fills both halves of layer 0's input itself");
        GV_REQUIRE(ib.n_passes >= 0 && ib.n_passes <= CH_MAX_PASSES, GV_ERR_SHAPE, "gv_made_chain_iafb: n_passes=%d (at most %d)", ib.n_passes,
                   CH_MAX_PASSES);
        {   // what the instance's epilogues assume (every MADE backward chain of the fused path is of this form)
            const gv_chain_layer& last = layers[n_layers - 1];
            bool ok = n_layers >= 2 && last.out_f32 && !last.out_bf16_t && !last.relu && !last.mask_bits && !last.out_bits && !last.iaf_z &&
                      (!last.add_src || last.add_colcount);
            for (int l = 0; l + 1 < n_layers; ++l)
                ok = ok && layers[l].out_bf16_t && layers[l].t_tile > 0 && !layers[l].out_f32 && !layers[l].add_src && !layers[l].out_bits &&
                     !layers[l].bias && !layers[l].out_bf16 && !layers[l].iaf_z;
            GV_REQUIRE(ok, GV_ERR_SHAPE, "gv_made_chain_iafb: hidden layers write tiled transposed copies alone (no bias), the last layer an fp32 output");
        }
        GV_REQUIRE(ib.n_passes <= 1 || layers[n_layers - 1].ldc == ib.ld, GV_ERR_SHAPE,
                   "gv_made_chain_iafb: with n_passes > 1 the last layer's fp32 output is the next pass's gx: ldc=%d must equal ld=%d",
                   layers[n_layers - 1].ldc, ib.ld);
        GV_REQUIRE((int64_t)m * ib.ld * 4 < (1ll << 32), GV_ERR_SHAPE, "gv_made_chain_iafb: m * ld = %lld elements exceed the stage's 32-bit byte offsets",
                   (long long)m * ib.ld);
    } else {
        GV_REQUIRE(ldx % 8 == 0 && aligned16(x) && ldx >= (layers[0].x_dup_half ? layers[0].k / 2 : layers[0].k), GV_ERR_ALIGN,
                   "gv_made_chain: x rows must be 16-B aligned pieces (ldx=%d)", ldx);
        GV_REQUIRE(!layers[0].x_dup_half || layers[0].k % 16 == 0, GV_ERR_SHAPE, "gv_made_chain: x_dup_half needs k %% 16 == 0");
    }
    ChainArgs p;
    for (int i = 0; i < n_layers; ++i) {
        const gv_chain_layer& L = layers[i];
        GV_REQUIRE(L.n > 0 && L.k > 0 && L.n % 8 == 0 && L.k % 8 == 0 && L.n <= CH_THREADS && (i == 0 || L.k == layers[i - 1].n), GV_ERR_SHAPE,
                   "gv_made_chain: layer %d is %d x %d (widths are multiples of 8, k = the previous layer's n)", i, L.n, L.k);
        GV_REQUIRE(L.w_packed && aligned16(L.w_packed), GV_ERR_NULL, "gv_made_chain: layer %d has no packed weight", i);
        GV_REQUIRE(L.out_bf16 || L.out_bf16_t || L.out_f32 || L.iaf_z || i + 1 < n_layers, GV_ERR_NULL, "gv_made_chain: the last layer stores nothing");
        GV_REQUIRE(!(L.out_bf16 && i + 1 == n_layers && !L.iaf_z), GV_ERR_SHAPE, "gv_made_chain: the last layer has no row-major bf16 output");
        GV_REQUIRE((!L.out_bits && !L.mask_bits) || (L.ldbits >= (L.n + 31) / 32 && !(L.mask_bits && (L.mask || L.mask_t)) && !L.iaf_z),
                   GV_ERR_SHAPE, "gv_made_chain: layer %d: mask bits are [m][ldbits >= ceil(n / 32)] words, one mask form per layer", i);
        GV_REQUIRE(!(L.mask && L.mask_t) && (!L.mask_t || (L.ldmask_t >= m && L.ldmask_t % 8 == 0 && aligned16(L.mask_t))), GV_ERR_ALIGN,
                   "gv_made_chain: layer %d: one mask form; the transposed one has 16-B aligned rows of >= m entries", i);
        GV_REQUIRE(!L.add_src || (i + 1 == n_layers && L.out_f32 && L.add_colcount && aligned16(L.add_src) && aligned16(L.add_colcount) &&
                                  L.n % 4 == 0), GV_ERR_NULL, "gv_made_chain: add_src belongs to the last layer's fp32 output, with its column counts");
        if (L.iaf_z) {
            GV_REQUIRE(i + 1 == n_layers && L.n % 16 == 0 && !L.relu && !L.mask && !L.accumulate, GV_ERR_SHAPE,
                       "gv_made_chain: the IAF update belongs to the last layer, n = 2 d with d %% 8 == 0, no ReLU / mask / accumulate");
            GV_REQUIRE(L.iaf_x_old && L.iaf_colcount && (L.iaf_x_new || L.iaf_ex || L.out_bf16 || L.out_bf16_t), GV_ERR_NULL,
                       "gv_made_chain: IAF layer needs x_old, colcount and an output");
            GV_REQUIRE(L.iaf_ld >= L.n / 2 && L.iaf_ld % 4 == 0 && aligned16(L.iaf_z) && aligned16(L.iaf_x_old) && aligned16(L.iaf_colcount) &&
                       (!L.iaf_x_new || aligned16(L.iaf_x_new)) && (!L.iaf_ex || aligned16(L.iaf_ex)) && (!L.iaf_alpha || aligned16(L.iaf_alpha)) &&
                       (!L.iaf_keep_colcount || aligned16(L.iaf_keep_colcount)),
                       GV_ERR_ALIGN, "gv_made_chain: IAF operands are [m][iaf_ld] fp32 with 16-B aligned rows");
        }
        GV_REQUIRE((!L.mask || (L.ldmask >= L.n && L.ldmask % 8 == 0 && aligned16(L.mask))) &&
                   (!L.out_bf16 || (L.ldb >= (L.iaf_z ? L.n / 2 : L.n) && L.ldb % 8 == 0 && aligned16(L.out_bf16))) &&
                   (!L.out_bf16_t || (L.t_tile == 0 ? L.ldt >= m : (L.ldt == CH_BM && L.t_tile >= CH_BM * (L.iaf_z ? L.n / 2 : L.n) &&
                                                                      (int64_t)L.t_tile * ((m + CH_BM - 1) / CH_BM) <= INT32_MAX))) &&
                   (!L.out_f32 || (L.ldc >= L.n && L.ldc % 4 == 0 && aligned16(L.out_f32))), GV_ERR_ALIGN, "gv_made_chain: layer %d: leading dimension / alignment", i);
        p.L[i] = L;
    }
    bool has_mask;
    int ldk = chain_ldk(n_layers, layers, &has_mask), ld0 = ldk;
    if (stage && n_layers >= 2) {
        // the stage's 2 d-wide tile over both activation buffers when the layers behind it are narrower (GV_CHAIN_WIDE0=0: off)
        static const bool wide_on = !(getenv("GV_CHAIN_WIDE0") && getenv("GV_CHAIN_WIDE0")[0] == '0');
        bool hm;
        const int ldn = chain_ldk(n_layers - 1, layers + 1, &hm);
        if (wide_on && ldn < ldk && CH_BM * ldk <= 2 * CH_BM * ldn && (layers[0].n + 31) / 32 <= CH_MMA_WAVES) ldk = ldn;
    }
    size_t bias_floats = 0;
    for (int i = 0; i < n_layers; ++i) bias_floats += (size_t)layers[i].n;
    if (layers[n_layers - 1].iaf_z) bias_floats += (size_t)layers[n_layers - 1].n / 2;      // its column counts
    else if (layers[n_layers - 1].add_src) bias_floats += (size_t)layers[n_layers - 1].n;
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].mask_bits) bias_floats += (size_t)CH_BM * ((layers[i].n + 31) / 32);      // its bit tile
    bias_floats = (bias_floats + 1) & ~(size_t)1;      // (the stage's block is read in 8-B pieces)
    const size_t lds = (size_t)(has_mask ? 3 : 2) * CH_BM * ldk * sizeof(uint16_t) + bias_floats * sizeof(float) +
                       (stage ? (size_t)2 * IB_COLS * IB_LD * sizeof(uint16_t) : 0);
    GV_REQUIRE(lds <= 160 * 1024, GV_ERR_SHAPE, "gv_made_chain: layers this wide need %zu B of LDS (160 KB per CU)", lds);
    p.x = x; p.ldx = ldx; p.m = m; p.n_layers = n_layers; p.ldk = ldk; p.has_mask = has_mask ? 1 : 0; p.stamps = g_chain_stamps;
    p.ib_lds_off = (int)bias_floats;
    p.ld0 = ld0;
    p.n_passes = 1;
    for (int q = 0; q < CH_MAX_PASSES; ++q) p.pass[q] = ChainArgs::Pass{};
    if (stage) {
        p.ib = *stage;
        const gv_chain_layer& last = layers[n_layers - 1];
        p.n_passes = stage->n_passes > 1 ? stage->n_passes : 1;
        for (int q = 0; q < p.n_passes; ++q) {
            ChainArgs::Pass& pq = p.pass[q];
            float* const of_prev = q ? last.out_f32 + (int64_t)(q - 1) * stage->of_step : nullptr;
            pq.ex = stage->ex + (int64_t)q * stage->rows_step * stage->ld;
            pq.gx = q ? of_prev : stage->gx;                 // pass q's dL/dx_new is pass q - 1's dL/dx_old
            pq.gld = q ? nullptr : stage->gld;
            pq.cc = stage->colcount + (int64_t)q * stage->cc_step;
            pq.gnt = stage->gnt + (int64_t)q * stage->tiles_step * stage->t_tile;
            pq.flags = q ? (stage->flags & ~3) : stage->flags;       // (bit 0: g_z starts here, bit 1: reversed gx -- the first pass alone)
            pq.of = last.out_f32 ? last.out_f32 + (int64_t)q * stage->of_step : nullptr;
            pq.add = q ? (last.add_src ? of_prev : nullptr) : last.add_src;
            for (int l = 0; l < n_layers; ++l) {
                pq.bits[l] = layers[l].mask_bits ? layers[l].mask_bits + (int64_t)q * stage->rows_step * layers[l].ldbits : nullptr;
                pq.out_t[l] = layers[l].out_bf16_t ? layers[l].out_bf16_t + (int64_t)q * stage->tiles_step * layers[l].t_tile : nullptr;
            }
        }
    } else {
        p.ib = gv_chain_iafb{};
    }
    bool full = false, bits = false;
    for (int i = 0; i < n_layers; ++i) {
        full = full || layers[i].mask || layers[i].mask_t || layers[i].accumulate;
        bits = bits || layers[i].mask_bits || layers[i].out_bits;
    }
    GV_REQUIRE(!(full && bits), GV_ERR_SHAPE, "gv_made_chain: mask bits do not combine with tile masks or an accumulating output in one chain");
    GV_REQUIRE(!(full && stage), GV_ERR_SHAPE, "gv_made_chain_iafb: not with tile masks or an accumulating output");
    static unsigned long long lds_full = 0, lds_lean = 0;
    if (!raise_dynamic_lds((const void*)k_made_chain<true>, 160 * 1024, lds_full, "gv_made_chain") ||
        !raise_dynamic_lds((const void*)k_made_chain<false>, 160 * 1024, lds_lean, "gv_made_chain"))
        return GV_ERR_SHAPE;
    const dim3 grid((unsigned)((m + CH_BM - 1) / CH_BM));
    static unsigned long long lds_ib = 0;
    if (stage && !raise_dynamic_lds((const void*)k_made_chain<false, true>, 160 * 1024, lds_ib, "gv_made_chain_iafb")) return GV_ERR_SHAPE;
    if (stage) hipLaunchKernelGGL((k_made_chain<false, true>), grid, dim3(CH_THREADS), lds, (hipStream_t)stream, p);
    else if (full) hipLaunchKernelGGL(k_made_chain<true>, grid, dim3(CH_THREADS), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_made_chain<false>, grid, dim3(CH_THREADS), lds, (hipStream_t)stream, p);
    return launch_status("gv_made_chain");
}

/* ALL passes of a MADE's forward (kgvae/flow_network.py:85-98, the loop over the index sets) in one launch: see include/gcnvae.h */
static int made_chain_fwd_launch(const uint16_t* x, int ldx, const gv_chain_fwd_row0* first, int m, int n_layers, const gv_chain_layer* layers,
                                 int n_passes, const gv_chain_fwd_pass* passes, void* stream);
extern "C" int gv_made_chain_fwd(const uint16_t* x, int ldx, int m, int n_layers, const gv_chain_layer* layers, int n_passes,
                                 const gv_chain_fwd_pass* passes, void* stream) {
    return made_chain_fwd_launch(x, ldx, nullptr, m, n_layers, layers, n_passes, passes, stream);
}
extern "C" int gv_made_chain_fwd_row0(const gv_chain_fwd_row0* first, int m, int n_layers, const gv_chain_layer* layers, int n_passes,
                                      const gv_chain_fwd_pass* passes, void* stream) {
    GV_REQUIRE(first, GV_ERR_NULL, "gv_made_chain_fwd_row0: NULL pointer");
    return made_chain_fwd_launch(nullptr, 0, first, m, n_layers, layers, n_passes, passes, stream);
}
static int made_chain_fwd_launch(const uint16_t* x, int ldx, const gv_chain_fwd_row0* first, int m, int n_layers, const gv_chain_layer* layers,
                                 int n_passes, const gv_chain_fwd_pass* passes, void* stream) {
    GV_REQUIRE(m >= 0 && n_layers >= 2 && n_layers <= GV_CHAIN_MAX_LAYERS && n_passes >= 1 && n_passes <= CH_MAX_PASSES, GV_ERR_SHAPE,
               "gv_made_chain_fwd: m=%d n_layers=%d n_passes=%d (2-%d layers, 1-%d passes)", m, n_layers, n_passes, GV_CHAIN_MAX_LAYERS, CH_MAX_PASSES);
    if (m == 0) return GV_OK;
    GV_REQUIRE((x || first) && layers && passes, GV_ERR_NULL, "gv_made_chain_fwd: NULL pointer");
    if (first) {
        const gv_chain_layer& lastL = layers[n_layers - 1];
        GV_REQUIRE(first->net_row && first->x_f32 && first->x_t && aligned16(first->net_row) && aligned16(first->x_f32) &&
                       (reinterpret_cast<uintptr_t>(first->x_t) & 7u) == 0, GV_ERR_NULL, "gv_made_chain_fwd_row0: net_row, x_f32, x_t (aligned)");
        GV_REQUIRE(layers[0].k == lastL.n / 2 && lastL.n % 8 == 0 && lastL.t_tile >= CH_BM * (lastL.n / 2) && (int64_t)m * lastL.iaf_ld * 4 < (1ll << 32),
                   GV_ERR_SHAPE, "gv_made_chain_fwd_row0: layer 0 reads the d = %d columns of x (k = %d), tiled copy t_tile = %d", lastL.n / 2,
                   layers[0].k, lastL.t_tile);
    } else {
        GV_REQUIRE(ldx % 8 == 0 && aligned16(x) && ldx >= layers[0].k, GV_ERR_ALIGN, "gv_made_chain_fwd: x rows must be 16-B aligned pieces (ldx=%d)", ldx);
    }
    const gv_chain_layer& last = layers[n_layers - 1];
    const int d = last.n / 2, tiles = (m + CH_BM - 1) / CH_BM;
    ChainArgs p;
    for (int i = 0; i < n_layers; ++i) {
        const gv_chain_layer& L = layers[i];
        GV_REQUIRE(L.n > 0 && L.k > 0 && L.n % 8 == 0 && L.k % 8 == 0 && L.n <= CH_THREADS && (i == 0 || L.k == layers[i - 1].n), GV_ERR_SHAPE,
                   "gv_made_chain_fwd: layer %d is %d x %d (widths are multiples of 8, k = the previous layer's n)", i, L.n, L.k);
        GV_REQUIRE(L.w_packed && aligned16(L.w_packed), GV_ERR_NULL, "gv_made_chain_fwd: layer %d has no packed weight", i);
        GV_REQUIRE(!L.mask && !L.mask_t && !L.mask_bits && !L.out_f32 && !L.accumulate && !L.add_src && !L.x_dup_half, GV_ERR_SHAPE,
                   "gv_made_chain_fwd: layer %d: no masks, fp32 outputs or add sources in this launch", i);
        if (i + 1 < n_layers)
            GV_REQUIRE(L.relu && !L.iaf_z && L.t_tile >= CH_BM * L.n && (int64_t)L.t_tile * tiles <= INT32_MAX && L.ldbits >= (L.n + 31) / 32,
                       GV_ERR_SHAPE, "gv_made_chain_fwd: hidden layer %d: ReLU, tiled copies (t_tile=%d >= 64 n), sign words (ldbits=%d)", i,
                       L.t_tile, L.ldbits);
        p.L[i] = L;
    }
    GV_REQUIRE(last.iaf_z && aligned16(last.iaf_z) && last.n % 16 == 0 && !last.relu && last.iaf_ld >= d && last.iaf_ld % 4 == 0 &&
                   (int64_t)CH_BM * last.iaf_ld * 4 < (1ll << 31),
               GV_ERR_SHAPE, "gv_made_chain_fwd: the last layer carries the IAF update: n = 2 d with d %% 8 == 0, z [m][iaf_ld]");
    GV_REQUIRE(last.iaf_reserved == 0, GV_ERR_SHAPE, "gv_made_chain_fwd: no probe switches");
    bool has_mask;
    const int ldk = chain_ldk(n_layers, layers, &has_mask);
    size_t bias_floats = (size_t)d;
    for (int i = 0; i < n_layers; ++i) bias_floats += (size_t)layers[i].n;
    bias_floats = (bias_floats + 1) & ~(size_t)1;      // (the row-0 stage's block is read in 8-B pieces)
    const size_t lds = (size_t)2 * CH_BM * ldk * sizeof(uint16_t) + bias_floats * sizeof(float) + (first ? (size_t)IB_COLS * IB_LD * sizeof(uint16_t) : 0);
    GV_REQUIRE(lds <= 160 * 1024, GV_ERR_SHAPE, "gv_made_chain_fwd: layers this wide need %zu B of LDS (160 KB per CU)", lds);
    p.x = x; p.ldx = ldx; p.m = m; p.n_layers = n_layers; p.ldk = ldk; p.has_mask = 0; p.stamps = g_chain_stamps;
    p.ib = gv_chain_iafb{}; p.ib_lds_off = (int)bias_floats; p.ld0 = ldk; p.n_passes = n_passes;
    p.row0 = first ? *first : gv_chain_fwd_row0{};
    for (int q = 0; q < CH_MAX_PASSES; ++q) p.pass[q] = ChainArgs::Pass{};
    for (int q = 0; q < n_passes; ++q) {
        const gv_chain_fwd_pass& f = passes[q];
        ChainArgs::Pass& pq = p.pass[q];
        GV_REQUIRE(f.x_old && f.colcount && f.ex && aligned16(f.x_old) && aligned16(f.colcount) && aligned16(f.ex) &&
                       (!f.x_new || aligned16(f.x_new)) && (!f.alpha || aligned16(f.alpha)) && (!f.keep_colcount || aligned16(f.keep_colcount)),
                   GV_ERR_NULL, "gv_made_chain_fwd: pass %d: x_old, colcount, ex (16-B aligned fp32 rows)", q);
        GV_REQUIRE((!f.out_bf16 || (last.ldb >= d && last.ldb % 8 == 0 && aligned16(f.out_bf16))) &&
                       (!f.out_bf16_t || (last.t_tile >= CH_BM * d && (int64_t)last.t_tile * tiles <= INT32_MAX)),
                   GV_ERR_ALIGN, "gv_made_chain_fwd: pass %d: x_new's bf16 copies (ldb=%d, t_tile=%d)", q, last.ldb, last.t_tile);
        pq.ex = f.ex; pq.gx = f.x_old; pq.cc = f.colcount; pq.gnt = f.out_bf16_t; pq.of = f.x_new; pq.add = f.alpha;
        pq.keep = f.keep_colcount; pq.ob = f.out_bf16;
        pq.flags = ((q + 1 < n_passes || f.out_bf16) ? 1 : 0) | ((f.flags & 1) ? 2 : 0);          // x_new into the LDS tile; reversed x_new
        for (int l = 0; l + 1 < n_layers; ++l) {
            GV_REQUIRE(f.act_t[l] && f.act_bits[l], GV_ERR_NULL, "gv_made_chain_fwd: pass %d: hidden layer %d has no tiled copy / sign words", q, l);
            pq.out_t[l] = f.act_t[l];
            pq.bits[l] = reinterpret_cast<const uint32_t*>(f.act_bits[l]);
        }
    }
    static unsigned long long lds_fw = 0;
    if (!raise_dynamic_lds((const void*)k_made_chain_fwd, 160 * 1024, lds_fw, "gv_made_chain_fwd")) return GV_ERR_SHAPE;
    hipLaunchKernelGGL(k_made_chain_fwd, dim3((unsigned)tiles), dim3(CH_THREADS), lds, (hipStream_t)stream, p);
    return launch_status("gv_made_chain_fwd");
}

/* 1 when gv_made_chain can run this chain (widths; LDS), 0 otherwise -- callers then launch product by product */
extern "C" int gv_made_chain_fits(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer, int any_mask) {
    if (n_layers < 1 || n_layers > GV_CHAIN_MAX_LAYERS || !n_of_layer || !k_of_layer) return 0;
    int kmax = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (n_of_layer[i] <= 0 || k_of_layer[i] <= 0 || n_of_layer[i] % 8 || k_of_layer[i] % 8 || n_of_layer[i] > 768) return 0;
        if (i > 0 && k_of_layer[i] != n_of_layer[i - 1]) return 0;
        kmax = max(kmax, (k_of_layer[i] + 15) & ~15);
        if (any_mask) kmax = max(kmax, n_of_layer[i]);
    }
    const int ldk = (kmax + 15) / 16 * 16 + 8;
    size_t bias_floats = 0;
    for (int i = 0; i < n_layers; ++i) bias_floats += (size_t)n_of_layer[i];
    bias_floats += (size_t)n_of_layer[n_layers - 1];      // room for the last layer's column counts (IAF update / handed-through gradient)
    return (size_t)(any_mask ? 3 : 2) * CH_BM * ldk * 2 + bias_floats * 4 <= 160 * 1024 ? 1 : 0;
}
