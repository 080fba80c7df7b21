// K4 in bf16: the masked-MLP products of MADE / IAF (kgvae/flow_network.py:7-98) with bf16 STORAGE of weights and
// activations, v_mfma_f32_32x32x16_bf16, fp32 accumulation (BASELINE configs[2]).
//
// gv_gemm_bf16 (k_gemm.hip) only rounds fp32 operands while staging them, so its tile loop stays bound by fp32 global
// loads.  Here the operands ARE bf16 in memory: a 64 x 64 output tile stages 64 x K bf16 rows of both
// operands (K <= 224 per chunk: the whole reduction of a 200-wide layer in ONE shot, one barrier pair per chunk), every
// fragment is one conflict-free ds_read_b128 (464-B padded LDS rows), and the epilogue fuses bias, ReLU, the ReLU mask of
// the backward pass (sign of the stored bf16 activation) and up to three stores: fp32 (optionally accumulating), bf16, and
// a bf16 TRANSPOSED copy -- four consecutive rows of an accumulator register group are four consecutive elements of a
// transposed row, one 8-B store.  With the transposed copies of activations and gradients at hand, every product of the
// masked MLP is the same NT kernel:
//   forward      a_l      = relu(a_{l-1} W_l^T + b)        A = a_{l-1} [M][in],   B = W_l   [out][in]
//   backward-x   g_{l-1}  = (g_l W_l) * [a_{l-1} > 0]      A = g_l     [M][out],  B = W_l^T [in][out]
//   backward-W   dW_l     = g_l^T a_{l-1}                  A = g_l^T   [out][M],  B = a_{l-1}^T [in][M]   (split over M)
// Semantics (pinned by the tests' CPU emulation): operands rounded to bf16 (round to nearest even), products and sums in fp32.
#include <cstdlib>
#include "common.h"

namespace gv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct BfGemm {
    const void* a;            // [M][lda] bf16, or fp32 (a_f32) rounded while staged
    int lda;
    const uint16_t* b;        // [N][ldb] bf16
    int ldb;
    int m, n, k;
    const float* bias;        // [N] or NULL
    int relu;
    const uint16_t* mask;     // [M][ldmask] bf16 or NULL: the result is kept where mask > 0, zero elsewhere
    int ldmask;
    float* c_f32;             // [M][ldc] or NULL
    int ldc;
    int accumulate;           // c_f32 += result
    uint16_t* c_bf;           // [M][ldcb] or NULL
    int ldcb;
    uint16_t* c_bft;          // [N][ldct] transposed, or NULL
    int ldct;
    float* partial;           // split-K: [gridDim.z][M][N] fp32 (then no epilogue, no other output)
    int k_per_split;
    float* rowsum_partial;    // whole-output kernel only: sums of A's rows over the block's K slice (split z at z * partial_stride), or NULL
    size_t partial_stride;    // floats between two splits' partial results
    unsigned a_tile, b_tile;  // whole-output kernel, TILES instance: the operands are given in 64-deep K TILES, element (row, kk) at
                              // [kk / 64][row][kk % 64] with `tile` elements between tiles (lda / ldb unused)
};

__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

template <bool A_F32>
__global__ __launch_bounds__(256) void k_gemm_bf16s(const BfGemm p) {
    constexpr int BM = 64, BN = 64, KC = 224, LDK = 232;        // 464-B LDS rows: conflict-free 16-B fragment reads
    __shared__ __attribute__((aligned(16))) uint16_t As[BM * LDK];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BN * LDK];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * p.k_per_split, k_end = min(p.k, k_begin + p.k_per_split);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1, r = lane & 31, h = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // Staging: a chunk is 64 rows x 28 pieces of 8 bf16 (16 B) per operand = 7 pieces per thread and operand.  All 14 loads of
    // a thread are issued before the first LDS write, so a block pays ONE global round trip per chunk, not fourteen.
    constexpr int PPR = KC / 8, PPT = BM * PPR / 256;
    static_assert(BM * PPR % 256 == 0 && BM == BN, "staging assumes whole pieces per thread");
    for (int k0 = k_begin; k0 < k_end; k0 += KC) {
        const int kc = min(KC, k_end - k0);          // multiple of 8 (host-checked)
        const int kc16 = (kc + 15) & ~15;
        uint4 va[PPT], vb[PPT];
        float4 fa[A_F32 ? PPT : 1][2];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int idx = threadIdx.x + j * 256, row = idx / PPR, kk = (idx - row * PPR) << 3;
            va[j] = make_uint4(0, 0, 0, 0);
            vb[j] = make_uint4(0, 0, 0, 0);
            if constexpr (A_F32) { fa[j][0] = make_float4(0.f, 0.f, 0.f, 0.f); fa[j][1] = fa[j][0]; }
            if (kk < kc) {
                if (m0 + row < p.m) {
                    if constexpr (A_F32) {
                        const float4* src = reinterpret_cast<const float4*>(static_cast<const float*>(p.a) +
                                                                            (size_t)(m0 + row) * p.lda + k0 + kk);
                        fa[j][0] = src[0];
                        fa[j][1] = src[1];
                    } else {
                        va[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.a) + (size_t)(m0 + row) * p.lda + k0 + kk);
                    }
                }
                if (n0 + row < p.n) vb[j] = *reinterpret_cast<const uint4*>(p.b + (size_t)(n0 + row) * p.ldb + k0 + kk);
            }
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int idx = threadIdx.x + j * 256, row = idx / PPR, kk = (idx - row * PPR) << 3;
            if (kk < kc16) {
                if constexpr (A_F32) {
                    const float4 lo = fa[j][0], hi = fa[j][1];
                    va[j].x = f2bf(lo.x) | ((uint32_t)f2bf(lo.y) << 16); va[j].y = f2bf(lo.z) | ((uint32_t)f2bf(lo.w) << 16);
                    va[j].z = f2bf(hi.x) | ((uint32_t)f2bf(hi.y) << 16); va[j].w = f2bf(hi.z) | ((uint32_t)f2bf(hi.w) << 16);
                }
                *reinterpret_cast<uint4*>(&As[row * LDK + kk]) = va[j];
                *reinterpret_cast<uint4*>(&Bs[row * LDK + kk]) = vb[j];
            }
        }
        __syncthreads();
        const uint16_t* ap = &As[(wm * 32 + r) * LDK + 8 * h];
        const uint16_t* bp = &Bs[(wn * 32 + r) * LDK + 8 * h];
        for (int kk = 0; kk < kc16; kk += 16) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap + kk);
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bp + kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        }
        __syncthreads();
    }

    if (p.partial) {
        const int col = n0 + wn * 32 + r;
        if (col >= p.n) return;
        float* dst = p.partial + (size_t)blockIdx.z * p.m * p.n;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (row < p.m) dst[(size_t)row * p.n + col] = acc[i];
        }
        return;
    }
    // Epilogue through an fp32 LDS tile (the operand buffers are free now): the accumulator layout has one column per lane,
    // which would make every global access a 2- or 4-byte one.  Phase 1: acc + bias (ReLU) -> tile.  Phase 2: 4-column
    // pieces, one per thread and step -- ReLU mask (8-B load), accumulate (16-B load), 16-B fp32 / 8-B bf16 stores, masked
    // value back into the tile.  Phase 3: the transposed copy, 4 consecutive rows of one column = one 8-B store.
    constexpr int LDC = 65;
    float* Ct = reinterpret_cast<float*>(As);
    static_assert(BM * LDC * 4 <= BM * LDK * 2, "the epilogue tile must fit the A buffer");
    {
        const int cl = wn * 32 + r;
        const float bv = (p.bias && n0 + cl < p.n) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = acc[i] + bv;
            if (p.relu) v = fmaxf(v, 0.f);
            Ct[(wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * LDC + cl] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = threadIdx.x + j * 256, rr = idx >> 4, cc = (idx & 15) << 2;
        const int row = m0 + rr, col = n0 + cc;
        if (row < p.m && col < p.n) {          // n % 4 == 0 (host-checked): a piece is inside or outside as a whole
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = Ct[rr * LDC + cc + e];
            if (p.mask) {
                const uint2 mv = *reinterpret_cast<const uint2*>(p.mask + (size_t)row * p.ldmask + col);
                if ((int16_t)(mv.x & 0xffff) <= 0) v[0] = 0.f;
                if ((int16_t)(mv.x >> 16) <= 0) v[1] = 0.f;
                if ((int16_t)(mv.y & 0xffff) <= 0) v[2] = 0.f;
                if ((int16_t)(mv.y >> 16) <= 0) v[3] = 0.f;
                if (p.c_bft) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ct[rr * LDC + cc + e] = v[e];
                }
            }
            if (p.c_f32) {
                float4* o = reinterpret_cast<float4*>(p.c_f32 + (size_t)row * p.ldc + col);
                float4 ov = make_float4(v[0], v[1], v[2], v[3]);
                if (p.accumulate) { const float4 old = *o; ov.x += old.x; ov.y += old.y; ov.z += old.z; ov.w += old.w; }
                *o = ov;
            }
            if (p.c_bf)
                *reinterpret_cast<uint2*>(p.c_bf + (size_t)row * p.ldcb + col) =
                    make_uint2(f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
        }
    }
    if (!p.c_bft) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = threadIdx.x + j * 256, cc = idx >> 4, rr = (idx & 15) << 2;
        const int col = n0 + cc, row = m0 + rr;
        if (col < p.n && row < p.m) {
            uint16_t* o = p.c_bft + (size_t)col * p.ldct + row;
            const uint16_t b0 = f2bf(Ct[rr * LDC + cc]), b1 = f2bf(Ct[(rr + 1) * LDC + cc]), b2 = f2bf(Ct[(rr + 2) * LDC + cc]),
                           b3 = f2bf(Ct[(rr + 3) * LDC + cc]);
            if (row + 3 < p.m) {
                *reinterpret_cast<uint2*>(o) = make_uint2(b0 | ((uint32_t)b1 << 16), b2 | ((uint32_t)b3 << 16));
            } else {
                o[0] = b0;
                if (row + 1 < p.m) o[1] = b1;
                if (row + 2 < p.m) o[2] = b2;
            }
        }
    }
}

// Split-K partial products for SMALL outputs over a LONG reduction (the weight gradients dW = g^T a: 200-400 x 200 outputs,
// 65 000 - 184 000 rows): a workgroup computes the WHOLE <= 224 x <= 224 output block of its K slice, so every operand element
// is read exactly once (the 64 x 64 tiling above re-reads both operands four times from L2 and is bound by that: 41 us for
// 52 MB at FB15k-237 size).  8 waves as 4 (M) x 2 (N), a wave owns 2 x 4 accumulator tiles of 32 x 32; 64-deep chunks staged
// through LDS (72-element rows: conflict-free ds_read_b128), the next chunk's global loads in flight during the MFMAs.
constexpr int TK_T = 224, TK_KC = 128, TK_LDK = 136, TK_THREADS = 512;      // 128-deep chunks: a chunk's MFMAs (~1.7 us) cover a memory round trip
constexpr int TK_PPR = TK_KC / 8, TK_PIECES = TK_T * TK_PPR, TK_PPT = (TK_PIECES + TK_THREADS - 1) / TK_THREADS;     // 16-B pieces per operand / thread
constexpr size_t TK_LDS_BYTES = (size_t)2 * TK_T * TK_LDK * sizeof(uint16_t);

// an UNCONDITIONAL load from a clamped (always valid) address; the piece is zeroed where it lies outside the operand only when it
// is written to LDS (tk_keep), a chunk later: a load under a condition, or a select right behind it, makes the compiler wait for
// the data on the spot instead of leaving it in flight during the MFMAs.
// TILES: the operand in 64-deep K tiles ([kk / 64][row][64], `tile` elements apart; 32-bit element offsets, checked on the host):
// the 128-deep chunk of a workgroup is then two contiguous blocks of 128 B x rows instead of a 256-B piece out of each of `rows`
// rows that lie k elements apart (WN18RR: 400 pieces 409 KB apart per chunk -- 80.6 -> 52.4 us per product).
template <bool TILES>
__device__ __forceinline__ uint4 tk_load(const uint16_t* base, int ld, unsigned tile, int rows, int row0, int idx, int k0, int k_end, int k_safe) {
    const int row = min(row0 + idx / TK_PPR, rows - 1), kk = (idx % TK_PPR) << 3, kc = k0 + kk < k_end ? k0 + kk : k_safe;
    if constexpr (TILES) return *reinterpret_cast<const uint4*>(base + ((unsigned)(kc >> 6) * tile + (unsigned)((row << 6) + (kc & 63))));
    else return *reinterpret_cast<const uint4*>(base + (size_t)row * ld + kc);
}
__device__ __forceinline__ bool tk_keep(int rows, int row0, int idx, int k0, int k_end) {
    return row0 + idx / TK_PPR < rows && k0 + ((idx % TK_PPR) << 3) < k_end;
}

template <bool TILES>
__global__ __launch_bounds__(TK_THREADS) void k_gemm_bf16_tallk(const BfGemm p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t tk_lds[];
    uint16_t* const As = tk_lds;
    uint16_t* const Bs = tk_lds + TK_T * TK_LDK;
    const int m0 = blockIdx.y * TK_T, n0 = blockIdx.x * TK_T;
    const int k_begin = blockIdx.z * p.k_per_split, k_end = min(p.k, k_begin + p.k_per_split);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid & 3, wn = wid >> 2, r = lane & 31, h = lane >> 5;
    const int mt_n = (min(TK_T, p.m - m0) + 31) >> 5, nt_n = (min(TK_T, p.n - n0) + 31) >> 5;      // 32-row tiles that exist
    const uint16_t* const a = static_cast<const uint16_t*>(p.a);
    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float rs0 = 0.f, rs1 = 0.f;
    uint4 va[TK_PPT], vb[TK_PPT];
#pragma unroll
    for (int j = 0; j < TK_PPT; ++j) {
        va[j] = tk_load<TILES>(a, p.lda, p.a_tile, p.m, m0, threadIdx.x + j * TK_THREADS, k_begin, k_end, k_begin);
        vb[j] = tk_load<TILES>(p.b, p.ldb, p.b_tile, p.n, n0, threadIdx.x + j * TK_THREADS, k_begin, k_end, k_begin);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += TK_KC) {
#pragma unroll
        for (int j = 0; j < TK_PPT; ++j) {
            const int idx = threadIdx.x + j * TK_THREADS;
            if (idx < TK_PIECES) {
                // (component-wise masks: a ?: between two uint4 objects becomes a select of ADDRESSES and sends the arrays to scratch)
                const uint32_t ka = tk_keep(p.m, m0, idx, k0, k_end) ? 0xffffffffu : 0u, kb = tk_keep(p.n, n0, idx, k0, k_end) ? 0xffffffffu : 0u;
                const int off = (idx / TK_PPR) * TK_LDK + ((idx % TK_PPR) << 3);
                *reinterpret_cast<uint4*>(&As[off]) = make_uint4(va[j].x & ka, va[j].y & ka, va[j].z & ka, va[j].w & ka);
                *reinterpret_cast<uint4*>(&Bs[off]) = make_uint4(vb[j].x & kb, vb[j].y & kb, vb[j].z & kb, vb[j].w & kb);
            }
        }
        __syncthreads();
        if (p.rowsum_partial && blockIdx.x == 0 && threadIdx.x < TK_T) {       // sum of A's row over this chunk (zero-padded in LDS)
            const uint16_t* ar = &As[threadIdx.x * TK_LDK];
#pragma unroll
            for (int c = 0; c < TK_KC; c += 8) {
                const uint4 v = *reinterpret_cast<const uint4*>(ar + c);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rs0 += __uint_as_float(w[j] << 16);
                    rs1 += __uint_as_float(w[j] & 0xffff0000u);
                }
            }
        }
        // the next chunk's loads fly while this one's MFMAs run (every lane loads, from clamped addresses; the only condition is
        // the workgroup-uniform "there is a next chunk": without it the last chunk re-read the slice's first one -- 1/7 more
        // traffic at WN18RR size; the [row][k] instance keeps the unconditional form, the branch costs it its last free registers)
        if (!TILES || k0 + TK_KC < k_end) {
#pragma unroll
            for (int j = 0; j < TK_PPT; ++j) {
                va[j] = tk_load<TILES>(a, p.lda, p.a_tile, p.m, m0, threadIdx.x + j * TK_THREADS, k0 + TK_KC, k_end, k_begin);
                vb[j] = tk_load<TILES>(p.b, p.ldb, p.b_tile, p.n, n0, threadIdx.x + j * TK_THREADS, k0 + TK_KC, k_end, k_begin);
            }
        }
#pragma unroll
        for (int kk = 0; kk < TK_KC; kk += 16) {
            bf16x8 af[2], bfr[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[((2 * wm + i) * 32 + r) * TK_LDK + kk + 8 * h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[((4 * wn + j) * 32 + r) * TK_LDK + kk + 8 * h]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (2 * wm + i < mt_n && 4 * wn + j < nt_n)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (p.rowsum_partial && blockIdx.x == 0 && threadIdx.x < TK_T && m0 + (int)threadIdx.x < p.m)
        p.rowsum_partial[(size_t)blockIdx.z * p.partial_stride + m0 + threadIdx.x] = rs0 + rs1;
    float* dst = p.partial + (size_t)blockIdx.z * p.partial_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + (4 * wn + j) * 32 + r;
            if (2 * wm + i >= mt_n || 4 * wn + j >= nt_n || col >= p.n) continue;      // (tile 7 belongs to the next block)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (2 * wm + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.m) dst[(size_t)row * p.n + col] = acc[i][j][e];
            }
        }
}
static_assert(TK_T == 7 * 32 && TK_LDS_BYTES <= 160 * 1024 && (TK_LDK / 2) % 8 == 4, "two 224-row chunks in LDS, conflict-free pitch");

// y[r][c] = bf16(x[r][c]) (row-major, ld ldy) and / or yT[c][r] (ld ldt); 64 x 64 tiles through LDS for the transposed copy
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ x, int ldx, int rows, int cols, uint16_t* y,
                                                   int ldy, uint16_t* yT, int ldt) {
    __shared__ uint16_t t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        uint16_t v = 0;
        if (r0 + rr < rows && c0 + cc < cols) {
            v = f2bf(x[(size_t)(r0 + rr) * ldx + c0 + cc]);
            if (y) y[(size_t)(r0 + rr) * ldy + c0 + cc] = v;
        }
        t[rr][cc] = v;
    }
    if (!yT) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, rr = i & 63;
        if (r0 + rr < rows && c0 + cc < cols) yT[(size_t)(c0 + cc) * ldt + r0 + rr] = t[rr][cc];
    }
}

// The IAF update (kgvae/flow_network.py:92-97) and its backward with the bf16 copies the NEXT products read written by the same
// launch (row-major + transposed through a 64 x 64 LDS tile) -- instead of an fp32 result followed by a k_cast_bf16 pass:
//   fwd: x_new = colcount > 0 ? z * exp(alpha + mu) : x_old    -> x_new (fp32), x_b (bf16), x_t (bf16, transposed)
//   bwd: g_net = [g_mu | g_alpha] -> bf16 row-major + transposed only; g_z ACCUMULATED (the passes' contributions add up);
//        g_xold fp32.  Same arithmetic as k_iaf_fwd / k_iaf_bwd (csrc/k_elem.hip).
__global__ __launch_bounds__(256) void k_iaf_fwd_bf16(const float* __restrict__ z, const float* __restrict__ net, int ld_net,
                                                      const float* __restrict__ xold, const int* __restrict__ colcount,
                                                      float* __restrict__ xnew, uint16_t* xb, int ldb, uint16_t* xt, int ldt,
                                                      int rows, int d) {
    __shared__ uint16_t t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        const int r = r0 + rr, c = c0 + cc;
        uint16_t v = 0;
        if (r < rows && c < d) {
            const size_t e = (size_t)r * d + c;
            float x;
            if (colcount[c] > 0) {
                const float mu = net[(size_t)r * ld_net + c], al = net[(size_t)r * ld_net + d + c];
                x = z[e] * expf(al + mu);
            } else {
                x = xold[e];
            }
            xnew[e] = x;
            v = f2bf(x);
            xb[(size_t)r * ldb + c] = v;
        }
        t[rr][cc] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, rr = i & 63;
        if (r0 + rr < rows && c0 + cc < d) xt[(size_t)(c0 + cc) * ldt + r0 + rr] = t[rr][cc];
    }
}

// EX: `net` holds expf(alpha + mu) [rows][ld_net] (what a chain with the fused update stores) instead of [mu | alpha];
// gxold may be NULL then (the backward chain adds the passed-through gradient itself)
template <bool EX>
__global__ __launch_bounds__(256) void k_iaf_bwd_bf16(const float* __restrict__ z, const float* __restrict__ net, int ld_net,
                                                      const int* __restrict__ colcount, const float* __restrict__ gx,
                                                      const float* __restrict__ gld, float* __restrict__ gz_acc,
                                                      uint16_t* gnb, int ldb, uint16_t* gnt, int ldt, float* __restrict__ gxold,
                                                      int rows, int d, int gz_overwrite) {
    // one thread element = column c of [0, d): its g_mu goes to column c and its g_alpha to column d + c of g_net (one exp per
    // element for both), two 64 x 64 LDS tiles for the two transposed copies
    __shared__ uint16_t tm[64][66];
    __shared__ uint16_t ta[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        const int r = r0 + rr, c = c0 + cc;
        uint16_t vm = 0, va = 0;
        if (r < rows && c < d) {
            const size_t e = (size_t)r * d + c;
            const int cnt = colcount[c];
            const float g = gx[e];
            float g_mu = 0.f, g_al = gld ? gld[r] : 0.f, g_z = 0.f, g_old = g;
            if (cnt > 0) {
                const float ex = EX ? net[(size_t)r * ld_net + c] : expf(net[(size_t)r * ld_net + d + c] + net[(size_t)r * ld_net + c]);
                const float gc = g * (float)cnt;
                g_z = gc * ex;
                g_mu = gc * z[e] * ex;
                g_al += g_mu;
                g_old = 0.f;
            }
            gz_acc[e] = (gz_overwrite & 1) ? g_z : gz_acc[e] + g_z;
            if (gxold) gxold[e] = g_old;
            vm = f2bf(g_mu);
            va = f2bf(g_al);
            gnb[(size_t)r * ldb + c] = vm;
            if (!(gz_overwrite & 2)) gnb[(size_t)r * ldb + d + c] = va;
        }
        tm[rr][cc] = vm;
        ta[rr][cc] = va;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, rr = i & 63;
        if (r0 + rr < rows && c0 + cc < d) {
            gnt[(size_t)(c0 + cc) * ldt + r0 + rr] = tm[rr][cc];
            gnt[(size_t)(d + c0 + cc) * ldt + r0 + rr] = ta[rr][cc];
        }
    }
}

// The same two updates with FOUR columns per thread (d % 4 == 0, 16-B aligned rows): 16-B loads and stores of the fp32 operands,
// 8-B stores of the bf16 row-major copies, and the transposed copies leave the LDS tile as four consecutive rows of a column
// per 8-B store (16 lanes = 128 contiguous bytes).  Same arithmetic, element by element, as the scalar kernels above.
// t_tile > 0: the destination in tiles of 64 rows ([row tile][column][64], t_tile elements apart: what gv_gemm_bf16_gradw_tiles reads)
__device__ __forceinline__ void iaf_tile_out(const uint16_t (*t)[68], uint16_t* dst_t, int ldt, int t_tile, int r0, int c0, int rows, int d) {
    const size_t roff = t_tile > 0 ? (size_t)blockIdx.y * t_tile : (size_t)r0;
    if (t_tile > 0) ldt = 64;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int cc = i >> 4, rq = (i & 15) << 2;
        // (tiled destination: the whole 64-row tile is written -- the LDS tile holds zeros in the rows past `rows`, which take part in
        // the weight-gradient reduction, so the caller's buffer needs no fill)
        if (c0 + cc >= d || (t_tile <= 0 && r0 + rq >= rows)) continue;
        uint16_t* o = dst_t + (size_t)(c0 + cc) * ldt + roff + rq;
        const uint2 v = *reinterpret_cast<const uint2*>(&t[cc][rq]);
        if (t_tile > 0 || r0 + rq + 3 < rows) {
            *reinterpret_cast<uint2*>(o) = v;
        } else {
            o[0] = (uint16_t)(v.x & 0xffff);
            if (r0 + rq + 1 < rows) o[1] = (uint16_t)(v.x >> 16);
            if (r0 + rq + 2 < rows) o[2] = (uint16_t)(v.y & 0xffff);
        }
    }
}

__global__ __launch_bounds__(256) void k_iaf_fwd_bf16_v4(const float* __restrict__ z, const float* __restrict__ net, int ld_net,
                                                         const float* __restrict__ xold, const int* __restrict__ colcount,
                                                         float* __restrict__ xnew, uint16_t* xb, int ldb, uint16_t* xt, int ldt,
                                                         int rows, int d, int t_tile) {
    __shared__ __attribute__((aligned(8))) uint16_t t[64][68];          // [column][row]
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int cq = (threadIdx.x & 15) << 2, c = c0 + cq;
    int4 cnt = make_int4(0, 0, 0, 0);
    if (c < d) cnt = *reinterpret_cast<const int4*>(colcount + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = (threadIdx.x >> 4) + 16 * j, r = r0 + rr;
        uint16_t b[4] = {0, 0, 0, 0};
        if (r < rows && c < d) {
            const size_t e = (size_t)r * d + c;
            const float4 zv = *reinterpret_cast<const float4*>(z + e), ov = *reinterpret_cast<const float4*>(xold + e);
            float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), al = mu;
            if (cnt.x > 0 || cnt.y > 0 || cnt.z > 0 || cnt.w > 0) {
                mu = *reinterpret_cast<const float4*>(net + (size_t)r * ld_net + c);
                al = *reinterpret_cast<const float4*>(net + (size_t)r * ld_net + d + c);
            }
            float4 x;
            x.x = cnt.x > 0 ? zv.x * expf(al.x + mu.x) : ov.x;
            x.y = cnt.y > 0 ? zv.y * expf(al.y + mu.y) : ov.y;
            x.z = cnt.z > 0 ? zv.z * expf(al.z + mu.z) : ov.z;
            x.w = cnt.w > 0 ? zv.w * expf(al.w + mu.w) : ov.w;
            *reinterpret_cast<float4*>(xnew + e) = x;
            b[0] = f2bf(x.x); b[1] = f2bf(x.y); b[2] = f2bf(x.z); b[3] = f2bf(x.w);
            *reinterpret_cast<uint2*>(xb + (size_t)r * ldb + c) = make_uint2(b[0] | ((uint32_t)b[1] << 16), b[2] | ((uint32_t)b[3] << 16));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) t[cq + e][rr] = b[e];
    }
    __syncthreads();
    iaf_tile_out(t, xt, ldt, t_tile, r0, c0, rows, d);
}

template <bool EX>
__global__ __launch_bounds__(256) void k_iaf_bwd_bf16_v4(const float* __restrict__ z, const float* __restrict__ net, int ld_net,
                                                         const int* __restrict__ colcount, const float* __restrict__ gx,
                                                         const float* __restrict__ gld, float* __restrict__ gz_acc,
                                                         uint16_t* gnb, int ldb, uint16_t* gnt, int ldt, float* __restrict__ gxold,
                                                         int rows, int d, int gz_overwrite) {
    const int t_tile = (gz_overwrite & 4) ? ldt : 0;
    __shared__ __attribute__((aligned(8))) uint16_t tm[64][68];
    __shared__ __attribute__((aligned(8))) uint16_t ta[64][68];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int cq = (threadIdx.x & 15) << 2, c = c0 + cq;
    int cn[4] = {0, 0, 0, 0};
    if (c < d) {
        const int4 cv = *reinterpret_cast<const int4*>(colcount + c);
        cn[0] = cv.x; cn[1] = cv.y; cn[2] = cv.z; cn[3] = cv.w;
    }
    const bool any = cn[0] > 0 || cn[1] > 0 || cn[2] > 0 || cn[3] > 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = (threadIdx.x >> 4) + 16 * j, r = r0 + rr;
        uint16_t bm[4] = {0, 0, 0, 0}, ba[4] = {0, 0, 0, 0};
        if (r < rows && c < d) {
            const size_t e = (size_t)r * d + c;
            const float4 g4 = *reinterpret_cast<const float4*>(gx + e);
            float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);      // gz_overwrite: the first pass of a backward starts g_z (no zero fill, no read)
            if (!(gz_overwrite & 1)) acc4 = *reinterpret_cast<const float4*>(gz_acc + e);
            float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), mu4 = z4, al4 = z4;
            if (any) {
                z4 = *reinterpret_cast<const float4*>(z + e);
                mu4 = *reinterpret_cast<const float4*>(net + (size_t)r * ld_net + c);
                if (!EX) al4 = *reinterpret_cast<const float4*>(net + (size_t)r * ld_net + d + c);
            }
            const float gl = gld ? gld[r] : 0.f;
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w}, mv[4] = {mu4.x, mu4.y, mu4.z, mu4.w},
                        av[4] = {al4.x, al4.y, al4.z, al4.w};
            float gz[4], go[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g = gv[q];
                float g_mu = 0.f, g_al = gl, g_z = 0.f, g_old = g;
                if (cn[q] > 0) {
                    const float ex = EX ? mv[q] : expf(av[q] + mv[q]);
                    const float gc = g * (float)cn[q];
                    g_z = gc * ex;
                    g_mu = gc * zv[q] * ex;
                    g_al += g_mu;
                    g_old = 0.f;
                }
                gz[q] = g_z;
                go[q] = g_old;
                bm[q] = f2bf(g_mu);
                ba[q] = f2bf(g_al);
            }
            acc4.x += gz[0]; acc4.y += gz[1]; acc4.z += gz[2]; acc4.w += gz[3];
            *reinterpret_cast<float4*>(gz_acc + e) = acc4;
            if (gxold) *reinterpret_cast<float4*>(gxold + e) = make_float4(go[0], go[1], go[2], go[3]);
            uint16_t* ob = gnb + (size_t)r * ldb + c;
            *reinterpret_cast<uint2*>(ob) = make_uint2(bm[0] | ((uint32_t)bm[1] << 16), bm[2] | ((uint32_t)bm[3] << 16));
            if (!(gz_overwrite & 2)) *reinterpret_cast<uint2*>(ob + d) = make_uint2(ba[0] | ((uint32_t)ba[1] << 16), ba[2] | ((uint32_t)ba[3] << 16));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tm[cq + q][rr] = bm[q];
            ta[cq + q][rr] = ba[q];
        }
    }
    __syncthreads();
    iaf_tile_out(tm, gnt, ldt, t_tile, r0, c0, rows, d);
    iaf_tile_out(ta, gnt + (size_t)d * (t_tile > 0 ? 64 : ldt), ldt, t_tile, r0, c0, rows, d);
}

// Sums over bf16 rows (bias gradients from the transposed gradient copies): stage 1, one wave per (row, 4096-column chunk),
// writes part[row][chunk]; stage 2 adds a row's chunk sums in order.  fp32 sums, fixed order.
constexpr int ROWSUM_CHUNK = 4096;
__global__ __launch_bounds__(256) void k_rowsum_bf16(const uint16_t* __restrict__ x, int ld, int rows, int cols, int nchunks,
                                                     float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= rows * nchunks) return;
    const int row = item / nchunks, ch = item - row * nchunks;
    const int c0 = ch * ROWSUM_CHUNK, c1 = min(cols, c0 + ROWSUM_CHUNK);
    const uint16_t* p = x + (size_t)row * ld;
    float s0 = 0.f, s1 = 0.f;
    int c = c0 + lane * 8;
    for (; c + 8 <= c1; c += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0 += __uint_as_float(w[j] << 16);
            s1 += __uint_as_float(w[j] & 0xffff0000u);
        }
    }
    if (c < c1)
        for (int cc = c; cc < c1; ++cc) s0 += __uint_as_float((uint32_t)p[cc] << 16);
    const float s = wave_sum(s0 + s1);
    if (lane == 0) part[item] = s;
}

__global__ __launch_bounds__(256) void k_rowsum_finish(const float* __restrict__ part, int rows, int nchunks, float* out,
                                                       int accumulate) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += part[(size_t)row * nchunks + c];
    out[row] = accumulate ? out[row] + s : s;
}

// the same finish with the rows cut into consecutive segments, each with its own output vector (the bias gradients of all
// layers of a MADE from ONE pass over their stacked transposed gradients)
struct RowsumSegs {
    float* out[GV_ROWSUM_SEG_MAX];
    int first[GV_ROWSUM_SEG_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void k_rowsum_finish_seg(const float* __restrict__ part, int rows, int nchunks, const RowsumSegs sg,
                                                           int accumulate) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    int t = 0;
#pragma unroll
    for (int i = 1; i < GV_ROWSUM_SEG_MAX; ++i)
        if (i < sg.count && row >= sg.first[i]) t = i;
    float* out = sg.out[0];
#pragma unroll
    for (int i = 1; i < GV_ROWSUM_SEG_MAX; ++i)
        if (t == i) out = sg.out[i];
    if (!out) return;
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += part[(size_t)row * nchunks + c];
    const int r = row - sg.first[t];
    out[r] = accumulate ? out[r] + s : s;
}

// out[i] (+)= sum_z partial[z][i]: a workgroup owns 64 consecutive outputs, its waves consecutive ranges of the splits (each summed
// in order, eight partials in flight), the waves' sums added in wave order -- the same bits on every run.  (One thread per output
// walking all ~230 splits of a MADE weight gradient was 29 dependent memory round trips on 157 workgroups: 15.5 us.)
// (a split's partial is `stride` floats: mn of the product, then -- out2 != NULL -- m2 row sums that go to out2, always added)
__global__ __launch_bounds__(1024) void k_splitk_sum(const float* __restrict__ partial, int splits, size_t mn, size_t stride, float* out,
                                                     int accumulate, size_t m2, float* out2) {
    __shared__ float share[16][64];
    const size_t total = mn + (out2 ? m2 : 0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int per = (splits + nw - 1) / nw, z0 = w * per, z1 = min(splits, z0 + per);
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < total) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(z + j) * stride + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; z < z1; ++z) s += partial[(size_t)z * stride + i];
    }
    if (nw > 1) {
        share[w][lane] = s;
        __syncthreads();
        if (w != 0) return;
        for (int j = 1; j < nw; ++j) s += share[j][lane];
    }
    if (i >= total) return;
    if (i < mn) out[i] = accumulate ? out[i] + s : s;
    else out2[i - mn] += s;
}

// ... four consecutive outputs per lane (16-B loads: a wave reads 1 KB of a partial at a time instead of 256 B; the same splits per
// wave and the same order of additions per output as above: bit-identical).  mn, stride, m2 multiples of 4, 16-B aligned operands.
__global__ __launch_bounds__(1024) void k_splitk_sum4(const float* __restrict__ partial, int splits, size_t mn, size_t stride, float* out,
                                                      int accumulate, size_t m2, float* out2) {
    __shared__ float4 share[16][64];
    const size_t total = mn + (out2 ? m2 : 0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int per = (splits + nw - 1) / nw, z0 = w * per, z1 = min(splits, z0 + per);
    const size_t i = ((size_t)blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(partial + (size_t)(z + j) * stride + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        for (; z < z1; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)z * stride + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (nw > 1) {
        share[w][lane] = s;
        __syncthreads();
        if (w != 0) return;
        for (int j = 1; j < nw; ++j) { const float4 v = share[j][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (i >= total) return;
    float4* o = reinterpret_cast<float4*>(i < mn ? out + i : out2 + (i - mn));
    if (i < mn ? accumulate != 0 : true) { const float4 v = *o; s.x = v.x + s.x; s.y = v.y + s.y; s.z = v.z + s.z; s.w = v.w + s.w; }
    *o = s;
}
static void launch_splitk_sum(hipStream_t st, const float* partial, int splits, size_t mn, size_t stride, float* out, int accumulate,
                              size_t m2, float* out2) {
    const size_t total = mn + (out2 ? m2 : 0);
    const int nw = splits >= 64 ? 16 : splits >= 16 ? 4 : 1;      // ~>= 4 partials per wave
    static const bool wide = !(getenv("GV_SPLITK_SUM4") && getenv("GV_SPLITK_SUM4")[0] == '0');
    if (wide && mn % 4 == 0 && stride % 4 == 0 && (!out2 || m2 % 4 == 0) && aligned16(partial) && aligned16(out) && (!out2 || aligned16(out2))) {
        hipLaunchKernelGGL(k_splitk_sum4, dim3((unsigned)((total / 4 + 63) / 64)), dim3(64 * nw), 0, st, partial, splits, mn, stride, out, accumulate,
                           m2, out2);
        return;
    }
    hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)((total + 63) / 64)), dim3(64 * nw), 0, st, partial, splits, mn, stride, out, accumulate,
                       m2, out2);
}


// ---- pass 0 of a MADE backward: the update fed by ONE broadcast [mu | alpha] row ------------------------------------------
// (kgvae/flow_network.py:85-98: the first pass's input is the zero matrix, so every node sees the same net output.)  The generic
// update backward materialises g_net (n x 2d fp32) only for its column sums to be taken -- the gradient w.r.t. the row --, writes a
// g_x_old nobody reads (x_old was the zero matrix) and leaves g_z to an axpby.  Here: exp(alpha + mu) once per column, g_z
// ACCUMULATED in place, the two column sums carried in registers: per element the same expressions as k_iaf_bwd
// (gc = g * cnt; g_z = gc * e; g_mu = gc * z * e; g_alpha = g_ld + g_mu), summed per column over row slices in a fixed order.
constexpr int ROW0_SLICES = 1024;

__global__ __launch_bounds__(256) void k_iaf_bwd_row0(const float* __restrict__ z, const float* __restrict__ net_row,
                                                      const int* __restrict__ colcount, const float* __restrict__ gx,
                                                      const float* __restrict__ gld, float* __restrict__ gz_acc,
                                                      float* __restrict__ part, int rows, int d, int rows_per_slice) {
    __shared__ float sm[256][8];
    const int d4 = d >> 2, rpi = 256 / d4;                 // column groups of 4; rows per sweep of the block
    const int cg = (int)threadIdx.x % d4, ri = (int)threadIdx.x / d4;
    const bool on = ri < rpi;
    const int c = cg << 2;
    float e[4], cn[4];
    {
        const float4 mu = *reinterpret_cast<const float4*>(net_row + c), al = *reinterpret_cast<const float4*>(net_row + d + c);
        const int4 cc = *reinterpret_cast<const int4*>(colcount + c);
        e[0] = expf(al.x + mu.x); e[1] = expf(al.y + mu.y); e[2] = expf(al.z + mu.z); e[3] = expf(al.w + mu.w);
        cn[0] = (float)cc.x; cn[1] = (float)cc.y; cn[2] = (float)cc.z; cn[3] = (float)cc.w;
    }
    float sm_mu[4] = {0.f, 0.f, 0.f, 0.f}, sm_al[4] = {0.f, 0.f, 0.f, 0.f};
    const int r_begin = blockIdx.x * rows_per_slice, r_end = min(rows, r_begin + rows_per_slice);
    if (on) {
        for (int r0 = r_begin + ri; r0 < r_end; r0 += 2 * rpi) {      // two rows in flight
            float4 g4[2], z4[2], a4[2];
            float gl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = min(r0 + u * rpi, r_end - 1);
                const size_t o = (size_t)r * d + c;
                g4[u] = *reinterpret_cast<const float4*>(gx + o);
                z4[u] = *reinterpret_cast<const float4*>(z + o);
                a4[u] = *reinterpret_cast<const float4*>(gz_acc + o);
                gl[u] = gld ? gld[r] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + u * rpi;
                if (r >= r_end) break;
                const float gv_[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w}, zv[4] = {z4[u].x, z4[u].y, z4[u].z, z4[u].w};
                float gz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float g_mu = 0.f, g_al = gl[u];
                    gz[q] = 0.f;
                    if (cn[q] > 0.f) {
                        const float gc = gv_[q] * cn[q];
                        gz[q] = gc * e[q];
                        g_mu = gc * zv[q] * e[q];
                        g_al += g_mu;
                    }
                    sm_mu[q] += g_mu;
                    sm_al[q] += g_al;
                }
                *reinterpret_cast<float4*>(gz_acc + (size_t)r * d + c) =
                    make_float4(a4[u].x + gz[0], a4[u].y + gz[1], a4[u].z + gz[2], a4[u].w + gz[3]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sm[threadIdx.x][q] = sm_mu[q];
        sm[threadIdx.x][4 + q] = sm_al[q];
    }
    __syncthreads();
    if (on && ri == 0) {          // the block's row lanes in order
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = sm[cg][q];
        for (int k = 1; k < rpi; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] += sm[k * d4 + cg][q];
        float* o = part + (size_t)blockIdx.x * 2 * d;
        *reinterpret_cast<float4*>(o + c) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4*>(o + d + c) = make_float4(t[4], t[5], t[6], t[7]);
    }
}

// the slices summed per column: 16 columns per 1024-thread block, 64 groups of slices each (common.h: sum_slices_16x64) -- two
// 256-thread blocks walking 1024 slices one after the other took 60 us
__global__ __launch_bounds__(1024) void k_iaf_row0_final(const float* __restrict__ part, int n2, int nsl, float* __restrict__ out) {
    __shared__ float sm[64][16];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const float acc = sum_slices_16x64(part, n2, nsl, c, sm);
    if ((threadIdx.x >> 4) == 0 && c < n2) out[c] = acc;
}

}  // namespace gv

using namespace gv;

extern "C" int64_t gv_gemm_bf16_nt_workspace_bytes(int m, int n, int split_k) {
    return split_k > 1 ? (int64_t)split_k * m * n * 4 : 0;
}

extern "C" int gv_gemm_bf16_nt(const void* a, int a_is_f32, int lda, const uint16_t* b, int ldb, int m, int n, int k,
                               const float* bias, int relu, const uint16_t* mask, int ldmask, float* c_f32, int ldc,
                               int accumulate, uint16_t* c_bf16, int ldcb, uint16_t* c_bf16_t, int ldct, int split_k,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(m >= 0 && n > 0 && k > 0 && split_k >= 1, GV_ERR_SHAPE, "gv_gemm_bf16_nt: m=%d n=%d k=%d split_k=%d", m, n, k, split_k);
    if (m == 0) return GV_OK;
    GV_REQUIRE(a && b && (c_f32 || c_bf16 || c_bf16_t), GV_ERR_NULL, "gv_gemm_bf16_nt: NULL pointer");
    GV_REQUIRE(k % 8 == 0 && lda >= k && ldb >= k && lda % (a_is_f32 ? 4 : 8) == 0 && ldb % 8 == 0 && aligned16(a) && aligned16(b),
               GV_ERR_ALIGN, "gv_gemm_bf16_nt: k, lda, ldb must allow 16-B row pieces (k=%d lda=%d ldb=%d)", k, lda, ldb);
    GV_REQUIRE((!c_f32 || ldc >= n) && (!c_bf16 || ldcb >= n) && (!c_bf16_t || (ldct >= m && ldct % 4 == 0)) &&
               (!mask || ldmask >= n), GV_ERR_SHAPE, "gv_gemm_bf16_nt: leading dimension too small");
    GV_REQUIRE(n % 4 == 0 && (!c_f32 || (ldc % 4 == 0 && aligned16(c_f32))) && (!c_bf16 || (ldcb % 4 == 0 && aligned16(c_bf16))) &&
               (!mask || (ldmask % 4 == 0 && aligned16(mask))) && (!c_bf16_t || aligned16(c_bf16_t)), GV_ERR_ALIGN,
               "gv_gemm_bf16_nt: n and the output / mask row pitches must be multiples of 4 elements");
    BfGemm p;
    p.a = a; p.lda = lda; p.b = b; p.ldb = ldb; p.m = m; p.n = n; p.k = k; p.bias = bias; p.relu = relu; p.mask = mask;
    p.ldmask = ldmask; p.c_f32 = c_f32; p.ldc = ldc; p.accumulate = accumulate; p.c_bf = c_bf16; p.ldcb = ldcb;
    p.c_bft = c_bf16_t; p.ldct = ldct; p.partial = nullptr; p.k_per_split = k; p.rowsum_partial = nullptr;
    p.partial_stride = (size_t)m * n;
    p.a_tile = p.b_tile = 0;
    hipStream_t st = (hipStream_t)stream;
    int splits = 1;
    bool tall = false;
    if (split_k > 1) {
        GV_REQUIRE(c_f32 && !c_bf16 && !c_bf16_t && !bias && !relu && !mask, GV_ERR_SHAPE,
                   "gv_gemm_bf16_nt: split-K writes a plain fp32 result only");
        // small outputs over a long reduction: whole-output workgroups (every operand element read once)
        static const bool tall_ok = !(getenv("GV_GEMM_TALL") && getenv("GV_GEMM_TALL")[0] == '0');
        tall = tall_ok && !a_is_f32 && n <= 2 * TK_T && m <= 4 * TK_T && k >= TK_KC * split_k;
        const int chunk = tall ? TK_KC : 224;
        int per = ((k + split_k - 1) / split_k + chunk - 1) / chunk * chunk;      // whole chunks per split
        splits = (k + per - 1) / per;
        GV_REQUIRE(workspace && workspace_bytes >= (int64_t)splits * m * n * 4, GV_ERR_WORKSPACE,
                   "gv_gemm_bf16_nt: workspace too small for %d splits", splits);
        p.partial = (float*)workspace; p.k_per_split = per;
    }
    if (tall) {
        static unsigned long long lds_done = 0;
        if (!raise_dynamic_lds((const void*)k_gemm_bf16_tallk<false>, (int)TK_LDS_BYTES, lds_done, "gv_gemm_bf16_nt")) return GV_ERR_SHAPE;
        hipLaunchKernelGGL(k_gemm_bf16_tallk<false>, dim3((n + TK_T - 1) / TK_T, (m + TK_T - 1) / TK_T, splits), dim3(TK_THREADS), TK_LDS_BYTES,
                           st, p);
    } else {
        dim3 grid((n + 63) / 64, (m + 63) / 64, splits), block(256);
        if (a_is_f32) hipLaunchKernelGGL(k_gemm_bf16s<true>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(k_gemm_bf16s<false>, grid, block, 0, st, p);
    }
    int rc = launch_status("gv_gemm_bf16_nt");
    if (rc != GV_OK || split_k <= 1) return rc;
    const size_t mn = (size_t)m * n;
    GV_REQUIRE(ldc == n, GV_ERR_SHAPE, "gv_gemm_bf16_nt: split-K needs a dense result (ldc == n)");
    launch_splitk_sum(st, (const float*)workspace, splits, mn, mn, c_f32, accumulate, (size_t)0, nullptr);
    return launch_status("gv_gemm_bf16_nt(split-k sum)");
}

/* The weight-gradient product of the masked MLP with its bias gradient from the same pass: c (+)= A B^T over a long reduction
 * on the whole-output kernel, a_rowsum[i] += sum_k A[i][k].  Both dense fp32; fits: small outputs, bf16 operands, enough k. */
static bool gradw_splits(int m, int n, int k, int split_k, int* per, int* splits) {
    if (split_k < 2 || n > 2 * TK_T || m > 4 * TK_T || k < TK_KC * split_k) return false;
    *per = ((k + split_k - 1) / split_k + TK_KC - 1) / TK_KC * TK_KC;
    *splits = (k + *per - 1) / *per;
    return true;
}

extern "C" int gv_gemm_bf16_gradw_fits(int m, int n, int k, int split_k) {
    int per, splits;
    return gradw_splits(m, n, k, split_k, &per, &splits) ? 1 : 0;
}

extern "C" int64_t gv_gemm_bf16_gradw_workspace_bytes(int m, int n, int split_k) {
    return split_k > 1 ? (int64_t)split_k * m * (n + 1) * 4 : 0;
}

template <bool TILES>
static int gemm_bf16_gradw(const char* who, const uint16_t* a, int lda, unsigned a_tile, const uint16_t* b, int ldb, unsigned b_tile,
                           int m, int n, int k, float* c_f32, int accumulate, float* a_rowsum, int split_k, void* workspace,
                           int64_t workspace_bytes, void* stream) {
    int per, splits;
    GV_REQUIRE(gradw_splits(m, n, k, split_k, &per, &splits), GV_ERR_SHAPE,
               "%s: %d x %d over k=%d with %d splits does not fit the whole-output kernel", who, m, n, k, split_k);
    GV_REQUIRE(workspace_bytes >= (int64_t)splits * m * (n + 1) * 4, GV_ERR_WORKSPACE, "%s: workspace too small", who);
    BfGemm p;
    p.a = a; p.lda = lda; p.b = b; p.ldb = ldb; p.m = m; p.n = n; p.k = k; p.bias = nullptr; p.relu = 0; p.mask = nullptr;
    p.ldmask = 0; p.c_f32 = c_f32; p.ldc = n; p.accumulate = accumulate; p.c_bf = nullptr; p.ldcb = 0; p.c_bft = nullptr; p.ldct = 0;
    p.partial = (float*)workspace; p.k_per_split = per;
    p.partial_stride = (size_t)m * (n + 1);         // a split's product, then its row sums
    p.rowsum_partial = a_rowsum ? (float*)workspace + (size_t)m * n : nullptr;
    p.a_tile = a_tile; p.b_tile = b_tile;
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long lds_done = 0;
    if (!raise_dynamic_lds((const void*)k_gemm_bf16_tallk<TILES>, (int)TK_LDS_BYTES, lds_done, "gv_gemm_bf16_gradw")) return GV_ERR_SHAPE;
    hipLaunchKernelGGL(k_gemm_bf16_tallk<TILES>, dim3((n + TK_T - 1) / TK_T, (m + TK_T - 1) / TK_T, splits), dim3(TK_THREADS), TK_LDS_BYTES, st, p);
    const size_t mn = (size_t)m * n;
    launch_splitk_sum(st, (const float*)workspace, splits, mn, p.partial_stride, c_f32, accumulate, (size_t)m, a_rowsum);
    return launch_status(who);
}

extern "C" int gv_gemm_bf16_gradw(const uint16_t* a, int lda, const uint16_t* b, int ldb, int m, int n, int k, float* c_f32,
                                  int accumulate, float* a_rowsum, int split_k, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
    GV_REQUIRE(m > 0 && n > 0 && k > 0, GV_ERR_SHAPE, "gv_gemm_bf16_gradw: m=%d n=%d k=%d", m, n, k);
    GV_REQUIRE(a && b && c_f32 && workspace, GV_ERR_NULL, "gv_gemm_bf16_gradw: NULL pointer");
    GV_REQUIRE(k % 8 == 0 && lda >= k && ldb >= k && lda % 8 == 0 && ldb % 8 == 0 && aligned16(a) && aligned16(b), GV_ERR_ALIGN,
               "gv_gemm_bf16_gradw: k, lda, ldb must allow 16-B row pieces (k=%d lda=%d ldb=%d)", k, lda, ldb);
    return gemm_bf16_gradw<false>("gv_gemm_bf16_gradw", a, lda, 0u, b, ldb, 0u, m, n, k, c_f32, accumulate, a_rowsum, split_k, workspace,
                                  workspace_bytes, stream);
}

/* ... with both operands in 64-deep K TILES: element (row, kk) of A at a[(kk / 64) * a_tile + row * 64 + kk % 64] (a_tile >= 64 m
 * elements between tiles; B likewise): a workgroup's K slice is a few CONTIGUOUS blocks of memory instead of one short piece
 * out of each of m + n rows that lie k elements apart. */
extern "C" int gv_gemm_bf16_gradw_tiles(const uint16_t* a, int64_t a_tile, const uint16_t* b, int64_t b_tile, int m, int n, int k,
                                        float* c_f32, int accumulate, float* a_rowsum, int split_k, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(m > 0 && n > 0 && k > 0, GV_ERR_SHAPE, "gv_gemm_bf16_gradw_tiles: m=%d n=%d k=%d", m, n, k);
    GV_REQUIRE(a && b && c_f32 && workspace, GV_ERR_NULL, "gv_gemm_bf16_gradw_tiles: NULL pointer");
    GV_REQUIRE(k % 64 == 0 && a_tile >= (int64_t)64 * m && b_tile >= (int64_t)64 * n && a_tile % 8 == 0 && b_tile % 8 == 0 &&
               aligned16(a) && aligned16(b), GV_ERR_ALIGN,
               "gv_gemm_bf16_gradw_tiles: k must be whole 64-deep tiles of >= 64 * rows elements (k=%d a_tile=%lld b_tile=%lld)", k,
               (long long)a_tile, (long long)b_tile);
    GV_REQUIRE((int64_t)(k / 64) * a_tile <= (int64_t)UINT32_MAX && (int64_t)(k / 64) * b_tile <= (int64_t)UINT32_MAX, GV_ERR_SHAPE,
               "gv_gemm_bf16_gradw_tiles: an operand spans more than 2^32 elements");
    return gemm_bf16_gradw<true>("gv_gemm_bf16_gradw_tiles", a, 0, (unsigned)a_tile, b, 0, (unsigned)b_tile, m, n, k, c_f32, accumulate,
                                 a_rowsum, split_k, workspace, workspace_bytes, stream);
}

extern "C" int gv_cast_bf16(const float* x, int ldx, int rows, int cols, uint16_t* y, int ldy, uint16_t* y_t, int ldt,
                            void* stream) {
    GV_REQUIRE(rows >= 0 && cols > 0, GV_ERR_SHAPE, "gv_cast_bf16: rows=%d cols=%d", rows, cols);
    if (rows == 0) return GV_OK;
    GV_REQUIRE(x && (y || y_t), GV_ERR_NULL, "gv_cast_bf16: NULL pointer");
    GV_REQUIRE(ldx >= cols && (!y || ldy >= cols) && (!y_t || ldt >= rows), GV_ERR_SHAPE, "gv_cast_bf16: leading dimension too small");
    hipLaunchKernelGGL(k_cast_bf16, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, rows,
                       cols, y, ldy, y_t, ldt);
    return launch_status("gv_cast_bf16");
}

static int iaf_update_fwd_bf16(const char* what, const float* z, const float* net, int ld_net, const float* x_old, const int32_t* colcount,
                               float* x_new, uint16_t* x_b, int ldb, uint16_t* x_t, int ldt, int64_t t_tile, int64_t n, int d, void* stream) {
    GV_REQUIRE(n >= 0 && d > 0 && n < (1ll << 31), GV_ERR_SHAPE, "%s: n=%lld d=%d", what, (long long)n, d);
    if (n == 0) return GV_OK;
    GV_REQUIRE(z && net && x_old && colcount && x_new && x_b && x_t, GV_ERR_NULL, "%s: NULL pointer", what);
    GV_REQUIRE(ldb >= d && (t_tile > 0 ? (t_tile >= (int64_t)64 * d && t_tile <= INT32_MAX) : ldt >= n), GV_ERR_SHAPE,
               "%s: leading dimension too small", what);
    const bool v4 = d % 4 == 0 && ld_net % 4 == 0 && ldb % 4 == 0 && (t_tile > 0 ? t_tile % 4 == 0 : ldt % 4 == 0) && aligned16(z) &&
                    aligned16(net) && aligned16(x_old) && aligned16(x_new) && aligned16(colcount) &&
                    (reinterpret_cast<uintptr_t>(x_b) & 7u) == 0 && (reinterpret_cast<uintptr_t>(x_t) & 7u) == 0;
    GV_REQUIRE(v4 || t_tile == 0, GV_ERR_ALIGN, "%s: the tiled copy needs d %% 4 == 0 and 16-B aligned fp32 rows", what);
    if (v4)
        hipLaunchKernelGGL(k_iaf_fwd_bf16_v4, dim3((d + 63) / 64, (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, z,
                           net, ld_net, x_old, colcount, x_new, x_b, ldb, x_t, ldt, (int)n, d, (int)t_tile);
    else
        hipLaunchKernelGGL(k_iaf_fwd_bf16, dim3((d + 63) / 64, (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, z,
                           net, ld_net, x_old, colcount, x_new, x_b, ldb, x_t, ldt, (int)n, d);
    return launch_status(what);
}

extern "C" int gv_iaf_update_fwd_bf16(const float* z, const float* net, int ld_net, const float* x_old, const int32_t* colcount,
                                     float* x_new, uint16_t* x_b, int ldb, uint16_t* x_t, int ldt, int64_t n, int d,
                                     void* stream) {
    return iaf_update_fwd_bf16("gv_iaf_update_fwd_bf16", z, net, ld_net, x_old, colcount, x_new, x_b, ldb, x_t, ldt, 0, n, d, stream);
}

extern "C" int gv_iaf_update_fwd_bf16_tiles(const float* z, const float* net, int ld_net, const float* x_old, const int32_t* colcount,
                                           float* x_new, uint16_t* x_b, int ldb, uint16_t* x_t, int64_t t_tile, int64_t n, int d,
                                           void* stream) {
    GV_REQUIRE(t_tile > 0, GV_ERR_SHAPE, "gv_iaf_update_fwd_bf16_tiles: t_tile=%lld", (long long)t_tile);
    return iaf_update_fwd_bf16("gv_iaf_update_fwd_bf16_tiles", z, net, ld_net, x_old, colcount, x_new, x_b, ldb, x_t, 0, t_tile, n, d, stream);
}

static int iaf_update_bwd_bf16(const char* what, bool ex, const float* z, const float* net, int ld_net, const int32_t* colcount,
                               const float* gx, const float* gld, float* gz_accumulate, uint16_t* gnet_b, int ldb, uint16_t* gnet_t,
                               int ldt, float* gx_old, int gz_overwrite, int64_t n, int d, void* stream) {
    GV_REQUIRE(n >= 0 && d > 0 && n < (1ll << 31), GV_ERR_SHAPE, "%s: n=%lld d=%d", what, (long long)n, d);
    if (n == 0) return GV_OK;
    GV_REQUIRE(z && net && colcount && gx && gz_accumulate && gnet_b && gnet_t && (gx_old || ex), GV_ERR_NULL, "%s: NULL pointer", what);
    const bool tiles = (gz_overwrite & 4) != 0;      // gnet_t in tiles of 64 rows, ldt elements apart
    GV_REQUIRE(ldb >= ((gz_overwrite & 2) ? d : 2 * d) && (tiles ? ldt >= 128 * d : ldt >= n) && ld_net >= (ex ? d : 2 * d), GV_ERR_SHAPE,
               "%s: leading dimension too small", what);
    GV_REQUIRE(!(gz_overwrite & 2) || !gld, GV_ERR_SHAPE, "%s: the mu-half-only form needs g_logdet == NULL (g_alpha == g_mu then)", what);
    const bool v4 = d % 4 == 0 && ld_net % 4 == 0 && ldb % 4 == 0 && ldt % 4 == 0 && aligned16(z) && aligned16(net) && aligned16(gx) &&
                    aligned16(gz_accumulate) && (!gx_old || aligned16(gx_old)) && aligned16(colcount) &&
                    (reinterpret_cast<uintptr_t>(gnet_b) & 7u) == 0 && (reinterpret_cast<uintptr_t>(gnet_t) & 7u) == 0;
    GV_REQUIRE(v4 || !tiles, GV_ERR_ALIGN, "%s: the tiled copy needs d %% 4 == 0 and 16-B aligned fp32 rows", what);
    const dim3 grid((d + 63) / 64, (unsigned)((n + 63) / 64));
#define GV_IAF_BWD(K)                                                                                                        \
    hipLaunchKernelGGL(K, grid, dim3(256), 0, (hipStream_t)stream, z, net, ld_net, colcount, gx, gld, gz_accumulate, gnet_b, ldb,   \
                       gnet_t, ldt, gx_old, (int)n, d, gz_overwrite)
    if (v4 && ex) GV_IAF_BWD(k_iaf_bwd_bf16_v4<true>);
    else if (v4) GV_IAF_BWD(k_iaf_bwd_bf16_v4<false>);
    else if (ex) GV_IAF_BWD(k_iaf_bwd_bf16<true>);
    else GV_IAF_BWD(k_iaf_bwd_bf16<false>);
#undef GV_IAF_BWD
    return launch_status(what);
}

extern "C" int gv_iaf_update_bwd_bf16(const float* z, const float* net, int ld_net, const int32_t* colcount, const float* gx,
                                     const float* gld, float* gz_accumulate, uint16_t* gnet_b, int ldb, uint16_t* gnet_t, int ldt,
                                     float* gx_old, int64_t n, int d, void* stream) {
    return iaf_update_bwd_bf16("gv_iaf_update_bwd_bf16", false, z, net, ld_net, colcount, gx, gld, gz_accumulate, gnet_b, ldb, gnet_t,
                               ldt, gx_old, 0, n, d, stream);
}

extern "C" int gv_iaf_update_bwd_bf16_ex(const float* z, const float* ex, int ld_ex, const int32_t* colcount, const float* gx,
                                        const float* gld, float* gz_accumulate, uint16_t* gnet_b, int ldb, uint16_t* gnet_t, int ldt,
                                        float* gx_old, int gz_overwrite, int64_t n, int d, void* stream) {
    return iaf_update_bwd_bf16("gv_iaf_update_bwd_bf16_ex", true, z, ex, ld_ex, colcount, gx, gld, gz_accumulate, gnet_b, ldb, gnet_t,
                               ldt, gx_old, gz_overwrite, n, d, stream);
}


extern "C" int64_t gv_iaf_update_bwd_row0_workspace_floats(int d) { return (int64_t)ROW0_SLICES * 2 * d; }

extern "C" int gv_iaf_update_bwd_row0(const float* z, const float* net_row, const int32_t* colcount, const float* gx, const float* gld,
                                      float* gz_accumulate, float* g_row, float* workspace, int64_t n, int d, void* stream) {
    GV_REQUIRE(n >= 0 && n < (1ll << 31) && d > 0 && d % 4 == 0 && d <= 1024, GV_ERR_SHAPE, "gv_iaf_update_bwd_row0: n=%lld d=%d (d %% 4 == 0, <= 1024)",
               (long long)n, d);
    GV_REQUIRE(z && net_row && colcount && gx && gz_accumulate && g_row && workspace, GV_ERR_NULL, "gv_iaf_update_bwd_row0: NULL pointer");
    GV_REQUIRE(aligned16(z) && aligned16(net_row) && aligned16(colcount) && aligned16(gx) && aligned16(gz_accumulate) && aligned16(workspace),
               GV_ERR_ALIGN, "gv_iaf_update_bwd_row0: 16-B aligned operands");
    hipStream_t st = (hipStream_t)stream;
    const int rpi = 256 / (d / 4);
    int per = (int)((n + ROW0_SLICES - 1) / ROW0_SLICES);
    per = (per + 2 * rpi - 1) / (2 * rpi) * (2 * rpi);          // whole sweeps of the block
    if (per < 2 * rpi) per = 2 * rpi;
    const int nsl = n > 0 ? (int)((n + per - 1) / per) : 0;
    if (nsl > 0)
        hipLaunchKernelGGL(k_iaf_bwd_row0, dim3(nsl), dim3(256), 0, st, z, net_row, colcount, gx, gld, gz_accumulate, workspace, (int)n, d, per);
    hipLaunchKernelGGL(k_iaf_row0_final, dim3((2 * d + 15) / 16), dim3(1024), 0, st, (const float*)workspace, 2 * d, nsl, g_row);
    return launch_status("gv_iaf_update_bwd_row0");
}

extern "C" int gv_rowsum_bf16(const uint16_t* x, int ld, int rows, int cols, float* out, int accumulate, float* workspace,
                              void* stream) {
    GV_REQUIRE(rows >= 0 && cols >= 0, GV_ERR_SHAPE, "gv_rowsum_bf16: rows=%d cols=%d", rows, cols);
    if (rows == 0) return GV_OK;
    GV_REQUIRE(x && out && workspace, GV_ERR_NULL, "gv_rowsum_bf16: NULL pointer");
    GV_REQUIRE(ld >= cols && ld % 8 == 0 && aligned16(x), GV_ERR_ALIGN, "gv_rowsum_bf16: rows must start on 16-B boundaries");
    const int nchunks = max(1, (cols + ROWSUM_CHUNK - 1) / ROWSUM_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rowsum_bf16, dim3((rows * nchunks + 3) / 4), dim3(256), 0, st, x, ld, rows, cols, nchunks, workspace);
    hipLaunchKernelGGL(k_rowsum_finish, dim3((rows + 255) / 256), dim3(256), 0, st, (const float*)workspace, rows, nchunks, out,
                       accumulate);
    return launch_status("gv_rowsum_bf16");
}

extern "C" int gv_rowsum_bf16_segments(const uint16_t* x, int ld, int rows, int cols, int count, float* const* outs,
                                       const int32_t* seg_rows, int accumulate, float* workspace, void* stream) {
    GV_REQUIRE(rows >= 0 && cols >= 0 && count >= 1 && count <= GV_ROWSUM_SEG_MAX, GV_ERR_SHAPE,
               "gv_rowsum_bf16_segments: rows=%d cols=%d count=%d", rows, cols, count);
    if (rows == 0) return GV_OK;
    GV_REQUIRE(x && outs && seg_rows && workspace, GV_ERR_NULL, "gv_rowsum_bf16_segments: NULL pointer");
    GV_REQUIRE(ld >= cols && ld % 8 == 0 && aligned16(x), GV_ERR_ALIGN, "gv_rowsum_bf16_segments: rows must start on 16-B boundaries");
    RowsumSegs sg;
    sg.count = count;
    int first = 0;
    for (int i = 0; i < count; ++i) {
        GV_REQUIRE(seg_rows[i] >= 0, GV_ERR_SHAPE, "gv_rowsum_bf16_segments: segment %d has %d rows", i, seg_rows[i]);
        sg.out[i] = outs[i];
        sg.first[i] = first;
        first += seg_rows[i];
    }
    sg.first[count] = first;
    GV_REQUIRE(first == rows, GV_ERR_SHAPE, "gv_rowsum_bf16_segments: segments cover %d of %d rows", first, rows);
    const int nchunks = max(1, (cols + ROWSUM_CHUNK - 1) / ROWSUM_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rowsum_bf16, dim3((rows * nchunks + 3) / 4), dim3(256), 0, st, x, ld, rows, cols, nchunks, workspace);
    hipLaunchKernelGGL(k_rowsum_finish_seg, dim3((rows + 255) / 256), dim3(256), 0, st, (const float*)workspace, rows, nchunks, sg,
                       accumulate);
    return launch_status("gv_rowsum_bf16_segments");
}

/* floats of workspace gv_rowsum_bf16 needs */
extern "C" int64_t gv_rowsum_bf16_workspace_floats(int rows, int cols) {
    return (int64_t)rows * ((cols + ROWSUM_CHUNK - 1) / ROWSUM_CHUNK + 1);
}
