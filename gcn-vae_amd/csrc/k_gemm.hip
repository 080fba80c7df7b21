// K2/K4: dense fp32 GEMM on the gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = act( op(A)[M,K] @ op(B)[K,N] + bias[N] ) (+ C)
//
// The MFMA result is bit-for-bit a k-ordered fp32 fma chain (no reduced-precision path exists on
// gfx950), so parity with the CPU reference is at fp32 rounding.  Block tile 128x64x16, four waves
// in a 2x2 grid, each wave owns a 64x32 patch = two 32x32 accumulators.  Operands are staged
// through LDS k-major (As[k][m], Bs[k][n], +1 padding) so that the MFMA operand read -- lane l
// needs A[m = l&31][k = l>>5] and B[k = l>>5][n = l&31] -- is a conflict-free ds_read_b32 for both
// 32-lane halves.  Global loads of tile t+1 are issued before the MFMAs of tile t (register
// staging); the shapes on this path are skinny (K = N = 200..400, M = nodes), so the kernel is
// sized for many small blocks rather than for a 256^2 pipeline.
// split-K (grid.z) writes raw partial tiles to a workspace that a second kernel sums in order.
#include <stdlib.h>

#include "common.h"

namespace gv {


typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float* a;
    const float* b;
    float* c;
    const float* bias;
    const float* a_mask;   // optional, same layout as A: A is read as (a_mask > 0 ? A : 0)  (ReLU backward folded in)
    float* ws;
    int m, n, k, lda, ldb, ldc;
    int act, accumulate, split_k, k_chunk;
    int vec_a, vec_b;
    const int* rows_dev;   // optional device scalar: only the first *rows_dev rows of the stored A exist (padding of a static-shape batch)
    // B known to be zero in whole blocks (an autoregressive mask folded into the weights): per 64-column tile of C one word, bit c set =
    // B holds non-zero entries in k-chunk c (16 wide); chunks with a clear bit are not loaded and not multiplied (BK = 16, no split-K)
    const unsigned long long* kmask;
    // C known to be zero in whole 64 x 64 tiles (the weight gradient of such a layer): bit (i * tmask_ld + j) set = tile (i, j) is wanted;
    // other tiles are stored as zeros (nothing when accumulating) without reading A or B
    const unsigned long long* tmask;
    int tmask_ld;
    int tmask_wanted;      // > 0 (split-K only): the launch holds ONE block row per wanted tile -- block x is the x-th set bit of tmask
};

// guarded load of VPT consecutive floats (VPT % 4 == 0) along the contiguous dimension; `lim` bounds that
// dimension, `ok` says the other coordinate is inside the matrix
template <int VPT>
__device__ __forceinline__ void load_run(const float* src, int pos, int lim, bool ok, bool vec, float (&r)[VPT], int off) {
#pragma unroll
    for (int h = 0; h < VPT / 4; ++h) {
        const int q = pos + 4 * h;
        if (ok && vec && q + 3 < lim) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * h);
            r[off + 4 * h] = v.x; r[off + 4 * h + 1] = v.y; r[off + 4 * h + 2] = v.z; r[off + 4 * h + 3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[off + 4 * h + i] = (ok && q + i < lim) ? src[4 * h + i] : 0.f;
        }
    }
}

// the same run of the mask matrix zeroes the entries whose mask value is not positive
template <int VPT>
__device__ __forceinline__ void mask_run(const float* msk, int pos, int lim, bool ok, bool vec, float (&r)[VPT]) {
    float mv[VPT];
    load_run<VPT>(msk, pos, lim, ok, vec, mv, 0);
#pragma unroll
    for (int i = 0; i < VPT; ++i) r[i] = mv[i] > 0.f ? r[i] : 0.f;
}

// ---- global -> registers: BM*BK/256 floats of A and BN*BK/256 floats of B per thread, zero outside the matrix
template <bool TA, int BM, int BK>
__device__ __forceinline__ void load_a(const GemmParams& p, int m0, int k0, int kend, float (&r)[BM * BK / 256]) {
    constexpr int APT = BM * BK / 256;
    const int t = threadIdx.x;
    if constexpr (!TA) {               // A is [M, K], K contiguous: BK/APT threads per row
        constexpr int TPR = BK / APT;
        const int m = m0 + t / TPR, kk = k0 + (t % TPR) * APT;
        load_run<APT>(p.a + (size_t)m * p.lda + kk, kk, kend, m < p.m, p.vec_a, r, 0);
        if (p.a_mask) mask_run<APT>(p.a_mask + (size_t)m * p.lda + kk, kk, kend, m < p.m, p.vec_a, r);
    } else {                           // A is stored [K, M], M contiguous: BM/APT threads per k
        constexpr int TPK = BM / APT;
        const int kq = k0 + t / TPK, m = m0 + (t % TPK) * APT;
        load_run<APT>(p.a + (size_t)kq * p.lda + m, m, p.m, kq < kend, p.vec_a, r, 0);
        if (p.a_mask) mask_run<APT>(p.a_mask + (size_t)kq * p.lda + m, m, p.m, kq < kend, p.vec_a, r);
    }
}

template <bool TB, int BN, int BK>
__device__ __forceinline__ void load_b(const GemmParams& p, int n0, int k0, int kend, float (&r)[BN * BK / 256]) {
    constexpr int BPT = BN * BK / 256;
    const int t = threadIdx.x;
    if constexpr (!TB) {               // B is [K, N], N contiguous: BN/BPT threads per k
        constexpr int TPK = BN / BPT;
        const int kq = k0 + t / TPK, n = n0 + (t % TPK) * BPT;
        load_run<BPT>(p.b + (size_t)kq * p.ldb + n, n, p.n, kq < kend, p.vec_b, r, 0);
    } else {                           // B is stored [N, K], K contiguous: BK/BPT threads per n
        constexpr int TPN = BK / BPT;
        const int n = n0 + t / TPN, kq = k0 + (t % TPN) * BPT;
        load_run<BPT>(p.b + (size_t)n * p.ldb + kq, kq, kend, n < p.n, p.vec_b, r, 0);
    }
}

template <bool TA, int BM, int BK>
__device__ __forceinline__ void stage_a(float* As, const float (&r)[BM * BK / 256]) {
    constexpr int APT = BM * BK / 256, LDA_S = BM + 1;
    const int t = threadIdx.x;
    if constexpr (!TA) {
        constexpr int TPR = BK / APT;
        const int m = t / TPR, kk = (t % TPR) * APT;
#pragma unroll
        for (int i = 0; i < APT; ++i) As[(kk + i) * LDA_S + m] = r[i];
    } else {
        constexpr int TPK = BM / APT;
        const int kq = t / TPK, m = (t % TPK) * APT;
#pragma unroll
        for (int i = 0; i < APT; ++i) As[kq * LDA_S + m + i] = r[i];
    }
}

template <bool TB, int BN, int BK>
__device__ __forceinline__ void stage_b(float* Bs, const float (&r)[BN * BK / 256]) {
    constexpr int BPT = BN * BK / 256, LDB_S = BN + 1;
    const int t = threadIdx.x;
    if constexpr (!TB) {
        constexpr int TPK = BN / BPT;
        const int kq = t / TPK, n = (t % TPK) * BPT;
#pragma unroll
        for (int i = 0; i < BPT; ++i) Bs[kq * LDB_S + n + i] = r[i];
    } else {
        constexpr int TPN = BK / BPT;
        const int n = t / TPN, kq = (t % TPN) * BPT;
#pragma unroll
        for (int i = 0; i < BPT; ++i) Bs[(kq + i) * LDB_S + n] = r[i];
    }
}

// MT / NT = 32-row / 32-column MFMA tiles per wave: block tile (64*MT) x (64*NT) x BK, four waves as 2 (M) x 2 (N).
template <bool TA, bool TB, int MT, int NT, int BK>
__global__ __launch_bounds__(256) void k_gemm_f32(const GemmParams p) {
    constexpr int BM = 64 * MT, BN = 64 * NT, LDA_S = BM + 1, LDB_S = BN + 1;
    __shared__ float As[BK * LDA_S];
    __shared__ float Bs[BK * LDB_S];
    // Workgroups are dealt round-robin to the 8 XCDs.  The n-tiles of one m-tile share the A rows: give them CONSECUTIVE slots of
    // the SAME XCD, so that a tall A (configs[4]: 800 MB, far beyond the Infinity Cache) crosses the fabric once instead of once
    // per n-tile.  (Whole groups of 8 m-tiles only; the ragged end keeps the plain order.)
    int mt_i = blockIdx.y, nt_i = blockIdx.x;
    if (p.tmask_wanted > 0) {
        // the x-th wanted 64 x 64 tile (MT = NT = 1): no block exists for the others -- with one block per tile, wanted or not, the CUs
        // that were dealt 4 wanted blocks set the launch's time whatever the others hold (2-4 per CU under a triangular mask)
        int rank = blockIdx.x, bit = -1;
        for (int w = 0; bit < 0; ++w) {
            unsigned long long word = p.tmask[w];
            const int pc = __popcll(word);
            if (rank >= pc) { rank -= pc; continue; }
            for (int i = 0; i < rank; ++i) word &= word - 1;
            bit = 64 * w + __builtin_ctzll(word);
        }
        mt_i = bit / p.tmask_ld;
        nt_i = bit - mt_i * p.tmask_ld;
    } else {
        const int ntn = gridDim.x, lin = blockIdx.y * ntn + blockIdx.x;
        const int full = (gridDim.y / 8) * 8 * ntn;
        if (lin < full) {
            const int xcd = lin & 7, j = lin >> 3;
            // (rotated by the m-tile group: an XCD deals its slots to its 32 CUs in turn, so with 8 column tiles a CU would receive the
            // SAME column tile every time -- and under k-chunk words the column tiles differ in length: 72 us instead of ~55 on the 500 x 500
            // MADE layer, the CUs holding the unmasked tiles finishing last)
            nt_i = (j + j / ntn) % ntn;
            mt_i = (j / ntn) * 8 + xcd;
            // wanted-tile words (a block-triangular weight gradient): XCD x would hold row-tile x of every column tile, i.e. 1 .. 8 wanted
            // tiles, and a CU -- dealt every 32nd slot of its XCD -- the SAME tile of every k-split.  Tiles with (row + column) % 8 = x
            // go to XCD x (4-5 wanted ones each under a triangular mask), the column rotated with the split so that a CU's blocks differ
            if (p.tmask) {
                const int z = blockIdx.z;
                nt_i = (nt_i + z + (z >> 2)) % ntn;
                mt_i = (j / ntn) * 8 + ((xcd - nt_i) & 7);
            }
        }
    }
    const int m0 = mt_i * BM, n0 = nt_i * BN;
    const int kbeg = blockIdx.z * p.k_chunk;
    int kend = min(p.k, kbeg + p.k_chunk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = (wid >> 1) * 32 * MT, wn = (wid & 1) * 32 * NT;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (p.rows_dev) {
        // a static-shape batch pads its node arrays: rows [*rows_dev, ...) of the stored A are padding.  A row-major A: a tile that
        // holds only padding rows is not computed -- zeros are stored (nothing when accumulating); A stored [K, M]: K ends there
        const int live = *p.rows_dev;
        if constexpr (TA) {
            kend = max(kbeg, min(kend, live));
        } else if (m0 >= live && p.split_k == 1) {
            if (!p.accumulate)
                for (int i = threadIdx.x; i < BM * BN; i += 256) {
                    const int row = m0 + i / BN, col = n0 + i % BN;
                    if (row < p.m && col < p.n) p.c[(size_t)row * p.ldc + col] = 0.f;
                }
            return;
        }
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

    bool wanted = true;
    if (p.tmask) {      // a tile of C that the caller knows to be zero: every 64 x 64 part of it is unwanted
        wanted = false;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int bit = (mt_i * MT + t) * p.tmask_ld + nt_i * NT + u;
                if ((mt_i * MT + t) * 64 < p.m && (nt_i * NT + u) * 64 < p.n) wanted = wanted || ((p.tmask[bit >> 6] >> (bit & 63)) & 1ull);
            }
    }
    // the k-chunks to walk: all of [kbeg, kend), or the ones whose bit is set in this column tile's word (NT == 1, BK == 16)
    unsigned long long todo = ~0ull;
    const bool sparse = p.kmask != nullptr;
    if (sparse) {
        todo = p.kmask[nt_i];                                  // bits count 16-wide chunks
        if (BK == 32) {                                        // a 32-deep step is walked when either of its halves is marked
            unsigned long long t = (todo | (todo >> 1)) & 0x5555555555555555ull, packed = 0ull;
            for (int i = 0; i < 32; ++i) packed |= ((t >> (2 * i)) & 1ull) << i;
            todo = packed;
        }
    }
    if (!wanted) todo = 0ull;
    auto next_chunk = [&](int from) -> int {      // first chunk >= from inside the k range that is to be walked, or kend
        if (!sparse && wanted) return from;
        const int c = from / BK;
        const unsigned long long rest = c < 64 ? (todo >> c) : 0ull;
        if (!rest) return kend;
        return (c + __builtin_ctzll(rest)) * BK;
    };

    float ra[BM * BK / 256], rb[BN * BK / 256];
    int k0 = min(next_chunk(kbeg), kend);
    if (k0 < kend) {
        load_a<TA, BM, BK>(p, m0, k0, kend, ra);
        load_b<TB, BN, BK>(p, n0, k0, kend, rb);
    }
    while (k0 < kend) {
        stage_a<TA, BM, BK>(As, ra);
        stage_b<TB, BN, BK>(Bs, rb);
        __syncthreads();
        const int k1 = min(next_chunk(k0 + BK), kend);
        if (k1 < kend) {  // next tile's loads fly under this tile's MFMAs
            load_a<TA, BM, BK>(p, m0, k1, kend, ra);
            load_b<TB, BN, BK>(p, n0, k1, kend, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float bf[NT], af[MT];
#pragma unroll
            for (int u = 0; u < NT; ++u) bf[u] = Bs[(kk + lhi) * LDB_S + wn + 32 * u + l31];
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = As[(kk + lhi) * LDA_S + wm + 32 * t + l31];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[u], acc[t][u], 0, 0, 0);
        }
        __syncthreads();
        k0 = k1;
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = n0 + wn + 32 * u + l31;
        if (col >= p.n) continue;
        const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            // accumulate: the 16 previous values of this tile column are requested TOGETHER (load, add, store per element in
            // source order makes the compiler wait for every load before the next: 16 round trips per tile)
            float old[16];
            const bool acc_old = p.accumulate && p.split_k == 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                old[r] = 0.f;
                if (acc_old && row < p.m) old[r] = p.c[(size_t)row * p.ldc + col];
            }
            // ... added in a loop of their own, and stored in a third: a store loop that reads no loaded register needs no wait
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][u][r];
                if (p.split_k == 1) {
                    v = apply_act(v + bv, p.act);
                    v = acc_old ? old[r] + v : v;
                }
                val[r] = v;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= p.m) continue;
                if (p.split_k > 1) p.ws[((size_t)blockIdx.z * p.m + row) * p.n + col] = val[r];
                else p.c[(size_t)row * p.ldc + col] = val[r];
            }
        }
    }
}

// ---- evaluation scorer with a rank-count epilogue (perturb_and_get_rank, kgvae/utils.py:180-221) ----
// S = Q (m x h) @ E^T (E: v x h), prob = sigmoid(S + *bias).  The reference materialises an (h, Eb, V) tensor, sorts every
// row and looks the target up; here the probabilities never leave the registers:
//   PASS 0  tgt[row]    = prob[row, target[row]]                      (written by the one lane that owns that element)
//   PASS 1  count[row] += #{ col != target[row] : prob[row, col] > tgt[row] }   (wave ballots, one int atomic per 32 columns)
// Both passes run the SAME k-ordered fp32 MFMA chain as k_gemm_f32<false, true, 1, 1, 16>, so tgt is bit-identical to the
// element the second pass recomputes and the count equals the one taken on a materialised score matrix.
struct RankParams {
    GemmParams g;            // a = Q, b = E (stored [v, h]), m, n = v, k = h
    const int* target;
    const float* bias;
    float* tgt;
    int* count;
};

template <int PASS>
__global__ __launch_bounds__(256) void k_rank_scores(const RankParams rp) {
    constexpr int BM = 64, BN = 64, BK = 16, LDA_S = BM + 1, LDB_S = BN + 1;
    __shared__ float As[BK * LDA_S];
    __shared__ float Bs[BK * LDB_S];
    const GemmParams& p = rp.g;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float ra[BM * BK / 256], rb[BN * BK / 256];
    load_a<false, BM, BK>(p, m0, 0, p.k, ra);
    load_b<true, BN, BK>(p, n0, 0, p.k, rb);
    for (int k0 = 0; k0 < p.k; k0 += BK) {
        stage_a<false, BM, BK>(As, ra);
        stage_b<true, BN, BK>(Bs, rb);
        __syncthreads();
        if (k0 + BK < p.k) {
            load_a<false, BM, BK>(p, m0, k0 + BK, p.k, ra);
            load_b<true, BN, BK>(p, n0, k0 + BK, p.k, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kk + lhi) * LDA_S + wm + l31], Bs[(kk + lhi) * LDB_S + wn + l31],
                                                       acc, 0, 0, 0);
        __syncthreads();
    }
    const float bv = rp.bias ? *rp.bias : 0.f;
    const int col = n0 + wn + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const bool in = row < p.m && col < p.n;
        // ranked on the LOGIT: monotone with the reference's sigmoid (kgvae/utils.py:208) but free of its saturation,
        // where every candidate ties at 1.0f.  count = 2 * #better + #equal, i.e. twice the mid-rank under ties; a
        // candidate that is NaN, or any candidate when the target itself is NaN, counts as better (never optimistic).
        const float logit = acc[r] + bv;
        const int tcol = row < p.m ? rp.target[row] : -1;
        if (PASS == 0) {
            if (in && col == tcol) rp.tgt[row] = logit;
        } else {
            const float t = row < p.m ? rp.tgt[row] : 0.f;
            const bool other = in && col != tcol;
            const bool above = other && !(logit <= t);          // greater, or either side NaN
            const bool equal = other && logit == t;
            const unsigned long long ma = __ballot(above), me = __ballot(equal);
            if (l31 == 0 && row < p.m) {
                const int c = 2 * __popc((unsigned)(lhi ? (ma >> 32) : (ma & 0xffffffffull))) +
                              __popc((unsigned)(lhi ? (me >> 32) : (me & 0xffffffffull)));
                if (c) atomicAdd(rp.count + row, c);
            }
        }
    }
}

// ---- relation-grouped products for dense per-relation weights (the `basis` regulariser) ----------------
// With W_r = sum_b w_comp[r, b] V_b a full (in x out) matrix, a message is a 2*in*out-flop mat-vec: MFMA work.  Edges are
// taken in BY-RELATION order (ops.RelationIndex.by_rel), so the messages of one relation are one GEMM whose A rows are
// GATHERED through an index (no materialised x[src] copy):
//   rows, TB = false  Msg[p]  = x[src_by_rel[p]] @ W_r           p in relation r's range       (forward)
//   rows, TB = true   Msg2[p] = g[dst_by_rel[p]] @ W_r^T                                       (backward w.r.t. x)
//   gradw             dW_r    = sum_{p in r} x[src_by_rel[p]]^T (norm_p g[dst_by_rel[p]])      (K = the relation's edge range)
// A 64-row tile never crosses a relation boundary: `tiles` lists (first row, end row, relation) per block row.  The
// per-destination sum of the messages (x norm) is the K1 aggregation with 1x1 blocks over Msg (k_bdd.hip), as in DistMult.
struct RelGemmParams {
    const float* a;          // feature table the A rows are gathered from
    const int* a_rows;       // row id per position (by-relation order)
    const float* w;          // [R, in, out] dense relation weights
    float* c;                // rows: [E, n]; gradw: [R, in, out]
    const int4* tiles;       // rows: (row0, row_end, relation, -)
    const float* b_feat;     // gradw: second gathered table (g)
    const int* b_rows;       // gradw: row id per position into b_feat
    const float* b_scale;    // gradw: per-position scale (edge norm in by-relation order), may be NULL
    const int* relptr;       // gradw: [R + 1] position ranges
    int lda, ldb_feat, in_feat, out_feat, n, k;      // rows: C is [*, n], inner dimension k
    int vec;
};

template <bool TB>
__global__ __launch_bounds__(256) void k_rel_rows(const RelGemmParams p) {
    constexpr int BM = 64, BN = 64, BK = 16, LDA_S = BM + 1, LDB_S = BN + 1, APT = 4, BPT = 4;
    __shared__ float As[BK * LDA_S];
    __shared__ float Bs[BK * LDB_S];
    const int4 tile = p.tiles[blockIdx.y];
    const int m0 = tile.x, m_end = tile.y;
    const float* wr = p.w + (size_t)tile.z * p.in_feat * p.out_feat;
    const int ldb = p.out_feat;                           // W_r is [in, out] row-major
    const int n0 = blockIdx.x * BN;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32, l31 = lane & 31, lhi = lane >> 5;
    // A: 4 threads per row (4 consecutive k each); the row pointer is resolved once
    const int am = m0 + t / 4, akk = (t % 4) * APT;
    const bool a_ok = am < m_end;
    const float* arow = p.a + (size_t)(a_ok ? p.a_rows[am] : 0) * p.lda;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float ra[APT], rb[BPT];
    auto load_tiles = [&](int k0) {
        load_run<APT>(arow + k0 + akk, k0 + akk, p.k, a_ok, p.vec, ra, 0);
        if (!TB) {          // B[k][n] = W_r[k][n]: 16 threads per k row
            const int kq = k0 + t / 16, n = n0 + (t % 16) * BPT;
            load_run<BPT>(wr + (size_t)kq * ldb + n, n, p.n, kq < p.k, p.vec, rb, 0);
        } else {            // B[k][n] = W_r[n][k]: 4 threads per n
            const int n = n0 + t / 4, kq = k0 + (t % 4) * BPT;
            load_run<BPT>(wr + (size_t)n * ldb + kq, kq, p.k, n < p.n, p.vec, rb, 0);
        }
    };
    load_tiles(0);
    for (int k0 = 0; k0 < p.k; k0 += BK) {
        stage_a<false, BM, BK>(As, ra);
        stage_b<TB, BN, BK>(Bs, rb);
        __syncthreads();
        if (k0 + BK < p.k) load_tiles(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kk + lhi) * LDA_S + wm + l31], Bs[(kk + lhi) * LDB_S + wn + l31],
                                                       acc, 0, 0, 0);
        __syncthreads();
    }
    const int col = n0 + wn + l31;
    if (col >= p.n) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < m_end) p.c[(size_t)row * p.n + col] = acc[r];
    }
}

// dW_r[i][o] = sum_p x[a_rows[p]][i] * scale_p * g[b_rows[p]][o] over the relation's positions; grid (out/64, in/64, R)
__global__ __launch_bounds__(256) void k_rel_gradw(const RelGemmParams p) {
    constexpr int BM = 64, BN = 64, BK = 16, LDA_S = BM + 1, LDB_S = BN + 1, APT = 4, BPT = 4;
    __shared__ float As[BK * LDA_S];
    __shared__ float Bs[BK * LDB_S];
    const int rel = blockIdx.z;
    const int kbeg = p.relptr[rel], kend = p.relptr[rel + 1];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32, l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float ra[APT], rb[BPT];
    auto load_tiles = [&](int k0) {      // both operands: 16 threads per position (k), 4 consecutive columns each
        const int kq = k0 + t / 16;
        const bool ok = kq < kend;
        const int m = m0 + (t % 16) * APT, n = n0 + (t % 16) * BPT;
        const float* xa = p.a + (size_t)(ok ? p.a_rows[kq] : 0) * p.lda + m;
        const float* gb = p.b_feat + (size_t)(ok ? p.b_rows[kq] : 0) * p.ldb_feat + n;
        load_run<APT>(xa, m, p.in_feat, ok, p.vec, ra, 0);
        load_run<BPT>(gb, n, p.out_feat, ok, p.vec, rb, 0);
        if (p.b_scale && ok) {
            const float sc = p.b_scale[kq];
#pragma unroll
            for (int i = 0; i < BPT; ++i) rb[i] *= sc;
        }
    };
    if (kbeg < kend) load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stage_a<true, BM, BK>(As, ra);
        stage_b<false, BN, BK>(Bs, rb);
        __syncthreads();
        if (k0 + BK < kend) load_tiles(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kk + lhi) * LDA_S + wm + l31], Bs[(kk + lhi) * LDB_S + wn + l31],
                                                       acc, 0, 0, 0);
        __syncthreads();
    }
    const int col = n0 + wn + l31;
    if (col >= p.out_feat) return;
    float* cr = p.c + (size_t)rel * p.in_feat * p.out_feat;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < p.in_feat) cr[(size_t)row * p.out_feat + col] = acc[r];
    }
}

// ---- bf16-operand variant (BASELINE configs[2]: bf16 with fp32 accumulation) --------------------------------------
// Same contract and epilogue as k_gemm_f32; A and B are read as fp32 from memory, rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulators:
// a bf16 x bf16 product is exact in fp32, so the result equals an fp32 GEMM of the rounded operands up to summation
// order.  Block tile 64 x 64 x 32, four waves 2 x 2, one 32x32 accumulator per wave, two MFMAs per tile.
// LDS holds both tiles k-contiguous, [row][32 k + 8 pad] bf16: the operand fragment of lane l (row l&31, k = 8*(l>>5)
// .. +7) is ONE 16-byte ds_read_b128, conflict-free at the 80-byte row pitch.  Staging: an operand whose k runs
// along memory (A [M,K]; B stored [N,K]) is read as 2 x 16 B per thread; one whose k is the row index (A stored [K,M];
// B [K,N]) as 8 coalesced 4-B loads (64 consecutive rows per wave instruction) -- either way 8 k-values of one row per
// thread, packed into one 16-byte LDS store.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int HB_BK = 32, HB_LD = 40;      // bf16 elements per LDS row (80 B)

// KC: element (row, k) at base[row * ld + k]
__device__ __forceinline__ void hb_load_kc(const float* base, const float* mask, int ld, int row0, int rows, int k0, int kend,
                                           bool vec, float (&r)[8]) {
    const int t = threadIdx.x;
    const int row = row0 + (t >> 2), kk = k0 + (t & 3) * 8;
    const bool ok = row < rows;
    const float* src = base + (size_t)row * ld + kk;
    load_run<8>(src, kk, kend, ok, vec, r, 0);
    if (mask) mask_run<8>(mask + (size_t)row * ld + kk, kk, kend, ok, vec, r);
}

// KS: element (row, k) at base[k * ld + row]
__device__ __forceinline__ void hb_load_ks(const float* base, const float* mask, int ld, int row0, int rows, int k0, int kend,
                                           float (&r)[8]) {
    const int t = threadIdx.x;
    const int row = row0 + (t & 63), kk = k0 + (t >> 6) * 8;
    const bool ok = row < rows;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool in = ok && kk + j < kend;
        float v = in ? base[(size_t)(kk + j) * ld + row] : 0.f;
        if (mask && in) v = mask[(size_t)(kk + j) * ld + row] > 0.f ? v : 0.f;
        r[j] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void hb_stage(__bf16* S, const float (&r)[8]) {
    const int t = threadIdx.x;
    const int row = KC ? (t >> 2) : (t & 63), kk = KC ? (t & 3) * 8 : (t >> 6) * 8;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)r[j];
    *reinterpret_cast<bf16x8*>(S + row * HB_LD + kk) = v;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_bf16(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 As[64 * HB_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[64 * HB_LD];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * p.k_chunk;
    const int kend = min(p.k, kbeg + p.k_chunk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    float ra[8], rb[8];
    auto load_tiles = [&](int k0) {
        if constexpr (!TA) hb_load_kc(p.a, p.a_mask, p.lda, m0, p.m, k0, kend, p.vec_a, ra);
        else hb_load_ks(p.a, p.a_mask, p.lda, m0, p.m, k0, kend, ra);
        if constexpr (TB) hb_load_kc(p.b, nullptr, p.ldb, n0, p.n, k0, kend, p.vec_b, rb);
        else hb_load_ks(p.b, nullptr, p.ldb, n0, p.n, k0, kend, rb);
    };
    load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += HB_BK) {
        hb_stage<!TA>(As, ra);
        hb_stage<TB>(Bs, rb);
        __syncthreads();
        if (k0 + HB_BK < kend) load_tiles(k0 + HB_BK);     // next tile's loads fly under this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < HB_BK; kk += 16) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(As + (wm + l31) * HB_LD + kk + 8 * lhi);
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Bs + (wn + l31) * HB_LD + kk + 8 * lhi);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn + l31;
    if (col >= p.n) return;
    const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row >= p.m) continue;
        float v = acc[r];
        if (p.split_k > 1) {
            p.ws[((size_t)blockIdx.z * p.m + row) * p.n + col] = v;
        } else {
            v = apply_act(v + bv, p.act);
            float* dst = p.c + (size_t)row * p.ldc + col;
            *dst = p.accumulate ? *dst + v : v;
        }
    }
}

__global__ __launch_bounds__(256) void k_gemm_splitk_reduce(const GemmParams p) {
    const size_t total = (size_t)p.m * p.n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // 8 partial tiles in flight (4 chains, fixed combine: the order does not depend on the launch geometry)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int z = 0;
        for (; z + 8 <= p.split_k; z += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = p.ws[(size_t)(z + q) * total + i];
            a0 += t[0]; a1 += t[1]; a2 += t[2]; a3 += t[3];
            a0 += t[4]; a1 += t[5]; a2 += t[6]; a3 += t[7];
        }
        for (; z < p.split_k; ++z) a0 += p.ws[(size_t)z * total + i];
        float v = (a0 + a1) + (a2 + a3);
        const int row = (int)(i / p.n), col = (int)(i - (size_t)row * p.n);
        if (p.tmask_wanted > 0) {      // no block wrote the partial tiles of an unwanted tile
            const int bit = (row >> 6) * p.tmask_ld + (col >> 6);
            if (!((p.tmask[bit >> 6] >> (bit & 63)) & 1ull)) v = 0.f;
        }
        if (p.bias) v += p.bias[col];
        v = apply_act(v, p.act);
        float* dst = p.c + (size_t)row * p.ldc + col;
        *dst = p.accumulate ? *dst + v : v;
    }
}

// column sums: grid (column tiles of 64, row slices); the 4 waves of a block interleave the slice's rows
// (4 loads in flight per lane), combine through LDS in wave order; a second kernel sums the slices in order.
constexpr int COLSUM_SLICES = 64;

__global__ __launch_bounds__(256) void k_colsum_part(const float* x, const float* msk, int64_t m, int n, int ld, float* part) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t per = (m + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = blockIdx.y * per, r1 = min(m, r0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < n) {
        int64_t r = r0 + w;
        if (msk) {
            for (; r + 12 < r1; r += 16) {
                const float v0 = x[r * ld + c], v1 = x[(r + 4) * ld + c], v2 = x[(r + 8) * ld + c], v3 = x[(r + 12) * ld + c];
                const float m0 = msk[r * ld + c], m1 = msk[(r + 4) * ld + c], m2 = msk[(r + 8) * ld + c], m3 = msk[(r + 12) * ld + c];
                a0 += m0 > 0.f ? v0 : 0.f; a1 += m1 > 0.f ? v1 : 0.f; a2 += m2 > 0.f ? v2 : 0.f; a3 += m3 > 0.f ? v3 : 0.f;
            }
            for (; r < r1; r += 4) a0 += msk[r * ld + c] > 0.f ? x[r * ld + c] : 0.f;
        } else {
            for (; r + 12 < r1; r += 16) {
                const float v0 = x[r * ld + c], v1 = x[(r + 4) * ld + c], v2 = x[(r + 8) * ld + c], v3 = x[(r + 12) * ld + c];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; r < r1; r += 4) a0 += x[r * ld + c];
        }
    }
    sm[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && c < n) part[(size_t)blockIdx.y * n + c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

__global__ __launch_bounds__(256) void k_colsum_final(const float* part, int n, int nsl, float* out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 4 <= nsl; s += 4) {
        const float v0 = part[(size_t)s * n + c], v1 = part[(size_t)(s + 1) * n + c];
        const float v2 = part[(size_t)(s + 2) * n + c], v3 = part[(size_t)(s + 3) * n + c];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; s < nsl; ++s) a0 += part[(size_t)s * n + c];
    const float acc = (a0 + a1) + (a2 + a3);
    out[c] = accumulate ? out[c] + acc : acc;
}

// final pass over many slices: 16 columns per 1024-thread block (common.h: sum_slices_16x64)
__global__ __launch_bounds__(1024) void k_colsum_final_wide(const float* part, int n, int nsl, float* out, int accumulate) {
    __shared__ float sm[64][16];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const float acc = sum_slices_16x64(part, n, nsl, c, sm);
    if ((threadIdx.x >> 4) == 0 && c < n) out[c] = accumulate ? out[c] + acc : acc;
}

}  // namespace gv

using namespace gv;

static int k_chunk_for(int k, int split_k) {
    int per = (k + split_k - 1) / split_k;
    return ((per + 31) / 32) * 32;
}

extern "C" int64_t gv_gemm_workspace_bytes(int m, int n, int k, int split_k) {
    (void)k;
    return split_k > 1 ? (int64_t)split_k * m * n * (int64_t)sizeof(float) : 0;
}

static int gemm_any(bool bf16, int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b,
                    int ldb, float* c, int ldc, const float* bias, int act, int accumulate, int split_k,
                    const float* a_relu_mask, void* workspace, int64_t workspace_bytes, void* stream, const int32_t* rows_dev = nullptr,
                    const uint64_t* kmask = nullptr, const uint64_t* tmask = nullptr, int tiles_wanted = 0) {
    GV_REQUIRE(m >= 0 && n >= 0 && k >= 0, GV_ERR_SHAPE, "gv_gemm_f32: negative size");
    if (m == 0 || n == 0) return GV_OK;
    GV_REQUIRE(a && b && c, GV_ERR_NULL, "gv_gemm_f32: NULL matrix");
    GV_REQUIRE(lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n) && ldc >= n, GV_ERR_SHAPE,
               "gv_gemm_f32: leading dimension too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    GV_REQUIRE(act == GV_ACT_NONE || act == GV_ACT_RELU, GV_ERR_SHAPE, "gv_gemm_f32: unknown act %d", act);
    if (split_k < 1) split_k = 1;
    if (k == 0) split_k = 1;
    GemmParams p;
    p.a = a; p.b = b; p.c = c; p.bias = bias; p.a_mask = a_relu_mask; p.ws = (float*)workspace;
    p.m = m; p.n = n; p.k = k; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate;
    p.k_chunk = k_chunk_for(k > 0 ? k : 1, split_k);
    split_k = k > 0 ? (k + p.k_chunk - 1) / p.k_chunk : 1;
    p.split_k = split_k;
    p.vec_a = aligned16(a) && (lda % 4 == 0) && (!a_relu_mask || aligned16(a_relu_mask));
    p.vec_b = aligned16(b) && (ldb % 4 == 0);
    p.rows_dev = bf16 ? nullptr : rows_dev;
    GV_REQUIRE(!kmask || (!bf16 && split_k == 1 && k <= 64 * 16), GV_ERR_SHAPE,
               "gv_gemm_f32_sparse: k-chunk words cover fp32 products of k <= 1024 without split-K (k=%d split_k=%d)", k, split_k);
    GV_REQUIRE(!tmask || !bf16, GV_ERR_SHAPE, "gv_gemm_f32_sparse: fp32 only");
    p.kmask = (const unsigned long long*)kmask;
    p.tmask = (const unsigned long long*)tmask;
    p.tmask_ld = (n + 63) / 64;
    GV_REQUIRE(tiles_wanted >= 0 && tiles_wanted <= ((m + 63) / 64) * ((n + 63) / 64), GV_ERR_SHAPE, "gv_gemm_f32_sparse: c_tiles_wanted=%d", tiles_wanted);
    // one block row per WANTED tile (and nothing for the others) where the partial tiles go through the split-K sum, which knows the words too
    p.tmask_wanted = (tmask && tiles_wanted > 0 && split_k > 1) ? tiles_wanted : 0;
    if (split_k > 1) {
        GV_REQUIRE(workspace, GV_ERR_NULL, "gv_gemm_f32: split_k needs a workspace");
        GV_REQUIRE(workspace_bytes >= gv_gemm_workspace_bytes(m, n, k, split_k), GV_ERR_WORKSPACE,
                   "gv_gemm_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                   (long long)gv_gemm_workspace_bytes(m, n, k, split_k));
    }
    hipStream_t st = (hipStream_t)stream;
    if (bf16) {
        dim3 grid((n + 63) / 64, (m + 63) / 64, split_k), block(256);
        if (!trans_a && !trans_b) hipLaunchKernelGGL((k_gemm_bf16<false, false>), grid, block, 0, st, p);
        else if (!trans_a && trans_b) hipLaunchKernelGGL((k_gemm_bf16<false, true>), grid, block, 0, st, p);
        else if (trans_a && !trans_b) hipLaunchKernelGGL((k_gemm_bf16<true, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((k_gemm_bf16<true, true>), grid, block, 0, st, p);
        int rcb = launch_status("gv_gemm_bf16");
        if (rcb != GV_OK) return rcb;
        if (split_k > 1) {
            const size_t total = (size_t)m * n;
            const int blocks = (int)min((size_t)2048, (total + 255) / 256);
            hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
            return launch_status("gv_gemm_bf16(split-k reduce)");
        }
        return GV_OK;
    }
    // Tile choice (measured on the C2 shapes, tools/microbench.py): 64-row tiles unless 128-row tiles still give every
    // CU >= 4 blocks; 64-column tiles; BK = 16 for the row-major-A
    // products (K = 200..400), 32 for the split-K weight-gradient products (A stored [K, M], K = nodes).
    static const int mt_env = getenv("GV_GEMM_MT") ? atoi(getenv("GV_GEMM_MT")) : 0;   // tuning knobs
    static const int nt_env = getenv("GV_GEMM_NT") ? atoi(getenv("GV_GEMM_NT")) : 0;
    static const int bk_env = getenv("GV_GEMM_BK") ? atoi(getenv("GV_GEMM_BK")) : 0;
    const int nt = (nt_env == 2 && !kmask && !(tmask && tiles_wanted > 0 && split_k > 1)) ? 2 : 1;        // 128-column tiles measured 8-12 % slower on every C2 shape: opt-in only
    const int bn = 64 * nt;
    const long blocks128 = (long)((n + bn - 1) / bn) * ((m + 127) / 128) * split_k;
    const int mt = p.tmask_wanted ? 1 : (mt_env ? mt_env : (blocks128 >= 1024 ? 2 : 1));
    // with k-chunk words: 16-deep steps = exactly the marked chunks (GV_GEMM_SPARSE_BK=32: 32-deep steps walked when either half is
    // marked -- measured the same, 66 us on the 500 x 500 MADE layer against 92 dense)
    static const int sparse_bk = getenv("GV_GEMM_SPARSE_BK") ? atoi(getenv("GV_GEMM_SPARSE_BK")) : 16;
    const int bk = kmask ? (sparse_bk == 32 ? 32 : 16) : (bk_env ? (bk_env == 16 ? 16 : 32) : (trans_a ? 32 : 16));
    dim3 grid((n + bn - 1) / bn, (m + 64 * mt - 1) / (64 * mt), split_k), block(256);
    if (p.tmask_wanted) { grid.x = p.tmask_wanted; grid.y = 1; }
#define GV_GEMM_CFG(TA_, TB_, MT_, NT_, BK_) \
    if (mt == MT_ && nt == NT_ && bk == BK_) hipLaunchKernelGGL((k_gemm_f32<TA_, TB_, MT_, NT_, BK_>), grid, block, 0, st, p);
#define GV_GEMM_LAUNCH(TA_, TB_)                                                                       \
    do {                                                                                               \
        GV_GEMM_CFG(TA_, TB_, 1, 1, 16) GV_GEMM_CFG(TA_, TB_, 1, 1, 32) GV_GEMM_CFG(TA_, TB_, 1, 2, 16) \
        GV_GEMM_CFG(TA_, TB_, 1, 2, 32) GV_GEMM_CFG(TA_, TB_, 2, 1, 16) GV_GEMM_CFG(TA_, TB_, 2, 1, 32) \
        GV_GEMM_CFG(TA_, TB_, 2, 2, 16) GV_GEMM_CFG(TA_, TB_, 2, 2, 32)                                 \
    } while (0)
    if (!trans_a && !trans_b) GV_GEMM_LAUNCH(false, false);
    else if (!trans_a && trans_b) GV_GEMM_LAUNCH(false, true);
    else if (trans_a && !trans_b) GV_GEMM_LAUNCH(true, false);
    else GV_GEMM_LAUNCH(true, true);
#undef GV_GEMM_LAUNCH
#undef GV_GEMM_CFG
    int rc = launch_status("gv_gemm_f32");
    if (rc != GV_OK) return rc;
    if (split_k > 1) {
        const size_t total = (size_t)m * n;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
        return launch_status("gv_gemm_f32(split-k reduce)");
    }
    return GV_OK;
}

extern "C" int gv_gemm_f32(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b,
                           int ldb, float* c, int ldc, const float* bias, int act, int accumulate, int split_k,
                           const float* a_relu_mask, void* workspace, int64_t workspace_bytes, void* stream) {
    return gemm_any(false, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, act, accumulate, split_k, a_relu_mask,
                    workspace, workspace_bytes, stream);
}

extern "C" int gv_gemm_f32_live_rows(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b,
                                     int ldb, float* c, int ldc, const float* bias, int act, int accumulate, int split_k,
                                     const float* a_relu_mask, void* workspace, int64_t workspace_bytes, const int32_t* rows_dev,
                                     void* stream) {
    return gemm_any(false, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, act, accumulate, split_k, a_relu_mask,
                    workspace, workspace_bytes, stream, rows_dev);
}

extern "C" int gv_gemm_f32_sparse(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                                  float* c, int ldc, const float* bias, int act, int accumulate, int split_k, const float* a_relu_mask,
                                  void* workspace, int64_t workspace_bytes, const int32_t* rows_dev, const uint64_t* b_k_chunks,
                                  const uint64_t* c_tiles, int c_tiles_wanted, void* stream) {
    return gemm_any(false, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, act, accumulate, split_k, a_relu_mask,
                    workspace, workspace_bytes, stream, rows_dev, b_k_chunks, c_tiles, c_tiles_wanted);
}

extern "C" int gv_gemm_bf16(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b,
                            int ldb, float* c, int ldc, const float* bias, int act, int accumulate, int split_k,
                            const float* a_relu_mask, void* workspace, int64_t workspace_bytes, void* stream) {
    return gemm_any(true, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, act, accumulate, split_k, a_relu_mask,
                    workspace, workspace_bytes, stream);
}

extern "C" int gv_rank_scores(const float* q, int ld_q, const float* e, int ld_e, const int* target, const float* bias,
                              float* tgt, int* count, int m, int v, int h, void* stream) {
    GV_REQUIRE(m >= 0 && v > 0 && h > 0, GV_ERR_SHAPE, "gv_rank_scores: m=%d v=%d h=%d", m, v, h);
    if (m == 0) return GV_OK;
    GV_REQUIRE(q && e && target && tgt && count, GV_ERR_NULL, "gv_rank_scores: NULL pointer");
    GV_REQUIRE(ld_q >= h && ld_e >= h, GV_ERR_SHAPE, "gv_rank_scores: leading dimension too small");
    RankParams rp;
    GemmParams& p = rp.g;
    p.a = q; p.b = e; p.c = nullptr; p.bias = nullptr; p.a_mask = nullptr; p.ws = nullptr;
    p.m = m; p.n = v; p.k = h; p.lda = ld_q; p.ldb = ld_e; p.ldc = v;
    p.act = GV_ACT_NONE; p.accumulate = 0; p.split_k = 1; p.k_chunk = h;
    p.vec_a = aligned16(q) && (ld_q % 4 == 0);
    p.vec_b = aligned16(e) && (ld_e % 4 == 0);
    p.rows_dev = nullptr; p.kmask = nullptr; p.tmask = nullptr; p.tmask_ld = 0; p.tmask_wanted = 0;
    rp.target = target; rp.bias = bias; rp.tgt = tgt; rp.count = count;
    hipStream_t st = (hipStream_t)stream;
    if (fill_words(count, 0u, (size_t)m * sizeof(int), st) != hipSuccess) return launch_status("gv_rank_scores(fill)");
    dim3 grid((v + 63) / 64, (m + 63) / 64), block(256);
    hipLaunchKernelGGL(k_rank_scores<0>, grid, block, 0, st, rp);
    hipLaunchKernelGGL(k_rank_scores<1>, grid, block, 0, st, rp);
    return launch_status("gv_rank_scores");
}

extern "C" int gv_rel_rows_gemm(const float* feat, int ld_feat, const int32_t* rows, const float* w, int num_rels, int in_feat,
                                int out_feat, int transpose_w, const int32_t* tiles, int n_tiles, float* msg, void* stream) {
    GV_REQUIRE(num_rels > 0 && in_feat > 0 && out_feat > 0 && n_tiles >= 0, GV_ERR_SHAPE, "gv_rel_rows_gemm: bad sizes");
    if (n_tiles == 0) return GV_OK;
    GV_REQUIRE(feat && rows && w && tiles && msg, GV_ERR_NULL, "gv_rel_rows_gemm: NULL pointer");
    const int k = transpose_w ? out_feat : in_feat, n = transpose_w ? in_feat : out_feat;
    GV_REQUIRE(ld_feat >= k, GV_ERR_SHAPE, "gv_rel_rows_gemm: ld_feat=%d < %d", ld_feat, k);
    RelGemmParams p{};
    p.a = feat; p.a_rows = rows; p.w = w; p.c = msg; p.tiles = (const int4*)tiles;
    p.lda = ld_feat; p.in_feat = in_feat; p.out_feat = out_feat; p.n = n; p.k = k;
    p.vec = aligned16(feat) && aligned16(w) && (ld_feat % 4 == 0) && (in_feat % 4 == 0) && (out_feat % 4 == 0);
    dim3 grid((n + 63) / 64, n_tiles), block(256);
    if (transpose_w) hipLaunchKernelGGL(k_rel_rows<true>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_rel_rows<false>, grid, block, 0, (hipStream_t)stream, p);
    return launch_status("gv_rel_rows_gemm");
}

extern "C" int gv_rel_gradw_gemm(const float* x, int ld_x, const int32_t* x_rows, const float* g, int ld_g, const int32_t* g_rows,
                                 const float* scale, const int32_t* relptr, int num_rels, int in_feat, int out_feat,
                                 float* grad_w, void* stream) {
    GV_REQUIRE(num_rels > 0 && in_feat > 0 && out_feat > 0, GV_ERR_SHAPE, "gv_rel_gradw_gemm: bad sizes");
    GV_REQUIRE(x && x_rows && g && g_rows && relptr && grad_w, GV_ERR_NULL, "gv_rel_gradw_gemm: NULL pointer");
    GV_REQUIRE(ld_x >= in_feat && ld_g >= out_feat, GV_ERR_SHAPE, "gv_rel_gradw_gemm: leading dimension too small");
    GV_REQUIRE(num_rels <= 65535, GV_ERR_SHAPE, "gv_rel_gradw_gemm: more than 65535 relation types");
    RelGemmParams p{};
    p.a = x; p.a_rows = x_rows; p.lda = ld_x; p.b_feat = g; p.b_rows = g_rows; p.ldb_feat = ld_g; p.b_scale = scale;
    p.relptr = relptr; p.c = grad_w; p.in_feat = in_feat; p.out_feat = out_feat;
    p.vec = aligned16(x) && aligned16(g) && (ld_x % 4 == 0) && (ld_g % 4 == 0);
    dim3 grid((out_feat + 63) / 64, (in_feat + 63) / 64, num_rels), block(256);
    hipLaunchKernelGGL(k_rel_gradw, grid, block, 0, (hipStream_t)stream, p);
    return launch_status("gv_rel_gradw_gemm");
}

extern "C" int gv_colsum_finish(const float* part, int n, int n_slices, float* out, int accumulate, void* stream) {
    GV_REQUIRE(part && out, GV_ERR_NULL, "gv_colsum_finish: NULL pointer");
    GV_REQUIRE(n > 0 && n_slices > 0, GV_ERR_SHAPE, "gv_colsum_finish: n=%d n_slices=%d", n, n_slices);
    hipLaunchKernelGGL(k_colsum_final_wide, dim3((n + 15) / 16), dim3(1024), 0, (hipStream_t)stream, part, n, n_slices, out,
                       accumulate);
    return launch_status("gv_colsum_finish");
}

extern "C" int gv_colsum(const float* x, const float* relu_mask, int64_t m, int n, int ld, float* out, float* workspace,
                         int accumulate, void* stream) {
    GV_REQUIRE(x && out && workspace, GV_ERR_NULL, "gv_colsum: NULL pointer");
    GV_REQUIRE(m >= 0 && n > 0 && ld >= n, GV_ERR_SHAPE, "gv_colsum: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int nsl = COLSUM_SLICES;     // workspace is sized for 64 slices
    hipLaunchKernelGGL(k_colsum_part, dim3((n + 63) / 64, nsl), dim3(256), 0, st, x, relu_mask, m, n, ld, workspace);
    hipLaunchKernelGGL(k_colsum_final, dim3((n + 63) / 64), dim3(64), 0, st, workspace, n, nsl, out, accumulate);
    return launch_status("gv_colsum");
}
