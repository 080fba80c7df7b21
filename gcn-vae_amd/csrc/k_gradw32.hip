// K4, fp32: the weight gradient of one masked-MLP layer over ALL stacked passes of a MADE (kgvae/flow_network.py:13-14, 85-98 under
// autograd):   dW[j][i] (+)= wmask[j][i] * ( sum_k G[k][j] A[k][i] + g0[j] a0[i] ),   db[j] (+)= sum_k G[k][j] + g0[j]
// with G [K][ldg] the (ReLU-masked) gradient w.r.t. the layer's output, A [K][lda] the layer's input, K = passes x nodes (50-200 k),
// (g0, a0) pass 0's single broadcast row, wmask the layer's 0/1 autoregressive mask.
//
// As launches of the generic GEMM this was, per layer: a 64 x 64-tile split-K product at 0.27 of the fp32 MFMA peak (L2-bound: 16
// flop per byte staged) + its split sum + a rank-1 product + an add + two column-sum launches + a share of the mask multiply:
// ~190 us, 2.9 ms of the 7.9 ms mini-batch step with 3 IAF blocks.  Here ONE workgroup owns the WHOLE (up to 224 x 224) output for
// its slice of K -- every operand element is staged once (36 flop per byte) -- and only the 32 x 32 output tiles in which the mask
// holds a non-zero are computed (create_masks: lower-triangular => 28-34 of 49); the bias gradient comes from the staged G chunk;
// a second launch sums the slices in a fixed order, adds pass 0's rank-1 term, applies the mask and stores or accumulates.
#include "common.h"

namespace gv {

typedef float f32x16g __attribute__((ext_vector_type(16)));

constexpr int GW_THREADS = 512, GW_WAVES = 8;
constexpr int GW_KC = 32;            // rows of K per staged chunk
constexpr int GW_BT = 7;             // 32-wide tiles per block side (224 columns)
constexpr int GW_LD = GW_BT * 32;    // LDS row pitch (floats): a half-wave reads 32 consecutive floats of one row
constexpr int GW_TPW = 7;            // tile slots per wave (49 tiles / 8 waves)
constexpr int GW_PIECES = GW_KC * GW_LD / 4 / GW_THREADS;      // 16-B pieces per thread and operand = 3.5 -> 4
static_assert(GW_KC * GW_LD / 4 <= 4 * GW_THREADS, "staging: four pieces per thread and operand");

struct GradW32Args {
    const float* g; const float* a;
    const int32_t* plan;         // [ceil(m / 32)][ceil(n / 32)] 1 = the tile holds a non-zero of the mask
    float* part;                 // [slices][mp][np]
    float* dbpart;               // [slices][mp] or NULL
    int ldg, lda, m, n, mp, np;
    int slices, gy, gz, reserved;   // the product's share of the launch: slices x gy x gz workgroups (slice fastest)
    long long k, k_per_slice;
};

// up to GW_MAXI products in ONE launch (the layers of a MADE): workgroups [first[q], first[q + 1]) belong to product q
constexpr int GW_MAXI = 8;
struct GradW32Multi {
    int count, first[GW_MAXI + 1];
    GradW32Args it[GW_MAXI];
};

__global__ __launch_bounds__(GW_THREADS) void k_gradw32(const GradW32Multi P) {
    int q_ = 0;
    while (q_ + 1 < P.count && (int)blockIdx.x >= P.first[q_ + 1]) ++q_;
    const GradW32Args& p = P.it[q_];
    const int local_ = (int)blockIdx.x - P.first[q_];
    const int bx = local_ % p.slices, by = (local_ / p.slices) % p.gy, bz = local_ / (p.slices * p.gy);
    extern __shared__ __attribute__((aligned(16))) float gw_lds[];
    float* const Gs = gw_lds;                               // [2][KC][LD]
    float* const As = gw_lds + 2 * GW_KC * GW_LD;           // [2][KC][LD]
    __shared__ int tiles[GW_BT * GW_BT];
    __shared__ int ntiles;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int jt0 = by * GW_BT, it0 = bz * GW_BT;           // first row / column tile of this block
    const int mt_all = (p.m + 31) >> 5, nt_all = (p.n + 31) >> 5;
    const int jts = min(GW_BT, mt_all - jt0), its = min(GW_BT, nt_all - it0);
    if (threadIdx.x < 64) {        // the block's active tiles, compacted in (j, i) order: one plan word per lane, ONE round trip
        const int t = threadIdx.x, j = t / GW_BT, i = t - j * GW_BT;
        const bool in = t < GW_BT * GW_BT && j < jts && i < its;
        const bool on = in && p.plan[in ? (jt0 + j) * nt_all + it0 + i : 0] != 0;
        const unsigned long long live = __ballot(on);
        if (on) tiles[__popcll(live & ((1ull << t) - 1ull))] = (j << 8) | i;
        if (t == 0) ntiles = __popcll(live);
    }
    __syncthreads();
    const int cnt = ntiles;
    // the block's active tiles dealt to the eight waves as evenly as they go (waves w and w + 4 share a SIMD)
    const int base = cnt / GW_WAVES, extra = cnt % GW_WAVES;
    const int first = wave * base + min(wave, extra), mine = base + (wave < extra ? 1 : 0);
    int tj[GW_TPW], ti[GW_TPW];
#pragma unroll
    for (int t = 0; t < GW_TPW; ++t) {
        const int e = tiles[min(first + min(t, max(mine - 1, 0)), max(cnt - 1, 0))];       // slots past the wave's last repeat it (computed, not stored)
        tj[t] = __builtin_amdgcn_readfirstlane(cnt ? (e >> 8) * 32 : 0);
        ti[t] = __builtin_amdgcn_readfirstlane(cnt ? (e & 0xff) * 32 : 0);
    }
    f32x16g acc[GW_TPW];
#pragma unroll
    for (int t = 0; t < GW_TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const long long k0 = (long long)bx * p.k_per_slice, k1 = min(p.k, k0 + p.k_per_slice);
    const int j0 = jt0 * 32, i0 = it0 * 32;
    // ---- staging: chunk rows [kc, kc + KC) x the block's columns, 16-B pieces along the rows, zero outside the operands ----
    float4 rg[4], ra[4];
    auto load_chunk = [&](long long kc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = (int)threadIdx.x + q * GW_THREADS;           // piece = (row, 4 columns)
            const int row = pc / (GW_LD / 4), c4 = (pc - row * (GW_LD / 4)) * 4;
            rg[q] = ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < GW_KC && kc + row < k1) {
                if (j0 + c4 < p.m) rg[q] = *reinterpret_cast<const float4*>(p.g + (size_t)(kc + row) * p.ldg + j0 + c4);      // (m, n % 4 == 0)
                if (i0 + c4 < p.n) ra[q] = *reinterpret_cast<const float4*>(p.a + (size_t)(kc + row) * p.lda + i0 + c4);
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = (int)threadIdx.x + q * GW_THREADS;
            if (pc < GW_KC * GW_LD / 4) {
                *reinterpret_cast<float4*>(Gs + buf * GW_KC * GW_LD + pc * 4) = rg[q];
                *reinterpret_cast<float4*>(As + buf * GW_KC * GW_LD + pc * 4) = ra[q];
            }
        }
    };
    float dbsum = 0.f;          // threads 0 .. 223 of the blocks with bz == 0: column j0 + threadIdx.x of G
    const bool do_db = p.dbpart && bz == 0 && (int)threadIdx.x < GW_LD;
    load_chunk(k0);
    int buf = 0;
    for (long long kc = k0; kc < k1; kc += GW_KC) {
        store_chunk(buf);
        __syncthreads();
        if (kc + GW_KC < k1) load_chunk(kc + GW_KC);           // the next chunk's loads fly under this chunk's MFMAs
        const float* G = Gs + buf * GW_KC * GW_LD + lhi * GW_LD + l31;
        const float* A = As + buf * GW_KC * GW_LD + lhi * GW_LD + l31;
        // two slots at a time: two independent accumulator chains per wave (and two waves per SIMD); a wave's odd last slot runs
        // as ONE chain (the SIMD's other wave fills the pipe) -- not as a repeated second tile that another MFMA per step pays for
#define GW_PAIR(T0, T1)                                                                                         \
        if (T1 < GW_TPW && T1 < mine) {                                                                         \
            _Pragma("unroll") for (int s = 0; s < GW_KC / 2; ++s) {                                             \
                const float g0_ = G[2 * s * GW_LD + tj[T0]], a0_ = A[2 * s * GW_LD + ti[T0]];                   \
                const float g1_ = G[2 * s * GW_LD + tj[T1 < GW_TPW ? T1 : T0]], a1_ = A[2 * s * GW_LD + ti[T1 < GW_TPW ? T1 : T0]]; \
                acc[T0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0_, a0_, acc[T0], 0, 0, 0);                     \
                acc[T1 < GW_TPW ? T1 : T0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1_, a1_, acc[T1 < GW_TPW ? T1 : T0], 0, 0, 0); \
            }                                                                                                   \
        } else if (T0 < mine) {                                                                                 \
            _Pragma("unroll") for (int s = 0; s < GW_KC / 2; ++s) {                                             \
                const float g0_ = G[2 * s * GW_LD + tj[T0]], a0_ = A[2 * s * GW_LD + ti[T0]];                   \
                acc[T0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0_, a0_, acc[T0], 0, 0, 0);                     \
            }                                                                                                   \
        }
        GW_PAIR(0, 1)
        GW_PAIR(2, 3)
        GW_PAIR(4, 5)
        GW_PAIR(6, 7)
#undef GW_PAIR
        if (do_db) {           // the bias gradient's share of this chunk: rows in order, one column per thread
            const float* col = Gs + buf * GW_KC * GW_LD + threadIdx.x;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < GW_KC; r += 2) { s0 += col[r * GW_LD]; s1 += col[(r + 1) * GW_LD]; }
            dbsum += s0 + s1;
        }
        buf ^= 1;
    }
    // ---- the slice's partial tiles (only the wave's own slots), the bias partial ----
    float* const part = p.part + (size_t)bx * p.mp * p.np;
#pragma unroll
    for (int t = 0; t < GW_TPW; ++t)
        if (t < mine) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = j0 + tj[t] + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                part[(size_t)row * p.np + i0 + ti[t] + l31] = acc[t][r];
            }
        }
    if (do_db && j0 + (int)threadIdx.x < p.mp) p.dbpart[(size_t)bx * p.mp + j0 + threadIdx.x] = dbsum;
}

struct GradW32Reduce {
    const float* part; const float* dbpart; const int32_t* plan;
    const float* wmask; const float* g0; const float* g0_act; const float* a0;
    float* out; float* db;
    int m, n, mp, np, ldw, ldo, slices, accumulate, db_accumulate, blocks;     // blocks: the product's workgroups in the launch
};
struct GradW32ReduceMulti {
    int count, first[GW_MAXI + 1];
    GradW32Reduce it[GW_MAXI];
};

// out[j][i] (+)= wmask * (ordered sum of the slices' partials + g0m[j] a0[i]); db[j] (+)= ordered sum + g0m[j];
// g0m[j] = g0_act == NULL || g0_act[j] > 0 ? g0[j] : 0 (pass 0's row gradient behind its ReLU mask).
// A workgroup owns 64 consecutive entries (of the m x n outputs, then of the m bias entries); its four waves each sum a quarter of
// the slices in order (8 partials in flight, 4 chains, fixed combine), the quarters are added in order through LDS: 35 MB of
// partials per 200 x 200 product are read by ~630 workgroups instead of 157 threads-in-series (27 -> ~9 us).
__global__ __launch_bounds__(256) void k_gradw32_reduce(const GradW32ReduceMulti P) {
    int q_ = 0;
    while (q_ + 1 < P.count && (int)blockIdx.x >= P.first[q_ + 1]) ++q_;
    const GradW32Reduce& p = P.it[q_];
    const int bx = (int)blockIdx.x - P.first[q_];
    __shared__ float sm[4][64];
    const int nt_all = (p.n + 31) >> 5;
    const size_t total = (size_t)p.m * p.n, slice = (size_t)p.mp * p.np;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int per = (p.slices + 3) >> 2, z0 = q * per, z1 = min(p.slices, z0 + per);
    const size_t nblk_w = (total + 63) / 64;
    for (size_t blk = bx; blk < nblk_w + (size_t)(p.db ? (p.m + 63) / 64 : 0); blk += p.blocks) {
        const bool is_b = blk >= nblk_w;
        const size_t e = (is_b ? blk - nblk_w : blk) * 64 + lane;
        const bool in = is_b ? e < (size_t)p.m : e < total;
        int j = 0, i = 0;
        const float* src = nullptr;
        size_t step = 0;
        if (in && is_b) {
            j = (int)e;
            src = p.dbpart + j;
            step = (size_t)p.mp;
        } else if (in) {
            j = (int)(e / p.n);
            i = (int)(e - (size_t)j * p.n);
            if (p.plan[(j >> 5) * nt_all + (i >> 5)]) src = p.part + (size_t)j * p.np + i;
            step = slice;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (src) {
            int z = z0;
            for (; z + 8 <= z1; z += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(z + u) * step];
                a0 += t[0]; a1 += t[1]; a2 += t[2]; a3 += t[3];
                a0 += t[4]; a1 += t[5]; a2 += t[6]; a3 += t[7];
            }
            for (; z < z1; ++z) a0 += src[(size_t)z * step];
        }
        sm[q][lane] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (q == 0 && in) {
            float v = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
            const float g0m = p.g0 ? ((!p.g0_act || p.g0_act[j] > 0.f) ? p.g0[j] : 0.f) : 0.f;
            if (is_b) {
                v += g0m;
                p.db[j] = p.db_accumulate ? p.db[j] + v : v;
            } else {
                if (p.g0) v += g0m * p.a0[i];
                if (p.wmask) v *= p.wmask[(size_t)j * p.ldw + i];
                float* dst = p.out + (size_t)j * p.ldo + i;
                *dst = p.accumulate ? *dst + v : v;
            }
        }
        __syncthreads();
    }
}

// plan[jt][it] = 1 when the 32 x 32 tile of the mask holds a non-zero (or there is no mask)
__global__ __launch_bounds__(256) void k_gradw32_plan(const float* wmask, int ldw, int m, int n, int32_t* plan) {
    const int nt_all = (n + 31) >> 5, mt_all = (m + 31) >> 5;
    for (int t = blockIdx.x; t < mt_all * nt_all; t += gridDim.x) {
        const int jt = t / nt_all, it = t - jt * nt_all;
        bool any = wmask == nullptr;
        if (wmask)
            for (int e = threadIdx.x; e < 1024; e += 256) {
                const int j = jt * 32 + (e >> 5), i = it * 32 + (e & 31);
                if (j < m && i < n && wmask[(size_t)j * ldw + i] != 0.f) any = true;
            }
        const int flag = __syncthreads_or(any ? 1 : 0);
        if (threadIdx.x == 0) plan[t] = flag ? 1 : 0;
    }
}

static int gw_slices(int m, int n, long long k) {
    const int blocks = ((m + 32 * GW_BT - 1) / (32 * GW_BT)) * ((n + 32 * GW_BT - 1) / (32 * GW_BT));
    long long s = max(1, 256 / blocks);
    s = min(s, max(1LL, k / (2 * GW_KC)));          // at least two chunks per slice
    return (int)s;
}

}  // namespace gv

using namespace gv;

extern "C" int64_t gv_made_gradw_f32_plan_words(int m, int n) { return (int64_t)((m + 31) / 32) * ((n + 31) / 32); }

extern "C" int gv_made_gradw_f32_plan(const float* wmask, int ldw, int m, int n, int32_t* plan, void* stream) {
    GV_REQUIRE(m > 0 && n > 0 && plan && (!wmask || ldw >= n), GV_ERR_SHAPE, "gv_made_gradw_f32_plan: m=%d n=%d ldw=%d", m, n, ldw);
    hipLaunchKernelGGL(k_gradw32_plan, dim3((unsigned)min((int64_t)256, gv_made_gradw_f32_plan_words(m, n))), dim3(256), 0, (hipStream_t)stream,
                       wmask, ldw, m, n, plan);
    return launch_status("gv_made_gradw_f32_plan");
}

extern "C" int64_t gv_made_gradw_f32_workspace_floats(int m, int n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int64_t mp = (m + 31) / 32 * 32, np = (n + 31) / 32 * 32;
    return ((int64_t)gw_slices(m, n, k) * (mp * np + mp) + 63) / 64 * 64;
}

namespace gv {
static int64_t gw_item_floats(const gv_gradw32_item& it) {
    if (it.m <= 0 || it.n <= 0 || it.k <= 0) return 0;
    const int64_t mp = (it.m + 31) / 32 * 32, np = (it.n + 31) / 32 * 32;
    return ((int64_t)gw_slices(it.m, it.n, it.k) * (mp * np + mp) + 63) / 64 * 64;
}
}  // namespace gv

extern "C" int64_t gv_made_gradw_f32_multi_workspace_floats(int count, const gv_gradw32_item* items) {
    int64_t total = 0;
    for (int q = 0; items && q < count; ++q) total += gw_item_floats(items[q]);
    return total;
}

extern "C" int gv_made_gradw_f32_multi(int count, const gv_gradw32_item* items, float* workspace, int64_t workspace_floats, void* stream) {
    GV_REQUIRE(count >= 1 && count <= GW_MAXI && items, GV_ERR_SHAPE, "gv_made_gradw_f32_multi: %d products (1 .. %d)", count, GW_MAXI);
    GV_REQUIRE(workspace && workspace_floats >= gv_made_gradw_f32_multi_workspace_floats(count, items), GV_ERR_SHAPE,
               "gv_made_gradw_f32_multi: workspace of %lld floats, %lld needed", (long long)workspace_floats,
               (long long)gv_made_gradw_f32_multi_workspace_floats(count, items));
    GradW32Multi P;
    GradW32ReduceMulti R;
    P.count = R.count = count;
    P.first[0] = R.first[0] = 0;
    float* ws = workspace;
    for (int q = 0; q < count; ++q) {
        const gv_gradw32_item& it = items[q];
        GV_REQUIRE(it.m > 0 && it.n > 0 && it.k > 0 && it.m % 4 == 0 && it.n % 4 == 0 && it.ldg >= it.m && it.lda >= it.n && it.ldg % 4 == 0 &&
                       it.lda % 4 == 0 && it.ldo >= it.n,
                   GV_ERR_SHAPE, "gv_made_gradw_f32: product %d: m=%d n=%d k=%lld ldg=%d lda=%d ldo=%d (widths and pitches are multiples of 4)", q,
                   it.m, it.n, (long long)it.k, it.ldg, it.lda, it.ldo);
        GV_REQUIRE(it.g && it.a && it.plan && it.out && aligned16(it.g) && aligned16(it.a), GV_ERR_NULL,
                   "gv_made_gradw_f32: product %d: NULL / unaligned pointer", q);
        GV_REQUIRE((!it.wmask || it.ldw >= it.n) && (!it.g0 || it.a0), GV_ERR_SHAPE,
                   "gv_made_gradw_f32: product %d: mask pitch / pass 0's row needs both vectors", q);
        GradW32Args& p = P.it[q];
        p.g = it.g; p.a = it.a; p.plan = it.plan; p.ldg = it.ldg; p.lda = it.lda; p.m = it.m; p.n = it.n; p.k = it.k;
        p.mp = (it.m + 31) / 32 * 32; p.np = (it.n + 31) / 32 * 32;
        p.slices = gw_slices(it.m, it.n, it.k);
        p.gy = (it.m + 32 * GW_BT - 1) / (32 * GW_BT);
        p.gz = (it.n + 32 * GW_BT - 1) / (32 * GW_BT);
        p.reserved = 0;
        p.k_per_slice = ((it.k + p.slices - 1) / p.slices + GW_KC - 1) / GW_KC * GW_KC;
        p.part = ws;
        p.dbpart = it.db ? ws + (size_t)p.slices * p.mp * p.np : nullptr;
        ws += gw_item_floats(it);
        P.first[q + 1] = P.first[q] + p.slices * p.gy * p.gz;
        GradW32Reduce& r = R.it[q];
        r.part = p.part; r.dbpart = p.dbpart; r.plan = it.plan; r.wmask = it.wmask; r.g0 = it.g0; r.g0_act = it.g0_act; r.a0 = it.a0;
        r.out = it.out; r.db = it.db;
        r.m = it.m; r.n = it.n; r.mp = p.mp; r.np = p.np; r.ldw = it.ldw; r.ldo = it.ldo; r.slices = p.slices;
        r.accumulate = it.accumulate; r.db_accumulate = it.db_accumulate;
        const size_t blocks = ((size_t)it.m * it.n + 63) / 64 + (it.db ? (it.m + 63) / 64 : 0);
        r.blocks = (int)min((size_t)4096, blocks);
        R.first[q + 1] = R.first[q] + r.blocks;
    }
    for (int q = count; q < GW_MAXI; ++q) {
        P.first[q + 1] = P.first[count]; R.first[q + 1] = R.first[count];
        P.it[q] = P.it[0]; R.it[q] = R.it[0];
    }
    const size_t lds = (size_t)4 * GW_KC * GW_LD * sizeof(float);
    static unsigned long long lds_done = 0;
    if (!raise_dynamic_lds((const void*)k_gradw32, (int)lds, lds_done, "gv_made_gradw_f32")) return GV_ERR_SHAPE;
    hipLaunchKernelGGL(k_gradw32, dim3((unsigned)P.first[count]), dim3(GW_THREADS), lds, (hipStream_t)stream, P);
    int rc = launch_status("gv_made_gradw_f32");
    if (rc != GV_OK) return rc;
    hipLaunchKernelGGL(k_gradw32_reduce, dim3((unsigned)R.first[count]), dim3(256), 0, (hipStream_t)stream, R);
    return launch_status("gv_made_gradw_f32(reduce)");
}

extern "C" int gv_made_gradw_f32(const float* g, int ldg, const float* a, int lda, int m, int n, int64_t k, const int32_t* plan,
                                 const float* wmask, int ldw, const float* g0, const float* g0_act, const float* a0, float* out, int ldo,
                                 int accumulate, float* db, int db_accumulate, float* workspace, int64_t workspace_floats, void* stream) {
    gv_gradw32_item it;
    it.g = g; it.a = a; it.plan = plan; it.wmask = wmask; it.g0 = g0; it.g0_act = g0_act; it.a0 = a0; it.out = out; it.db = db;
    it.ldg = ldg; it.lda = lda; it.m = m; it.n = n; it.ldw = ldw; it.ldo = ldo; it.accumulate = accumulate; it.db_accumulate = db_accumulate;
    it.k = k;
    return gv_made_gradw_f32_multi(1, &it, workspace, workspace_floats, stream);
}
