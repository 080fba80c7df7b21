// Index construction on the device (SURVEY 8(f-1), 8(b) "gv_build_csr"): the reference builds every batch's graph with
// python `sorted(zip(dst, src, rel))` and numpy (kgvae/utils.py:127-150); the kernels here need three orderings of the
// edge list (by destination, by source, by relation) and two of the triplet list, each with a CSR pointer and a list of
// <= chunk-edge work items.  One C call builds a whole index with NO host synchronisation: stable LSD radix sorts
// (rocPRIM through hipCUB), row pointers by binary search over the sorted keys, work-item counts + one exclusive scan of
// a packed {items, slots, fix-ups} triple, item lists sized by their upper bounds and pre-filled with -1 (the K1 kernels
// skip such entries).  Results are bit-identical to the torch formulation in ops.py (stable sort, searchsorted, cumsum),
// which stays as the exact-size builder for graphs that are indexed once.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace gv {

struct Tri {
    int a, b, c;
    __host__ __device__ Tri operator+(const Tri& o) const { return Tri{a + o.a, b + o.b, c + o.c}; }
};

__global__ void k_iota(int* x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = (int)i;
}

// iota + the keys narrowed to 16 bits (segment ids below 65 536): rocPRIM sorts 2-byte keys of >= 100 000 items with its
// onesweep radix sort (a histogram pass + two scatter passes) instead of ~10 merge passes
__global__ void k_iota_narrow(int* x, const int* keys, unsigned short* k16, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { x[i] = (int)i; k16[i] = (unsigned short)keys[i]; }
}

__global__ void k_iota_copy(int* x, const int* keys, int* copy, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { x[i] = (int)i; copy[i] = keys[i]; }
}

// rowptr[s] = number of sorted keys < s  (s = 0 .. n_seg)
template <typename KeyT>
__device__ __forceinline__ void lower_bound_of(const KeyT* keys_sorted, int64_t n, int n_seg, int* rowptr, int s) {
    if (s > n_seg) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int)keys_sorted[mid] < s) lo = mid + 1; else hi = mid;
    }
    rowptr[s] = (int)lo;
}

template <typename KeyT>
__global__ void k_lower_bounds(const KeyT* keys_sorted, int64_t n, int n_seg, int* rowptr) {
    lower_bound_of<KeyT>(keys_sorted, n, n_seg, rowptr, blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void k_item_counts(const int* rowptr, int n_seg, int chunk, Tri* counts) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const int deg = rowptr[s + 1] - rowptr[s];
    const int nch = max(1, (deg + chunk - 1) / chunk);
    counts[s] = Tri{nch, nch > 1 ? nch : 0, nch > 1 ? 1 : 0};
}

// same contents as k_items_fill (k_bdd.hip); entries beyond the capacities are dropped (cannot happen for lists sized by
// index_caps, kept as a guard against a caller's smaller buffers)
__global__ void k_items_fill_packed(const int* rowptr, int n_seg, int chunk, const Tri* offs, int4* items, int items_cap,
                                    int4* fix, int fix_cap) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const int beg = rowptr[s], end = rowptr[s + 1];
    const int nch = max(1, (end - beg + chunk - 1) / chunk);
    const Tri o = offs[s];
    for (int k = 0; k < nch; ++k) {
        const int b = beg + k * chunk;
        if (o.a + k < items_cap) items[o.a + k] = make_int4(s, b, min(end, b + chunk), nch > 1 ? o.b + k : -1);
    }
    if (nch > 1 && o.c < fix_cap) fix[o.c] = make_int4(s, o.b, nch, 0);
}

// ---- own stable LSD radix sort for segment ids below 65 536 -------------------------------------------------------------
// Kernels only (nothing a hipGraph would record as a memset / memcpy node) and few of them: a batch's index is ~0.3-0.7 M
// entries, so every pass is launch-latency bound and what counts is the number of dependent launches -- per pass one
// histogram kernel and one scatter kernel that redoes the (tiny) cross-block scan itself; ids of <= 9 bits sort in one pass.
// A block owns `tile` consecutive entries, a wave tile/16 consecutive ones of those, so (digit, block, wave, lane) order is
// input order and the sort is stable.
constexpr int RS_T = 1024, RS_W = 16, RS_MAXB = 512;

// (the bodies take the block's number b among the B blocks of ITS ordering: the batched launches below run several orderings in one grid)
template <typename KIn>
__device__ __forceinline__ void rs_hist_body(const KIn* keys, int n, int tile, int shift, int nbits, int* hist, const int b, const int B) {
    __shared__ int bins[RS_MAXB];
    const int nbins = 1 << nbits;
    for (int d = threadIdx.x; d < nbins; d += RS_T) bins[d] = 0;
    __syncthreads();
    const int beg = b * tile, end = min(n, beg + tile);
    for (int i = beg + threadIdx.x; i < end; i += RS_T) atomicAdd(&bins[((int)keys[i] >> shift) & (nbins - 1)], 1);
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += RS_T) hist[d * B + b] = bins[d];      // [digit][block]: the scan order
}

template <typename KIn>
__global__ __launch_bounds__(1024) void k_rs_hist(const KIn* keys, int n, int tile, int shift, int nbits, int* hist) {
    rs_hist_body<KIn>(keys, n, tile, shift, nbits, hist, blockIdx.x, gridDim.x);
}

// one more array carried through the final pass's permutation (the gathers the builders need anyway)
struct RsCarry {
    const int* src[3];
    int* out[3];
};

// MAXR = the most rounds (64 entries per wave each) a wave makes over its share of the tile; keys / values sit in registers from
// one batch of loads.  The pass is bound by memory TRANSACTIONS, not bytes (a wave's 64 entries go to ~64 different lines), so
// with MAXR = 4 (tiles of <= 4096 entries: everything up to 0.5 M entries) the block first puts its tile in digit order in LDS
// and then writes it out with consecutive threads on consecutive addresses (runs of tile / nbins entries per digit).
template <typename KIn, bool IOTA, int MAXR>
__device__ __forceinline__ void rs_scatter_body(const KIn* keys, const int* vals, int n, int tile, int shift, int nbits,
                                                const int* hist, unsigned short* keys_out, int* vals_out, const RsCarry& carry,
                                                const int b, const int B) {
    constexpr bool STAGE = MAXR <= 4;
    constexpr int STAGED = STAGE ? MAXR * RS_T : 1;
    __shared__ int wh[RS_W][RS_MAXB];      // per (wave, digit): count, then the wave's first position for the digit
    __shared__ int tot[RS_MAXB];           // per digit: entries in all blocks, then this block's first GLOBAL position (- local, staged)
    __shared__ int pre[RS_MAXB];           // per digit: entries in the blocks before this one
    __shared__ int ltot[RS_MAXB];          // per digit: entries in this block
    __shared__ int sval[STAGED];
    __shared__ unsigned short skey[STAGED];
    const int nbins = 1 << nbits, mask = nbins - 1;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int beg = b * tile, end = min(n, beg + tile);
    const int rounds = tile / RS_T;
    const int wbeg = beg + w * rounds * 64 + lane;
    int kreg[MAXR], vreg[MAXR], preg[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int i = wbeg + r * 64;
        const bool valid = r < rounds && i < end;
        kreg[r] = valid ? (int)keys[i] : -1;
        vreg[r] = IOTA ? i : (valid ? vals[i] : 0);
    }
    for (int i = t; i < RS_W * RS_MAXB; i += RS_T) (&wh[0][0])[i] = 0;
    {   // the [digit][block] histogram: G = 1024 / nbins threads per digit (nbins in 16 .. 512)
        const int G = RS_T >> nbits;
        const int d = t / G, j = t - d * G;
        int ts = 0, ps = 0;
        for (int bb0 = j; bb0 < B; bb0 += 8 * G) {      // eight loads in flight: other XCDs wrote these, they come from memory
            int h[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int bb = bb0 + k * G;
                h[k] = bb < B ? hist[d * B + bb] : 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ts += h[k];
                ps += bb0 + k * G < b ? h[k] : 0;
            }
        }
        for (int o = 1; o < G; o <<= 1) {
            ts += __shfl_xor(ts, o);
            ps += __shfl_xor(ps, o);
        }
        if (j == 0) { tot[d] = ts; pre[d] = ps; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
        if (kreg[r] >= 0) atomicAdd(&wh[w][(kreg[r] >> shift) & mask], 1);
    __syncthreads();
    if (t < nbins) {      // counts -> exclusive prefix over the waves
        int run = 0;
        for (int ww = 0; ww < RS_W; ++ww) {
            const int c = wh[ww][t];
            wh[ww][t] = run;
            run += c;
        }
        ltot[t] = run;
    }
    __syncthreads();
    if (w == 0) {   // exclusive scans over the digits: all blocks' totals (global start of a digit) and this block's (local start)
        const int per = nbins > 64 ? nbins >> 6 : 1;
        int gl[8], lo[8], gs = 0, ls = 0;
        for (int k = 0; k < 8; ++k) {
            const int d = lane * per + k;
            const bool in = k < per && d < nbins;
            gl[k] = in ? tot[d] : 0;
            lo[k] = in ? ltot[d] : 0;
            gs += gl[k];
            ls += lo[k];
        }
        int ginc = gs, linc = ls;
        for (int o = 1; o < 64; o <<= 1) {
            const int gv_ = __shfl_up(ginc, o), lv = __shfl_up(linc, o);
            if (lane >= o) { ginc += gv_; linc += lv; }
        }
        int grun = ginc - gs, lrun = linc - ls;
        for (int k = 0; k < 8; ++k) {
            const int d = lane * per + k;
            if (k < per && d < nbins) {
                const int gstart = grun + pre[d];                 // where this block's entries of digit d start in the output
                tot[d] = STAGE ? gstart - lrun : gstart;          // staged: output position = position in the tile + tot[d]
                ltot[d] = STAGE ? lrun : 0;
                grun += gl[k];
                lrun += lo[k];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r >= rounds) break;                       // uniform
        const bool valid = kreg[r] >= 0;
        const int d = (kreg[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool set = (d >> bit) & 1;
            const unsigned long long bm = __ballot(valid && set);
            peers &= set ? bm : ~bm;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        volatile int* c = &wh[w][d];
        const int base = valid ? *c : 0;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) *c = base + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
        preg[r] = base + rank + (STAGE ? ltot[d] : tot[d]);
    }
    if constexpr (STAGE) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (kreg[r] < 0) continue;
            skey[preg[r]] = (unsigned short)kreg[r];
            sval[preg[r]] = vreg[r];
        }
        __syncthreads();
        const int count = end - beg;
        int pos[MAXR], cr[3][MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int j = r * RS_T + t;
            const bool valid = j < count;
            kreg[r] = valid ? (int)skey[j] : -1;
            vreg[r] = valid ? sval[j] : 0;
            pos[r] = j + tot[(kreg[r] >> shift) & mask];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {      // all gathers of the carried arrays before any store (loads and stores share one counter)
            if (!carry.src[a]) continue;
#pragma unroll
            for (int r = 0; r < MAXR; ++r) cr[a][r] = kreg[r] >= 0 ? carry.src[a][vreg[r]] : 0;
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (kreg[r] < 0) continue;
            keys_out[pos[r]] = (unsigned short)kreg[r];
            vals_out[pos[r]] = vreg[r];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!carry.src[a]) continue;
#pragma unroll
            for (int r = 0; r < MAXR; ++r)
                if (kreg[r] >= 0) carry.out[a][pos[r]] = cr[a][r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (kreg[r] < 0) continue;
            keys_out[preg[r]] = (unsigned short)kreg[r];
            vals_out[preg[r]] = vreg[r];
            if (carry.src[0]) carry.out[0][preg[r]] = carry.src[0][vreg[r]];
            if (carry.src[1]) carry.out[1][preg[r]] = carry.src[1][vreg[r]];
            if (carry.src[2]) carry.out[2][preg[r]] = carry.src[2][vreg[r]];
        }
    }
}

template <typename KIn, bool IOTA, int MAXR>
__global__ __launch_bounds__(1024) void k_rs_scatter(const KIn* keys, const int* vals, int n, int tile, int shift, int nbits,
                                                     const int* hist, unsigned short* keys_out, int* vals_out, RsCarry carry) {
    rs_scatter_body<KIn, IOTA, MAXR>(keys, vals, n, tile, shift, nbits, hist, keys_out, vals_out, carry, blockIdx.x, gridDim.x);
}

// work-item counts, their exclusive scan, the item / fix-up lists and the -1 padding behind them in ONE launch (n_seg <=
// 65 536): a block owns 1024 consecutive segments and adds up the counts of all segments before its own itself (the rowptr is
// <= 256 KB and L2-resident: cheaper than a scan kernel in between).  Same lists as k_item_counts + scan + k_items_fill_packed.
__device__ inline Tri item_count(int deg, int chunk, int chunk_shift) {
    const int nch = max(1, chunk_shift >= 0 ? (deg + chunk - 1) >> chunk_shift : (deg + chunk - 1) / chunk);
    return Tri{nch, nch > 1 ? nch : 0, nch > 1 ? 1 : 0};
}

__device__ inline Tri tri_shfl_up(const Tri& v, int o) { return Tri{__shfl_up(v.a, o), __shfl_up(v.b, o), __shfl_up(v.c, o)}; }
__device__ inline Tri tri_shfl_xor(const Tri& v, int o) { return Tri{__shfl_xor(v.a, o), __shfl_xor(v.b, o), __shfl_xor(v.c, o)}; }

__device__ __forceinline__ void items_blocks_body(const int* rowptr, int n_seg, int chunk, int chunk_shift, int4* items,
                                                  int items_cap, int4* fix, int fix_cap, const int b, const int B) {
    __shared__ Tri red[3][RS_W];
    __shared__ int s_oa[RS_T + 1], s_ob[RS_T], s_beg[RS_T], s_end[RS_T];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int base = b * RS_T;
    Tri before{0, 0, 0}, total{0, 0, 0};
    const int s = base + t;
    int beg = 0, end = 0;
    // sixteen independent pairs of loads in flight: up to 16 384 segments (a sampled batch's padded node rows) in ONE round trip -- the
    // row pointers were written by another launch on other XCDs and come from memory; this thread's own segment is among them
    for (int s0 = 0; s0 < n_seg; s0 += 16 * RS_T) {
        int lo[16], hi[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = s0 + k * RS_T + t;
            lo[k] = q < n_seg ? rowptr[q] : 0;
            hi[k] = q < n_seg ? rowptr[q + 1] : 0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = s0 + k * RS_T + t;
            if (q >= n_seg) continue;
            const Tri c = item_count(hi[k] - lo[k], chunk, chunk_shift);
            total = total + c;
            if (q < base) before = before + c;
            if (q == s) { beg = lo[k]; end = hi[k]; }
        }
    }
    const Tri mine = s < n_seg ? item_count(end - beg, chunk, chunk_shift) : Tri{0, 0, 0};
    Tri inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const Tri v = tri_shfl_up(inc, o);
        if (lane >= o) inc = inc + v;
        before = before + tri_shfl_xor(before, o);
        total = total + tri_shfl_xor(total, o);
    }
    if (lane == 63) { red[0][w] = inc; red[1][w] = before; red[2][w] = total; }
    __syncthreads();
    Tri o{inc.a - mine.a, inc.b - mine.b, inc.c - mine.c}, first{0, 0, 0}, tot{0, 0, 0};
    int block_items = 0;
    for (int ww = 0; ww < RS_W; ++ww) {
        if (ww < w) o = o + red[0][ww];
        block_items += red[0][ww].a;
        first = first + red[1][ww];
        tot = tot + red[2][ww];
    }
    // the lists are written item by item (a relation's or a hub's row is hundreds of items): item i of this block belongs to the
    // last segment whose first item is <= i
    s_oa[t] = o.a;
    s_ob[t] = o.b + first.b;
    s_beg[t] = beg;
    s_end[t] = end;
    if (t == 0) s_oa[RS_T] = block_items;
    __syncthreads();
    const int nch = mine.a;
    if (nch > 1 && o.c + first.c < fix_cap) fix[o.c + first.c] = make_int4(s, o.b + first.b, nch, 0);
    for (int i = t; i < block_items; i += RS_T) {
        int lo = 0, hi = RS_T;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_oa[mid] <= i) lo = mid + 1; else hi = mid;
        }
        const int seg = lo - 1, k = i - s_oa[seg], n_it = s_oa[seg + 1] - s_oa[seg];
        const int bb = s_beg[seg] + k * chunk;
        if (first.a + i < items_cap)
            items[first.a + i] = make_int4(base + seg, bb, min(s_end[seg], bb + chunk), n_it > 1 ? s_ob[seg] + k : -1);
    }
    const int4 none = make_int4(-1, -1, -1, -1);
    const int stride = B * RS_T;
    for (int i = tot.a + base + t; i < items_cap; i += stride) items[i] = none;
    for (int i = tot.c + base + t; i < fix_cap; i += stride) fix[i] = none;
}

__global__ __launch_bounds__(1024) void k_items_blocks(const int* rowptr, int n_seg, int chunk, int chunk_shift, int4* items,
                                                       int items_cap, int4* fix, int fix_cap) {
    items_blocks_body(rowptr, n_seg, chunk, chunk_shift, items, items_cap, fix, fix_cap, blockIdx.x, gridDim.x);
}

__global__ void k_gather3_i32(const int* s1, int* o1, const int* s2, int* o2, const int* s3, int* o3, const int* idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx ? idx[i] : (int)i;
    o1[i] = s1[j];
    if (s2) o2[i] = s2[j];
    if (s3) o3[i] = s3[j];
}

// two fills in one launch (the -1 padding of the work-item and fix-up lists)
__global__ void k_fill2_u32(unsigned* p1, long long n1, unsigned* p2, long long n2, unsigned v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * 256) {
        if (i < n1) p1[i] = v; else p2[i - n1] = v;
    }
}

// one array through two permutations (idx NULL = identity)
__global__ void k_gather_two_i32(const int* src, const int* idx1, int* out1, const int* idx2, int* out2, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out1[i] = idx1 ? src[idx1[i]] : src[i];
    out2[i] = idx2 ? src[idx2[i]] : src[i];
}

// triplet incidence list: entry i < T = (subject, object, rel, i); entry T + i = (object, subject, rel, i)
__global__ void k_triplet_incidence(const int* trip, int64_t T, int* ent, int* other, int* rel2, int* tid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * T) return;
    const int64_t t = i < T ? i : i - T;
    const int s = trip[3 * t], r = trip[3 * t + 1], o = trip[3 * t + 2];
    ent[i] = i < T ? s : o;
    other[i] = i < T ? o : s;
    rel2[i] = r;
    tid[i] = (int)t;
}

__global__ void k_triplet_columns(const int* trip, int64_t T, int* s, int* r, int* o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    s[i] = trip[3 * i];
    r[i] = trip[3 * i + 1];
    o[i] = trip[3 * i + 2];
}

// ---- several orderings in the SAME launches (gv_build_csr_batch) -----------------------------------------------------------------
// A sampled batch needs five orderings (graph by destination / source / relation, triplets by entity / relation) of 20-60 k entries:
// every pass of every ordering is one short launch, ~30 dependent launches of ~4 us each.  The orderings do not depend on each
// other, so pass p of all of them runs as ONE grid: a block finds its ordering from the block ranges below and runs the same body
// as the single-ordering kernel.  Six launches for any number of orderings (histogram + scatter per pass, row pointers, items).
constexpr int IDX_MAX_JOBS = 8;

struct IdxJob {
    const int* keys;
    int n, n_seg, chunk, chunk_shift;
    int tile, blocks, passes, bits0, bits1;      // passes 0 = the keys are sorted already (perm NULL: identity)
    int* perm;
    unsigned short* k16_a;
    unsigned short* k16_b;
    int* iota;
    int* hist;
    RsCarry carry;
    int* rowptr;
    int4* items;
    int4* fix;
    int items_cap, fix_cap;
    int lb_blocks, copy_blocks, it_blocks;       // blocks of this ordering in the row-pointer (+ identity carry) and items launches
};

struct IdxBatch {
    IdxJob j[IDX_MAX_JOBS];
    int n;
};

template <int PASS>
__global__ __launch_bounds__(1024) void k_rs_hist_batch(const IdxBatch bt) {
    int b = blockIdx.x;
    for (int q = 0; q < bt.n; ++q) {
        const IdxJob& J = bt.j[q];
        const int nb = J.passes > PASS ? J.blocks : 0;
        if (b < nb) {
            if (PASS == 0) rs_hist_body<int>(J.keys, J.n, J.tile, 0, J.bits0, J.hist, b, nb);
            else rs_hist_body<unsigned short>(J.k16_a, J.n, J.tile, J.bits0, J.bits1, J.hist, b, nb);
            return;
        }
        b -= nb;
    }
}

template <int PASS>
__global__ __launch_bounds__(1024) void k_rs_scatter_batch(const IdxBatch bt) {
    int b = blockIdx.x;
    for (int q = 0; q < bt.n; ++q) {
        const IdxJob& J = bt.j[q];
        const int nb = J.passes > PASS ? J.blocks : 0;
        if (b < nb) {
            RsCarry none;
            for (int a = 0; a < 3; ++a) { none.src[a] = nullptr; none.out[a] = nullptr; }
            if (PASS == 0) {
                const bool last = J.passes == 1;
                rs_scatter_body<int, true, 4>(J.keys, nullptr, J.n, J.tile, 0, J.bits0, J.hist, J.k16_a, last ? J.perm : J.iota,
                                              last ? J.carry : none, b, nb);
            } else {
                rs_scatter_body<unsigned short, false, 4>(J.k16_a, J.iota, J.n, J.tile, J.bits0, J.bits1, J.hist, J.k16_b, J.perm,
                                                          J.carry, b, nb);
            }
            return;
        }
        b -= nb;
    }
}

// row pointers of every ordering; an ordering whose keys came sorted (no permutation) copies its carried arrays here
__global__ __launch_bounds__(256) void k_lower_bounds_batch(const IdxBatch bt) {
    int b = blockIdx.x;
    for (int q = 0; q < bt.n; ++q) {
        const IdxJob& J = bt.j[q];
        if (b < J.lb_blocks) {
            const int s = b * 256 + threadIdx.x;
            if (J.passes == 0) lower_bound_of<int>(J.keys, J.n, J.n_seg, J.rowptr, s);
            else lower_bound_of<unsigned short>(J.passes == 1 ? J.k16_a : J.k16_b, J.n, J.n_seg, J.rowptr, s);
            return;
        }
        b -= J.lb_blocks;
        if (b < J.copy_blocks) {
            const int i = b * 256 + threadIdx.x;
            if (i < J.n)
                for (int a = 0; a < 3; ++a)
                    if (J.carry.src[a]) J.carry.out[a][i] = J.carry.src[a][i];
            return;
        }
        b -= J.copy_blocks;
    }
}

__global__ __launch_bounds__(1024) void k_items_blocks_batch(const IdxBatch bt) {
    int b = blockIdx.x;
    for (int q = 0; q < bt.n; ++q) {
        const IdxJob& J = bt.j[q];
        if (b < J.it_blocks) {
            items_blocks_body(J.rowptr, J.n_seg, J.chunk, J.chunk_shift, J.items, J.items_cap, J.fix, J.fix_cap, b, J.it_blocks);
            return;
        }
        b -= J.it_blocks;
    }
}

// the incidence list and the columns of a triplet batch in one launch (the inputs of its two orderings)
template <typename TripT>
__global__ void k_triplet_lists(const TripT* trip, int64_t T, int* ent, int* other, int* rel2, int* tid, int* cs, int* cr, int* co,
                                int* trip32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * T) return;
    const int64_t t = i < T ? i : i - T;
    const int s = (int)trip[3 * t], r = (int)trip[3 * t + 1], o = (int)trip[3 * t + 2];
    if (trip32 && i < T) { trip32[3 * t] = s; trip32[3 * t + 1] = r; trip32[3 * t + 2] = o; }
    ent[i] = i < T ? s : o;
    other[i] = i < T ? o : s;
    rel2[i] = r;
    tid[i] = (int)t;
    if (i < T) { cs[i] = s; cr[i] = r; co[i] = o; }
}

// two int32 arrays to int64 in one launch (node ids / row picks that the reference's interfaces carry as int64)
__global__ void k_widen2_i32(const int* a, long long* a_out, int64_t na, const int* b, long long* b_out, int64_t nb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a_out[i] = a[i];
    else if (i < na + nb) b_out[i - na] = b[i - na];
}

}  // namespace gv

using namespace gv;

namespace {

inline int bits_for(int n_seg) {
    int b = 1;
    while (b < 31 && (1ll << b) < (long long)n_seg) ++b;
    return b;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct KeyLess {
    template <typename T>
    __host__ __device__ bool operator()(const T& a, const T& b) const { return a < b; }
};

// Which sort orders the entries.  'own' (k_rs_*, above) whenever the ids fit 16 bits and the entries fit <= 512 tiles: the same
// kernels eagerly and under capture.  Otherwise rocPRIM: its onesweep radix sort clears its counters with hipMemsetAsync,
// which a hipGraph records as MEMSET NODES, and on this stack such nodes can replay with stale parameters once other runtime
// work ran between two replays ('Memory access fault by GPU') -- so while the stream is being captured the fallback is
// rocPRIM's merge sort (kernels only; ~10 passes instead of 3).  GV_INDEX_SORT = own | radix | merge overrides (tests).
struct RsPlan {
    int tile, blocks, passes, bits[2];
};

bool rs_plan(int64_t n, int n_seg, RsPlan& p) {
    if (n <= 0 || n_seg > 65536) return false;
    int64_t tile = (((n + 127) / 128) + RS_T - 1) / RS_T * RS_T;
    tile = tile < RS_T ? RS_T : (tile > 16384 ? 16384 : tile);
    const int64_t blocks = (n + tile - 1) / tile;
    if (blocks > 512) return false;
    p.tile = (int)tile;
    p.blocks = (int)blocks;
    int bits = 1;
    while ((1ll << bits) < (long long)n_seg) ++bits;
    bits = bits < 4 ? 4 : bits;
    if (bits <= 9) { p.passes = 1; p.bits[0] = bits; p.bits[1] = 0; }
    else { p.passes = 2; p.bits[0] = (bits + 1) / 2; p.bits[1] = bits - p.bits[0]; }
    return true;
}

enum SortMode { SORT_OWN, SORT_RADIX, SORT_MERGE };

SortMode sort_mode(hipStream_t st, int64_t n, int n_seg) {
    const char* mode = getenv("GV_INDEX_SORT");      // read per build (tests flip it)
    if (mode && mode[0] == 'r') return SORT_RADIX;
    if (mode && mode[0] == 'm') return SORT_MERGE;
    RsPlan p;
    if (rs_plan(n, n_seg, p)) return SORT_OWN;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return SORT_RADIX; }
    return cs == hipStreamCaptureStatusActive ? SORT_MERGE : SORT_RADIX;
}

size_t cub_temp_bytes(int64_t n, int n_seg) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                       (int)n, 0, bits_for(n_seg));
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, b, (const Tri*)nullptr, (Tri*)nullptr, hipcub::Sum(), Tri{0, 0, 0}, n_seg + 1);
    size_t c = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, c, (const unsigned short*)nullptr, (unsigned short*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, (int)n, 0, 16);
    a = a > c ? a : c;
    size_t d = 0, e = 0;      // the merge-sort form (no memset nodes: used while the stream is being captured)
    (void)hipcub::DeviceMergeSort::StableSortPairs(nullptr, d, (unsigned short*)nullptr, (int*)nullptr, (int)n, KeyLess(), (hipStream_t)0);
    (void)hipcub::DeviceMergeSort::StableSortPairs(nullptr, e, (int*)nullptr, (int*)nullptr, (int)n, KeyLess(), (hipStream_t)0);
    d = d > e ? d : e;
    a = a > d ? a : d;
    const size_t own = (size_t)512 * RS_MAXB * sizeof(int);      // the own sort's [digit][block] histogram at its largest
    a = a > own ? a : own;
    return align256(a > b ? a : b);
}

// carve-up of the caller's workspace for ONE ordering of n entries into n_seg segments
struct Scratch {
    int* keys_sorted;
    int* iota;
    Tri* counts;
    Tri* offs;
    void* cub;
    size_t cub_bytes;
    static size_t bytes(int64_t n, int n_seg) {
        return 2 * align256((size_t)n * sizeof(int)) + 2 * align256((size_t)(n_seg + 1) * sizeof(Tri)) + cub_temp_bytes(n, n_seg);
    }
    Scratch(void* base, int64_t n, int n_seg) {
        char* p = (char*)base;
        keys_sorted = (int*)p; p += align256((size_t)n * sizeof(int));
        iota = (int*)p; p += align256((size_t)n * sizeof(int));
        counts = (Tri*)p; p += align256((size_t)(n_seg + 1) * sizeof(Tri));
        offs = (Tri*)p; p += align256((size_t)(n_seg + 1) * sizeof(Tri));
        cub = p;
        cub_bytes = cub_temp_bytes(n, n_seg);
    }
};

#define GV_HIP_OK(expr, what)                                          \
    do {                                                               \
        if ((expr) != hipSuccess) return launch_status(what);          \
    } while (0)

// up to three arrays through the same permutation in one launch (s2 / s3 may be NULL)
inline void gather3(const int* s1, int* o1, const int* s2, int* o2, const int* s3, int* o3, const int* idx, int64_t n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_gather3_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s1, o1, s2, o2, s3, o3, idx, n);
}

inline RsCarry carry_of(const int* s1, int* o1, const int* s2 = nullptr, int* o2 = nullptr, const int* s3 = nullptr,
                        int* o3 = nullptr) {
    RsCarry c;
    c.src[0] = s1; c.out[0] = o1; c.src[1] = s2; c.out[1] = o2; c.src[2] = s3; c.out[2] = o3;
    return c;
}

// the own sort: (keys, 0 .. n-1) -> (16-bit sorted keys, perm), `carry` arrays permuted along in the last pass
const unsigned short* own_sort(const int* keys, int64_t n, const RsPlan& pl, int* perm, const Scratch& sc, const RsCarry& carry,
                               hipStream_t st) {
    unsigned short* k16_a = (unsigned short*)sc.keys_sorted;                     // the two halves of the 4 n-byte key area
    unsigned short* k16_b = k16_a + ((n + 7) & ~(int64_t)7);
    int* hist = (int*)sc.cub;
    const dim3 grid((unsigned)pl.blocks), block(RS_T);
    const RsCarry none = carry_of(nullptr, nullptr);
    hipLaunchKernelGGL(k_rs_hist<int>, grid, block, 0, st, keys, (int)n, pl.tile, 0, pl.bits[0], hist);
    const bool few = pl.tile <= 4 * RS_T;
#define GV_RS_SCATTER(KIN, IOTA, ...)                                                                                   \
    do {                                                                                                                \
        if (few) hipLaunchKernelGGL((k_rs_scatter<KIN, IOTA, 4>), grid, block, 0, st, __VA_ARGS__);                     \
        else hipLaunchKernelGGL((k_rs_scatter<KIN, IOTA, 16>), grid, block, 0, st, __VA_ARGS__);                        \
    } while (0)
    if (pl.passes == 1) {
        GV_RS_SCATTER(int, true, keys, (const int*)nullptr, (int)n, pl.tile, 0, pl.bits[0], (const int*)hist, k16_a, perm, carry);
        return k16_a;
    }
    GV_RS_SCATTER(int, true, keys, (const int*)nullptr, (int)n, pl.tile, 0, pl.bits[0], (const int*)hist, k16_a, sc.iota, none);
    hipLaunchKernelGGL(k_rs_hist<unsigned short>, grid, block, 0, st, (const unsigned short*)k16_a, (int)n, pl.tile, pl.bits[0],
                       pl.bits[1], hist);
    GV_RS_SCATTER(unsigned short, false, (const unsigned short*)k16_a, (const int*)sc.iota, (int)n, pl.tile, pl.bits[0], pl.bits[1],
                  (const int*)hist, k16_b, perm, carry);
#undef GV_RS_SCATTER
    return k16_b;
}

// perm (optional: NULL = keys are already sorted, identity order) + rowptr + work items for one ordering; the `carry` arrays
// (optional) come out permuted the same way (carry.out[k][i] = carry.src[k][perm[i]])
int order_and_items(const int* keys, int64_t n, int n_seg, int chunk, int* perm, int* rowptr, int* items, int items_cap,
                    int* fix, int fix_cap, const Scratch& sc, hipStream_t st, const RsCarry* carry = nullptr) {
    const int* sorted = keys;
    const unsigned short* sorted16 = nullptr;
    bool carried = false;
    const SortMode mode = (perm && n > 0) ? sort_mode(st, n, n_seg) : SORT_RADIX;
    if (perm && n > 0 && n_seg <= 65536 && (mode == SORT_OWN || (mode == SORT_RADIX && n >= 100000))) {
        // two 16-bit halves in the 4 n-byte key area: the second starts n shorts in, rounded up to 16 B; it ends at most
        // 4 n + 14 bytes in, and whenever that rounding adds anything (n % 8 = m > 0) the area's own padding,
        // 256 - 4 (n % 64) >= 32 - 4 m bytes, covers the 16 - 2 m added -- the sort never writes into sc.iota behind it
        if ((size_t)(((n + 7) & ~(int64_t)7) + n) * sizeof(unsigned short) > align256((size_t)n * sizeof(int)))
            GV_REQUIRE(false, GV_ERR_SHAPE, "gv index: 16-bit key halves do not fit the key area (n=%lld)", (long long)n);
    }
    if (perm && n > 0 && mode == SORT_OWN) {
        RsPlan pl;
        rs_plan(n, n_seg, pl);
        // the carried arrays ride in the last pass only for small inputs: its <= 128 workgroups have too few gathers in flight for
        // more, and a separate full-occupancy gather is faster there (GV_INDEX_FUSE_CARRY = the largest n that fuses)
        const char* lim = getenv("GV_INDEX_FUSE_CARRY");
        const bool fuse = carry && n <= (lim ? atoll(lim) : 262144);
        sorted16 = own_sort(keys, n, pl, perm, sc, fuse ? *carry : carry_of(nullptr, nullptr), st);
        carried = fuse;
    } else if (perm && n > 0 && mode == SORT_MERGE) {      // stable merge sort in place: (keys, perm = iota)
        size_t tb = sc.cub_bytes;
        if (n_seg <= 65536) {
            unsigned short* k16 = (unsigned short*)sc.keys_sorted;
            hipLaunchKernelGGL(k_iota_narrow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, perm, keys, k16, n);
            GV_HIP_OK(hipcub::DeviceMergeSort::StableSortPairs(sc.cub, tb, k16, perm, (int)n, KeyLess(), st),
                      "gv index: merge sort (16-bit keys)");
            sorted16 = k16;
        } else {
            hipLaunchKernelGGL(k_iota_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, perm, keys, sc.keys_sorted, n);
            GV_HIP_OK(hipcub::DeviceMergeSort::StableSortPairs(sc.cub, tb, sc.keys_sorted, perm, (int)n, KeyLess(), st),
                      "gv index: merge sort");
            sorted = sc.keys_sorted;
        }
    } else if (perm && n >= 100000 && n_seg <= 65536) {      // rocPRIM onesweep on 2-byte keys (in / out halves of the key area)
        unsigned short* k16_in = (unsigned short*)sc.keys_sorted;
        unsigned short* k16_out = k16_in + ((n + 7) & ~(int64_t)7);
        hipLaunchKernelGGL(k_iota_narrow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sc.iota, keys, k16_in, n);
        size_t tb = sc.cub_bytes;
        GV_HIP_OK(hipcub::DeviceRadixSort::SortPairs(sc.cub, tb, (const unsigned short*)k16_in, k16_out, (const int*)sc.iota, perm,
                                                     (int)n, 0, bits_for(n_seg), st),
                  "gv index: radix sort (16-bit keys)");
        sorted16 = k16_out;
    } else if (perm && n > 0) {
        hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sc.iota, n);
        size_t tb = sc.cub_bytes;
        GV_HIP_OK(hipcub::DeviceRadixSort::SortPairs(sc.cub, tb, keys, sc.keys_sorted, (const int*)sc.iota, perm, (int)n, 0,
                                                     bits_for(n_seg), st),
                  "gv index: radix sort");
        sorted = sc.keys_sorted;
    }
    if (carry && !carried) gather3(carry->src[0], carry->out[0], carry->src[1], carry->out[1], carry->src[2], carry->out[2], perm, n, st);
    if (sorted16)
        hipLaunchKernelGGL(k_lower_bounds<unsigned short>, dim3((n_seg + 1 + 255) / 256), dim3(256), 0, st, sorted16, n, n_seg, rowptr);
    else
        hipLaunchKernelGGL(k_lower_bounds<int>, dim3((n_seg + 1 + 255) / 256), dim3(256), 0, st, sorted, n, n_seg, rowptr);
    if (n_seg <= 65536) {      // counts + scan + lists + padding in one launch
        int chunk_shift = -1;
        for (int k = 0; k < 31; ++k)
            if (chunk == (1 << k)) chunk_shift = k;
        hipLaunchKernelGGL(k_items_blocks, dim3(n_seg > 0 ? (n_seg + RS_T - 1) / RS_T : 1), dim3(RS_T), 0, st, (const int*)rowptr, n_seg,
                           chunk, chunk_shift, (int4*)items, items_cap, (int4*)fix, fix_cap);
        return launch_status("gv index: order_and_items");
    }
    {
        const long long n1 = (long long)items_cap * 4, n2 = (long long)fix_cap * 4;
        const unsigned blocks = (unsigned)((n1 + n2 + 255) / 256 > 4096 ? 4096 : (n1 + n2 + 255) / 256);
        if (n1 + n2 > 0)
            hipLaunchKernelGGL(k_fill2_u32, dim3(blocks), dim3(256), 0, st, (unsigned*)items, n1, (unsigned*)fix, n2, 0xFFFFFFFFu);
    }
    if (n_seg > 0) {
        hipLaunchKernelGGL(k_item_counts, dim3((n_seg + 255) / 256), dim3(256), 0, st, rowptr, n_seg, chunk, sc.counts);
        size_t tb = sc.cub_bytes;
        GV_HIP_OK(hipcub::DeviceScan::ExclusiveScan(sc.cub, tb, (const Tri*)sc.counts, sc.offs, hipcub::Sum(), Tri{0, 0, 0},
                                                    n_seg, st),
                  "gv index: scan");
        hipLaunchKernelGGL(k_items_fill_packed, dim3((n_seg + 255) / 256), dim3(256), 0, st, rowptr, n_seg, chunk, sc.offs,
                           (int4*)items, items_cap, (int4*)fix, fix_cap);
    }
    return launch_status("gv index: order_and_items");
}

}  // namespace

extern "C" void gv_index_caps(int64_t n_entries, int n_seg, int chunk, int* items_cap, int* fix_cap, int* slots_cap) {
    const int64_t extra = n_entries / (chunk > 0 ? chunk : 1) + 1;
    if (items_cap) *items_cap = (int)(n_seg + extra);
    if (fix_cap) *fix_cap = (int)((int64_t)n_seg < extra ? n_seg : extra);
    if (fix_cap && *fix_cap < 1) *fix_cap = 1;
    if (slots_cap) *slots_cap = (int)(2 * extra);
}

extern "C" int64_t gv_index_workspace_bytes(int64_t n_entries, int n_seg_max) {
    // room for one ordering at a time plus the incidence / column temporaries of the triplet index (6 int arrays)
    return (int64_t)(Scratch::bytes(n_entries, n_seg_max) + 6 * align256((size_t)n_entries * sizeof(int)));
}

extern "C" int gv_build_csr(const int32_t* keys, int64_t n, int n_seg, int chunk, int32_t* perm, int32_t* rowptr,
                            int32_t* items, int items_cap, int32_t* fix, int fix_cap, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    GV_REQUIRE(n >= 0 && n < (1ll << 31) && n_seg >= 0 && chunk > 0, GV_ERR_SHAPE, "gv_build_csr: n=%lld n_seg=%d chunk=%d",
               (long long)n, n_seg, chunk);
    GV_REQUIRE((keys || n == 0) && rowptr && items && fix && workspace, GV_ERR_NULL, "gv_build_csr: NULL pointer");
    GV_REQUIRE(workspace_bytes >= (int64_t)Scratch::bytes(n, n_seg), GV_ERR_WORKSPACE, "gv_build_csr: workspace too small");
    GV_REQUIRE(items_cap >= 1 && fix_cap >= 1, GV_ERR_SHAPE, "gv_build_csr: empty item buffers");
    Scratch sc(workspace, n, n_seg);
    return order_and_items(keys, n, n_seg, chunk, perm, rowptr, items, items_cap, fix, fix_cap, sc, (hipStream_t)stream);
}

extern "C" int gv_graph_index_build(const int32_t* src, const int32_t* dst, int64_t n_edges, int n_dst, int n_src,
                                    int dst_sorted, int chunk, int32_t* perm_d, int32_t* nbr_by_dst, int32_t* rowptr_d,
                                    int32_t* items_d, int items_d_cap, int32_t* fix_d, int fix_d_cap, int32_t* perm_s,
                                    int32_t* nbr_by_src, int32_t* rowptr_s, int32_t* items_s, int items_s_cap,
                                    int32_t* fix_s, int fix_s_cap, void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(n_edges >= 0 && n_edges < (1ll << 31) && n_dst >= 0 && n_src >= 0 && chunk > 0, GV_ERR_SHAPE,
               "gv_graph_index_build: bad sizes");
    GV_REQUIRE((src && dst) || n_edges == 0, GV_ERR_NULL, "gv_graph_index_build: NULL edge arrays");
    GV_REQUIRE(rowptr_d && items_d && fix_d && rowptr_s && items_s && fix_s && workspace, GV_ERR_NULL,
               "gv_graph_index_build: NULL output");
    GV_REQUIRE((nbr_by_dst && perm_s && nbr_by_src && (dst_sorted || perm_d)) || n_edges == 0, GV_ERR_NULL,
               "gv_graph_index_build: NULL per-edge output (perm_d is needed for unsorted destinations)");
    const int seg_max = n_dst > n_src ? n_dst : n_src;
    GV_REQUIRE(workspace_bytes >= (int64_t)Scratch::bytes(n_edges, seg_max), GV_ERR_WORKSPACE,
               "gv_graph_index_build: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Scratch sc(workspace, n_edges, seg_max);
    const RsCarry by_dst = carry_of(src, nbr_by_dst), by_src = carry_of(dst, nbr_by_src);
    int rc = order_and_items(dst, n_edges, n_dst, chunk, dst_sorted ? nullptr : perm_d, rowptr_d, items_d, items_d_cap, fix_d,
                             fix_d_cap, sc, st, &by_dst);
    if (rc != GV_OK) return rc;
    rc = order_and_items(src, n_edges, n_src, chunk, perm_s, rowptr_s, items_s, items_s_cap, fix_s, fix_s_cap, sc, st, &by_src);
    if (rc != GV_OK) return rc;
    return launch_status("gv_graph_index_build");
}

extern "C" int gv_relation_index_build(const int32_t* src, const int32_t* dst, const int32_t* etype, const int32_t* perm_d,
                                       const int32_t* perm_s, int64_t n_edges, int n_rel, int chunk, int32_t* et_by_dst,
                                       int32_t* et_by_src, int32_t* perm_r, int32_t* src_by_rel, int32_t* dst_by_rel,
                                       int32_t* rowptr_r, int32_t* items_r, int items_cap, int32_t* fix_r, int fix_cap,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(n_edges >= 0 && n_edges < (1ll << 31) && n_rel >= 0 && chunk > 0, GV_ERR_SHAPE, "gv_relation_index_build: bad sizes");
    GV_REQUIRE((src && dst && etype && perm_s) || n_edges == 0, GV_ERR_NULL, "gv_relation_index_build: NULL input");
    GV_REQUIRE(rowptr_r && items_r && fix_r && workspace, GV_ERR_NULL, "gv_relation_index_build: NULL output");
    GV_REQUIRE((et_by_dst && et_by_src && perm_r && src_by_rel && dst_by_rel) || n_edges == 0, GV_ERR_NULL,
               "gv_relation_index_build: NULL per-edge output");
    GV_REQUIRE(workspace_bytes >= (int64_t)Scratch::bytes(n_edges, n_rel), GV_ERR_WORKSPACE,
               "gv_relation_index_build: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Scratch sc(workspace, n_edges, n_rel);
    if (n_edges > 0)
        hipLaunchKernelGGL(k_gather_two_i32, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, st, etype, perm_d, et_by_dst,
                           perm_s, et_by_src, n_edges);
    const RsCarry by_rel = carry_of(src, src_by_rel, dst, dst_by_rel);
    int rc = order_and_items(etype, n_edges, n_rel, chunk, perm_r, rowptr_r, items_r, items_cap, fix_r, fix_cap, sc, st, &by_rel);
    if (rc != GV_OK) return rc;
    return launch_status("gv_relation_index_build");
}

extern "C" int gv_triplet_index_build(const int32_t* trip, int64_t T, int n_ent, int n_rel, int chunk, int chunk_rel,
                                      int32_t* inc_other, int32_t* inc_rel, int32_t* inc_tid, int32_t* rowptr_inc,
                                      int32_t* items_inc, int items_inc_cap, int32_t* fix_inc, int fix_inc_cap,
                                      int32_t* rel_s, int32_t* rel_o, int32_t* rel_tid, int32_t* rowptr_rel,
                                      int32_t* items_rel, int items_rel_cap, int32_t* fix_rel, int fix_rel_cap,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(T >= 0 && 2 * T < (1ll << 31) && n_ent >= 0 && n_rel >= 0 && chunk > 0 && chunk_rel > 0, GV_ERR_SHAPE,
               "gv_triplet_index_build: bad sizes");
    GV_REQUIRE(trip || T == 0, GV_ERR_NULL, "gv_triplet_index_build: NULL triplets");
    GV_REQUIRE(rowptr_inc && items_inc && fix_inc && rowptr_rel && items_rel && fix_rel && workspace, GV_ERR_NULL,
               "gv_triplet_index_build: NULL output");
    GV_REQUIRE((inc_other && inc_rel && inc_tid && rel_s && rel_o && rel_tid) || T == 0, GV_ERR_NULL,
               "gv_triplet_index_build: NULL per-triplet output");
    const int seg_max = n_ent > n_rel ? n_ent : n_rel;
    GV_REQUIRE(workspace_bytes >= gv_index_workspace_bytes(2 * T, seg_max), GV_ERR_WORKSPACE,
               "gv_triplet_index_build: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n2 = 2 * T;
    Scratch sc(workspace, n2, seg_max);
    char* p = (char*)workspace + Scratch::bytes(n2, seg_max);
    const size_t step = align256((size_t)n2 * sizeof(int));
    int* ent = (int*)p; int* other = (int*)(p + step); int* rel2 = (int*)(p + 2 * step); int* tid = (int*)(p + 3 * step);
    int* perm = (int*)(p + 4 * step); int* col = (int*)(p + 5 * step);
    if (n2 > 0)
        hipLaunchKernelGGL(k_triplet_incidence, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, trip, T, ent, other, rel2, tid);
    const RsCarry inc = carry_of(other, inc_other, rel2, inc_rel, tid, inc_tid);
    int rc = order_and_items(ent, n2, n_ent, chunk, perm, rowptr_inc, items_inc, items_inc_cap, fix_inc, fix_inc_cap, sc, st, &inc);
    if (rc != GV_OK) return rc;
    // by relation: columns of the triplet list (reusing the incidence temporaries), stable sort by relation
    int* cs = ent; int* cr = other; int* co = rel2;
    if (T > 0) hipLaunchKernelGGL(k_triplet_columns, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, trip, T, cs, cr, co);
    const RsCarry by_rel = carry_of(cs, rel_s, co, rel_o);
    rc = order_and_items(cr, T, n_rel, chunk_rel, rel_tid, rowptr_rel, items_rel, items_rel_cap, fix_rel, fix_rel_cap, sc, st, &by_rel);
    if (rc != GV_OK) return rc;
    (void)col;
    return launch_status("gv_triplet_index_build");
}

// ---- gv_build_csr_batch: see the batched kernels above ----------------------------------------------------------------------------
namespace {

bool batch_job_fits(const gv_csr_job& j, RsPlan& pl) {
    if (j.n <= 0 || j.n >= (1ll << 31) || j.n_seg <= 0 || j.n_seg > 65536 || j.chunk <= 0) return false;
    if (!j.perm) { pl.tile = RS_T; pl.blocks = 0; pl.passes = 0; pl.bits[0] = pl.bits[1] = 0; return true; }
    return rs_plan(j.n, j.n_seg, pl) && pl.tile <= 4 * RS_T;
}

}  // namespace

extern "C" int64_t gv_build_csr_batch_workspace_bytes(const gv_csr_job* jobs, int n_jobs) {
    if (!jobs || n_jobs <= 0) return 0;
    size_t total = 0;
    for (int q = 0; q < n_jobs; ++q) total += Scratch::bytes(jobs[q].n > 0 ? jobs[q].n : 0, jobs[q].n_seg > 0 ? jobs[q].n_seg : 0);
    return (int64_t)total;
}

extern "C" int gv_build_csr_batch(const gv_csr_job* jobs, int n_jobs, void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(n_jobs >= 0 && n_jobs <= IDX_MAX_JOBS, GV_ERR_SHAPE, "gv_build_csr_batch: %d orderings (at most %d)", n_jobs, IDX_MAX_JOBS);
    if (n_jobs == 0) return GV_OK;
    GV_REQUIRE(jobs && workspace, GV_ERR_NULL, "gv_build_csr_batch: NULL pointer");
    GV_REQUIRE(workspace_bytes >= gv_build_csr_batch_workspace_bytes(jobs, n_jobs), GV_ERR_WORKSPACE, "gv_build_csr_batch: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    bool batched = getenv("GV_INDEX_SORT") == nullptr && !(getenv("GV_INDEX_BATCH") && atoi(getenv("GV_INDEX_BATCH")) == 0);
    RsPlan plans[IDX_MAX_JOBS];
    for (int q = 0; q < n_jobs; ++q) {
        const gv_csr_job& j = jobs[q];
        GV_REQUIRE(j.n >= 0 && j.n < (1ll << 31) && j.n_seg >= 0 && j.chunk > 0, GV_ERR_SHAPE, "gv_build_csr_batch: ordering %d: n=%lld n_seg=%d chunk=%d",
                   q, (long long)j.n, j.n_seg, j.chunk);
        GV_REQUIRE((j.keys || j.n == 0) && j.rowptr && j.items && j.fix, GV_ERR_NULL, "gv_build_csr_batch: ordering %d: NULL pointer", q);
        GV_REQUIRE(j.items_cap >= 1 && j.fix_cap >= 1, GV_ERR_SHAPE, "gv_build_csr_batch: ordering %d: empty item buffers", q);
        for (int a = 0; a < 3; ++a)
            GV_REQUIRE(!j.carry_src[a] || j.carry_out[a], GV_ERR_NULL, "gv_build_csr_batch: ordering %d: carried array %d has no output", q, a);
        batched = batched && batch_job_fits(j, plans[q]);
    }
    char* base = (char*)workspace;
    if (!batched) {      // some ordering is outside the batched kernels (empty, > 65 536 segments, > 0.5 M entries): one after the other
        for (int q = 0; q < n_jobs; ++q) {
            const gv_csr_job& j = jobs[q];
            Scratch sc(base, j.n, j.n_seg);
            base += Scratch::bytes(j.n, j.n_seg);
            const RsCarry carry = carry_of(j.carry_src[0], j.carry_out[0], j.carry_src[1], j.carry_out[1], j.carry_src[2], j.carry_out[2]);
            const int rc = order_and_items(j.keys, j.n, j.n_seg, j.chunk, j.perm, j.rowptr, j.items, j.items_cap, j.fix, j.fix_cap, sc, st,
                                           j.carry_src[0] ? &carry : nullptr);
            if (rc != GV_OK) return rc;
        }
        return launch_status("gv_build_csr_batch");
    }
    IdxBatch bt;
    bt.n = n_jobs;
    int sort_blocks[2] = {0, 0}, lb_total = 0, it_total = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const gv_csr_job& j = jobs[q];
        const RsPlan& pl = plans[q];
        Scratch sc(base, j.n, j.n_seg);
        base += Scratch::bytes(j.n, j.n_seg);
        IdxJob& J = bt.j[q];
        J.keys = j.keys; J.n = (int)j.n; J.n_seg = j.n_seg; J.chunk = j.chunk;
        J.chunk_shift = -1;
        for (int k = 0; k < 31; ++k)
            if (j.chunk == (1 << k)) J.chunk_shift = k;
        J.tile = pl.tile; J.blocks = pl.blocks; J.passes = pl.passes; J.bits0 = pl.bits[0]; J.bits1 = pl.bits[1];
        J.perm = j.perm;
        J.k16_a = (unsigned short*)sc.keys_sorted;
        J.k16_b = J.k16_a + ((j.n + 7) & ~(int64_t)7);
        J.iota = sc.iota;
        J.hist = (int*)sc.cub;
        J.carry = carry_of(j.carry_src[0], j.carry_out[0], j.carry_src[1], j.carry_out[1], j.carry_src[2], j.carry_out[2]);
        J.rowptr = j.rowptr; J.items = (int4*)j.items; J.fix = (int4*)j.fix; J.items_cap = j.items_cap; J.fix_cap = j.fix_cap;
        J.lb_blocks = (j.n_seg + 1 + 255) / 256;
        J.copy_blocks = (pl.passes == 0 && j.carry_src[0]) ? (int)((j.n + 255) / 256) : 0;
        J.it_blocks = (j.n_seg + RS_T - 1) / RS_T;
        if (pl.passes >= 1) sort_blocks[0] += pl.blocks;
        if (pl.passes >= 2) sort_blocks[1] += pl.blocks;
        lb_total += J.lb_blocks + J.copy_blocks;
        it_total += J.it_blocks;
    }
    if (sort_blocks[0] > 0) {
        hipLaunchKernelGGL(k_rs_hist_batch<0>, dim3(sort_blocks[0]), dim3(RS_T), 0, st, bt);
        hipLaunchKernelGGL(k_rs_scatter_batch<0>, dim3(sort_blocks[0]), dim3(RS_T), 0, st, bt);
    }
    if (sort_blocks[1] > 0) {
        hipLaunchKernelGGL(k_rs_hist_batch<1>, dim3(sort_blocks[1]), dim3(RS_T), 0, st, bt);
        hipLaunchKernelGGL(k_rs_scatter_batch<1>, dim3(sort_blocks[1]), dim3(RS_T), 0, st, bt);
    }
    hipLaunchKernelGGL(k_lower_bounds_batch, dim3(lb_total), dim3(256), 0, st, bt);
    hipLaunchKernelGGL(k_items_blocks_batch, dim3(it_total), dim3(RS_T), 0, st, bt);
    return launch_status("gv_build_csr_batch");
}

extern "C" int gv_triplet_lists(const void* trip, int trip_is_int64, int64_t T, int32_t* ent, int32_t* other, int32_t* rel2, int32_t* tid,
                                int32_t* col_s, int32_t* col_r, int32_t* col_o, int32_t* trip32, void* stream) {
    GV_REQUIRE(T >= 0 && 2 * T < (1ll << 31), GV_ERR_SHAPE, "gv_triplet_lists: T=%lld", (long long)T);
    if (T == 0) return GV_OK;
    GV_REQUIRE(trip && ent && other && rel2 && tid && col_s && col_r && col_o, GV_ERR_NULL, "gv_triplet_lists: NULL pointer");
    const dim3 grid((unsigned)((2 * T + 255) / 256)), block(256);
    if (trip_is_int64)
        hipLaunchKernelGGL(k_triplet_lists<long long>, grid, block, 0, (hipStream_t)stream, (const long long*)trip, T, ent, other, rel2, tid,
                           col_s, col_r, col_o, trip32);
    else
        hipLaunchKernelGGL(k_triplet_lists<int>, grid, block, 0, (hipStream_t)stream, (const int*)trip, T, ent, other, rel2, tid, col_s, col_r,
                           col_o, trip32);
    return launch_status("gv_triplet_lists");
}

extern "C" int gv_widen2_i32(const int32_t* a, int64_t* a_out, int64_t na, const int32_t* b, int64_t* b_out, int64_t nb, void* stream) {
    GV_REQUIRE(na >= 0 && nb >= 0, GV_ERR_SHAPE, "gv_widen2_i32: na=%lld nb=%lld", (long long)na, (long long)nb);
    if (na + nb == 0) return GV_OK;
    GV_REQUIRE((na == 0 || (a && a_out)) && (nb == 0 || (b && b_out)), GV_ERR_NULL, "gv_widen2_i32: NULL pointer");
    hipLaunchKernelGGL(k_widen2_i32, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, (long long*)a_out, na, b,
                       (long long*)b_out, nb);
    return launch_status("gv_widen2_i32");
}
