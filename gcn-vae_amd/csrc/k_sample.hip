// Mini-batch preparation on the device: the reference samples edges, relabels nodes, draws negatives and
// builds the message-passing graph with numpy and python loops on the host (kgvae/utils.py:79-171).  Four entry points do
// the same on the GPU with a handful of launches:
//
//   gv_perm_sample          k distinct indices of [0, n)        (np.random.choice(n, k, replace=False), utils.py:79-82)
//   gv_relabel_pairs        np.unique((a, b), return_inverse)   (utils.py:103-105): sorted unique ids + the relabelled pair
//   gv_negative_sampling    utils.negative_sampling (:158-171): positives followed by neg_rate corrupted copies, labels
//   gv_graph_from_triplets  utils.build_graph_from_triplets (:135-150) + comp_deg_norm (:127-132): reverse edges, the
//                           (dst, src, rel) edge order, 1/in-degree of every edge's destination
//
// The deterministic parts (relabel, negatives from given draws, graph build) reproduce the host pipeline -- itself pinned
// by vectors captured from the reference -- array for array.  The random draws are counter-based (Philox4x32-10, the
// generator of gv_rng_fill), so a batch is a pure function of (seed, tick): the index sample is the first k outputs of a
// keyed PERMUTATION of [0, n) (4-round unbalanced Feistel network over ceil(log2 n) bits with cycle walking: distinct by
// construction, no sort of n keys as torch.randperm / np.random.choice do); a negative's entity is mulhi(u32, n_entities)
// and its coin is the top bit of a second u32.  That is this library's random stream, not numpy's.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace gv {

__device__ __forceinline__ uint32_t feistel_permute(uint32_t x, int bits, uint32_t k0, uint32_t k1, uint32_t stream,
                                                    uint32_t t0, uint32_t t1) {
    int la = bits / 2, rb = bits - la;                     // left has la bits, right has rb bits
    uint32_t l = x >> rb, r = x & ((1u << rb) - 1u);
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const uint32_t f = philox4x32_10(k0, k1, r, stream + 0x10000u * (uint32_t)(round + 1), t0, t1).x & ((1u << la) - 1u);
        const uint32_t nl = r, nr = l ^ f;                 // new left: rb bits, new right: la bits
        l = nl; r = nr;
        const int t = la; la = rb; rb = t;
    }
    return (l << rb) | r;
}

// tick_dev / n_dev (optional): the batch counter and the range are read on the device at execution time, so a launch recorded
// in a hipGraph draws a fresh sample on every replay (the counter is advanced by gv_rng_tick inside the same graph)
__global__ void k_perm_sample(long long n, long long k, int bits, uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long tick,
                              const unsigned long long* tick_dev, const int* n_dev, int* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    if (tick_dev) tick += *tick_dev;
    const uint32_t t0 = (uint32_t)tick, t1 = (uint32_t)(tick >> 32);
    if (n_dev) {
        n = *n_dev;
        bits = 2;
        while ((1ll << bits) < n) ++bits;
        if (i >= n) { out[i] = (int)(i % max(n, 1ll)); return; }        // k > n: no k distinct values exist; stay in range
    }
    uint32_t x = (uint32_t)i;
    do {
        x = feistel_permute(x, bits, k0, k1, stream, t0, t1);
    } while ((long long)x >= n);                           // cycle walking: the walk of i < n returns to [0, n)
    out[i] = (int)x;
}

__global__ void k_mark_pairs(const int* a, const int* b, long long k, int* flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    flags[a[i]] = 1;
    flags[b[i]] = 1;
}

__global__ void k_compact_ids(const int* flags, const int* rank, int num_ids, int* uniq, int cap, int* count) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id == 0) *count = rank[num_ids];
    if (id >= num_ids) return;
    if (flags[id] && rank[id] < cap) uniq[rank[id]] = id;
}

__global__ void k_map_pairs(const int* a, const int* b, long long k, const int* rank, int* a_local, int* b_local) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    a_local[i] = rank[a[i]];
    b_local[i] = rank[b[i]];
}

// gv_relabel_pairs as ONE launch when the id range fits a CU's LDS (num_ids + 1 ints: up to ~38 000 ids): one 1 024-thread
// workgroup keeps the flags, then their exclusive ranks, in LDS -- clear, mark, block scan, compact, map, with barriers where the
// six-launch form (fill, mark, rocPRIM's two scan kernels, compact, map) has launch boundaries.  A sampled batch is ~20 000 pairs
// over 14 541 ids: every one of those launches is at the ~5 us floor of a graph node.  Same results (integer work).
__global__ __launch_bounds__(1024) void k_relabel_one(const int* __restrict__ a, const int* __restrict__ b, long long k, int num_ids,
                                                      int* __restrict__ uniq, int cap, int* __restrict__ a_local,
                                                      int* __restrict__ b_local, int* __restrict__ count) {
    extern __shared__ int lds_rank[];              // [num_ids + 1]: flags, then exclusive ranks (rank[num_ids] = how many ids occur)
    __shared__ int wtot[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int n = num_ids + 1;
    for (int i = t; i < n; i += 1024) lds_rank[i] = 0;
    __syncthreads();
    for (long long i = t; i < k; i += 1024) {
        lds_rank[a[i]] = 1;
        lds_rank[b[i]] = 1;
    }
    __syncthreads();
    const int per = (n + 1023) / 1024, lo = min(n, t * per), hi = min(n, lo + per);      // thread t: a contiguous block of ids
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += lds_rank[i];
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int ww = 0; ww < w; ++ww) run += wtot[ww];
    for (int i = lo; i < hi; ++i) {
        const int f = lds_rank[i];
        lds_rank[i] = run;
        run += f;
    }
    __syncthreads();
    if (t == 0) *count = lds_rank[num_ids];
    for (int id = t; id < num_ids; id += 1024) {
        const int r = lds_rank[id];
        if (lds_rank[id + 1] != r && r < cap) uniq[r] = id;
    }
    for (long long i = t; i < k; i += 1024) {
        a_local[i] = lds_rank[a[i]];
        b_local[i] = lds_rank[b[i]];
    }
}

// samples (k * (neg_rate + 1), 3) int64: rows [0, k) = positives, row k + j*k + p = positive p with its subject
// (hit_subject) or object replaced by values[j*k + p]  (np.tile(pos, (neg_rate, 1)) order); labels 1 / 0
__global__ void k_negative_sampling(const int* s, const int* r, const int* o, long long k, int neg_rate, const int* n_ent_dev,
                                    const int* values, const uint8_t* hit, uint32_t k0, uint32_t k1, uint32_t stream,
                                    unsigned long long tick, const unsigned long long* tick_dev, long long* samples,
                                    float* labels) {
    if (tick_dev) tick += *tick_dev;
    const uint32_t t0 = (uint32_t)tick, t1 = (uint32_t)(tick >> 32);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = k * (neg_rate + 1);
    if (i >= total) return;
    long long* row = samples + 3 * i;
    if (i < k) {
        row[0] = s[i]; row[1] = r[i]; row[2] = o[i];
        labels[i] = 1.f;
        return;
    }
    const long long q = i - k, p = q % k;
    int v;
    bool subj;
    if (values) {
        v = values[q];
        subj = hit[q] != 0;
    } else {
        const uint4 x = philox4x32_10(k0, k1, (uint32_t)q, stream, t0, t1);
        v = (int)__umulhi(x.x, (uint32_t)(*n_ent_dev));
        subj = (x.y >> 31) != 0u;
    }
    row[0] = subj ? v : s[p];
    row[1] = r[p];
    row[2] = subj ? o[p] : v;
    labels[i] = 0.f;
}

__global__ void k_edge_keys(const int* s, const int* r, const int* o, const int* keep, long long m, long long bound,
                            int num_rels, unsigned long long* keys, int* src2, int* dst2, int* rel2, int* iota) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * m) return;
    const long long p = i < m ? i : i - m;
    const long long t = keep ? keep[p] : p;
    const int a = s[t], b = o[t], rr = r[t];
    const int sv = i < m ? a : b, dv = i < m ? b : a, rv = i < m ? rr : rr + num_rels;      // reverse edges: rel + num_rels
    src2[i] = sv; dst2[i] = dv; rel2[i] = rv;
    keys[i] = ((unsigned long long)dv * (unsigned long long)bound + (unsigned long long)sv) * (2ull * num_rels) + rv;
    iota[i] = (int)i;
}

__global__ void k_gather3(const int* perm, long long n, const int* a, const int* b, const int* c, int* oa, int* ob, int* oc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = perm[i];
    oa[i] = a[p]; ob[i] = b[p]; oc[i] = c[p];
}

// dst sorted ascending: in-degree of d = #entries equal to d, by two binary searches; norm = 1 / in-degree (fp32 division)
__global__ void k_edge_norm(const int* dst_sorted, long long n, float* norm) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = dst_sorted[i];
    long long lo = 0, hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (dst_sorted[mid] < d) lo = mid + 1; else hi = mid; }
    const long long first = lo;
    hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (dst_sorted[mid] <= d) lo = mid + 1; else hi = mid; }
    norm[i] = 1.0f / (float)(lo - first);
}

}  // namespace gv

using namespace gv;

namespace {
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

int key_bits(long long bound, int num_rels) {
    const unsigned long long top = (unsigned long long)bound * bound * 2ull * (unsigned long long)(num_rels > 0 ? num_rels : 1);
    int b = 1;
    while (b < 64 && (1ull << b) < top) ++b;
    return b;
}

size_t relabel_temp(int num_ids) {
    size_t t = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t, (const int*)nullptr, (int*)nullptr, num_ids + 1);
    return al256(t);
}

size_t graph_temp(long long n2, long long bound, int num_rels) {
    size_t t = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                             (const int*)nullptr, (int*)nullptr, (int)n2, 0, key_bits(bound, num_rels));
    return al256(t);
}
}  // namespace

extern "C" int gv_perm_sample(int64_t n, int64_t k, uint64_t seed, uint64_t tick, const uint64_t* tick_dev,
                              const int32_t* n_dev, uint32_t stream_id, int32_t* out, void* stream) {
    GV_REQUIRE(n >= 0 && k >= 0 && k <= n && n < (1ll << 31), GV_ERR_SHAPE, "gv_perm_sample: n=%lld k=%lld", (long long)n,
               (long long)k);
    if (k == 0) return GV_OK;
    GV_REQUIRE(out, GV_ERR_NULL, "gv_perm_sample: NULL output");
    int bits = 2;
    while ((1ll << bits) < n) ++bits;
    hipLaunchKernelGGL(k_perm_sample, dim3(blocks_for(k)), dim3(256), 0, (hipStream_t)stream, (long long)n, (long long)k, bits,
                       (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, (unsigned long long)tick,
                       (const unsigned long long*)tick_dev, n_dev, out);
    return launch_status("gv_perm_sample");
}

extern "C" int64_t gv_relabel_workspace_bytes(int num_ids) {
    return (int64_t)(2 * al256((size_t)(num_ids + 1) * sizeof(int)) + relabel_temp(num_ids));
}

extern "C" int gv_relabel_pairs(const int32_t* a, const int32_t* b, int64_t k, int num_ids, int32_t* uniq, int uniq_cap,
                                int32_t* a_local, int32_t* b_local, int32_t* count, void* workspace, int64_t workspace_bytes,
                                void* stream) {
    GV_REQUIRE(k >= 0 && num_ids > 0 && uniq_cap >= 0, GV_ERR_SHAPE, "gv_relabel_pairs: k=%lld num_ids=%d", (long long)k, num_ids);
    GV_REQUIRE(count && workspace && ((a && b && a_local && b_local && uniq) || k == 0), GV_ERR_NULL, "gv_relabel_pairs: NULL pointer");
    GV_REQUIRE(workspace_bytes >= gv_relabel_workspace_bytes(num_ids), GV_ERR_WORKSPACE, "gv_relabel_pairs: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    {   // the one-launch form while the id range fits LDS (GV_SAMPLER_ONE_LAUNCH=0: the six-launch form, as beyond that range)
        static const int one = getenv("GV_SAMPLER_ONE_LAUNCH") ? atoi(getenv("GV_SAMPLER_ONE_LAUNCH")) : 1;
        const size_t lds = (size_t)(num_ids + 1) * sizeof(int);
        static unsigned long long lds_done = 0;
        if (one && lds <= 150 * 1024 &&
            (lds <= 48 * 1024 || raise_dynamic_lds((const void*)k_relabel_one, 150 * 1024, lds_done, "gv_relabel_pairs"))) {
            hipLaunchKernelGGL(k_relabel_one, dim3(1), dim3(1024), lds, st, a, b, (long long)k, num_ids, uniq, uniq_cap, a_local, b_local,
                               count);
            return launch_status("gv_relabel_pairs");
        }
    }
    char* p = (char*)workspace;
    int* flags = (int*)p; p += al256((size_t)(num_ids + 1) * sizeof(int));
    int* rank = (int*)p; p += al256((size_t)(num_ids + 1) * sizeof(int));
    size_t tb = relabel_temp(num_ids);
    if (fill_words(flags, 0u, (size_t)(num_ids + 1) * sizeof(int), st) != hipSuccess) return launch_status("gv_relabel_pairs(memset)");
    if (k > 0) hipLaunchKernelGGL(k_mark_pairs, dim3(blocks_for(k)), dim3(256), 0, st, a, b, (long long)k, flags);
    if (hipcub::DeviceScan::ExclusiveSum(p, tb, (const int*)flags, rank, num_ids + 1, st) != hipSuccess)
        return launch_status("gv_relabel_pairs(scan)");
    hipLaunchKernelGGL(k_compact_ids, dim3(blocks_for(num_ids)), dim3(256), 0, st, flags, rank, num_ids, uniq, uniq_cap, count);
    if (k > 0) hipLaunchKernelGGL(k_map_pairs, dim3(blocks_for(k)), dim3(256), 0, st, a, b, (long long)k, rank, a_local, b_local);
    return launch_status("gv_relabel_pairs");
}

extern "C" int gv_negative_sampling(const int32_t* s, const int32_t* r, const int32_t* o, int64_t k, int neg_rate,
                                    const int32_t* n_entities_dev, const int32_t* values, const uint8_t* hit_subject,
                                    uint64_t seed, uint64_t tick, const uint64_t* tick_dev, uint32_t stream_id, int64_t* samples,
                                    float* labels, void* stream) {
    GV_REQUIRE(k >= 0 && neg_rate >= 0, GV_ERR_SHAPE, "gv_negative_sampling: k=%lld neg_rate=%d", (long long)k, neg_rate);
    if (k == 0) return GV_OK;
    GV_REQUIRE(s && r && o && samples && labels, GV_ERR_NULL, "gv_negative_sampling: NULL pointer");
    GV_REQUIRE((values && hit_subject) || (!values && !hit_subject && n_entities_dev), GV_ERR_NULL,
               "gv_negative_sampling: pass both draws (values, hit_subject) or neither (then n_entities_dev)");
    hipLaunchKernelGGL(k_negative_sampling, dim3(blocks_for(k * (neg_rate + 1))), dim3(256), 0, (hipStream_t)stream, s, r, o,
                       (long long)k, neg_rate, n_entities_dev, values, hit_subject, (uint32_t)seed, (uint32_t)(seed >> 32),
                       stream_id, (unsigned long long)tick, (const unsigned long long*)tick_dev, (long long*)samples, labels);
    return launch_status("gv_negative_sampling");
}

extern "C" int64_t gv_graph_from_triplets_workspace_bytes(int64_t m, int n_nodes_bound, int num_rels) {
    const long long n2 = 2 * m;
    return (int64_t)(2 * al256((size_t)n2 * 8) + 5 * al256((size_t)n2 * 4) + graph_temp(n2, n_nodes_bound, num_rels));
}

extern "C" int gv_graph_from_triplets(const int32_t* s, const int32_t* r, const int32_t* o, const int32_t* keep, int64_t m,
                                      int n_nodes_bound, int num_rels, int32_t* src2, int32_t* dst2, int32_t* rel2,
                                      float* norm, void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(m >= 0 && 2 * m < (1ll << 31) && n_nodes_bound > 0 && num_rels > 0, GV_ERR_SHAPE, "gv_graph_from_triplets: bad sizes");
    if (m == 0) return GV_OK;
    GV_REQUIRE(s && r && o && src2 && dst2 && rel2 && norm && workspace, GV_ERR_NULL, "gv_graph_from_triplets: NULL pointer");
    GV_REQUIRE(workspace_bytes >= gv_graph_from_triplets_workspace_bytes(m, n_nodes_bound, num_rels), GV_ERR_WORKSPACE,
               "gv_graph_from_triplets: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long long n2 = 2 * m;
    char* p = (char*)workspace;
    unsigned long long* keys = (unsigned long long*)p; p += al256((size_t)n2 * 8);
    unsigned long long* keys_sorted = (unsigned long long*)p; p += al256((size_t)n2 * 8);
    int* ts = (int*)p; p += al256((size_t)n2 * 4);
    int* td = (int*)p; p += al256((size_t)n2 * 4);
    int* tr = (int*)p; p += al256((size_t)n2 * 4);
    int* iota = (int*)p; p += al256((size_t)n2 * 4);
    int* perm = (int*)p; p += al256((size_t)n2 * 4);
    size_t tb = graph_temp(n2, n_nodes_bound, num_rels);
    hipLaunchKernelGGL(k_edge_keys, dim3(blocks_for(n2)), dim3(256), 0, st, s, r, o, keep, (long long)m, (long long)n_nodes_bound,
                       num_rels, keys, ts, td, tr, iota);
    if (hipcub::DeviceRadixSort::SortPairs(p, tb, (const unsigned long long*)keys, keys_sorted, (const int*)iota, perm, (int)n2, 0,
                                           key_bits(n_nodes_bound, num_rels), st) != hipSuccess)
        return launch_status("gv_graph_from_triplets(sort)");
    hipLaunchKernelGGL(k_gather3, dim3(blocks_for(n2)), dim3(256), 0, st, perm, n2, ts, td, tr, src2, dst2, rel2);
    hipLaunchKernelGGL(k_edge_norm, dim3(blocks_for(n2)), dim3(256), 0, st, dst2, n2, norm);
    return launch_status("gv_graph_from_triplets");
}

extern "C" int gv_gather3_i32(const int32_t* idx, int64_t n, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* out_a,
                              int32_t* out_b, int32_t* out_c, void* stream) {
    GV_REQUIRE(n >= 0, GV_ERR_SHAPE, "gv_gather3_i32: n=%lld", (long long)n);
    if (n == 0) return GV_OK;
    GV_REQUIRE(idx && a && b && c && out_a && out_b && out_c, GV_ERR_NULL, "gv_gather3_i32: NULL pointer");
    hipLaunchKernelGGL(k_gather3, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, idx, (long long)n, a, b, c, out_a, out_b, out_c);
    return launch_status("gv_gather3_i32");
}

// ---------------------------------------------------------------------------------------------
// Neighbourhood-expansion edge sampler (kgvae/utils.py:33-76 sample_edge_neighborhood, selected by --edge-sampler
// neighbor, kgvae/link_predict.py:311).  Inherently sequential -- every draw conditions the next -- so ONE 1024-thread
// workgroup walks the sample_size draws; what is parallel is each draw's inverse-CDF search over the vertices:
// thread t owns a contiguous block of vertices and keeps its block's weight sum  sum(budget * seen)  (and the count of
// vertices with budget left, for the reference's "nothing seen has budget" fallback) in registers; a block scan locates
// the owner of the drawn position, the owner walks its few vertices, picks the edge (rejection loop over the vertex's
// incidence list, as the reference) and updates budget / seen / picked; only the two touched owners re-sum.
// Draws: Philox4x32-10(seed; counter = (draw i, stream + 0x10000 * attempt a, tick lo, tick hi)).x -- a = 0 selects the vertex as
// floor(u * total weight / 2^32) in the integer weight CDF, a >= 1 the incidence-list entry floor(u * degree / 2^32).
namespace gv {
__device__ __forceinline__ uint32_t nbr_draw(uint32_t k0, uint32_t k1, uint32_t i, uint32_t attempt, uint32_t stream, uint64_t tick) {
    return philox4x32_10(k0, k1, i, stream + 0x10000u * attempt, (uint32_t)tick, (uint32_t)(tick >> 32)).x;
}

__global__ __launch_bounds__(1024) void k_neighborhood_sample(const int* __restrict__ adj_ptr, const int* __restrict__ adj_edge,
                                                              const int* __restrict__ adj_other, int n_vertices, int sample_size,
                                                              uint32_t k0, uint32_t k1, uint32_t stream, uint64_t tick,
                                                              const unsigned long long* tick_dev, int* budget_g, uint8_t* seen_g, uint8_t* picked, int* edges,
                                                              int state_in_lds) {
    __shared__ long long wave_tot[16];
    __shared__ int sh_v, sh_other;
    extern __shared__ int lds_state[];              // budget (V ints) then seen (V bytes) when they fit: no global round trips
    if (tick_dev) tick += *tick_dev;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n_vertices + 1023) / 1024;
    const int lo = min(n_vertices, tid * per), hi = min(n_vertices, lo + per);
    int* budget = budget_g;
    uint8_t* seen = seen_g;
    if (state_in_lds) {
        budget = lds_state;
        seen = (uint8_t*)(lds_state + n_vertices);
        for (int v = tid; v < n_vertices; v += 1024) {
            budget[v] = budget_g[v];
            seen[v] = 0;
        }
        __syncthreads();
    }
    long long ws = 0, cs = 0;                       // this thread's block: sum(budget * seen), #(budget > 0)
    for (int v = lo; v < hi; ++v) {
        ws += seen[v] ? budget[v] : 0;
        cs += budget[v] > 0;
    }
    for (int i = 0; i < sample_size; ++i) {
        // ---- total weight and this thread's exclusive prefix (two-level scan; a second one for the fallback) ----
        long long val = ws, excl = 0, total = 0;
        bool use_cs = false;
        for (;;) {
            long long inc = val;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const long long up = __shfl_up(inc, d);
                if (lane >= d) inc += up;
            }
            if (lane == 63) wave_tot[wid] = inc;
            __syncthreads();
            long long base = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const long long t = wave_tot[w];
                if (w < wid) base += t;
                total += t;
            }
            excl = base + inc - val;
            __syncthreads();
            if (total > 0 || use_cs) break;
            use_cs = true;                         // nothing seen has budget left: uniform over the vertices that do
            val = cs;
        }
        if (total <= 0) {                          // every edge is picked (sample_size > number of triplets): stop
            if (tid == 0) for (int j = i; j < sample_size; ++j) edges[j] = -1;
            break;
        }
        const long long pos = (long long)(((unsigned long long)nbr_draw(k0, k1, (uint32_t)i, 0u, stream, tick) *
                                           (unsigned long long)total) >> 32);
        if (pos >= excl && pos < excl + val) {      // the owner of the drawn position
            long long run = excl;
            int v = lo;
            for (; v < hi; ++v) {
                const long long w = use_cs ? (budget[v] > 0 ? 1 : 0) : (seen[v] ? budget[v] : 0);
                if (pos < run + w) break;
                run += w;
            }
            seen[v] = 1;
            const int a0 = adj_ptr[v], deg = adj_ptr[v + 1] - a0;
            int e, j;
            uint32_t attempt = 1;
            do {
                if (attempt > 4096u) {             // sampling.NEIGHBOR_MAX_ATTEMPTS: take the first unpicked entry
                    for (j = 0; picked[adj_edge[a0 + j]]; ++j) {}
                } else {
                    j = (int)(((unsigned long long)nbr_draw(k0, k1, (uint32_t)i, attempt, stream, tick) * (unsigned long long)deg) >> 32);
                }
                ++attempt;
                e = adj_edge[a0 + j];
            } while (picked[e]);
            const int other = adj_other[a0 + j];
            edges[i] = e;
            picked[e] = 1;
            budget[v] -= 1;
            budget[other] -= 1;
            seen[other] = 1;
            sh_v = v;
            sh_other = other;
        }
        __syncthreads();
        const int cv = sh_v, co = sh_other;
        if ((cv >= lo && cv < hi) || (co >= lo && co < hi)) {      // the touched owners re-sum their blocks
            ws = 0; cs = 0;
            for (int v = lo; v < hi; ++v) {
                ws += seen[v] ? budget[v] : 0;
                cs += budget[v] > 0;
            }
        }
        __syncthreads();
    }
}
}  // namespace gv

extern "C" int gv_neighborhood_sample(const int32_t* adj_ptr, const int32_t* adj_edge, const int32_t* adj_other,
                                      const int32_t* degrees, int num_vertices, int64_t num_triplets, int sample_size,
                                      uint64_t seed, uint64_t tick, const uint64_t* tick_dev, uint32_t stream_id, int32_t* edges,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    GV_REQUIRE(num_vertices > 0 && num_triplets > 0 && sample_size >= 0, GV_ERR_SHAPE,
               "gv_neighborhood_sample: num_vertices=%d num_triplets=%lld sample_size=%d", num_vertices, (long long)num_triplets,
               sample_size);
    if (sample_size == 0) return GV_OK;
    GV_REQUIRE(adj_ptr && adj_edge && adj_other && degrees && edges && workspace, GV_ERR_NULL, "gv_neighborhood_sample: NULL pointer");
    const size_t b_budget = al256((size_t)num_vertices * sizeof(int)), b_seen = al256((size_t)num_vertices),
                 b_picked = al256((size_t)num_triplets);
    GV_REQUIRE(workspace_bytes >= (int64_t)(b_budget + b_seen + b_picked), GV_ERR_WORKSPACE,
               "gv_neighborhood_sample: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    int* budget = (int*)p; p += b_budget;
    uint8_t* seen = (uint8_t*)p; p += b_seen;
    uint8_t* picked = (uint8_t*)p;
    if (copy_words(budget, degrees, (size_t)num_vertices * sizeof(int), st) != hipSuccess ||
        fill_words(seen, 0u, b_seen + b_picked, st) != hipSuccess)
        return launch_status("gv_neighborhood_sample(init)");
    const size_t lds = (size_t)num_vertices * 5 + 16;
    const int in_lds = lds <= 150 * 1024;
    if (in_lds && lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)k_neighborhood_sample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return launch_status("gv_neighborhood_sample(lds)");
    hipLaunchKernelGGL(k_neighborhood_sample, dim3(1), dim3(1024), in_lds ? lds : 0, st, adj_ptr, adj_edge, adj_other, num_vertices,
                       sample_size, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, tick, (const unsigned long long*)tick_dev,
                       budget, seen, picked, edges, in_lds);
    return launch_status("gv_neighborhood_sample");
}

extern "C" int64_t gv_neighborhood_sample_workspace_bytes(int num_vertices, int64_t num_triplets) {
    return (int64_t)(al256((size_t)num_vertices * sizeof(int)) + al256((size_t)num_vertices) + al256((size_t)num_triplets));
}
