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
__global__ void k_lower_bounds(const KeyT* keys_sorted, int64_t n, int n_seg, int* rowptr) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int)keys_sorted[mid] < s) lo = mid + 1; else hi = mid;
    }
    rowptr[s] = (int)lo;
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

__global__ void k_gather_i32(const int* src, const int* idx, int* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = idx ? src[idx[i]] : src[i];
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

// rocPRIM's onesweep radix sort clears its counters with hipMemsetAsync, which a hipGraph records as MEMSET NODES; on this
// stack such nodes can replay with stale parameters once other runtime work ran between two replays ('Memory access fault by
// GPU').  While the stream is being captured the orderings therefore come from rocPRIM's merge sort (kernels only; ~10
// passes instead of 3); GV_INDEX_SORT = radix | merge overrides.
bool use_merge_sort(hipStream_t st) {
    const char* mode = getenv("GV_INDEX_SORT");      // read per build (tests flip it)
    if (mode && mode[0] == 'r') return false;
    if (mode && mode[0] == 'm') return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs == hipStreamCaptureStatusActive;
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

// perm (optional: NULL = keys are already sorted, identity order) + rowptr + work items for one ordering
int order_and_items(const int* keys, int64_t n, int n_seg, int chunk, int* perm, int* rowptr, int* items, int items_cap,
                    int* fix, int fix_cap, const Scratch& sc, hipStream_t st) {
    const int* sorted = keys;
    const unsigned short* sorted16 = nullptr;
    if (perm && n > 0 && use_merge_sort(st)) {      // stable merge sort in place: (keys, perm = iota); no memset / memcpy nodes
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
    } else if (perm && n >= 100000 && n_seg <= 65536) {      // 2-byte keys: the two halves of the keys_sorted area hold them (in / out)
        // the area is align256(4 n) bytes; the output half starts n shorts in, rounded up to 16 B: it ends at most
        // 4 n + 14 bytes in, and whenever that rounding adds anything (n % 8 = m > 0) the area's own padding,
        // 256 - 4 (n % 64) >= 32 - 4 m bytes, covers the 16 - 2 m added -- the sort never writes into sc.iota behind it
        unsigned short* k16_in = (unsigned short*)sc.keys_sorted;
        const int64_t out_at = (n + 7) & ~(int64_t)7;
        if ((size_t)(out_at + n) * sizeof(unsigned short) > align256((size_t)n * sizeof(int)))
            GV_REQUIRE(false, GV_ERR_SHAPE, "gv index: 16-bit key halves do not fit the key area (n=%lld)", (long long)n);
        unsigned short* k16_out = k16_in + out_at;
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
    if (sorted16)
        hipLaunchKernelGGL(k_lower_bounds<unsigned short>, dim3((n_seg + 1 + 255) / 256), dim3(256), 0, st, sorted16, n, n_seg, rowptr);
    else
        hipLaunchKernelGGL(k_lower_bounds<int>, dim3((n_seg + 1 + 255) / 256), dim3(256), 0, st, sorted, n, n_seg, rowptr);
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

inline void gather(const int* src, const int* idx, int* out, int64_t n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_gather_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, idx, out, n);
}
// up to three arrays through the same permutation in one launch (s2 / s3 may be NULL)
inline void gather3(const int* s1, int* o1, const int* s2, int* o2, const int* s3, int* o3, const int* idx, int64_t n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_gather3_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s1, o1, s2, o2, s3, o3, idx, n);
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
    int rc = order_and_items(dst, n_edges, n_dst, chunk, dst_sorted ? nullptr : perm_d, rowptr_d, items_d, items_d_cap, fix_d,
                             fix_d_cap, sc, st);
    if (rc != GV_OK) return rc;
    gather(src, dst_sorted ? nullptr : perm_d, nbr_by_dst, n_edges, st);
    rc = order_and_items(src, n_edges, n_src, chunk, perm_s, rowptr_s, items_s, items_s_cap, fix_s, fix_s_cap, sc, st);
    if (rc != GV_OK) return rc;
    gather(dst, perm_s, nbr_by_src, n_edges, st);
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
    gather(etype, perm_d, et_by_dst, n_edges, st);
    gather(etype, perm_s, et_by_src, n_edges, st);
    int rc = order_and_items(etype, n_edges, n_rel, chunk, perm_r, rowptr_r, items_r, items_cap, fix_r, fix_cap, sc, st);
    if (rc != GV_OK) return rc;
    gather3(src, src_by_rel, dst, dst_by_rel, nullptr, nullptr, perm_r, n_edges, st);
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
    int rc = order_and_items(ent, n2, n_ent, chunk, perm, rowptr_inc, items_inc, items_inc_cap, fix_inc, fix_inc_cap, sc, st);
    if (rc != GV_OK) return rc;
    gather3(other, inc_other, rel2, inc_rel, tid, inc_tid, perm, n2, st);
    // by relation: columns of the triplet list (reusing the incidence temporaries), stable sort by relation
    int* cs = ent; int* cr = other; int* co = rel2;
    if (T > 0) hipLaunchKernelGGL(k_triplet_columns, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, trip, T, cs, cr, co);
    rc = order_and_items(cr, T, n_rel, chunk_rel, rel_tid, rowptr_rel, items_rel, items_rel_cap, fix_rel, fix_rel_cap, sc, st);
    if (rc != GV_OK) return rc;
    gather3(cs, rel_s, co, rel_o, nullptr, nullptr, rel_tid, T, st);
    (void)col;
    return launch_status("gv_triplet_index_build");
}
