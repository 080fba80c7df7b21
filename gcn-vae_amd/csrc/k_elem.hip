// Row-wise / element-wise kernels of the path: layer epilogue, embedding gather/scatter,
// reparameterisation (K3), IAF update (K4 glue), clip+Adam (a-11).  All HBM/cache-bound
// streaming kernels: 16-B per lane where the shape allows, grid-stride, <= 2048 blocks.
#include "common.h"

namespace gv {

static inline int grid_for(int64_t n, int per_block) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

#define GV_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ __launch_bounds__(256) void k_epilogue_fwd(const float* agg, const float* addend, int act,
                                                      const uint8_t* keep, float scale, float* out, int64_t n) {
    GV_GRID_STRIDE(i, n) {
        float v = agg[i];
        if (addend) v += addend[i];
        v = apply_act(v, act);
        if (keep) v = keep[i] ? v * scale : 0.f;
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_epilogue_bwd(const float* out, const float* gout, int act,
                                                      const uint8_t* keep, float scale, float* g, int64_t n) {
    GV_GRID_STRIDE(i, n) {
        float v = gout[i];
        if (keep) v = keep[i] ? v * scale : 0.f;
        if (act == GV_ACT_RELU && !(out[i] > 0.f)) v = 0.f;
        g[i] = v;
    }
}

// epilogue backward that also emits the column sums of g (the bias gradient).  Block b owns a contiguous row range;
// a thread owns one float4 column group and every (256 / groups)-th row of the range, so each pass of the block is a
// coalesced sweep of whole rows; the row lanes' sums are combined through LDS in lane order and written as slice b
// of `part` (EPI_SLICES x n); gv_colsum_finish adds the slices in order.
constexpr int EPI_SLICES = 1024;
__global__ __launch_bounds__(256) void k_epilogue_bwd_colsum(const float* __restrict__ out, const float* __restrict__ gout,
                                                             int act, const uint8_t* __restrict__ keep, float scale,
                                                             float* __restrict__ g, int64_t m, int n, float* part) {
    __shared__ float4 sm[256];
    const int groups = n >> 2;                                  // float4 column groups per row (n % 4 == 0, <= 256)
    const int rows_par = 256 / groups;
    const int rl = threadIdx.x / groups, cg = threadIdx.x - rl * groups;
    const int64_t per = (m + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = blockIdx.x * per, r1 = min(m, r0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < rows_par) {
        for (int64_t r = r0 + rl; r < r1; r += rows_par) {
            const int64_t i = r * n + 4 * cg;
            float4 v = *reinterpret_cast<const float4*>(gout + i);
            if (keep) {
                const uchar4 k4 = *reinterpret_cast<const uchar4*>(keep + i);
                v.x = k4.x ? v.x * scale : 0.f; v.y = k4.y ? v.y * scale : 0.f;
                v.z = k4.z ? v.z * scale : 0.f; v.w = k4.w ? v.w * scale : 0.f;
            }
            if (act == GV_ACT_RELU) {
                const float4 o = *reinterpret_cast<const float4*>(out + i);
                if (!(o.x > 0.f)) v.x = 0.f;
                if (!(o.y > 0.f)) v.y = 0.f;
                if (!(o.z > 0.f)) v.z = 0.f;
                if (!(o.w > 0.f)) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(g + i) = v;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        sm[threadIdx.x] = acc;
    }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rows_par; ++k) {
            const float4 t = sm[k * groups + cg];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * n + 4 * cg) = acc;
    }
}

// one wave per row; int64 ids as torch's embedding takes them
__global__ __launch_bounds__(256) void k_gather_rows(const float* table, const int64_t* ids, float* out, int64_t n,
                                                     int h, uint64_t* rng_state) {
    if (rng_state && blockIdx.x == 0 && threadIdx.x == 0) rng_state[1] += 1;   // start-of-forward RNG tick (nobody reads it here)
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        const float* src = table + ids[r] * (int64_t)h;
        float* dst = out + r * (int64_t)h;
        for (int c = lane; c < h; c += 64) dst[c] = src[c];
    }
}

__global__ __launch_bounds__(256) void k_scatter_add_rows(const float* gout, const int64_t* ids, float* gtable,
                                                          int64_t n, int h) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        const float* src = gout + r * (int64_t)h;
        float* dst = gtable + ids[r] * (int64_t)h;
        for (int c = lane; c < h; c += 64) atomicAdd(dst + c, src[c]);   // ids may repeat; 256-B contiguous adds
    }
}

__device__ __forceinline__ float softplus_t(float x) {  // torch: beta=1, threshold=20
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void k_reparam_fwd(const float* h2, const float* eps, float* z, float* v, float* mout,
                                                     int64_t n, int h) {
    const int64_t total = n * h;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / h;
        const int c = (int)(i - r * h);
        const float m = h2[r * 2 * h + c];
        const float var = softplus_t(h2[r * 2 * h + h + c]) + 1e-8f;
        v[i] = var;
        if (mout) mout[i] = m;
        z[i] = m + eps[i] * sqrtf(var);
    }
}

__global__ __launch_bounds__(256) void k_reparam_bwd(const float* h2, const float* eps, const float* v, const float* gz,
                                                     const float* gm, const float* gv, float* gh2, int64_t n, int h) {
    const int64_t total = n * h;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / h;
        const int c = (int)(i - r * h);
        const float g = gz ? gz[i] : 0.f;
        gh2[r * 2 * h + c] = g + (gm ? gm[i] : 0.f);
        const float raw = h2[r * 2 * h + h + c];
        float dv = g * eps[i] * 0.5f / sqrtf(v[i]) + (gv ? gv[i] : 0.f);
        // d softplus / d raw = sigmoid(raw) below the threshold, 1 above it (torch's softplus_backward)
        gh2[r * 2 * h + h + c] = raw > 20.f ? dv : dv * sigmoid_f(raw);
    }
}

__global__ __launch_bounds__(256) void k_axpby(int64_t n, const float* a, float alpha, const float* x, float beta,
                                               float* y) {
    const float s = a ? alpha * (*a) : alpha;
    GV_GRID_STRIDE(i, n) y[i] = beta == 0.f ? s * x[i] : s * x[i] + beta * y[i];
}

__global__ __launch_bounds__(256) void k_mul(int64_t n, const float* a, const float* b, float* out) {
    GV_GRID_STRIDE(i, n) out[i] = a[i] * b[i];
}

// several element-wise products in ONE launch (blockIdx.y = which): the mask folds of a MADE's layers
struct MulMulti {
    const float* a[GV_MUL_MULTI_MAX];
    const float* b[GV_MUL_MULTI_MAX];
    float* out[GV_MUL_MULTI_MAX];
    int64_t n[GV_MUL_MULTI_MAX];
};
__global__ __launch_bounds__(256) void k_mul_multi(const MulMulti p) {
    const int t = blockIdx.y;
    const float* a = p.a[t];
    const float* b = p.b[t];
    float* out = p.out[t];
    const int64_t n = p.n[t];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] * b[i];
}

__global__ __launch_bounds__(256) void k_iaf_fwd(const float* z, const float* net, int ld_net, const float* xold,
                                                 const int* colcount, float* xnew, int64_t n, int d) {
    const int64_t total = n * d;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        if (colcount[c] > 0) {
            const float mu = net[r * ld_net + c], al = net[r * ld_net + d + c];
            xnew[i] = z[i] * expf(al + mu);
        } else {
            xnew[i] = xold[i];
        }
    }
}

__global__ __launch_bounds__(256) void k_iaf_bwd(const float* z, const float* net, int ld_net, const int* colcount,
                                                 const float* gx, const float* gld, float* gz, float* gnet,
                                                 float* gxold, int64_t n, int d) {
    const int64_t total = n * d;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        const int cnt = colcount[c];
        const float g = gx[i];
        float g_mu = 0.f, g_al = gld ? gld[r] : 0.f, g_z = 0.f, g_old = g;
        if (cnt > 0) {
            const float e = expf(net[r * ld_net + d + c] + net[r * ld_net + c]);
            const float gc = g * (float)cnt;  // autograd gives the column's gradient to every duplicate index
            g_z = gc * e;
            g_mu = gc * z[i] * e;
            g_al += g_mu;
            g_old = 0.f;
        }
        gz[i] = g_z;
        gnet[r * 2 * d + c] = g_mu;
        gnet[r * 2 * d + d + c] = g_al;
        gxold[i] = g_old;
    }
}

// the same, four columns per thread (16-B accesses) and with dL/dz ADDED in place where the caller accumulates it over the passes
// (the fp32 MADE node: one launch instead of this kernel + an axpby per pass)
__global__ __launch_bounds__(256) void k_iaf_bwd_v4(const float* z, const float* net, int ld_net, const int* colcount,
                                                    const float* gx, const float* gld, float* gz, int gz_accumulate, float* gnet,
                                                    float* gxold, int64_t n, int d) {
    const int d4 = d >> 2;
    const int64_t total = n * d4;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / d4;
        const int c = (int)(i - r * d4) << 2;
        const int4 cnt = *reinterpret_cast<const int4*>(colcount + c);
        const float4 g = *reinterpret_cast<const float4*>(gx + r * d + c);
        const float4 zz = *reinterpret_cast<const float4*>(z + r * d + c);
        const float4 mu = *reinterpret_cast<const float4*>(net + r * ld_net + c), al = *reinterpret_cast<const float4*>(net + r * ld_net + d + c);
        float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gz_accumulate) old = *reinterpret_cast<const float4*>(gz + r * d + c);
        const float gl = gld ? gld[r] : 0.f;
        const int cn[4] = {cnt.x, cnt.y, cnt.z, cnt.w};
        const float gv[4] = {g.x, g.y, g.z, g.w}, zv[4] = {zz.x, zz.y, zz.z, zz.w}, mv[4] = {mu.x, mu.y, mu.z, mu.w}, av[4] = {al.x, al.y, al.z, al.w};
        float o_z[4], o_mu[4], o_al[4], o_old[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g_mu = 0.f, g_al = gl, g_z = 0.f, g_old = gv[e];
            if (cn[e] > 0) {
                const float ex = expf(av[e] + mv[e]);
                const float gc = gv[e] * (float)cn[e];
                g_z = gc * ex;
                g_mu = gc * zv[e] * ex;
                g_al += g_mu;
                g_old = 0.f;
            }
            o_z[e] = g_z; o_mu[e] = g_mu; o_al[e] = g_al; o_old[e] = g_old;
        }
        *reinterpret_cast<float4*>(gz + r * d + c) = make_float4(old.x + o_z[0], old.y + o_z[1], old.z + o_z[2], old.w + o_z[3]);
        *reinterpret_cast<float4*>(gnet + r * 2 * d + c) = make_float4(o_mu[0], o_mu[1], o_mu[2], o_mu[3]);
        *reinterpret_cast<float4*>(gnet + r * 2 * d + d + c) = make_float4(o_al[0], o_al[1], o_al[2], o_al[3]);
        *reinterpret_cast<float4*>(gxold + r * d + c) = make_float4(o_old[0], o_old[1], o_old[2], o_old[3]);
    }
}

__global__ __launch_bounds__(256) void k_rowsum(const float* x, int ld, int col0, int ncols, float* out, int64_t n) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        const float* src = x + r * (int64_t)ld + col0;
        float acc = 0.f;
        for (int c = lane; c < ncols; c += 64) acc += src[c];
        acc = wave_sum(acc);
        if (lane == 0) out[r] = acc;
    }
}

// flow_log_prob = mean over the rows that exist of the sum of the IAF blocks' per-row log-determinants (kgvae/model.py:116-123:
// log_det_sum = sum_flows log_det; flow_log_prob = mean(log_det_sum)).  ONE 1 024-thread workgroup, ordered: thread t adds its rows
// r = t, t + 1024, ... (per row the blocks in order), the 1 024 partials are added in a fixed tree.  rows_dev (optional): only
// the first *rows_dev of the n rows exist (static-shape batch); the mean divides by that count.
struct MeanRowsArgs { const float* x[8]; int count; };
constexpr int MEAN_ROWS_BLOCKS = 64;
// stage 1: block b sums rows [b * per, (b + 1) * per) (thread t its rows in order, the 256 thread sums in a fixed tree) -> part[b];
// stage 2 (one block): the 64 partials in order, divided by the row count.  Same result whatever the launch geometry.
__global__ __launch_bounds__(256) void k_mean_rows_part(const MeanRowsArgs a, int64_t n, const int* rows_dev, float* part) {
    __shared__ float sm[256];
    const int64_t live = rows_dev ? min((int64_t)*rows_dev, n) : n;
    const int64_t per = (n + MEAN_ROWS_BLOCKS - 1) / MEAN_ROWS_BLOCKS;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(live, r0 + per);
    float acc = 0.f;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        float s = 0.f;
        for (int i = 0; i < a.count; ++i) s += a.x[i][r];
        acc += s;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(64) void k_mean_rows_final(const float* part, int64_t n, const int* rows_dev, float* out) {
    const int64_t live = rows_dev ? min((int64_t)*rows_dev, n) : n;
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int b = 0; b < MEAN_ROWS_BLOCKS; ++b) acc += part[b];
        *out = acc / (float)live;
    }
}
// its backward: every block's per-row gradient is the same vector  g / rows  on the rows that exist, 0 on the padding rows
__global__ __launch_bounds__(256) void k_mean_rows_bwd(const float* g, int64_t len, int64_t n, const int* rows_dev, float* out) {
    const int64_t live = rows_dev ? min((int64_t)*rows_dev, n) : n;
    const float v = *g / (float)live;
    GV_GRID_STRIDE(i, len) out[i] = i < live ? v : 0.f;
}

__global__ __launch_bounds__(256) void k_reverse_cols(const float* x, float* out, int64_t n, int d) {
    const int64_t total = n * d;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        out[i] = x[r * d + (d - 1 - c)];
    }
}

// ---- counter-based RNG: every random draw of one forward pass in ONE launch ---------------------------------
// Philox4x32-10 (Salmon et al., SC'11): key = the 64-bit seed, counter = (group, stream, tick_lo, tick_hi); one call
// yields the 4 outputs of elements 4*group .. 4*group+3 of a job.  `tick` is a device-side step counter (state[1]),
// so a captured hipGraph draws fresh numbers on every replay; `stream` separates the jobs of one tick.
//   kind 0: keep mask, byte = (u32 >= floor(p_drop * 2^32))            (nn.Dropout's Bernoulli(1-p) decision)
//   kind 1: standard normals by Box-Muller, (u1, u2) = ((x0 + 1) * 2^-32, x1 * 2^-32)  (torch.randn_like)
struct RngJobs {
    void* ptr[GV_RNG_MAX_JOBS];
    long long n[GV_RNG_MAX_JOBS];
    long long group0[GV_RNG_MAX_JOBS + 1];     // prefix of ceil(n/4): job j owns global groups [group0[j], group0[j+1])
    int kind[GV_RNG_MAX_JOBS];
    uint32_t thresh[GV_RNG_MAX_JOBS];
    uint32_t stream[GV_RNG_MAX_JOBS];
    int n_jobs;
};

__global__ __launch_bounds__(256) void k_rng_fill(const uint64_t* state, const RngJobs jobs) {
    const uint64_t seed = state[0], tick = state[1];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), t0 = (uint32_t)tick, t1 = (uint32_t)(tick >> 32);
    const long long total = jobs.group0[jobs.n_jobs];
    for (long long gg = (long long)blockIdx.x * 256 + threadIdx.x; gg < total; gg += (long long)gridDim.x * 256) {
        int j = 0;
#pragma unroll
        for (int q = 1; q < GV_RNG_MAX_JOBS; ++q)
            if (q < jobs.n_jobs && gg >= jobs.group0[q]) j = q;
        const long long grp = gg - jobs.group0[j];
        const uint4 x = philox4x32_10(k0, k1, (uint32_t)grp, jobs.stream[j], t0, t1);
        const long long e0 = grp * 4, left = jobs.n[j] - e0;
        if (jobs.kind[j] == 0) {
            const uint32_t th = jobs.thresh[j];
            uint8_t* o = (uint8_t*)jobs.ptr[j] + e0;
            const uchar4 b = make_uchar4(x.x >= th, x.y >= th, x.z >= th, x.w >= th);
            if (left >= 4) {
                *reinterpret_cast<uchar4*>(o) = b;
            } else {
                o[0] = b.x;
                if (left > 1) o[1] = b.y;
                if (left > 2) o[2] = b.z;
            }
        } else {
            // hardware transcendentals: v_log_f32 (base 2), v_sin/v_cos_f32 (argument in revolutions) -- ~1e-6 absolute
            // on the result, far inside what noise needs; the precise libm sincosf tripled this kernel's time
            const float s = 2.3283064365386963e-10f;       // 2^-32
            const float r0 = sqrtf(-1.3862943611198906f * __log2f(((float)x.x + 1.f) * s));    // -2 ln u = -2 ln2 log2 u
            const float r1 = sqrtf(-1.3862943611198906f * __log2f(((float)x.z + 1.f) * s));
            const float a0 = (float)x.y * s, a1 = (float)x.w * s;                                 // revolutions in [0, 1]
            const float s0 = __builtin_amdgcn_sinf(a0), c0 = __builtin_amdgcn_cosf(a0);
            const float s1 = __builtin_amdgcn_sinf(a1), c1 = __builtin_amdgcn_cosf(a1);
            const float4 nrm = make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
            float* o = (float*)jobs.ptr[j] + e0;
            if (left >= 4) {
                *reinterpret_cast<float4*>(o) = nrm;
            } else {
                o[0] = nrm.x;
                if (left > 1) o[1] = nrm.y;
                if (left > 2) o[2] = nrm.z;
            }
        }
    }
}

__global__ void k_rng_tick(uint64_t* state) {
    if (blockIdx.x == 0 && threadIdx.x == 0) state[1] += 1;
}

// first pass of the gradient norm: per-block sums of g^2 into part[]; thread 0 of block 0 also advances the
// optimiser's step counter (nothing else reads it during this launch)
__global__ __launch_bounds__(256) void k_gradsq_part(const float* g, int64_t n, float* part, float* step) {
    __shared__ float sm[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        if (step && blockIdx.x == 0) *step += 1.f;
    }
}

// Adam over the flat arena.  n_part > 0: every block first adds the n_part partial sums of g^2 in index order (the
// second pass of the norm, no separate launch) and block 0 publishes the total; zero_g: the gradient is cleared as it
// is consumed (the next step's zero_grad).
__global__ __launch_bounds__(256) void k_adam(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq,
                                              const float* part, int n_part, float* sumsq_out, float max_norm, float lr,
                                              float b1, float b2, float eps, const float* step, int zero_g) {
    __shared__ float sm[4];
    float clip = 1.f;
    if (part && n_part > 0) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < n_part; i += 256) acc += part[i];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
        __syncthreads();
        const float tot = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        if (sumsq_out && blockIdx.x == 0 && threadIdx.x == 0) *sumsq_out = tot;
        if (max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(tot) + 1e-6f));
    } else if (sumsq && max_norm > 0.f) {
        clip = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    }
    const float t = *step;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    GV_GRID_STRIDE(i, n) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        if (zero_g) g[i] = 0.f;
    }
}

}  // namespace gv

using namespace gv;

#define GV_ST ((hipStream_t)stream)

extern "C" int gv_rgcn_epilogue_fwd(const float* agg, const float* addend, int act, const uint8_t* keep,
                                    float keep_scale, float* out, int64_t n_rows, int n_cols, void* stream) {
    GV_REQUIRE(agg && out, GV_ERR_NULL, "gv_rgcn_epilogue_fwd: NULL pointer");
    const int64_t n = n_rows * n_cols;
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_epilogue_fwd, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, agg, addend, act, keep,
                       keep_scale, out, n);
    return launch_status("gv_rgcn_epilogue_fwd");
}

extern "C" int gv_rgcn_epilogue_bwd(const float* out, const float* grad_out, int act, const uint8_t* keep,
                                    float keep_scale, float* g, int64_t n_rows, int n_cols, float* colsum_part,
                                    void* stream) {
    GV_REQUIRE(grad_out && g && (act == GV_ACT_NONE || out), GV_ERR_NULL, "gv_rgcn_epilogue_bwd: NULL pointer");
    const int64_t n = n_rows * n_cols;
    if (n <= 0) return GV_OK;
    if (colsum_part) {      // also write GV_EPILOGUE_COLSUM_SLICES row-slice partials of the column sums
        GV_REQUIRE(n_cols % 4 == 0 && n_cols <= 1024 && aligned16(grad_out) && aligned16(g) && (!out || aligned16(out)) &&
                       aligned16(colsum_part) && (!keep || (reinterpret_cast<uintptr_t>(keep) & 3u) == 0),
                   GV_ERR_ALIGN, "gv_rgcn_epilogue_bwd(colsum): needs n_cols %% 4 == 0, n_cols <= 1024 and aligned buffers");
        static_assert(EPI_SLICES == GV_EPILOGUE_COLSUM_SLICES, "header constant");
        hipLaunchKernelGGL(k_epilogue_bwd_colsum, dim3(EPI_SLICES), dim3(256), 0, GV_ST, out, grad_out, act, keep, keep_scale,
                           g, n_rows, n_cols, colsum_part);
        return launch_status("gv_rgcn_epilogue_bwd(colsum)");
    }
    hipLaunchKernelGGL(k_epilogue_bwd, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, out, grad_out, act, keep,
                       keep_scale, g, n);
    return launch_status("gv_rgcn_epilogue_bwd");
}

extern "C" int gv_gather_rows(const float* table, const int64_t* ids, float* out, int64_t n, int h, void* stream) {
    GV_REQUIRE(table && ids && out, GV_ERR_NULL, "gv_gather_rows: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n, 4)), dim3(256), 0, GV_ST, table, ids, out, n, h, (uint64_t*)nullptr);
    return launch_status("gv_gather_rows");
}

extern "C" int gv_gather_rows_rng_tick(const float* table, const int64_t* ids, float* out, int64_t n, int h,
                                       uint64_t* rng_state, void* stream) {
    GV_REQUIRE(table && ids && out && rng_state, GV_ERR_NULL, "gv_gather_rows_rng_tick: NULL pointer");
    GV_REQUIRE(n > 0, GV_ERR_SHAPE, "gv_gather_rows_rng_tick: n=%lld (the tick rides on a non-empty gather)", (long long)n);
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n, 4)), dim3(256), 0, GV_ST, table, ids, out, n, h, rng_state);
    return launch_status("gv_gather_rows_rng_tick");
}

extern "C" int gv_rng_tick(uint64_t* rng_state, void* stream) {
    GV_REQUIRE(rng_state, GV_ERR_NULL, "gv_rng_tick: NULL pointer");
    hipLaunchKernelGGL(k_rng_tick, dim3(1), dim3(64), 0, GV_ST, rng_state);
    return launch_status("gv_rng_tick");
}

extern "C" int gv_rng_fill(const uint64_t* rng_state, int n_jobs, void* const* ptrs, const int64_t* counts, const int32_t* kinds,
                           const float* drop_p, const uint32_t* streams, void* stream) {
    GV_REQUIRE(rng_state && ptrs && counts && kinds && drop_p && streams, GV_ERR_NULL, "gv_rng_fill: NULL pointer");
    GV_REQUIRE(n_jobs > 0 && n_jobs <= GV_RNG_MAX_JOBS, GV_ERR_SHAPE, "gv_rng_fill: n_jobs=%d (1..%d)", n_jobs, GV_RNG_MAX_JOBS);
    RngJobs jb;
    long long g = 0;
    for (int j = 0; j < GV_RNG_MAX_JOBS; ++j) {
        jb.group0[j] = g;
        if (j < n_jobs) {
            GV_REQUIRE(ptrs[j] && counts[j] > 0 && counts[j] < (1ll << 34), GV_ERR_SHAPE, "gv_rng_fill: job %d: bad buffer/count", j);
            GV_REQUIRE(kinds[j] == GV_RNG_KEEP_MASK || kinds[j] == GV_RNG_NORMAL, GV_ERR_SHAPE, "gv_rng_fill: job %d: kind %d", j, kinds[j]);
            GV_REQUIRE((reinterpret_cast<uintptr_t>(ptrs[j]) & (kinds[j] == GV_RNG_NORMAL ? 15u : 3u)) == 0, GV_ERR_ALIGN,
                       "gv_rng_fill: job %d: buffer alignment", j);
            GV_REQUIRE(kinds[j] == GV_RNG_NORMAL || (drop_p[j] >= 0.f && drop_p[j] < 1.f), GV_ERR_SHAPE, "gv_rng_fill: job %d: p=%f", j, drop_p[j]);
            jb.ptr[j] = ptrs[j]; jb.n[j] = counts[j]; jb.kind[j] = kinds[j]; jb.stream[j] = streams[j];
            const double th = (double)drop_p[j] * 4294967296.0;
            jb.thresh[j] = th >= 4294967295.0 ? 4294967295u : (uint32_t)th;
            g += (counts[j] + 3) / 4;
        } else {
            jb.ptr[j] = nullptr; jb.n[j] = 0; jb.kind[j] = 0; jb.stream[j] = 0; jb.thresh[j] = 0;
        }
    }
    jb.group0[GV_RNG_MAX_JOBS] = g;
    for (int j = n_jobs; j <= GV_RNG_MAX_JOBS; ++j) jb.group0[j] = g;
    jb.n_jobs = n_jobs;
    hipLaunchKernelGGL(k_rng_fill, dim3(grid_for(g, 1024)), dim3(256), 0, GV_ST, rng_state, jb);
    return launch_status("gv_rng_fill");
}

extern "C" int gv_scatter_add_rows(const float* grad_out, const int64_t* ids, float* grad_table, int64_t n, int h,
                                   void* stream) {
    GV_REQUIRE(grad_out && ids && grad_table, GV_ERR_NULL, "gv_scatter_add_rows: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_scatter_add_rows, dim3(grid_for(n, 4)), dim3(256), 0, GV_ST, grad_out, ids, grad_table, n, h);
    return launch_status("gv_scatter_add_rows");
}

extern "C" int gv_reparam_fwd(const float* h2, const float* eps, float* z, float* v, float* m_out, int64_t n, int h,
                              void* stream) {
    GV_REQUIRE(h2 && eps && z && v, GV_ERR_NULL, "gv_reparam_fwd: NULL pointer");
    if (n * h <= 0) return GV_OK;
    hipLaunchKernelGGL(k_reparam_fwd, dim3(grid_for(n * h, 1024)), dim3(256), 0, GV_ST, h2, eps, z, v, m_out, n, h);
    return launch_status("gv_reparam_fwd");
}

extern "C" int gv_reparam_bwd(const float* h2, const float* eps, const float* v, const float* gz, const float* gm,
                              const float* gv, float* grad_h2, int64_t n, int h, void* stream) {
    GV_REQUIRE(h2 && eps && v && grad_h2, GV_ERR_NULL, "gv_reparam_bwd: NULL pointer");
    if (n * h <= 0) return GV_OK;
    hipLaunchKernelGGL(k_reparam_bwd, dim3(grid_for(n * h, 1024)), dim3(256), 0, GV_ST, h2, eps, v, gz, gm, gv,
                       grad_h2, n, h);
    return launch_status("gv_reparam_bwd");
}

extern "C" int gv_axpby(int64_t n, const float* a, float alpha, const float* x, float beta, float* y, void* stream) {
    GV_REQUIRE(x && y, GV_ERR_NULL, "gv_axpby: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_axpby, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, n, a, alpha, x, beta, y);
    return launch_status("gv_axpby");
}

extern "C" int gv_mul(int64_t n, const float* a, const float* b, float* out, void* stream) {
    GV_REQUIRE(a && b && out, GV_ERR_NULL, "gv_mul: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_mul, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, n, a, b, out);
    return launch_status("gv_mul");
}

extern "C" int gv_mul_multi(int count, const float* const* a, const float* const* b, float* const* out, const int64_t* n,
                            void* stream) {
    GV_REQUIRE(count >= 0 && count <= GV_MUL_MULTI_MAX, GV_ERR_SHAPE, "gv_mul_multi: count=%d (at most %d)", count, GV_MUL_MULTI_MAX);
    if (count == 0) return GV_OK;
    GV_REQUIRE(a && b && out && n, GV_ERR_NULL, "gv_mul_multi: NULL table");
    MulMulti p;
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        GV_REQUIRE(n[i] >= 0 && (n[i] == 0 || (a[i] && b[i] && out[i])), GV_ERR_NULL, "gv_mul_multi: NULL pointer in entry %d", i);
        p.a[i] = a[i]; p.b[i] = b[i]; p.out[i] = out[i]; p.n[i] = n[i];
        nmax = n[i] > nmax ? n[i] : nmax;
    }
    if (nmax == 0) return GV_OK;
    hipLaunchKernelGGL(k_mul_multi, dim3(grid_for(nmax, 256), count), dim3(256), 0, GV_ST, p);
    return launch_status("gv_mul_multi");
}

extern "C" int gv_iaf_update_fwd(const float* z, const float* net, int ld_net, const float* x_old,
                                 const int32_t* colcount, float* x_new, int64_t n, int d, void* stream) {
    GV_REQUIRE(z && net && x_old && colcount && x_new, GV_ERR_NULL, "gv_iaf_update_fwd: NULL pointer");
    GV_REQUIRE(ld_net == 0 || ld_net >= 2 * d, GV_ERR_SHAPE, "gv_iaf_update_fwd: ld_net=%d (0 = broadcast one row, else >= 2d)", ld_net);
    if (n * d <= 0) return GV_OK;
    hipLaunchKernelGGL(k_iaf_fwd, dim3(grid_for(n * d, 1024)), dim3(256), 0, GV_ST, z, net, ld_net, x_old, colcount, x_new, n, d);
    return launch_status("gv_iaf_update_fwd");
}

extern "C" int gv_iaf_update_bwd(const float* z, const float* net, int ld_net, const int32_t* colcount,
                                 const float* g_xnew, const float* g_logdet, float* g_z, float* g_net, float* g_xold,
                                 int64_t n, int d, void* stream) {
    GV_REQUIRE(z && net && colcount && g_xnew && g_z && g_net && g_xold, GV_ERR_NULL, "gv_iaf_update_bwd: NULL pointer");
    GV_REQUIRE(ld_net == 0 || ld_net >= 2 * d, GV_ERR_SHAPE, "gv_iaf_update_bwd: ld_net=%d", ld_net);
    if (n * d <= 0) return GV_OK;
    hipLaunchKernelGGL(k_iaf_bwd, dim3(grid_for(n * d, 1024)), dim3(256), 0, GV_ST, z, net, ld_net, colcount, g_xnew,
                       g_logdet, g_z, g_net, g_xold, n, d);
    return launch_status("gv_iaf_update_bwd");
}

extern "C" int gv_iaf_update_bwd_acc(const float* z, const float* net, int ld_net, const int32_t* colcount, const float* g_xnew,
                                     const float* g_logdet, float* g_z, int gz_accumulate, float* g_net, float* g_xold, int64_t n, int d,
                                     void* stream) {
    GV_REQUIRE(z && net && colcount && g_xnew && g_z && g_net && g_xold, GV_ERR_NULL, "gv_iaf_update_bwd_acc: NULL pointer");
    GV_REQUIRE(d > 0 && d % 4 == 0 && ld_net >= 2 * d && ld_net % 4 == 0, GV_ERR_SHAPE, "gv_iaf_update_bwd_acc: d=%d ld_net=%d (multiples of 4)", d, ld_net);
    GV_REQUIRE(aligned16(z) && aligned16(net) && aligned16(colcount) && aligned16(g_xnew) && aligned16(g_z) && aligned16(g_net) && aligned16(g_xold),
               GV_ERR_ALIGN, "gv_iaf_update_bwd_acc: 16-B aligned operands");
    if (n * d <= 0) return GV_OK;
    hipLaunchKernelGGL(k_iaf_bwd_v4, dim3(grid_for(n * (d / 4), 2048)), dim3(256), 0, GV_ST, z, net, ld_net, colcount, g_xnew, g_logdet, g_z,
                       gz_accumulate, g_net, g_xold, n, d);
    return launch_status("gv_iaf_update_bwd_acc");
}

extern "C" int gv_rowsum(const float* x, int ld, int col0, int ncols, float* out, int64_t n, void* stream) {
    GV_REQUIRE(x && out, GV_ERR_NULL, "gv_rowsum: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_rowsum, dim3(grid_for(n, 4)), dim3(256), 0, GV_ST, x, ld, col0, ncols, out, n);
    return launch_status("gv_rowsum");
}

extern "C" int gv_mean_rows_multi(int count, const float* const* x, int64_t n, const int32_t* rows_dev, float* out, float* workspace,
                                  void* stream) {
    GV_REQUIRE(count >= 1 && count <= 8 && x && out && workspace && n > 0, GV_ERR_SHAPE, "gv_mean_rows_multi: count=%d n=%lld", count, (long long)n);
    MeanRowsArgs a;
    a.count = count;
    for (int i = 0; i < 8; ++i) a.x[i] = i < count ? x[i] : nullptr;
    for (int i = 0; i < count; ++i) GV_REQUIRE(x[i], GV_ERR_NULL, "gv_mean_rows_multi: NULL vector %d", i);
    hipLaunchKernelGGL(k_mean_rows_part, dim3(MEAN_ROWS_BLOCKS), dim3(256), 0, GV_ST, a, n, rows_dev, workspace);
    hipLaunchKernelGGL(k_mean_rows_final, dim3(1), dim3(64), 0, GV_ST, workspace, n, rows_dev, out);
    return launch_status("gv_mean_rows_multi");
}

extern "C" int gv_mean_rows_bwd(const float* g, int64_t len, int64_t n, const int32_t* rows_dev, float* out, void* stream) {
    GV_REQUIRE(g && out && n > 0 && len >= n, GV_ERR_NULL, "gv_mean_rows_bwd: NULL pointer / len < n");
    hipLaunchKernelGGL(k_mean_rows_bwd, dim3(grid_for(len, 1024)), dim3(256), 0, GV_ST, g, len, n, rows_dev, out);
    return launch_status("gv_mean_rows_bwd");
}

extern "C" int gv_reverse_cols(const float* x, float* out, int64_t n, int d, void* stream) {
    GV_REQUIRE(x && out, GV_ERR_NULL, "gv_reverse_cols: NULL pointer");
    if (n * d <= 0) return GV_OK;
    hipLaunchKernelGGL(k_reverse_cols, dim3(grid_for(n * d, 1024)), dim3(256), 0, GV_ST, x, out, n, d);
    return launch_status("gv_reverse_cols");
}

extern "C" int gv_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n,
                            const float* sumsq, float max_norm, float lr, float beta1, float beta2, float eps,
                            const float* step, void* stream) {
    GV_REQUIRE(p && g && exp_avg && exp_avg_sq && step, GV_ERR_NULL, "gv_adam_step: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_adam, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, p, const_cast<float*>(g), exp_avg, exp_avg_sq,
                       n, sumsq, (const float*)nullptr, 0, (float*)nullptr, max_norm, lr, beta1, beta2, eps, step, 0);
    return launch_status("gv_adam_step");
}

extern "C" int gv_clip_adam_step(float* p, float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float* workspace,
                                 float* sumsq_out, float max_norm, float lr, float beta1, float beta2, float eps,
                                 float* step, int zero_grad, void* stream) {
    GV_REQUIRE(p && g && exp_avg && exp_avg_sq && step && workspace, GV_ERR_NULL, "gv_clip_adam_step: NULL pointer");
    if (n <= 0) return GV_OK;
    int64_t nb64 = (n + 4095) / 4096;
    const int nb = (int)(nb64 < 1 ? 1 : (nb64 > 1024 ? 1024 : nb64));
    hipLaunchKernelGGL(k_gradsq_part, dim3(nb), dim3(256), 0, GV_ST, g, n, workspace, step);
    hipLaunchKernelGGL(k_adam, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, p, g, exp_avg, exp_avg_sq, n,
                       (const float*)nullptr, workspace, nb, sumsq_out, max_norm, lr, beta1, beta2, eps, step, zero_grad);
    return launch_status("gv_clip_adam_step");
}
