// K5 DistMult + BCE-with-logits, K6 fused reductions (regulariser, KL to the mixture prior).
// All scalar results are produced by an ordered two-pass reduction (per-block partials, then one
// block sums them in index order) -> bitwise reproducible, no float atomics.
#include "common.h"

namespace gv {

constexpr int RED_BLOCKS = 1024;  // partial slots of the two-pass scalar reductions

__device__ __forceinline__ float block_sum_256(float v, float* sm /*[4]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// out (+)= scale * sum(part[0..n))
__global__ __launch_bounds__(256) void k_final_sum(const float* part, int n, float scale, float* out, int accumulate) {
    __shared__ float sm[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    const float tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) *out = accumulate ? *out + scale * tot : scale * tot;
}

// the four loss terms' final sums and their weighted combination in one block:
//   scal = {s0*sum(p0), s1*sum(p1), s2*sum(p2), s3*sum(p3)},  loss = scal[0] + w1*scal[1] + w2*scal[2] + w3*scal[3]
__global__ __launch_bounds__(256) void k_loss_combine(const float* p0, int n0, float s0, const float* p1, int n1, float s1,
                                                      const float* p2, int n2, float s2, const float* p3, int n3, float s3,
                                                      float w1, float w2, float w3, float* scal, float* loss,
                                                      const int* rows_dev, float rows_host) {
    __shared__ float sm[4];
    const float* ps[4] = {p0, p1, p2, p3};
    const int ns[4] = {n0, n1, n2, n3};
    if (rows_dev) s2 *= rows_host / (float)*rows_dev;          // the KL mean runs over the rows that exist
    const float ss[4] = {s0, s1, s2, s3};
    float term[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
        if (ps[j])
            for (int i = threadIdx.x; i < ns[j]; i += 256) acc += ps[j][i];
        term[j] = ss[j] * block_sum_256(acc, sm);
    }
    if (threadIdx.x == 0) {
        if (scal) { scal[0] = term[0]; scal[1] = term[1]; scal[2] = term[2]; scal[3] = term[3]; }
        float v = term[0];
        if (p1) v += w1 * term[1];
        if (p2) v += w2 * term[2];
        if (p3) v += w3 * term[3];
        *loss = v;
    }
}

// ---- DistMult: 16 lanes per triplet (4 triplets per wave); three 800-B row gathers per triplet
// `order` (optional): the triplets sorted by subject.  Workgroups are dealt round-robin to the 8 XCDs; workgroup b takes
// positions of the range [T/8 * (b%8), T/8 * (b%8 + 1)) of that order, so the subject rows an XCD gathers come from one
// window of the embedding table and stay in its L2 (consecutive triplets often share the subject outright).
__global__ __launch_bounds__(256) void k_distmult_bce(const float* e, int ld_e, const float* w, int ld_w,
                                                      const int* trip, const int* order, const float* labels,
                                                      const float* bias, float* score, float* part, int64_t T, int h) {
    __shared__ float sm[4];
    const int sub = threadIdx.x & 15;
    const float bv = bias ? *bias : 0.f;
    const bool vec = (h % 4 == 0) && (ld_e % 4 == 0) && (ld_w % 4 == 0);
    float lsum = 0.f;
    // 16 triplets per block iteration; every lane runs the same trip count (shuffles below)
    const int n_x = gridDim.x >= 8 ? 8 : 1;                       // XCD-major split of the position range
    const int64_t per = (T + n_x - 1) / n_x;
    const int64_t lo = (blockIdx.x % n_x) * per, hi = min(T, lo + per);
    const int brank = blockIdx.x / n_x, bper = (gridDim.x + n_x - 1 - (blockIdx.x % n_x)) / n_x;   // blocks of this XCD
    // one triplet per 16-lane group and iteration; its dot product over h columns
    auto triplet_dot = [&](int64_t t) -> float {
        const int s = trip[3 * t], r = trip[3 * t + 1], o = trip[3 * t + 2];
        const float* es = e + (size_t)s * ld_e;
        const float* eo = e + (size_t)o * ld_e;
        const float* wr = w + (size_t)r * ld_w;
        float acc = 0.f;
        if (vec && h <= 256) {
            // all of a triplet's row pieces in flight at once (up to 4 x 3 16-B loads per lane), then the products
            float4 a[4], b[4], d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = sub * 4 + 64 * q;
                if (c < h) {
                    a[q] = *reinterpret_cast<const float4*>(es + c);
                    b[q] = *reinterpret_cast<const float4*>(wr + c);
                    d[q] = *reinterpret_cast<const float4*>(eo + c);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (sub * 4 + 64 * q < h) {
                    acc = fmaf(a[q].x * b[q].x, d[q].x, acc);
                    acc = fmaf(a[q].y * b[q].y, d[q].y, acc);
                    acc = fmaf(a[q].z * b[q].z, d[q].z, acc);
                    acc = fmaf(a[q].w * b[q].w, d[q].w, acc);
                }
            }
        } else if (vec) {
            for (int c = sub * 4; c < h; c += 64) {
                const float4 a = *reinterpret_cast<const float4*>(es + c);
                const float4 b = *reinterpret_cast<const float4*>(wr + c);
                const float4 d = *reinterpret_cast<const float4*>(eo + c);
                acc = fmaf(a.x * b.x, d.x, acc);
                acc = fmaf(a.y * b.y, d.y, acc);
                acc = fmaf(a.z * b.z, d.z, acc);
                acc = fmaf(a.w * b.w, d.w, acc);
            }
        } else {
            for (int c = sub; c < h; c += 16) acc = fmaf(es[c] * wr[c], eo[c], acc);
        }
        return acc;
    };
    auto finish = [&](float acc, bool live, int64_t t) {
        acc = row16_sum(acc);          // the triplet's 16 lanes are one DPP row
        if (live && sub == 0) {
            const float x = acc + bv;
            score[t] = x;
            const float y = labels[t];
            lsum += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        }
    };
    const int64_t stride = (int64_t)bper * 16;
    for (int64_t p0 = lo + (int64_t)brank * 16; p0 < hi; p0 += 2 * stride) {      // two triplets in flight per group
        const int64_t pa = p0 + (threadIdx.x >> 4), pb = pa + stride;
        const bool la = pa < hi, lb = pb < hi;
        const int64_t ta = la ? (order ? order[pa] : pa) : 0, tb = lb ? (order ? order[pb] : pb) : 0;
        const float xa = la ? triplet_dot(ta) : 0.f;
        const float xb = lb ? triplet_dot(tb) : 0.f;
        finish(xa, la, ta);
        finish(xb, lb, tb);
    }
    const float tot = block_sum_256(lsum, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// pos3 (optional, int32 [T][3]): where triplet t sits in the entity-incidence list (as subject, as object) and in the
// by-relation list of the backward's two K1 launches; d is then also written there, so those launches read their edge
// coefficient in their own order instead of through an index (one dependent load less per 64-edge batch).
__global__ __launch_bounds__(256) void k_bce_grad(const float* score, const float* labels, const float* gloss,
                                                  float* dscore, const int* pos3, float* d_inc, float* d_rel, float* part,
                                                  int64_t T) {
    __shared__ float sm[4];
    const float g = (gloss ? *gloss : 1.f) / (float)T;
    float acc = 0.f;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (int64_t)gridDim.x * 256) {
        const float d = g * (1.f / (1.f + expf(-score[t])) - labels[t]);
        dscore[t] = d;
        if (pos3) {
            d_inc[pos3[3 * t]] = d;
            d_inc[pos3[3 * t + 1]] = d;
            d_rel[pos3[3 * t + 2]] = d;
        }
        acc += d;
    }
    const float tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_sumsq_part(const float* x, int64_t n, float* part) {
    __shared__ float sm[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(x[i], x[i], acc);
    const float tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// ---- KL -----------------------------------------------------------------------------------------
// workspace layout (floats): mix[3][k*h] (mu_j, 1/(2 v_j), log sqrt v_j + log sqrt 2pi) | terms[n] | part[...]
constexpr float LOG_SQRT_2PI = 0.9189385332046727f;

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(256) void k_kl_mix(const float* z_pre, int k, int h, float* mix) {
    const int total = k * h;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const float v = softplus_t(z_pre[total + i]) + 1e-8f;
        mix[i] = z_pre[i];
        mix[total + i] = 1.f / (2.f * v);
        mix[2 * total + i] = logf(sqrtf(v)) + LOG_SQRT_2PI;
    }
}

constexpr int KL_KMAX = 64;   // mixture components (lane j keeps component j's log-density)

// one wave per node; mixture table staged in LDS once per block; CPL = columns per lane (h <= 64*CPL)
template <int CPL>
__global__ __launch_bounds__(256) void k_kl_fwd(const float* z, const float* m, int ld_m, const float* v,
                                                const float* mix, const float* flp, float* resp, float* part,
                                                int64_t n, int h, int k, const int* rows_dev, const float* h2 = nullptr,
                                                const float* eps = nullptr, float* z_out = nullptr, float* v_out = nullptr,
                                                float* m_out = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float wsum[4];
    if (rows_dev) n = *rows_dev;               // rows beyond it are padding of a static-shape batch
    __shared__ float lconst[KL_KMAX];          // sum_c (log sqrt v_jc + log sqrt 2 pi): independent of the node
    const int kh = k * h;
    if ((kh & 3) == 0) {
        for (int i = threadIdx.x; i < 2 * kh / 4; i += 256)
            reinterpret_cast<float4*>(sm)[i] = reinterpret_cast<const float4*>(mix)[i];
    } else {
        for (int i = threadIdx.x; i < 2 * kh; i += 256) sm[i] = mix[i];
    }
    for (int j = threadIdx.x >> 6; j < k; j += 4) {
        float t = 0.f;
        for (int c = threadIdx.x & 63; c < h; c += 64) t += mix[2 * kh + j * h + c];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) lconst[j] = t;
    }
    __syncthreads();
    const float* mu = sm;
    const float* i2v = sm + kh;
    const int lane = threadIdx.x & 63;
    const float fl = flp ? *flp : 0.f;
    const float logk = logf((float)k);
    float my_terms = 0.f;                       // lane 0 of each wave: sum of its nodes' terms, in node order
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        float zz[CPL];
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            zz[i] = 0.f;
            if (c < h) {
                float mm, vv;
                if (h2) {      // fused reparameterisation (K3): (m, v, z) are made here from h2 and eps, and stored for the rest of the step
                    mm = h2[r * 2 * h + c];
                    vv = softplus_t(h2[r * 2 * h + h + c]) + 1e-8f;
                    zz[i] = mm + eps[r * h + c] * sqrtf(vv);
                    z_out[r * h + c] = zz[i];
                    v_out[r * h + c] = vv;
                    if (m_out) m_out[r * h + c] = mm;
                } else {
                    zz[i] = z[r * h + c];
                    mm = m[r * ld_m + c];
                    vv = v[r * h + c];
                }
                const float d = zz[i] - mm;
                a += -(d * d) / (2.f * vv) - logf(sqrtf(vv)) - LOG_SQRT_2PI;
            }
        }
        a = wave_sum(a);
        float my_l = -INFINITY;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = lane + 64 * i;
                if (c < h) {
                    const float dj = zz[i] - mu[j * h + c];
                    acc = fmaf(-(dj * dj), i2v[j * h + c], acc);
                }
            }
            acc = wave_sum(acc) - lconst[j];
            if (lane == j) my_l = acc;
        }
        const float mx = wave_max(my_l);
        const float e = lane < k ? expf(my_l - mx) : 0.f;
        const float se = wave_sum(e);
        if (lane < k) resp[r * k + lane] = e / se;
        my_terms += a + fl - (mx + logf(se) - logk);
    }
    if (lane == 0) wsum[threadIdx.x >> 6] = my_terms;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// The same pass with a lane owning FOUR CONSECUTIVE columns per group (h % 4 == 0, G = ceil(h / 256) groups): 16-B loads and
// stores of the node rows, one ds_read_b128 per operand and component instead of four ds_read_b32, no per-element column
// guards (a group is inside or outside as a whole).  The per-lane kernel above issues ~1700 VALU instructions per node -- it is
// VALU-bound, 4 cycles per wave64 instruction -- most of them in the component loop; this form halves that.  Per element the
// same expressions; the sums over a row's columns run in another (fixed) order.
template <int G>
__global__ __launch_bounds__(256) void k_kl_fwd_v4(const float* z, const float* m, int ld_m, const float* v,
                                                   const float* mix, const float* flp, float* resp, float* part,
                                                   int64_t n, int h, int k, const int* rows_dev, const float* h2,
                                                   const float* eps, float* z_out, float* v_out, float* m_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float wsum[4];
    __shared__ float lconst[KL_KMAX];
    if (rows_dev) n = *rows_dev;
    const int kh = k * h;
    for (int i = threadIdx.x; i < 2 * kh / 4; i += 256) reinterpret_cast<float4*>(sm)[i] = reinterpret_cast<const float4*>(mix)[i];
    for (int j = threadIdx.x >> 6; j < k; j += 4) {
        float t = 0.f;
        for (int c = threadIdx.x & 63; c < h; c += 64) t += mix[2 * kh + j * h + c];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) lconst[j] = t;
    }
    __syncthreads();
    const float* mu = sm;
    const float* i2v = sm + kh;
    const int lane = threadIdx.x & 63;
    const float fl = flp ? *flp : 0.f;
    const float logk = logf((float)k);
    float my_terms = 0.f;
    bool on[G];
    int col[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        col[g] = 4 * (lane + 64 * g);
        on[g] = col[g] < h;
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        float zz[G][4];
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) zz[g][i] = 0.f;
            if (!on[g]) continue;
            float mm[4], vv[4];
            if (h2) {      // fused reparameterisation (K3)
                const float4 m4 = *reinterpret_cast<const float4*>(h2 + r * 2 * h + col[g]);
                const float4 r4 = *reinterpret_cast<const float4*>(h2 + r * 2 * h + h + col[g]);
                const float4 e4 = *reinterpret_cast<const float4*>(eps + r * h + col[g]);
                const float raw[4] = {r4.x, r4.y, r4.z, r4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
                mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vv[i] = softplus_t(raw[i]) + 1e-8f;
                    zz[g][i] = mm[i] + ee[i] * sqrtf(vv[i]);
                }
                *reinterpret_cast<float4*>(z_out + r * h + col[g]) = make_float4(zz[g][0], zz[g][1], zz[g][2], zz[g][3]);
                *reinterpret_cast<float4*>(v_out + r * h + col[g]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                if (m_out) *reinterpret_cast<float4*>(m_out + r * h + col[g]) = m4;
            } else {
                const float4 z4 = *reinterpret_cast<const float4*>(z + r * h + col[g]);
                const float4 m4 = *reinterpret_cast<const float4*>(m + r * ld_m + col[g]);
                const float4 v4 = *reinterpret_cast<const float4*>(v + r * h + col[g]);
                zz[g][0] = z4.x; zz[g][1] = z4.y; zz[g][2] = z4.z; zz[g][3] = z4.w;
                mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
                vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = zz[g][i] - mm[i];
                a += -(d * d) / (2.f * vv[i]) - logf(sqrtf(vv[i])) - LOG_SQRT_2PI;
            }
        }
        a = wave_sum(a);
        float my_l = -INFINITY;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!on[g]) continue;
                const float4 mu4 = *reinterpret_cast<const float4*>(mu + j * h + col[g]);
                const float4 iv4 = *reinterpret_cast<const float4*>(i2v + j * h + col[g]);
                const float d0 = zz[g][0] - mu4.x, d1 = zz[g][1] - mu4.y, d2 = zz[g][2] - mu4.z, d3 = zz[g][3] - mu4.w;
                acc = fmaf(-(d0 * d0), iv4.x, acc);
                acc = fmaf(-(d1 * d1), iv4.y, acc);
                acc = fmaf(-(d2 * d2), iv4.z, acc);
                acc = fmaf(-(d3 * d3), iv4.w, acc);
            }
            acc = wave_sum(acc) - lconst[j];
            if (lane == j) my_l = acc;
        }
        const float mx = wave_max(my_l);
        const float e = lane < k ? expf(my_l - mx) : 0.f;
        const float se = wave_sum(e);
        if (lane < k) resp[r * k + lane] = e / se;
        my_terms += a + fl - (mx + logf(se) - logk);
    }
    if (lane == 0) wsum[threadIdx.x >> 6] = my_terms;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// FOUR NODES PER WAVE (h <= 64 G, h % 4 == 0, k <= 16): a row of 16 lanes owns one node, lane l of the row the float4 columns
// l, l + 16, ... (G = 4 or 8 of them).  The form above spends most of its ~850 VALU instructions per node on k + 2 whole-wave reductions
// (four DPP steps, four v_readlane and three adds each, serialised by the loop over the components); a row's reduction is the
// four DPP steps alone, the four nodes of a wave share every instruction, and the component loop's LDS reads serve four nodes.
// (Measured at FB15k-237 size: 34.4 -> 32.4 us.  The kernel is bound by its per-element arithmetic -- ablations, NOTES.md round 4:
// softplus + sqrt 13 us, the component loop 5, the own-density term 4, the three row stores 4 of 37 -- and fewer, longer workgroups
// with the next rows in flight are SLOWER: 256 / 512 / 768 workgroups 54 / 40 / 34 us.)  Per element the same expressions as k_kl_fwd / k_kl_fwd_v4; the sums over a node's columns and over the nodes run in another
// (fixed) order.
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_1032>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_2301>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return v;
}

template <int G>
__global__ __launch_bounds__(256) void k_kl_fwd_v5(const float* z, const float* m, int ld_m, const float* v,
                                                   const float* mix, const float* flp, float* resp, float* part,
                                                   int64_t n, int h, int k, const int* rows_dev, const float* h2,
                                                   const float* eps, float* z_out, float* v_out, float* m_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float wsum[4];
    __shared__ float lconst[KL_KMAX];
    if (rows_dev) n = *rows_dev;
    const int kh = k * h;
    for (int i = threadIdx.x; i < 2 * kh / 4; i += 256) reinterpret_cast<float4*>(sm)[i] = reinterpret_cast<const float4*>(mix)[i];
    for (int j = threadIdx.x >> 6; j < k; j += 4) {
        float t = 0.f;
        for (int c = threadIdx.x & 63; c < h; c += 64) t += mix[2 * kh + j * h + c];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) lconst[j] = t;
    }
    __syncthreads();
    const float* mu = sm;
    const float* i2v = sm + kh;
    const int lane = threadIdx.x & 63, l16 = lane & 15, rw = lane >> 4, wv = threadIdx.x >> 6;
    const float fl = flp ? *flp : 0.f;
    const float logk = logf((float)k);
    bool on[G];
    int col[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        col[g] = 4 * (l16 + 16 * g);
        on[g] = col[g] < h;
    }
    float my_terms = 0.f;                       // every lane of a row: the sum of the row's nodes' terms, in node order
    for (int64_t base = (int64_t)blockIdx.x * 16 + wv * 4; base < n; base += (int64_t)gridDim.x * 16) {      // (wave-uniform)
        const bool live = base + rw < n;
        const int64_t r = live ? base + rw : n - 1;      // rows past the end shadow the last node: uniform control flow, no effects
        float zz[G][4];
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) zz[g][i] = 0.f;
            if (!on[g]) continue;
            float mm[4], vv[4];
            if (h2) {      // fused reparameterisation (K3)
                const float4 m4 = *reinterpret_cast<const float4*>(h2 + r * 2 * h + col[g]);
                const float4 r4 = *reinterpret_cast<const float4*>(h2 + r * 2 * h + h + col[g]);
                const float4 e4 = *reinterpret_cast<const float4*>(eps + r * h + col[g]);
                const float raw[4] = {r4.x, r4.y, r4.z, r4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
                mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vv[i] = softplus_t(raw[i]) + 1e-8f;
                    zz[g][i] = mm[i] + ee[i] * sqrtf(vv[i]);
                }
                if (live) {
                    *reinterpret_cast<float4*>(z_out + r * h + col[g]) = make_float4(zz[g][0], zz[g][1], zz[g][2], zz[g][3]);
                    *reinterpret_cast<float4*>(v_out + r * h + col[g]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    if (m_out) *reinterpret_cast<float4*>(m_out + r * h + col[g]) = m4;
                }
            } else {
                const float4 z4 = *reinterpret_cast<const float4*>(z + r * h + col[g]);
                const float4 m4 = *reinterpret_cast<const float4*>(m + r * ld_m + col[g]);
                const float4 v4 = *reinterpret_cast<const float4*>(v + r * h + col[g]);
                zz[g][0] = z4.x; zz[g][1] = z4.y; zz[g][2] = z4.z; zz[g][3] = z4.w;
                mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
                vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = zz[g][i] - mm[i];
                a += -(d * d) / (2.f * vv[i]) - logf(sqrtf(vv[i])) - LOG_SQRT_2PI;
            }
        }
        a = row16_sum(a);
        float my_l = -INFINITY;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!on[g]) continue;
                const float4 mu4 = *reinterpret_cast<const float4*>(mu + j * h + col[g]);
                const float4 iv4 = *reinterpret_cast<const float4*>(i2v + j * h + col[g]);
                const float d0 = zz[g][0] - mu4.x, d1 = zz[g][1] - mu4.y, d2 = zz[g][2] - mu4.z, d3 = zz[g][3] - mu4.w;
                acc = fmaf(-(d0 * d0), iv4.x, acc);
                acc = fmaf(-(d1 * d1), iv4.y, acc);
                acc = fmaf(-(d2 * d2), iv4.z, acc);
                acc = fmaf(-(d3 * d3), iv4.w, acc);
            }
            acc = row16_sum(acc) - lconst[j];
            if (l16 == j) my_l = acc;
        }
        const float mx = row16_max(my_l);
        const float e = l16 < k ? expf(my_l - mx) : 0.f;
        const float se = row16_sum(e);
        if (live && l16 < k) resp[r * k + l16] = e / se;
        if (live) my_terms += a + fl - (mx + logf(se) - logk);
    }
    const float t = (rl_bcast_f(my_terms, 0) + rl_bcast_f(my_terms, 16)) + (rl_bcast_f(my_terms, 32) + rl_bcast_f(my_terms, 48));
    if (lane == 0) wsum[wv] = t;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// per-node gradients gz, gm, gv (scaled by *gkl / n)
__global__ __launch_bounds__(256) void k_kl_bwd_nodes(const float* z, const float* m, int ld_m, const float* v,
                                                      const float* mix, const float* resp, const float* gkl, float gscale,
                                                      float z_extra, float* gz, float* gm, float* gv, int64_t n, int h, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int kh = k * h;
    if ((kh & 3) == 0) {
        for (int i = threadIdx.x; i < 2 * kh / 4; i += 256)
            reinterpret_cast<float4*>(sm)[i] = reinterpret_cast<const float4*>(mix)[i];
    } else {
        for (int i = threadIdx.x; i < 2 * kh; i += 256) sm[i] = mix[i];
    }
    __syncthreads();
    const float* mu = sm;
    const float* i2v = sm + kh;
    const int lane = threadIdx.x & 63;
    const float cg = gscale * (gkl ? *gkl : 1.f) / (float)n;
    const float cz = z_extra * (gkl ? *gkl : 1.f);
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        const float* rp = resp + r * k;
        for (int c = lane; c < h; c += 64) {
            const float zz = z[r * h + c], mm = m[r * ld_m + c], vv = v[r * h + c];
            const float d = zz - mm;
            float mixg = 0.f;
            for (int j = 0; j < k; ++j) mixg += rp[j] * (zz - mu[j * h + c]) * 2.f * i2v[j * h + c];
            gz[r * h + c] = fmaf(cz, zz, cg * (-d / vv + mixg));
            gm[r * h + c] = cg * (d / vv);
            gv[r * h + c] = cg * (d * d / (2.f * vv * vv) - 0.5f / vv);
        }
    }
}

// Fused KL backward for k <= KT components: ONE pass over z / m / v / resp produces the per-node gradients AND the
// row-slice partial sums of the mixture-parameter gradients (the two-kernel form reads z ten more times, once per
// component).  Grid (64-column tile, row slice); lane <-> column, the block's 4 waves interleave the slice's rows; a
// lane keeps its column's k (mu_j, 1/(2 v_j)) pairs and 2k accumulators in registers; the row's k responsibilities are a
// wave-uniform read.  The 4 waves' accumulators are combined through LDS in wave order and written as slice `blockIdx.y`
// of `part`, which k_kl_bwd_mix_final then sums in slice order (deterministic).
template <int KT>
__global__ __launch_bounds__(256) void k_kl_bwd_fused(const float* z, const float* m, int ld_m, const float* v, const float* mix,
                                                      const float* resp, const float* gkl, float gscale, float z_extra,
                                                      float* gz, float* gm, float* gv, float* part, int64_t n, int h, int k,
                                                      const int* rows_dev, const float* h2 = nullptr, const float* eps = nullptr,
                                                      const float* gz_up = nullptr, float* gh2 = nullptr) {
    __shared__ float sm[2 * KT][4][64];
    // static-shape batches: rows [*rows_dev, n) are padding -- zero gradients, no share in the sums; the 1/n factors of the
    // two means (KL and the regulariser's z_extra) become 1/*rows_dev
    const int64_t n_real = rows_dev ? (int64_t)*rows_dev : n;
    const float refit = (float)n / (float)n_real;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // uniform: resp rows become scalar loads
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < h;
    const int kh = k * h;
    const int nsl = gridDim.y;
    const int64_t per = (n + nsl - 1) / nsl;
    const int64_t r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    const float cg = gscale * (gkl ? *gkl : 1.f) / (float)n * refit;
    const float cz = z_extra * (gkl ? *gkl : 1.f) * refit;   // an extra upstream * z_extra * z term on gz (the regulariser)
    float mu[KT], i2[KT], amu[KT], av[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        mu[j] = (ok && j < k) ? mix[j * h + c] : 0.f;
        i2[j] = (ok && j < k) ? mix[kh + j * h + c] : 0.f;
        amu[j] = 0.f;
        av[j] = 0.f;
    }
    // four rows in flight per wave (independent loads issued together: the loop is latency-bound)
    constexpr int RU = 4;
    for (int64_t rb = r0 + w; rb < r1; rb += 4 * RU) {
        float zz[RU], mm[RU], vv[RU];
        // the 4 rows' responsibilities: lanes 16u .. 16u+k-1 load row u's k values (one 64-lane load), read back by v_readlane
        float rv = 0.f;
        {
            const int64_t rr = rb + 4 * (lane >> 4);
            if ((lane & 15) < k && rr < r1 && rr < n_real) rv = resp[rr * k + (lane & 15)];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t r = rb + 4 * u;
            zz[u] = 0.f; mm[u] = 0.f; vv[u] = 1.f;
            if (ok && r < r1) {
                zz[u] = z[r * h + c];
                mm[u] = m[r * ld_m + c];
                vv[u] = v[r * h + c];
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t r = rb + 4 * u;
            if (r >= r1) break;                            // wave-uniform
            if (r >= n_real) {                             // padding row (wave-uniform)
                if (ok) { gz[r * h + c] = 0.f; gm[r * h + c] = 0.f; gv[r * h + c] = 0.f; }
                continue;
            }
            const float d = zz[u] - mm[u];
            float mixg = 0.f;
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if (j < k) {
                    const float ra = rl_bcast_f(rv, 16 * u + j);
                    const float dj = zz[u] - mu[j];
                    mixg = fmaf(ra * dj, 2.f * i2[j], mixg);
                    amu[j] = fmaf(ra, dj, amu[j]);
                    av[j] = fmaf(ra, dj * dj * 2.f * i2[j] * i2[j] - i2[j], av[j]);
                }
            }
            if (ok) {
                const float gzk = fmaf(cz, zz[u], cg * (-d / vv[u] + mixg));
                const float gmk = cg * (d / vv[u]);
                const float gvk = cg * (d * d / (2.f * vv[u] * vv[u]) - 0.5f / vv[u]);
                if (gh2) {     // fused reparameterisation backward (K3): the three node gradients never reach memory
                    const float g = gzk + (gz_up ? gz_up[r * h + c] : 0.f);
                    const float raw = h2[r * 2 * h + h + c];
                    const float dv = g * eps[r * h + c] * 0.5f / sqrtf(vv[u]) + gvk;
                    gh2[r * 2 * h + c] = g + gmk;
                    gh2[r * 2 * h + h + c] = raw > 20.f ? dv : dv / (1.f + expf(-raw));
                } else {
                    gz[r * h + c] = gzk;
                    gm[r * h + c] = gmk;
                    gv[r * h + c] = gvk;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        sm[j][w][lane] = amu[j];
        sm[KT + j][w][lane] = av[j];
    }
    __syncthreads();
    if (w == 0 && ok) {
        for (int j = 0; j < k; ++j) {
            const float gmu = (sm[j][0][lane] + sm[j][1][lane]) + (sm[j][2][lane] + sm[j][3][lane]);
            const float gvv = (sm[KT + j][0][lane] + sm[KT + j][1][lane]) + (sm[KT + j][2][lane] + sm[KT + j][3][lane]);
            part[((size_t)blockIdx.y * 2 * k + j) * h + c] = -gmu * 2.f * i2[j];       // same terms as k_kl_bwd_mix_part
            part[((size_t)blockIdx.y * 2 * k + k + j) * h + c] = -gvv;
        }
    }
}

// The fused backward with a lane on FOUR CONSECUTIVE COLUMNS (h % 4 == 0, k <= KT; column tiles of 256 on grid.x): one wave covers a
// whole row (tile), so the
// node rows move in 16-B accesses (the lane-per-column form above reads and writes 256 B per wave-instruction, and its fourth
// 64-column tile of a 200-wide row keeps 8 of 64 lanes busy), the row's responsibilities are read once instead of once per tile,
// and the mixture table comes from LDS.  Grid = KL_SLICES workgroups of 8 waves that interleave the slice's rows, two rows in
// flight per wave; the waves' 2 k partial sums are combined through LDS two components at a time, in wave order.  Per element the
// same expressions; a slice's rows are summed in another (fixed) order.
constexpr int KLB_WAVES = 8, KLB_KR = 2;
template <int KT>
__global__ __launch_bounds__(64 * KLB_WAVES) void k_kl_bwd_cols4(const float* z, const float* m, int ld_m, const float* v, const float* mix,
                                                                  const float* resp, const float* gkl, float gscale, float z_extra,
                                                                  float* gz, float* gm, float* gv, float* part, int64_t n, int h, int k,
                                                                  const int* rows_dev, const float* h2, const float* eps,
                                                                  const float* gz_up, float* gh2) {
    extern __shared__ __attribute__((aligned(16))) float4 sm4[];      // [2][k * h / 4] (mu_j, 1/(2 v_j)), then red[2 KR][waves][64]
    const int64_t n_real = rows_dev ? (int64_t)*rows_dev : n;
    const float refit = (float)n / (float)n_real;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h4 = h >> 2, kh4 = k * h4;
    const int col = (int)blockIdx.x * 256 + 4 * lane;            // grid.x: tiles of 256 columns (h <= 256: one)
    const bool ok = col < h;
    for (int i = threadIdx.x; i < 2 * kh4; i += 64 * KLB_WAVES) sm4[i] = reinterpret_cast<const float4*>(mix)[i];
    const float4* smu = sm4 + (ok ? (col >> 2) : 0);
    const float4* si2 = smu + kh4;
    float4* red = sm4 + 2 * kh4;
    const int nsl = gridDim.y;
    const int64_t per = (n + nsl - 1) / nsl;
    const int64_t r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    const float cg = gscale * (gkl ? *gkl : 1.f) / (float)n * refit;
    const float cz = z_extra * (gkl ? *gkl : 1.f) * refit;
    float amu[KT][4], av[KT][4];
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) { amu[j][i] = 0.f; av[j][i] = 0.f; }
    __syncthreads();
    constexpr int RU = 2;
    for (int64_t rb = r0 + w; rb < r1; rb += KLB_WAVES * RU) {
        // the rows' responsibilities: lanes 16u .. 16u + k - 1 load row u's k values, read back by v_readlane
        float rv = 0.f;
        {
            const int64_t rr = rb + KLB_WAVES * (lane >> 4);
            if ((lane & 15) < k && (lane >> 4) < RU && rr < r1 && rr < n_real) rv = resp[rr * k + (lane & 15)];
        }
        float zz[RU][4], mm[RU][4], vv[RU][4], rawv[RU][4], ee[RU][4], gu[RU][4];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t r = rb + KLB_WAVES * u;
            float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = z4, v4 = make_float4(1.f, 1.f, 1.f, 1.f), r4 = z4, e4 = z4, g4 = z4;
            if (ok && r < r1) {
                z4 = *reinterpret_cast<const float4*>(z + r * h + col);
                m4 = *reinterpret_cast<const float4*>(m + r * ld_m + col);
                v4 = *reinterpret_cast<const float4*>(v + r * h + col);
                if (gh2) {
                    r4 = *reinterpret_cast<const float4*>(h2 + r * 2 * h + h + col);
                    e4 = *reinterpret_cast<const float4*>(eps + r * h + col);
                    if (gz_up) g4 = *reinterpret_cast<const float4*>(gz_up + r * h + col);
                }
            }
            zz[u][0] = z4.x; zz[u][1] = z4.y; zz[u][2] = z4.z; zz[u][3] = z4.w;
            mm[u][0] = m4.x; mm[u][1] = m4.y; mm[u][2] = m4.z; mm[u][3] = m4.w;
            vv[u][0] = v4.x; vv[u][1] = v4.y; vv[u][2] = v4.z; vv[u][3] = v4.w;
            rawv[u][0] = r4.x; rawv[u][1] = r4.y; rawv[u][2] = r4.z; rawv[u][3] = r4.w;
            ee[u][0] = e4.x; ee[u][1] = e4.y; ee[u][2] = e4.z; ee[u][3] = e4.w;
            gu[u][0] = g4.x; gu[u][1] = g4.y; gu[u][2] = g4.z; gu[u][3] = g4.w;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t r = rb + KLB_WAVES * u;
            if (r >= r1) break;                            // wave-uniform
            if (r >= n_real) {                             // padding row (wave-uniform): zero gradients, no share in the sums
                if (ok) {
                    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gh2) {
                        *reinterpret_cast<float4*>(gh2 + r * 2 * h + col) = zero;
                        *reinterpret_cast<float4*>(gh2 + r * 2 * h + h + col) = zero;
                    } else {
                        *reinterpret_cast<float4*>(gz + r * h + col) = zero;
                        *reinterpret_cast<float4*>(gm + r * h + col) = zero;
                        *reinterpret_cast<float4*>(gv + r * h + col) = zero;
                    }
                }
                continue;
            }
            float mixg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if (j < k) {
                    const float ra = rl_bcast_f(rv, 16 * u + j);
                    const float4 mu4 = smu[j * h4], i24 = si2[j * h4];
                    const float mu_[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, i2_[4] = {i24.x, i24.y, i24.z, i24.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float dj = zz[u][i] - mu_[i];
                        mixg[i] = fmaf(ra * dj, 2.f * i2_[i], mixg[i]);
                        amu[j][i] = fmaf(ra, dj, amu[j][i]);
                        av[j][i] = fmaf(ra, dj * dj * 2.f * i2_[i] * i2_[i] - i2_[i], av[j][i]);
                    }
                }
            }
            if (ok) {
                float o0[4], o1[4], o2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = zz[u][i] - mm[u][i];
                    const float gzk = fmaf(cz, zz[u][i], cg * (-d / vv[u][i] + mixg[i]));
                    const float gmk = cg * (d / vv[u][i]);
                    const float gvk = cg * (d * d / (2.f * vv[u][i] * vv[u][i]) - 0.5f / vv[u][i]);
                    if (gh2) {     // fused reparameterisation backward (K3)
                        const float g = gzk + gu[u][i];
                        const float dv = g * ee[u][i] * 0.5f / sqrtf(vv[u][i]) + gvk;
                        o0[i] = g + gmk;
                        o1[i] = rawv[u][i] > 20.f ? dv : dv / (1.f + expf(-rawv[u][i]));
                    } else {
                        o0[i] = gzk; o1[i] = gmk; o2[i] = gvk;
                    }
                }
                if (gh2) {
                    *reinterpret_cast<float4*>(gh2 + r * 2 * h + col) = make_float4(o0[0], o0[1], o0[2], o0[3]);
                    *reinterpret_cast<float4*>(gh2 + r * 2 * h + h + col) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                } else {
                    *reinterpret_cast<float4*>(gz + r * h + col) = make_float4(o0[0], o0[1], o0[2], o0[3]);
                    *reinterpret_cast<float4*>(gm + r * h + col) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                    *reinterpret_cast<float4*>(gv + r * h + col) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                }
            }
        }
    }
    // the waves' sums, KR components per round: red[(2 c + kind)][wave][lane]; wave q < 2 KR adds entry q's 8 partials in wave order
    for (int j0 = 0; j0 < k; j0 += KLB_KR) {
        __syncthreads();                                   // (the previous round's readers are done; round 0: the row loop's LDS reads)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if (j >= j0 && j < j0 + KLB_KR && j < k) {
                red[((2 * (j - j0)) * KLB_WAVES + w) * 64 + lane] = make_float4(amu[j][0], amu[j][1], amu[j][2], amu[j][3]);
                red[((2 * (j - j0) + 1) * KLB_WAVES + w) * 64 + lane] = make_float4(av[j][0], av[j][1], av[j][2], av[j][3]);
            }
        }
        __syncthreads();
        const int j = j0 + (w >> 1), kind = w & 1;
        if (w < 2 * KLB_KR && j < k && ok) {
            float4 t = red[((2 * (j - j0) + kind) * KLB_WAVES) * 64 + lane];
#pragma unroll
            for (int q = 1; q < KLB_WAVES; ++q) {
                const float4 x = red[((2 * (j - j0) + kind) * KLB_WAVES + q) * 64 + lane];
                t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
            }
            float4 o;
            if (kind == 0) {       // same terms as k_kl_bwd_mix_part
                const float4 i24 = si2[j * h4];
                o = make_float4(-t.x * 2.f * i24.x, -t.y * 2.f * i24.y, -t.z * 2.f * i24.z, -t.w * 2.f * i24.w);
            } else {
                o = make_float4(-t.x, -t.y, -t.z, -t.w);
            }
            *reinterpret_cast<float4*>(part + ((size_t)blockIdx.y * 2 * k + (kind ? k : 0) + j) * h + col) = o;
        }
    }
}

// mixture-parameter gradients: grid (component j, 64-column tile, node slice); the block's 4 waves
// interleave the slice's nodes (2 rows in flight per lane) and combine through LDS in wave order.
__global__ __launch_bounds__(256) void k_kl_bwd_mix_part(const float* z, const float* mix, const float* resp,
                                                         float* part, int64_t n, int h, int k) {
    __shared__ float sm[2][4][64];
    const int j = blockIdx.x, slice = blockIdx.z, nsl = gridDim.z;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int64_t per = (n + nsl - 1) / nsl;
    const int64_t r0 = slice * per, r1 = min(n, r0 + per);
    const int kh = k * h;
    float gmu0 = 0.f, gmu1 = 0.f, gv0 = 0.f, gv1 = 0.f;
    const bool ok = c < h;
    const float muj = ok ? mix[j * h + c] : 0.f, i2 = ok ? mix[kh + j * h + c] : 0.f;   // i2 = 1/(2 v_j)
    if (ok) {
        int64_t r = r0 + w;
        for (; r + 4 < r1; r += 8) {
            const float ra = resp[r * k + j], rb = resp[(r + 4) * k + j];
            const float da = z[r * h + c] - muj, db = z[(r + 4) * h + c] - muj;
            gmu0 += ra * da;
            gmu1 += rb * db;
            gv0 += ra * (da * da * 2.f * i2 * i2 - i2);      // d^2/(2 v^2) - 1/(2 v)
            gv1 += rb * (db * db * 2.f * i2 * i2 - i2);
        }
        for (; r < r1; r += 4) {
            const float ra = resp[r * k + j];
            const float da = z[r * h + c] - muj;
            gmu0 += ra * da;
            gv0 += ra * (da * da * 2.f * i2 * i2 - i2);
        }
    }
    sm[0][w][lane] = gmu0 + gmu1;
    sm[1][w][lane] = gv0 + gv1;
    __syncthreads();
    if (w == 0 && ok) {
        const float gmu = (sm[0][0][lane] + sm[0][1][lane]) + (sm[0][2][lane] + sm[0][3][lane]);
        const float gvv = (sm[1][0][lane] + sm[1][1][lane]) + (sm[1][2][lane] + sm[1][3][lane]);
        // d(-lme)/dmu = -resp * d / v ; d(-lme)/dv = -resp * (...)
        part[((size_t)slice * 2 * k + j) * h + c] = -gmu * 2.f * i2;
        part[((size_t)slice * 2 * k + k + j) * h + c] = -gvv;
    }
}

__global__ __launch_bounds__(1024) void k_kl_bwd_mix_final(const float* part, const float* z_pre, const float* gkl,
                                                           float gscale, float* g_zpre, int accumulate, int64_t n, int h,
                                                           int k, int nsl, const int* rows_dev) {
    __shared__ float sm[64][16];
    const int total = 2 * k * h, kh = k * h;
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    float acc = sum_slices_16x64(part, total, nsl, i, sm);     // slices added in a fixed order
    if ((threadIdx.x >> 4) != 0 || i >= total) return;
    const float cg = gscale * (gkl ? *gkl : 1.f) / (float)(rows_dev ? (int64_t)*rows_dev : n);
    if (i >= kh) {  // chain through v_j = softplus(raw) + 1e-8
        const float raw = z_pre[i];
        acc *= raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw));
    }
    g_zpre[i] = accumulate ? g_zpre[i] + cg * acc : cg * acc;
}

// ---- MMD (KGVAE.get_mmd / compute_kernel, kgvae/model.py:71-80, :89-102) -------------------------------
//   K(a, b) = exp(-mean_d (a_d - b_d)^2 / h) ;  mmd = mean Kxx + mean Kyy - 2 mean Kxy
// grid: one block per row of x (first sx blocks) or of y; the block's 4 waves walk the partner rows.
// 16 waves per block, each wave takes partner rows j = w, w+16, ... four at a time (independent loads and
// wave reductions in flight: the loop is latency-bound, the whole problem is ~100 MFLOP).
constexpr int MMD_WAVES = 16;

template <int CPL>
__device__ __forceinline__ void mmd_load_row(const float* p, int h, int lane, float (&v)[CPL]) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = (lane + 64 * i < h) ? p[lane + 64 * i] : 0.f;
}

template <int CPL>
__device__ __forceinline__ float mmd_d2(const float (&a)[CPL], const float (&b)[CPL]) {
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { const float d = a[i] - b[i]; d2 = fmaf(d, d, d2); }
    return d2;
}

template <int CPL>
__global__ __launch_bounds__(64 * MMD_WAVES) void k_mmd_fwd(const float* x, const float* y, const int64_t* yidx, int sx, int sy,
                                                             int h, float* part) {
    __shared__ float sm[MMD_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool is_x = (int)blockIdx.x < sx;
    // sample row j of a set: x is dense; y is dense or, with yidx, row yidx[j] of a larger matrix (no gathered copy)
    auto row_of = [&](bool from_y, int j) -> const float* {
        return from_y ? y + (size_t)(yidx ? yidx[j] : j) * h : x + (size_t)j * h;
    };
    float av[CPL];
    mmd_load_row<CPL>(row_of(!is_x, is_x ? (int)blockIdx.x : (int)blockIdx.x - sx), h, lane, av);
    const float inv = 1.f / ((float)h * (float)h);
    float tot = 0.f;
    // x rows: + Kxx/sx^2 - 2 Kxy/(sx sy);   y rows: + Kyy/sy^2
    for (int pass = 0; pass < (is_x ? 2 : 1); ++pass) {
        const bool b_is_y = pass == 0 ? !is_x : true;
        const int nb = pass == 0 ? (is_x ? sx : sy) : sy;
        const float wt = pass == 0 ? 1.f / ((float)nb * (float)nb) : -2.f / ((float)sx * (float)sy);
        for (int j = w; j < nb; j += 4 * MMD_WAVES) {
            float b0[CPL], b1[CPL], b2[CPL], b3[CPL];
            const int j1 = j + MMD_WAVES, j2 = j + 2 * MMD_WAVES, j3 = j + 3 * MMD_WAVES;
            mmd_load_row<CPL>(row_of(b_is_y, j), h, lane, b0);
            mmd_load_row<CPL>(row_of(b_is_y, min(j1, nb - 1)), h, lane, b1);
            mmd_load_row<CPL>(row_of(b_is_y, min(j2, nb - 1)), h, lane, b2);
            mmd_load_row<CPL>(row_of(b_is_y, min(j3, nb - 1)), h, lane, b3);
            const float d0 = wave_sum(mmd_d2<CPL>(av, b0)), d1 = wave_sum(mmd_d2<CPL>(av, b1));
            const float d2 = wave_sum(mmd_d2<CPL>(av, b2)), d3 = wave_sum(mmd_d2<CPL>(av, b3));
            tot += wt * expf(-d0 * inv);
            if (j1 < nb) tot += wt * expf(-d1 * inv);
            if (j2 < nb) tot += wt * expf(-d2 * inv);
            if (j3 < nb) tot += wt * expf(-d3 * inv);
        }
    }
    if (lane == 0) sm[w] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MMD_WAVES; ++i) s += sm[i];
        part[blockIdx.x] = s;
    }
}

template <int CPL>
__global__ __launch_bounds__(64 * MMD_WAVES) void k_mmd_bwd(const float* x, const float* y, const int64_t* yidx, int sx, int sy,
                                                             int h, const float* gmmd, float gscale, float* gx, float* gy) {
    __shared__ float sm[MMD_WAVES][64 * CPL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool is_x = (int)blockIdx.x < sx;
    const int row = is_x ? blockIdx.x : blockIdx.x - sx;
    auto row_of = [&](bool from_y, int j) -> const float* {
        return from_y ? y + (size_t)(yidx ? yidx[j] : j) * h : x + (size_t)j * h;
    };
    float av[CPL], acc[CPL];
    mmd_load_row<CPL>(row_of(!is_x, row), h, lane, av);
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
    const float inv = 1.f / ((float)h * (float)h);
    const float g = gscale * (gmmd ? *gmmd : 1.f);
    const int n_same = is_x ? sx : sy, n_other = is_x ? sy : sx;
    // d mmd / d a = (-2/h^2) [ (2/n_same^2) sum_j K(a,same_j)(a - same_j) - (2/(sx sy)) sum_j K(a,other_j)(a - other_j) ]
    const float c_same = g * (-2.f * inv) * 2.f / ((float)n_same * (float)n_same);
    const float c_other = g * (-2.f * inv) * (-2.f) / ((float)sx * (float)sy);
    for (int pass = 0; pass < 2; ++pass) {
        const bool b_is_y = pass == 0 ? !is_x : is_x;
        const int nb = pass == 0 ? n_same : n_other;
        const float cf = pass == 0 ? c_same : c_other;
        for (int j = w; j < nb; j += 2 * MMD_WAVES) {
            float b0[CPL], b1[CPL];
            const int j1 = j + MMD_WAVES;
            mmd_load_row<CPL>(row_of(b_is_y, j), h, lane, b0);
            mmd_load_row<CPL>(row_of(b_is_y, min(j1, nb - 1)), h, lane, b1);
            const float d0 = wave_sum(mmd_d2<CPL>(av, b0)), d1 = wave_sum(mmd_d2<CPL>(av, b1));
            const float k0 = cf * expf(-d0 * inv), k1 = j1 < nb ? cf * expf(-d1 * inv) : 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) acc[i] = fmaf(k0, av[i] - b0[i], fmaf(k1, av[i] - b1[i], acc[i]));
        }
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) sm[w][lane + 64 * i] = acc[i];
    __syncthreads();
    // x rows and dense y rows are stored; indexed y rows ADD into row yidx[row] of gy (a gradient that already holds
    // other terms; float atomics, since an index may repeat)
    const bool indexed = !is_x && yidx;
    float* o = is_x ? gx + (size_t)row * h : gy + (size_t)(yidx ? yidx[row] : row) * h;
    for (int c = threadIdx.x; c < h; c += 64 * MMD_WAVES) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MMD_WAVES; ++i) s += sm[i][c];
        if (indexed) atomicAdd(o + c, s);
        else o[c] = s;
    }
}

// The same two passes with A ROW OF 16 LANES PER PARTNER ROW (h % 4 == 0; G float4 columns per lane: h <= 64 G):
// a wave compares its block's row with four partner rows at once, each lane loads 16-B pieces, and a squared distance is a
// row reduction (four DPP steps) instead of a whole-wave one (DPP steps + four v_readlane + adds, once per partner row).  64
// partner rows are in flight per workgroup as before (16 waves x 4), eight 16-B loads per lane and iteration.  Per element the
// same expressions; the sums run in another (fixed) order.
__device__ __forceinline__ float rl_dyn(float v, int src_lane) { return __shfl(v, src_lane); }

template <int G>
__device__ __forceinline__ void mmd_load_row4(const float* p, const bool (&on)[G], const int (&col)[G], float4 (&v)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = on[g] ? *reinterpret_cast<const float4*>(p + col[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int G>
__device__ __forceinline__ float mmd_d2_4(const float4 (&a)[G], const float4 (&b)[G]) {
    float d2 = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float d0 = a[g].x - b[g].x, d1 = a[g].y - b[g].y, d2_ = a[g].z - b[g].z, d3 = a[g].w - b[g].w;
        d2 = fmaf(d0, d0, d2); d2 = fmaf(d1, d1, d2); d2 = fmaf(d2_, d2_, d2); d2 = fmaf(d3, d3, d2);
    }
    return d2;
}

template <int G>
__global__ __launch_bounds__(64 * MMD_WAVES) void k_mmd_fwd_g(const float* x, const float* y, const int64_t* yidx, int sx, int sy,
                                                               int h, float* part) {
    __shared__ float sm[MMD_WAVES];
    const int lane = threadIdx.x & 63, l16 = lane & 15, w = threadIdx.x >> 6;
    const int grp = (int)(threadIdx.x >> 4);                  // 64 groups of 16 lanes
    constexpr int NG = 4 * MMD_WAVES;
    const bool is_x = (int)blockIdx.x < sx;
    auto row_of = [&](bool from_y, int j) -> const float* {
        return from_y ? y + (size_t)(yidx ? yidx[j] : j) * h : x + (size_t)j * h;
    };
    bool on[G];
    int col[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { col[g] = 4 * (l16 + 16 * g); on[g] = col[g] < h; }
    float4 av[G];
    mmd_load_row4<G>(row_of(!is_x, is_x ? (int)blockIdx.x : (int)blockIdx.x - sx), on, col, av);
    const float inv = 1.f / ((float)h * (float)h);
    float tot = 0.f;
    for (int pass = 0; pass < (is_x ? 2 : 1); ++pass) {
        const bool b_is_y = pass == 0 ? !is_x : true;
        const int nb = pass == 0 ? (is_x ? sx : sy) : sy;
        const float wt = pass == 0 ? 1.f / ((float)nb * (float)nb) : -2.f / ((float)sx * (float)sy);
        for (int j0 = 0; j0 < nb; j0 += 2 * NG) {               // (uniform trip count: the row reductions run in every lane)
            const int j = j0 + grp, j1 = j + NG;
            float4 b0[G], b1[G];
            mmd_load_row4<G>(row_of(b_is_y, min(j, nb - 1)), on, col, b0);
            mmd_load_row4<G>(row_of(b_is_y, min(j1, nb - 1)), on, col, b1);
            const float d0 = row16_sum(mmd_d2_4<G>(av, b0)), d1 = row16_sum(mmd_d2_4<G>(av, b1));
            if (j < nb) tot += wt * expf(-d0 * inv);
            if (j1 < nb) tot += wt * expf(-d1 * inv);
        }
    }
    const float t = (rl_bcast_f(tot, 0) + rl_bcast_f(tot, 16)) + (rl_bcast_f(tot, 32) + rl_bcast_f(tot, 48));
    if (lane == 0) sm[w] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MMD_WAVES; ++i) s += sm[i];
        part[blockIdx.x] = s;
    }
}

template <int G>
__global__ __launch_bounds__(64 * MMD_WAVES) void k_mmd_bwd_g(const float* x, const float* y, const int64_t* yidx, int sx, int sy,
                                                               int h, const float* gmmd, float gscale, float* gx, float* gy) {
    __shared__ float4 sm[MMD_WAVES][16 * G];                    // a wave's sum over its four groups, [float4 column]
    const int lane = threadIdx.x & 63, l16 = lane & 15, w = threadIdx.x >> 6;
    const int grp = (int)(threadIdx.x >> 4);
    constexpr int NG = 4 * MMD_WAVES;
    const bool is_x = (int)blockIdx.x < sx;
    const int row = is_x ? blockIdx.x : blockIdx.x - sx;
    auto row_of = [&](bool from_y, int j) -> const float* {
        return from_y ? y + (size_t)(yidx ? yidx[j] : j) * h : x + (size_t)j * h;
    };
    bool on[G];
    int col[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { col[g] = 4 * (l16 + 16 * g); on[g] = col[g] < h; }
    float4 av[G], acc[G];
    mmd_load_row4<G>(row_of(!is_x, row), on, col, av);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv = 1.f / ((float)h * (float)h);
    const float gg = gscale * (gmmd ? *gmmd : 1.f);
    const int n_same = is_x ? sx : sy, n_other = is_x ? sy : sx;
    const float c_same = gg * (-2.f * inv) * 2.f / ((float)n_same * (float)n_same);
    const float c_other = gg * (-2.f * inv) * (-2.f) / ((float)sx * (float)sy);
    for (int pass = 0; pass < 2; ++pass) {
        const bool b_is_y = pass == 0 ? !is_x : is_x;
        const int nb = pass == 0 ? n_same : n_other;
        const float cf = pass == 0 ? c_same : c_other;
        for (int j0 = 0; j0 < nb; j0 += 2 * NG) {
            const int j = j0 + grp, j1 = j + NG;
            float4 b0[G], b1[G];
            mmd_load_row4<G>(row_of(b_is_y, min(j, nb - 1)), on, col, b0);
            mmd_load_row4<G>(row_of(b_is_y, min(j1, nb - 1)), on, col, b1);
            const float d0 = row16_sum(mmd_d2_4<G>(av, b0)), d1 = row16_sum(mmd_d2_4<G>(av, b1));
            const float k0 = j < nb ? cf * expf(-d0 * inv) : 0.f, k1 = j1 < nb ? cf * expf(-d1 * inv) : 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[g].x = fmaf(k0, av[g].x - b0[g].x, fmaf(k1, av[g].x - b1[g].x, acc[g].x));
                acc[g].y = fmaf(k0, av[g].y - b0[g].y, fmaf(k1, av[g].y - b1[g].y, acc[g].y));
                acc[g].z = fmaf(k0, av[g].z - b0[g].z, fmaf(k1, av[g].z - b1[g].z, acc[g].z));
                acc[g].w = fmaf(k0, av[g].w - b0[g].w, fmaf(k1, av[g].w - b1[g].w, acc[g].w));
            }
        }
    }
    // a wave's four groups hold the same columns in lanes l16, 16 + l16, 32 + l16, 48 + l16: added in group order, then the 16 waves
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float4 t;
        t.x = (rl_dyn(acc[g].x, l16) + rl_dyn(acc[g].x, 16 + l16)) + (rl_dyn(acc[g].x, 32 + l16) + rl_dyn(acc[g].x, 48 + l16));
        t.y = (rl_dyn(acc[g].y, l16) + rl_dyn(acc[g].y, 16 + l16)) + (rl_dyn(acc[g].y, 32 + l16) + rl_dyn(acc[g].y, 48 + l16));
        t.z = (rl_dyn(acc[g].z, l16) + rl_dyn(acc[g].z, 16 + l16)) + (rl_dyn(acc[g].z, 32 + l16) + rl_dyn(acc[g].z, 48 + l16));
        t.w = (rl_dyn(acc[g].w, l16) + rl_dyn(acc[g].w, 16 + l16)) + (rl_dyn(acc[g].w, 32 + l16) + rl_dyn(acc[g].w, 48 + l16));
        if (lane < 16) sm[w][l16 + 16 * g] = t;
    }
    __syncthreads();
    const bool indexed = !is_x && yidx;
    float* o = is_x ? gx + (size_t)row * h : gy + (size_t)(yidx ? yidx[row] : row) * h;
    for (int q = threadIdx.x; q < (h >> 2); q += 64 * MMD_WAVES) {
        float4 s4 = sm[0][q];
#pragma unroll
        for (int i = 1; i < MMD_WAVES; ++i) { const float4 t = sm[i][q]; s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w; }
        if (indexed) {
            atomicAdd(o + 4 * q, s4.x); atomicAdd(o + 4 * q + 1, s4.y); atomicAdd(o + 4 * q + 2, s4.z); atomicAdd(o + 4 * q + 3, s4.w);
        } else {
            *reinterpret_cast<float4*>(o + 4 * q) = s4;
        }
    }
}

// prior samples of get_mmd: z_pri[i] = mu[i % k] + eps[i] * sqrt(softplus(raw[i % k]) + 1e-8)   (z_pre = [mu; raw], (2k, h))
__global__ __launch_bounds__(256) void k_prior_sample_fwd(const float* z_pre, const float* eps, float* out, int s, int k, int h) {
    const int total = s * h;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / h, c = i - r * h, j = r % k;
        const float v = softplus_t(z_pre[(size_t)(k + j) * h + c]) + 1e-8f;
        out[i] = z_pre[(size_t)j * h + c] + eps[i] * sqrtf(v);
    }
}

__global__ __launch_bounds__(256) void k_prior_sample_bwd(const float* z_pre, const float* eps, const float* g, float* gz_pre,
                                                          int accumulate, int s, int k, int h) {
    const int total = k * h;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i / h, c = i - j * h;
        const float raw = z_pre[(size_t)(k + j) * h + c];
        const float v = softplus_t(raw) + 1e-8f;
        float gm = 0.f, gs = 0.f;
        for (int r = j; r < s; r += k) {
            gm += g[(size_t)r * h + c];
            gs += g[(size_t)r * h + c] * eps[(size_t)r * h + c];
        }
        const float gr = gs * 0.5f / sqrtf(v) * (raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw)));
        gz_pre[i] = accumulate ? gz_pre[i] + gm : gm;
        gz_pre[total + i] = accumulate ? gz_pre[total + i] + gr : gr;
    }
}

constexpr int KL_SLICES = 256;      // row slices of the mixture-gradient partial sums
constexpr int KL_FUSED_KT = 16;     // (also the 16-lane group that carries one row's responsibilities)     // the fused backward keeps 4*KT floats per lane: k <= 16 (the reference's mog_k is 10)

}  // namespace gv

using namespace gv;

#define GV_ST ((hipStream_t)stream)

static int red_blocks(int64_t n, int per_block) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > RED_BLOCKS ? RED_BLOCKS : b));
}

static int sq2_blocks(int64_t n, int cap) {
    const int b = red_blocks(n, 4096);
    return b > cap ? cap : b;
}

static int kl_blocks(int64_t n) { return (int)((n + 3) / 4 > RED_BLOCKS ? RED_BLOCKS : (n + 3) / 4); }

extern "C" int gv_distmult_bce_fwd(const float* embed, int ld_e, const float* w_rel, int ld_w,
                                   const int32_t* triplets, const int32_t* order, const float* labels, const float* bias,
                                   float* score, float* loss, float* workspace, int64_t t, int h, void* stream) {
    GV_REQUIRE(embed && w_rel && triplets && labels && score && workspace, GV_ERR_NULL,
               "gv_distmult_bce_fwd: NULL pointer");
    GV_REQUIRE(t > 0 && h > 0 && ld_e >= h && ld_w >= h, GV_ERR_SHAPE, "gv_distmult_bce_fwd: bad shape");
    const int nb = red_blocks(t, 16);
    hipLaunchKernelGGL(k_distmult_bce, dim3(nb), dim3(256), 0, GV_ST, embed, ld_e, w_rel, ld_w, triplets, order, labels,
                       bias, score, workspace, t, h);
    if (loss) hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, GV_ST, workspace, nb, 1.f / (float)t, loss, 0);
    return launch_status("gv_distmult_bce_fwd");
}

extern "C" int gv_bce_grad(const float* score, const float* labels, const float* gloss, float* dscore,
                           const int32_t* pos3, float* dscore_inc, float* dscore_rel, float* dbias, float* workspace,
                           int64_t t, void* stream) {
    GV_REQUIRE(score && labels && dscore && workspace, GV_ERR_NULL, "gv_bce_grad: NULL pointer");
    GV_REQUIRE(!pos3 || (dscore_inc && dscore_rel), GV_ERR_NULL, "gv_bce_grad: pos3 needs dscore_inc and dscore_rel");
    GV_REQUIRE(t > 0, GV_ERR_SHAPE, "gv_bce_grad: t=%lld", (long long)t);
    const int nb = red_blocks(t, 256);
    hipLaunchKernelGGL(k_bce_grad, dim3(nb), dim3(256), 0, GV_ST, score, labels, gloss, dscore, pos3, dscore_inc,
                       dscore_rel, workspace, t);
    if (dbias) hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, GV_ST, workspace, nb, 1.f, dbias, 0);
    return launch_status("gv_bce_grad");
}

extern "C" int gv_mean_sq(const float* x, int64_t n, float scale, float* out, float* workspace, int accumulate,
                          void* stream) {
    GV_REQUIRE(x && out && workspace, GV_ERR_NULL, "gv_mean_sq: NULL pointer");
    GV_REQUIRE(n > 0, GV_ERR_SHAPE, "gv_mean_sq: n=%lld", (long long)n);
    const int nb = red_blocks(n, 4096);
    hipLaunchKernelGGL(k_sumsq_part, dim3(nb), dim3(256), 0, GV_ST, x, n, workspace);
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, GV_ST, workspace, nb, scale, out, accumulate);
    return launch_status("gv_mean_sq");
}

__global__ __launch_bounds__(256) void k_sumsq2_part(const float* x1, int64_t n1, float s1, int nb1, const float* x2,
                                                     int64_t n2, float s2, float* part, const int* rows_dev, int64_t rows_host) {
    __shared__ float sm[4];
    if (rows_dev) {                        // x1 = rows_host rows of which only the first *rows_dev exist: mean over those
        const int64_t rows = *rows_dev;
        n1 = n1 / rows_host * rows;
        s1 *= (float)rows_host / (float)rows;
    }
    const bool first = (int)blockIdx.x < nb1;
    const float* x = first ? x1 : x2;
    const int64_t n = first ? n1 : n2;
    const int b = first ? blockIdx.x : blockIdx.x - nb1, nb = first ? nb1 : gridDim.x - nb1;
    float acc = 0.f;
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) acc = fmaf(x[i], x[i], acc);
    const float tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot * (first ? s1 : s2);
}

extern "C" int gv_mean_sq2(const float* x1, int64_t n1, float scale1, const float* x2, int64_t n2, float scale2, float* out,
                           float* workspace, const int32_t* rows_dev, int64_t rows_host, void* stream) {
    GV_REQUIRE(x1 && x2 && workspace, GV_ERR_NULL, "gv_mean_sq2: NULL pointer");
    GV_REQUIRE(n1 > 0 && n2 > 0, GV_ERR_SHAPE, "gv_mean_sq2: empty input");
    GV_REQUIRE(!rows_dev || (rows_host > 0 && n1 % rows_host == 0), GV_ERR_SHAPE, "gv_mean_sq2: rows_host=%lld does not divide n1",
               (long long)rows_host);
    const int nb1 = sq2_blocks(n1, 768), nb2 = sq2_blocks(n2, 255);
    hipLaunchKernelGGL(k_sumsq2_part, dim3(nb1 + nb2), dim3(256), 0, GV_ST, x1, n1, scale1, nb1, x2, n2, scale2, workspace, rows_dev,
                       rows_host);
    if (out) hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, GV_ST, workspace, nb1 + nb2, 1.f, out, 0);
    return launch_status("gv_mean_sq2");
}

extern "C" int64_t gv_kl_workspace_bytes(int64_t n, int h, int k) {
    (void)n;
    return (int64_t)sizeof(float) * (3 * (int64_t)k * h + RED_BLOCKS + (int64_t)KL_SLICES * 2 * k * h);
}

// the float4-column form where rows are 16-B aligned (h % 4 == 0, every operand 16-B aligned; GV_KL_V4=0 keeps the per-lane form)
static bool kl_fwd_v4(int nb, size_t lds, hipStream_t st, const float* z, const float* m, int ld_m, const float* v, const float* mix,
                      const float* flp, float* resp, float* part, int64_t n, int h, int k, const int* rows_dev, const float* h2,
                      const float* eps, float* z_out, float* v_out, float* m_out) {
    static const int env = getenv("GV_KL_V4") ? atoi(getenv("GV_KL_V4")) : 1;
    const bool al = aligned16(mix) && (!z || aligned16(z)) && (!m || (aligned16(m) && ld_m % 4 == 0)) && (!v || aligned16(v)) &&
                    (!h2 || aligned16(h2)) && (!eps || aligned16(eps)) && (!z_out || aligned16(z_out)) && (!v_out || aligned16(v_out)) &&
                    (!m_out || aligned16(m_out));
    if (!env || h % 4 != 0 || h > 1024 || !al) return false;
    static const int v5 = getenv("GV_KL_V5") ? atoi(getenv("GV_KL_V5")) : 1;
    // four nodes per wave (a lane: 4 float4 columns; the 8-column instance measured SLOWER than the form below at h = 500: 67.9 vs 61.9 us)
    if (v5 && h <= (v5 == 2 ? 512 : 256) && k <= 16) {
        if (h <= 256)
            hipLaunchKernelGGL(k_kl_fwd_v5<4>, dim3(nb), dim3(256), lds, st, z, m, ld_m, v, mix, flp, resp, part, n, h, k, rows_dev, h2,
                               eps, z_out, v_out, m_out);
        else
            hipLaunchKernelGGL(k_kl_fwd_v5<8>, dim3(nb), dim3(256), lds, st, z, m, ld_m, v, mix, flp, resp, part, n, h, k, rows_dev, h2,
                               eps, z_out, v_out, m_out);
        return true;
    }
#define GV_KL_V4(G_) hipLaunchKernelGGL(k_kl_fwd_v4<G_>, dim3(nb), dim3(256), lds, st, z, m, ld_m, v, mix, flp, resp, part, n, h, k, rows_dev, h2, eps, z_out, v_out, m_out)
    if (h <= 256) GV_KL_V4(1);
    else if (h <= 512) GV_KL_V4(2);
    else GV_KL_V4(4);
#undef GV_KL_V4
    return true;
}

extern "C" int gv_kl_fwd(const float* z, const float* m, int ld_m, const float* v, const float* z_pre,
                         const float* flp, float* resp, float* kl, float* workspace, int64_t n, int h, int k,
                         const int32_t* rows_dev, void* stream) {
    GV_REQUIRE(z && m && v && z_pre && resp && workspace, GV_ERR_NULL, "gv_kl_fwd: NULL pointer");
    GV_REQUIRE(n > 0 && h > 0 && k > 0 && k <= KL_KMAX && ld_m >= h, GV_ERR_SHAPE, "gv_kl_fwd: n=%lld h=%d k=%d",
               (long long)n, h, k);
    const size_t lds = (size_t)2 * k * h * sizeof(float);
    GV_REQUIRE(lds <= 64 * 1024, GV_ERR_SHAPE, "gv_kl_fwd: mixture table %zu B exceeds the 64 KiB LDS budget", lds);
    GV_REQUIRE(!(rows_dev && kl), GV_ERR_SHAPE, "gv_kl_fwd: with rows_dev the mean is finished by gv_loss_combine (kl must be NULL)");
    float* mix = workspace;
    float* part = workspace + 3 * (size_t)k * h;
    hipLaunchKernelGGL(k_kl_mix, dim3((k * h + 255) / 256), dim3(256), 0, GV_ST, z_pre, k, h, mix);
    const int nb = kl_blocks(n);
    GV_REQUIRE(h <= 1024, GV_ERR_SHAPE, "gv_kl_fwd: h=%d > 1024 unsupported", h);
#define GV_KL_FWD(CPL_) hipLaunchKernelGGL(k_kl_fwd<CPL_>, dim3(nb), dim3(256), lds, GV_ST, z, m, ld_m, v, mix, flp, resp, part, n, h, k, rows_dev)
    if (kl_fwd_v4(nb, lds, GV_ST, z, m, ld_m, v, mix, flp, resp, part, n, h, k, rows_dev, nullptr, nullptr, nullptr, nullptr, nullptr)) {}
    else if (h <= 64) GV_KL_FWD(1);
    else if (h <= 128) GV_KL_FWD(2);
    else if (h <= 256) GV_KL_FWD(4);
    else if (h <= 512) GV_KL_FWD(8);
    else GV_KL_FWD(16);
#undef GV_KL_FWD
    // mean over nodes: the per-block partials (fixed node->wave->block assignment) summed in block order
    if (kl) hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, GV_ST, part, nb, 1.f / (float)n, kl, 0);
    return launch_status("gv_kl_fwd");
}

// the float4-column backward where rows are 16-B aligned (GV_KL_BWD_V4=0 keeps the lane-per-column form)
static bool kl_bwd_cols4(hipStream_t st, const float* z, const float* m, int ld_m, const float* v, const float* mix, const float* resp,
                         const float* gkl, float gscale, float z_extra, float* gz, float* gm, float* gv, float* part, int64_t n, int h,
                         int k, const int* rows_dev, const float* h2, const float* eps, const float* gz_up, float* gh2) {
    static const int env = getenv("GV_KL_BWD_V4") ? atoi(getenv("GV_KL_BWD_V4")) : 1;
    const bool al = aligned16(z) && aligned16(m) && ld_m % 4 == 0 && aligned16(v) && aligned16(mix) && aligned16(part) &&
                    (!gz || aligned16(gz)) && (!gm || aligned16(gm)) && (!gv || aligned16(gv)) && (!h2 || aligned16(h2)) &&
                    (!eps || aligned16(eps)) && (!gz_up || aligned16(gz_up)) && (!gh2 || aligned16(gh2));
    if (!env || h % 4 != 0 || h > 1024 || k > 10 || !al) return false;      // (a 16-component instance spills: k > 10 keeps the other form)
    const size_t lds = ((size_t)2 * k * h / 4 + (size_t)2 * KLB_KR * KLB_WAVES * 64) * sizeof(float4);
    if (lds > 128 * 1024) return false;
    static unsigned long long lds_done = 0;
    if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)k_kl_bwd_cols4<10>, 128 * 1024, lds_done, "gv_kl_bwd")) return false;
#define GV_KLB(KT_) hipLaunchKernelGGL(k_kl_bwd_cols4<KT_>, dim3((h + 255) / 256, KL_SLICES), dim3(64 * KLB_WAVES), lds, st, z, m, ld_m, v, mix, resp, gkl, \
                                       gscale, z_extra, gz, gm, gv, part, n, h, k, rows_dev, h2, eps, gz_up, gh2)
    GV_KLB(10);
#undef GV_KLB
    return true;
}

extern "C" int gv_kl_bwd(const float* z, const float* m, int ld_m, const float* v, const float* z_pre,
                         const float* resp, const float* gkl, float gscale, float z_extra, float* gz, float* gm,
                         float* gv, float* g_zpre, int accumulate_zpre, float* workspace, int mix_ready, int64_t n, int h,
                         int k, const int32_t* rows_dev, void* stream) {
    GV_REQUIRE(z && m && v && z_pre && resp && gz && gm && gv && g_zpre && workspace, GV_ERR_NULL,
               "gv_kl_bwd: NULL pointer");
    GV_REQUIRE(n > 0 && h > 0 && k > 0 && k <= KL_KMAX && ld_m >= h, GV_ERR_SHAPE, "gv_kl_bwd: bad shape");
    float* mix = workspace;  // mix_ready: the caller hands back the workspace gv_kl_fwd filled for the same z_pre
    if (!mix_ready) hipLaunchKernelGGL(k_kl_mix, dim3((k * h + 255) / 256), dim3(256), 0, GV_ST, z_pre, k, h, mix);
    float* part = workspace + 3 * (size_t)k * h + RED_BLOCKS;
    if (kl_bwd_cols4(GV_ST, z, m, ld_m, v, mix, resp, gkl, gscale, z_extra, gz, gm, gv, part, n, h, k, rows_dev, nullptr, nullptr,
                     nullptr, nullptr)) {
    } else if (k <= KL_FUSED_KT) {
        hipLaunchKernelGGL(k_kl_bwd_fused<KL_FUSED_KT>, dim3((h + 63) / 64, KL_SLICES), dim3(256), 0, GV_ST, z, m, ld_m, v, mix,
                           resp, gkl, gscale, z_extra, gz, gm, gv, part, n, h, k, rows_dev);
    } else {
        GV_REQUIRE(!rows_dev, GV_ERR_SHAPE, "gv_kl_bwd: rows_dev needs k <= %d mixture components", KL_FUSED_KT);
        const size_t lds = (size_t)2 * k * h * sizeof(float);
        const int nb = (int)((n + 3) / 4 > 1024 ? 1024 : (n + 3) / 4);
        hipLaunchKernelGGL(k_kl_bwd_nodes, dim3(nb), dim3(256), lds, GV_ST, z, m, ld_m, v, mix, resp, gkl, gscale, z_extra,
                           gz, gm, gv, n, h, k);
        hipLaunchKernelGGL(k_kl_bwd_mix_part, dim3(k, (h + 63) / 64, KL_SLICES), dim3(256), 0, GV_ST, z, mix, resp, part, n,
                           h, k);
    }
    hipLaunchKernelGGL(k_kl_bwd_mix_final, dim3((2 * k * h + 15) / 16), dim3(1024), 0, GV_ST, part, z_pre, gkl,
                       gscale, g_zpre, accumulate_zpre, n, h, k, KL_SLICES, rows_dev);
    return launch_status("gv_kl_bwd");
}

// K3 fused into K6 (the reparameterisation and the KL term read the same node rows):
//   fwd: (m, v, z) = reparameterise(h2, eps) AND the KL forward pass over them (resp, per-block partial sums) in one sweep;
//   bwd: KL's node gradients, the regulariser's z_extra * z and the upstream dL/dz are chained through the reparameterisation
//        in registers: gh2 is the only node-sized output (gz / gm / gv of gv_kl_bwd are never materialised).
extern "C" int gv_reparam_kl_fwd(const float* h2, const float* eps, const float* z_pre, float* z, float* v, float* m_out,
                                 float* resp, float* workspace, int64_t n, int h, int k, void* stream) {
    GV_REQUIRE(h2 && eps && z_pre && z && v && resp && workspace, GV_ERR_NULL, "gv_reparam_kl_fwd: NULL pointer");
    GV_REQUIRE(n > 0 && h > 0 && h <= 1024 && k > 0 && k <= KL_KMAX, GV_ERR_SHAPE, "gv_reparam_kl_fwd: n=%lld h=%d k=%d", (long long)n,
               h, k);
    const size_t lds = (size_t)2 * k * h * sizeof(float);
    GV_REQUIRE(lds <= 64 * 1024, GV_ERR_SHAPE, "gv_reparam_kl_fwd: mixture table %zu B exceeds the 64 KiB LDS budget", lds);
    float* mix = workspace;
    float* part = workspace + 3 * (size_t)k * h;
    hipLaunchKernelGGL(k_kl_mix, dim3((k * h + 255) / 256), dim3(256), 0, GV_ST, z_pre, k, h, mix);
    const int nb = kl_blocks(n);
    const float* none = nullptr;
    const int* no_rows = nullptr;
#define GV_RKL_FWD(CPL_)                                                                                                       \
    hipLaunchKernelGGL(k_kl_fwd<CPL_>, dim3(nb), dim3(256), lds, GV_ST, none, none, h, none, (const float*)mix, none, resp, part, n, h, \
                       k, no_rows, h2, eps, z, v, m_out)
    if (kl_fwd_v4(nb, lds, GV_ST, none, none, h, none, mix, none, resp, part, n, h, k, no_rows, h2, eps, z, v, m_out)) {}
    else if (h <= 64) GV_RKL_FWD(1);
    else if (h <= 128) GV_RKL_FWD(2);
    else if (h <= 256) GV_RKL_FWD(4);
    else if (h <= 512) GV_RKL_FWD(8);
    else GV_RKL_FWD(16);
#undef GV_RKL_FWD
    return launch_status("gv_reparam_kl_fwd");
}

extern "C" int gv_reparam_kl_bwd(const float* z, const float* h2, const float* v, const float* eps, const float* z_pre,
                                 const float* resp, const float* gkl, float gscale, float z_extra, const float* gz_up,
                                 float* gh2, float* g_zpre, int accumulate_zpre, float* workspace, int64_t n, int h, int k,
                                 void* stream) {
    GV_REQUIRE(z && h2 && v && eps && z_pre && resp && gh2 && g_zpre && workspace, GV_ERR_NULL, "gv_reparam_kl_bwd: NULL pointer");
    GV_REQUIRE(n > 0 && h > 0 && k > 0 && k <= KL_FUSED_KT, GV_ERR_SHAPE, "gv_reparam_kl_bwd: n=%lld h=%d k=%d (k <= %d)", (long long)n,
               h, k, KL_FUSED_KT);
    float* mix = workspace;                    // the table gv_reparam_kl_fwd left there for the same z_pre
    float* part = workspace + 3 * (size_t)k * h + RED_BLOCKS;
    float* none = nullptr;
    const int* no_rows = nullptr;
    if (!kl_bwd_cols4(GV_ST, z, h2, 2 * h, v, mix, resp, gkl, gscale, z_extra, none, none, none, part, n, h, k, no_rows, h2, eps, gz_up,
                      gh2))
        hipLaunchKernelGGL(k_kl_bwd_fused<KL_FUSED_KT>, dim3((h + 63) / 64, KL_SLICES), dim3(256), 0, GV_ST, z, h2, 2 * h, v,
                           (const float*)mix, resp, gkl, gscale, z_extra, none, none, none, part, n, h, k, no_rows, h2, eps, gz_up, gh2);
    hipLaunchKernelGGL(k_kl_bwd_mix_final, dim3((2 * k * h + 15) / 16), dim3(1024), 0, GV_ST, (const float*)part, z_pre, gkl, gscale,
                       g_zpre, accumulate_zpre, n, h, k, KL_SLICES, no_rows);
    return launch_status("gv_reparam_kl_bwd");
}

__global__ void k_lincomb(const float* a0, float c0, const float* a1, float c1, const float* a2, float c2, const float* a3,
                          float c3, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float v = 0.f;
        if (a0) v += c0 * *a0;
        if (a1) v += c1 * *a1;
        if (a2) v += c2 * *a2;
        if (a3) v += c3 * *a3;
        *out = v;
    }
}

extern "C" int gv_lincomb4(const float* a0, float c0, const float* a1, float c1, const float* a2, float c2,
                           const float* a3, float c3, float* out, void* stream) {
    GV_REQUIRE(out, GV_ERR_NULL, "gv_lincomb4: NULL output");
    hipLaunchKernelGGL(k_lincomb, dim3(1), dim3(64), 0, GV_ST, a0, c0, a1, c1, a2, c2, a3, c3, out);
    return launch_status("gv_lincomb4");
}

extern "C" int gv_mmd_fwd(const float* x, const float* y, const int64_t* y_index, int sx, int sy, int h, float* mmd,
                          float* workspace, void* stream) {
    GV_REQUIRE(x && y && workspace, GV_ERR_NULL, "gv_mmd_fwd: NULL pointer");
    GV_REQUIRE(sx > 0 && sy > 0 && h > 0 && h <= 1024 && sx + sy <= RED_BLOCKS, GV_ERR_SHAPE,
               "gv_mmd_fwd: sx=%d sy=%d h=%d (need h <= 1024, sx+sy <= %d)", sx, sy, h, RED_BLOCKS);
    static const int v2 = getenv("GV_MMD_ROWS16") ? atoi(getenv("GV_MMD_ROWS16")) : 1;
    const bool al = h % 4 == 0 && aligned16(x) && aligned16(y);
    if (v2 && al && h <= 256) hipLaunchKernelGGL(k_mmd_fwd_g<4>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, workspace);
    else if (v2 && al && h <= 512) hipLaunchKernelGGL(k_mmd_fwd_g<8>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, workspace);
    else if (h <= 256) hipLaunchKernelGGL(k_mmd_fwd<4>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, workspace);
    else hipLaunchKernelGGL(k_mmd_fwd<16>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, workspace);
    if (mmd) hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, GV_ST, workspace, sx + sy, 1.f, mmd, 0);
    return launch_status("gv_mmd_fwd");
}

extern "C" int gv_loss_combine(const float* ws_pred, int64_t t, const float* ws_reg, int64_t n_embed, int64_t n_wrel,
                               const float* ws_kl, int64_t n_nodes, int h, int k, const float* ws_mmd, int sx, int sy,
                               float reg_w, float kl_w, float mmd_w, float* scal, float* loss, const int32_t* rows_dev,
                               void* stream) {
    GV_REQUIRE(ws_pred && loss, GV_ERR_NULL, "gv_loss_combine: NULL pointer");
    GV_REQUIRE(t > 0, GV_ERR_SHAPE, "gv_loss_combine: t=%lld", (long long)t);
    const int n0 = red_blocks(t, 16);
    const int n1 = ws_reg ? sq2_blocks(n_embed, 768) + sq2_blocks(n_wrel, 255) : 0;
    const int n2 = ws_kl ? kl_blocks(n_nodes) : 0;
    const int n3 = ws_mmd ? sx + sy : 0;
    const float* kl_part = ws_kl ? ws_kl + 3 * (size_t)k * h : nullptr;     // partials follow the mixture table
    hipLaunchKernelGGL(k_loss_combine, dim3(1), dim3(256), 0, GV_ST, ws_pred, n0, 1.f / (float)t, ws_reg, n1, 1.f, kl_part,
                       n2, ws_kl ? 1.f / (float)n_nodes : 0.f, ws_mmd, n3, 1.f, reg_w, kl_w, mmd_w, scal, loss, rows_dev,
                       (float)n_nodes);
    return launch_status("gv_loss_combine");
}

extern "C" int gv_mmd_bwd(const float* x, const float* y, const int64_t* y_index, int sx, int sy, int h, const float* gmmd,
                          float gscale, float* gx, float* gy, void* stream) {
    GV_REQUIRE(x && y && gx && gy, GV_ERR_NULL, "gv_mmd_bwd: NULL pointer");
    GV_REQUIRE(sx > 0 && sy > 0 && h > 0 && h <= 1024, GV_ERR_SHAPE, "gv_mmd_bwd: bad shape");
    static const int v2 = getenv("GV_MMD_ROWS16") ? atoi(getenv("GV_MMD_ROWS16")) : 1;
    const bool al = h % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gx) && aligned16(gy);
    if (v2 && al && h <= 256) hipLaunchKernelGGL(k_mmd_bwd_g<4>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, gmmd, gscale, gx, gy);
    // (an eight-column-group instance of the backward needs more than the 128 registers a 1 024-thread workgroup leaves a lane)
    else if (h <= 256) hipLaunchKernelGGL(k_mmd_bwd<4>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, gmmd, gscale, gx, gy);
    else hipLaunchKernelGGL(k_mmd_bwd<16>, dim3(sx + sy), dim3(64 * MMD_WAVES), 0, GV_ST, x, y, y_index, sx, sy, h, gmmd, gscale, gx, gy);
    return launch_status("gv_mmd_bwd");
}

extern "C" int gv_prior_sample_fwd(const float* z_pre, const float* eps, float* out, int s, int k, int h, void* stream) {
    GV_REQUIRE(z_pre && eps && out, GV_ERR_NULL, "gv_prior_sample_fwd: NULL pointer");
    GV_REQUIRE(s > 0 && k > 0 && h > 0, GV_ERR_SHAPE, "gv_prior_sample_fwd: bad shape");
    hipLaunchKernelGGL(k_prior_sample_fwd, dim3((s * h + 255) / 256), dim3(256), 0, GV_ST, z_pre, eps, out, s, k, h);
    return launch_status("gv_prior_sample_fwd");
}

extern "C" int gv_prior_sample_bwd(const float* z_pre, const float* eps, const float* g, float* gz_pre, int accumulate,
                                   int s, int k, int h, void* stream) {
    GV_REQUIRE(z_pre && eps && g && gz_pre, GV_ERR_NULL, "gv_prior_sample_bwd: NULL pointer");
    GV_REQUIRE(s > 0 && k > 0 && h > 0, GV_ERR_SHAPE, "gv_prior_sample_bwd: bad shape");
    hipLaunchKernelGGL(k_prior_sample_bwd, dim3((k * h + 255) / 256), dim3(256), 0, GV_ST, z_pre, eps, g, gz_pre, accumulate, s, k, h);
    return launch_status("gv_prior_sample_bwd");
}
