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

// epilogue backward that also emits the column sums of g (the bias gradient): grid (64-column tiles, row slices);
// the block's 4 waves interleave the slice's rows, combine in LDS in wave order; gv_colsum's final kernel sums slices
__global__ __launch_bounds__(256) void k_epilogue_bwd_colsum(const float* out, const float* gout, int act, const uint8_t* keep,
                                                             float scale, float* g, int64_t m, int n, float* part) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t per = (m + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = blockIdx.y * per, r1 = min(m, r0 + per);
    float acc = 0.f;
    if (c < n) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            const int64_t i = r * n + c;
            float v = gout[i];
            if (keep) v = keep[i] ? v * scale : 0.f;
            if (act == GV_ACT_RELU && !(out[i] > 0.f)) v = 0.f;
            g[i] = v;
            acc += v;
        }
    }
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < n) part[(size_t)blockIdx.y * n + c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// one wave per row; int64 ids as torch's embedding takes them
__global__ __launch_bounds__(256) void k_gather_rows(const float* table, const int64_t* ids, float* out, int64_t n,
                                                     int h) {
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

__global__ __launch_bounds__(256) void k_reverse_cols(const float* x, float* out, int64_t n, int d) {
    const int64_t total = n * d;
    GV_GRID_STRIDE(i, total) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        out[i] = x[r * d + (d - 1 - c)];
    }
}

__global__ __launch_bounds__(256) void k_adam(float* p, const float* g, float* m, float* v, int64_t n,
                                              const float* sumsq, float max_norm, float lr, float b1, float b2,
                                              float eps, const float* step) {
    float clip = 1.f;
    if (sumsq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
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
    if (colsum_part) {      // also write the 64 row-slice partials of the column sums (finish with gv_colsum_finish)
        hipLaunchKernelGGL(k_epilogue_bwd_colsum, dim3((n_cols + 63) / 64, 64), dim3(256), 0, GV_ST, out, grad_out, act, keep,
                           keep_scale, g, n_rows, n_cols, colsum_part);
        return launch_status("gv_rgcn_epilogue_bwd(colsum)");
    }
    hipLaunchKernelGGL(k_epilogue_bwd, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, out, grad_out, act, keep,
                       keep_scale, g, n);
    return launch_status("gv_rgcn_epilogue_bwd");
}

extern "C" int gv_gather_rows(const float* table, const int64_t* ids, float* out, int64_t n, int h, void* stream) {
    GV_REQUIRE(table && ids && out, GV_ERR_NULL, "gv_gather_rows: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n, 4)), dim3(256), 0, GV_ST, table, ids, out, n, h);
    return launch_status("gv_gather_rows");
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

extern "C" int gv_rowsum(const float* x, int ld, int col0, int ncols, float* out, int64_t n, void* stream) {
    GV_REQUIRE(x && out, GV_ERR_NULL, "gv_rowsum: NULL pointer");
    if (n <= 0) return GV_OK;
    hipLaunchKernelGGL(k_rowsum, dim3(grid_for(n, 4)), dim3(256), 0, GV_ST, x, ld, col0, ncols, out, n);
    return launch_status("gv_rowsum");
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
    hipLaunchKernelGGL(k_adam, dim3(grid_for(n, 1024)), dim3(256), 0, GV_ST, p, g, exp_avg, exp_avg_sq, n, sumsq,
                       max_norm, lr, beta1, beta2, eps, step);
    return launch_status("gv_adam_step");
}
