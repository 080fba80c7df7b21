// K4, pass 0 of MADE (kgvae/flow_network.py:85-98): the first autoregressive pass runs the masked MLP on an all-zero input, so
// every node sees the SAME row -- the products are 1 x k by k x n.  As launches of the tiled GEMM that is 15 launches forward and
// ~27 backward per IAF block at 7-15 us each (0.5 ms of a 5.3 ms step); here the whole chain of one row is ONE workgroup:
//   forward   y_l = act(r(y_{l-1}) . r(W_l)^T + b_l)                              a wave per output, lanes over 16-B pieces of k
//   backward  gm_l = g_l * [y_l > 0];  gb_l = gm_l;  gW_l = r(gm_l)^T r(y_{l-1});  g_{l-1} = r(gm_l) . r(W_l)
// r = round to bf16 (the operand precision of BASELINE configs[2]; fp32 products and sums, fixed summation order).
#include "common.h"

namespace gv {

constexpr int ROW_THREADS = 1024, ROW_MAXW = 512;

struct RowArgs {
    const float* x;          // forward: input row [k_0] or NULL (zeros); backward: gradient of the last layer's output [n_last]
    float* gx;               // backward: gradient w.r.t. the input row [k_0] or NULL
    int n_layers, exact;
    gv_row_layer L[GV_CHAIN_MAX_LAYERS];
};

// ex != 0: exact fp32 operands (the fp32 MADE node: layers[0].reserved = 1) instead of the bf16 rounding
__device__ __forceinline__ float rbf(float v, int ex) { return ex ? v : (float)(__bf16)v; }
__device__ __forceinline__ float4 rbf4(float4 v, int ex) { return make_float4(rbf(v.x, ex), rbf(v.y, ex), rbf(v.z, ex), rbf(v.w, ex)); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

constexpr int ROW_WAVES = ROW_THREADS / 64;
// weight rows per wave in flight at a time (one workgroup: the launch is bound by round trips, not by bytes): 16 when every
// layer is at most 256 deep (one 16-B piece per lane and row), 8 otherwise
template <bool WIDE> struct RowJB { static constexpr int v = WIDE ? 8 : 16; };

// a wave's lane owns columns [4 lane, 4 lane + 4) and [256 + 4 lane, ...): one or two 16-B loads per weight row
// UNCONDITIONAL loads from clamped (always valid) addresses, zeroed afterwards where the lane is outside the matrix: a load
// under a condition into a pre-zeroed register makes the compiler wait for each load before issuing the next (16 round trips
// per layer instead of one)
// (the layer's fields are passed as values read ONCE per layer: through the kernel-argument struct the compiler re-reads them with
// a scalar load + wait at every use that follows a store)
template <bool WIDE>
__device__ __forceinline__ void row_load(const float* w, int ld, int n, int k, int j, int lane, float4& a, float4& b) {
    const float* wr = w + (size_t)min(j, n - 1) * ld;
    const bool oka = j < n && 4 * lane < k, okb = WIDE && j < n && 256 + 4 * lane < k;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 va = *reinterpret_cast<const float4*>(wr + (4 * lane < k ? 4 * lane : 0));
    a = oka ? va : z;
    b = z;
    if (WIDE) {
        const float4 vb = *reinterpret_cast<const float4*>(wr + (256 + 4 * lane < k ? 256 + 4 * lane : 0));
        b = okb ? vb : z;
    }
}

template <bool WIDE>
__global__ __launch_bounds__(ROW_THREADS) void k_row_fwd(const RowArgs p) {
    constexpr int ROW_JB = RowJB<WIDE>::v;
    __shared__ __attribute__((aligned(16))) float xs[2][ROW_MAXW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < ROW_MAXW; i += ROW_THREADS) {
        xs[0][i] = (p.x && i < p.L[0].k) ? rbf(p.x[i], p.exact) : 0.f;
        xs[1][i] = 0.f;
    }
    __syncthreads();
    for (int l = 0; l < p.n_layers; ++l) {
        const float* const w = p.L[l].w;
        const float* const bias = p.L[l].bias;
        float* const outp = p.L[l].out;
        const int n = p.L[l].n, k = p.L[l].k, ld = p.L[l].ld, relu = p.L[l].relu;
        const float* xin = xs[l & 1];
        float* xout = xs[(l + 1) & 1];
        const float4 xa = *reinterpret_cast<const float4*>(xin + 4 * lane), xb = *reinterpret_cast<const float4*>(xin + 256 + 4 * lane);
        for (int j0 = wave; j0 < n; j0 += ROW_WAVES * ROW_JB) {
            float4 wa[ROW_JB], wb[ROW_JB];
            // the biases of this wave's rows: ONE vector load (lane jb holds row jb's) in flight with the weight rows -- a load per
            // row at a wave-uniform address becomes a scalar load the compiler waits for on the spot
            const int jl = j0 + (lane & (ROW_JB - 1)) * ROW_WAVES;
            float bvec = 0.f;
            if (bias) bvec = bias[min(jl, n - 1)];
#pragma unroll
            for (int jb = 0; jb < ROW_JB; ++jb) row_load<WIDE>(w, ld, n, k, j0 + jb * ROW_WAVES, lane, wa[jb], wb[jb]);
#pragma unroll
            for (int jb = 0; jb < ROW_JB; ++jb) {
                const int j = j0 + jb * ROW_WAVES;
                const float s = wave_sum(dot4(xa, rbf4(wa[jb], p.exact)) + dot4(xb, rbf4(wb[jb], p.exact)));
                const float bj = rl_bcast_f(bvec, jb);
                if (lane == 0 && j < n) {
                    float y = s + bj;
                    if (relu) y = fmaxf(y, 0.f);
                    if (outp) outp[j] = y;
                    xout[j] = rbf(y, p.exact);
                }
            }
        }
        __syncthreads();
        // columns past this layer's width must read as zero in the next layer's 16-B pieces
        for (int i = n + (int)threadIdx.x; i < ROW_MAXW; i += ROW_THREADS) xout[i] = 0.f;
        __syncthreads();
    }
}

template <bool WIDE>
__global__ __launch_bounds__(ROW_THREADS) void k_row_bwd(const RowArgs p) {
    constexpr int ROW_JB = WIDE ? 4 : 8;          // more rows in flight would spill here (128 registers per lane at 1024 threads)
    __shared__ __attribute__((aligned(16))) float g[2][ROW_MAXW];          // gradient w.r.t. a layer's output, fp32
    __shared__ __attribute__((aligned(16))) float gm[ROW_MAXW];            // masked and rounded to bf16
    __shared__ __attribute__((aligned(16))) float rin[ROW_MAXW];           // the layer's input row, rounded
    __shared__ __attribute__((aligned(16))) float part[ROW_WAVES][ROW_MAXW];
    const int nl = p.n_layers, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < p.L[nl - 1].n; i += ROW_THREADS) g[(nl - 1) & 1][i] = p.x[i];
    __syncthreads();
    for (int l = nl - 1; l >= 0; --l) {
        const float* const w = p.L[l].w;
        const float* const act = p.L[l].act;
        const float* const inp = p.L[l].inp;
        float* const gw = p.L[l].gw;
        float* const gb = p.L[l].gb;
        const int n = p.L[l].n, k = p.L[l].k, ld = p.L[l].ld, ldgw = p.L[l].ldgw;
        const float* gl = g[l & 1];
        for (int j = threadIdx.x; j < n; j += ROW_THREADS) {
            float v = gl[j];
            if (act && !(act[j] > 0.f)) v = 0.f;
            if (gb) gb[j] = v;
            gm[j] = rbf(v, p.exact);
        }
        for (int c = threadIdx.x; c < ROW_MAXW; c += ROW_THREADS) rin[c] = (inp && c < k) ? rbf(inp[c], p.exact) : 0.f;
        __syncthreads();
        const bool need_g = l > 0 || p.gx;
        const float4 ra = *reinterpret_cast<const float4*>(rin + 4 * lane), rb = *reinterpret_cast<const float4*>(rin + 256 + 4 * lane);
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
        for (int j0 = wave; j0 < n; j0 += ROW_WAVES * ROW_JB) {
            float4 wa[ROW_JB], wb[ROW_JB];
            if (need_g) {
#pragma unroll
                for (int jb = 0; jb < ROW_JB; ++jb) row_load<WIDE>(w, ld, n, k, j0 + jb * ROW_WAVES, lane, wa[jb], wb[jb]);
            }
#pragma unroll
            for (int jb = 0; jb < ROW_JB; ++jb) {
                const int j = j0 + jb * ROW_WAVES;
                if (j >= n) continue;
                const float gj = gm[j];
                if (gw) {        // outer product with the layer's input row (all-zero row: zeros)
                    float* o = gw + (size_t)j * ldgw;
                    if (4 * lane < k) *reinterpret_cast<float4*>(o + 4 * lane) = make_float4(gj * ra.x, gj * ra.y, gj * ra.z, gj * ra.w);
                    if (256 + 4 * lane < k)
                        *reinterpret_cast<float4*>(o + 256 + 4 * lane) = make_float4(gj * rb.x, gj * rb.y, gj * rb.z, gj * rb.w);
                }
                if (need_g) {       // g_{l-1} += gm[j] W[j][:]: this wave's rows in order
                    const float4 a = rbf4(wa[jb], p.exact), b = rbf4(wb[jb], p.exact);
                    sa.x += gj * a.x; sa.y += gj * a.y; sa.z += gj * a.z; sa.w += gj * a.w;
                    sb.x += gj * b.x; sb.y += gj * b.y; sb.z += gj * b.z; sb.w += gj * b.w;
                }
            }
        }
        if (need_g) {
            *reinterpret_cast<float4*>(&part[wave][4 * lane]) = sa;
            *reinterpret_cast<float4*>(&part[wave][256 + 4 * lane]) = sb;
            __syncthreads();
            if ((int)threadIdx.x < k) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < ROW_WAVES; ++q) v += part[q][threadIdx.x];      // the 16 waves' partial sums, in order
                if (l > 0) g[(l - 1) & 1][threadIdx.x] = v;
                else p.gx[threadIdx.x] = v;
            }
        }
        __syncthreads();
    }
}

static int row_check(const char* who, int n_layers, const gv_row_layer* layers, bool backward) {
    GV_REQUIRE(n_layers >= 1 && n_layers <= GV_CHAIN_MAX_LAYERS && layers, GV_ERR_SHAPE, "%s: n_layers=%d", who, n_layers);
    for (int i = 0; i < n_layers; ++i) {
        const gv_row_layer& L = layers[i];
        GV_REQUIRE(L.n > 0 && L.k > 0 && L.n <= ROW_MAXW && L.k <= ROW_MAXW && L.ld >= L.k && (i == 0 || L.k == layers[i - 1].n),
                   GV_ERR_SHAPE, "%s: layer %d is %d x %d (ld %d; widths <= %d, k = the previous layer's n)", who, i, L.n, L.k, L.ld, ROW_MAXW);
        GV_REQUIRE(L.w, GV_ERR_NULL, "%s: layer %d has no weight", who, i);
        GV_REQUIRE(L.k % 4 == 0 && L.ld % 4 == 0 && aligned16(L.w) && (!backward || !L.gw || (L.ldgw % 4 == 0 && aligned16(L.gw))),
                   GV_ERR_ALIGN, "%s: layer %d: rows of the weight (and of gw) must be whole 16-B pieces", who, i);
        GV_REQUIRE(!backward || !L.gw || L.ldgw >= L.k, GV_ERR_SHAPE, "%s: layer %d: ldgw too small", who, i);
    }
    return GV_OK;
}

}  // namespace gv

using namespace gv;

extern "C" int gv_made_row_fwd(const float* x, int n_layers, const gv_row_layer* layers, void* stream) {
    int rc = row_check("gv_made_row_fwd", n_layers, layers, false);
    if (rc != GV_OK) return rc;
    RowArgs p;
    p.x = x; p.gx = nullptr; p.n_layers = n_layers; p.exact = layers[0].reserved ? 1 : 0;
    for (int i = 0; i < n_layers; ++i) p.L[i] = layers[i];
    bool wide = false;
    for (int i = 0; i < n_layers; ++i) wide = wide || layers[i].k > 256;
    if (wide) hipLaunchKernelGGL(k_row_fwd<true>, dim3(1), dim3(ROW_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_row_fwd<false>, dim3(1), dim3(ROW_THREADS), 0, (hipStream_t)stream, p);
    return launch_status("gv_made_row_fwd");
}

extern "C" int gv_made_row_bwd(const float* g_out, int n_layers, const gv_row_layer* layers, float* g_x, void* stream) {
    int rc = row_check("gv_made_row_bwd", n_layers, layers, true);
    if (rc != GV_OK) return rc;
    GV_REQUIRE(g_out, GV_ERR_NULL, "gv_made_row_bwd: NULL gradient");
    RowArgs p;
    p.x = g_out; p.gx = g_x; p.n_layers = n_layers; p.exact = layers[0].reserved ? 1 : 0;
    for (int i = 0; i < n_layers; ++i) p.L[i] = layers[i];
    bool wide = false;
    for (int i = 0; i < n_layers; ++i) wide = wide || layers[i].k > 256;
    if (wide) hipLaunchKernelGGL(k_row_bwd<true>, dim3(1), dim3(ROW_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_row_bwd<false>, dim3(1), dim3(ROW_THREADS), 0, (hipStream_t)stream, p);
    return launch_status("gv_made_row_bwd");
}
