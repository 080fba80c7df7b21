// Calibration probe: issue rate of v_mfma_f32_32x32x2_f32 per SIMD for W waves per SIMD, each with A accumulator chains.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int A>
__global__ void k(float* out, int iters, float a, float b) {
    f32x16 acc[A];
    for (int u = 0; u < A; ++u)
        for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int u = 0; u < A; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
    }
    float s = 0.f;
    for (int u = 0; u < A; ++u)
        for (int i = 0; i < 16; ++i) s += acc[u][i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int A>
void run(int waves_per_simd, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * waves_per_simd), block(256);          // one 4-wave block per CU per wave-per-SIMD
    k<A><<<grid, block>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<A><<<grid, block>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * A * waves_per_simd;
    const double flops = mfma_per_simd * 1024 * 4096.0;
    printf("waves/SIMD %d  chains/wave %d: %.1f us, %.1f ns per MFMA per SIMD (64 cyc @2.4GHz = 26.7 ns), %.1f TFLOP/s\n",
           waves_per_simd, A, ms * 1e3, ms * 1e6 / mfma_per_simd, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    for (int w = 1; w <= 4; ++w) { run<1>(w, out); run<2>(w, out); }
    return 0;
}
