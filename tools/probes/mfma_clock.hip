// What the fp32 MFMA pipe of this box sustains: pure v_mfma_f32_16x16x4_f32 / 32x32x2 loops (no memory), 1-4 waves per SIMD on every
// CU; TFLOP/s from hipEvents, shader clock from s_memtime against s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_clock.hip -o gpurun_out/mfma_clock && gpurun_out/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    const long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, s = 0.f;
    if (KIND == 0) {
        f4 acc[13];
        for (int i = 0; i < 13; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 13; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 13; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f16v acc[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    const long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
void run(int wgs, int iters, const char* name) {
    float* out; long long* clk;
    hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&clk, (size_t)wgs * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * wgs);
        hipMemcpy(h.data(), clk, (size_t)wgs * 16, hipMemcpyDeviceToHost);
        const double per = KIND == 0 ? 13 * 2048.0 : 4 * 4096.0;
        const double flops = per * iters * 4.0 * wgs;
        printf("%s wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s   wave clocks %lld  realtime ticks %lld -> %.2f GHz\n", name, wgs, iters, ms,
               flops / ms / 1e9, h[0], h[1], (double)h[0] / ((double)h[1] * 10.0));
    }
    hipFree(out); hipFree(clk);
}

int main() {
    run<0>(256, 20000, "16x16x4 x13");
    run<0>(512, 20000, "16x16x4 x13");
    run<0>(1024, 20000, "16x16x4 x13");
    run<1>(256, 50000, "32x32x2 x4 ");
    run<1>(1024, 50000, "32x32x2 x4 ");
    run<0>(256, 500, "16x16x4 x13 short");
    run<0>(228, 350, "16x16x4 x13 20us");
    return 0;
}
