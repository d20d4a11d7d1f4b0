// mfma_probe.hip -- issue rate of v_mfma_f32_16x16x4_f32 on MI355X: cycles per instruction per SIMD as a function of
// waves per SIMD, independent accumulators per wave, and where the A / B operands come from (constant registers vs fresh
// LDS reads, the dW kernel's pattern).   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int LDS>
__global__ void probe(float *out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f + 1e-3f * i;
    __syncthreads();
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane, b = 2.0f - lane;
    for (int it = 0; it < iters; ++it) {
        float av[4], bv[4];
        for (int q = 0; q < 4; ++q) {
            if (LDS) { av[q] = lds[((it & 7) * 8 + q) * 64 + lane]; bv[q] = lds[((it & 7) * 8 + 4 + q) * 64 + lane]; }
            else { av[q] = a; bv[q] = b; }
        }
        for (int q = 0; q < 4; ++q)
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q] + n, bv[q], acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int LDS>
void run(int waves_per_simd) {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<NACC, LDS>), dim3(256), dim3(threads), 32768, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<NACC, LDS>), dim3(256), dim3(threads), 32768, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfma_per_simd = (double)iters * 4 * NACC * waves_per_simd;
    const double tflops = mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12;
    printf("acc=%2d lds=%d waves/SIMD=%d : %8.3f ms  %6.1f ns per MFMA per SIMD  (%6.1f cycles at 2.4 GHz)  %6.1f TFLOP/s\n", NACC, LDS, waves_per_simd, ms,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, tflops);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) { run<1, 0>(w); run<2, 0>(w); run<8, 0>(w); run<8, 1>(w); }
    return 0;
}
