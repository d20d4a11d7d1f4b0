// lds_atomic_probe.hip -- cost of ds_add_f32 (no return) per wave instruction vs bank-conflict degree and vs plain ds_write
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0: conflict-free atomic, 1: 4-way (stride-32 rows), 2: plain read-add-write (non atomic), 3: 2-way
__global__ void probe(float* out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    int base;
    if (MODE == 0) base = g * 16 + j;                 // 64 consecutive floats
    else if (MODE == 1) base = g * 4 * 32 + j;        // rows 4g of a stride-32 matrix
    else if (MODE == 3) base = g * 4 * 93 + j;        // stride 93
    else base = g * 16 + j;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        const int off = ((it * 67) & 127) * 64;
        if (MODE == 2) { lds[(base + off) & 16383] += v; }
        else atomicAdd(&lds[(base + off) & 16383], v);
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x];
}
template <int MODE> void run(const char* name, int waves) {
    float* out; hipMalloc(&out, 256 * 64 * 4);
    const int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves), 65536, 0, out, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves), 65536, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // cycles per wave-instruction per CU (LDS is per CU): time * 2.4e9 / (iters * waves)
    printf("%-28s waves/CU=%2d : %7.3f ms  -> %6.1f cycles per wave-instr (CU-serialised)\n", name, waves, ms, ms * 1e-3 * 2.4e9 / (iters * (double)waves));
    hipFree(out);
}
int main() {
    for (int w : {1, 4, 8}) {
        run<0>("ds_add_f32 conflict-free", w); run<1>("ds_add_f32 4-way (stride32)", w); run<3>("ds_add_f32 stride93", w); run<2>("plain rmw (non-atomic)", w);
    }
    return 0;
}
