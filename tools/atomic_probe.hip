// atomic_probe.hip -- what does a global f32 atomic cost on MI355X as a function of (lanes per 64B line, table size)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// each wave instruction: 64/LPL distinct random 128-byte voxels, LPL lanes each covering LPL consecutive floats
template <int LPL>
__global__ void probe(float* tab, unsigned nvox, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63, wave = tid >> 6;
    const unsigned grp = lane / LPL, sub = lane % LPL;
    for (int it = 0; it < iters; ++it) {
        const unsigned v = hash(wave * 977u + it * 131071u + grp * 7919u) % nvox;
        unsafeAtomicAdd(tab + (size_t)v * 32 + sub + (it & 1) * (32 - LPL), 1.0f);
    }
}
template <int LPL>
void run(float* tab, unsigned nvox, const char* name) {
    const int blocks = 3072, threads = 256, iters = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<LPL>, dim3(blocks), dim3(threads), 0, 0, tab, nvox, 8);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<LPL>, dim3(blocks), dim3(threads), 0, 0, tab, nvox, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lanes = (double)blocks * threads * iters;
    printf("%-10s nvox=%8u  LPL=%2d : %7.3f ms  %7.2f G lane-atomics/s  %7.2f G line-requests/s\n", name, nvox, LPL, ms,
           lanes / ms / 1e6, lanes / LPL / ms / 1e6);
}
int main() {
    const unsigned big = 178192;  // Replica fine grid voxels
    float* tab; hipMalloc(&tab, (size_t)big * 32 * 4); hipMemset(tab, 0, (size_t)big * 32 * 4);
    for (unsigned nvox : {616u, 21756u, big}) {
        run<1>(tab, nvox, "1/line"); run<4>(tab, nvox, "4/line"); run<16>(tab, nvox, "16/line"); run<32>(tab, nvox, "32/128B");
    }
    return 0;
}
