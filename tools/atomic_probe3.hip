// atomic_probe3.hip -- measurement tooling: does the FORM of the scatter's atomic matter?  global_atomic_add_f32 (64-bit address per lane)
// against buffer_atomic_add_f32 (scalar base + 32-bit offset per lane), full and half-filled instructions, one wave alone up to the whole chip.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_atomic_probe3 tools/atomic_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int FORM, int HALF>
__global__ void k(float* tab, unsigned nvox, int iters, long long* stamps) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63, wave = tid >> 6;
    const unsigned grp = lane >> 5, sub = lane & 31;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tab, 0, 0x7fffffff, 0x00020000);
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const unsigned v = hash(wave * 977u + it * 131071u + grp * 7919u) % nvox;
        if (HALF && grp != (unsigned)(it & 1)) continue;                 // only one half wave active per instruction
        if (FORM == 0) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(tab + (size_t)v * 32 + sub), "v"(1.0f) : "memory");
        else asm volatile("buffer_atomic_add_f32 %0, %1, %2, 0 offen" ::"v"(1.0f), "v"((v * 32u + sub) * 4u), "s"(r) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}
static double span_us(const std::vector<long long>& st, int nb) {
    long long lo = st[0], hi = st[1];
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, st[2 * b]); hi = std::max(hi, st[2 * b + 1]); }
    return (hi - lo) / 100.0;
}
template <int FORM, int HALF>
void run(float* tab, unsigned nvox, long long* stamps, int blocks, int threads) {
    std::vector<long long> st(2 * blocks);
    const int iters = 256;
    hipLaunchKernelGGL((k<FORM, HALF>), dim3(blocks), dim3(threads), 0, 0, tab, nvox, 8, stamps);
    hipLaunchKernelGGL((k<FORM, HALF>), dim3(blocks), dim3(threads), 0, 0, tab, nvox, iters, stamps);
    hipDeviceSynchronize();
    hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
    const double us = span_us(st, blocks);
    const double instr = (double)blocks * (threads / 64) * iters, lines = instr * (HALF ? 2 : 4);
    printf("   %-6s %-4s blocks %4d x %2d waves : %8.1f us  %6.2f G lines/s  %7.2f instr/us per wave\n", FORM ? "buffer" : "global", HALF ? "half" : "full",
           blocks, threads / 64, us, lines / us / 1e3, (double)iters / us);
}
int main() {
    const unsigned nvox = 178192;
    float* tab; hipMalloc(&tab, (size_t)nvox * 32 * 4); hipMemset(tab, 0, (size_t)nvox * 32 * 4);
    long long* stamps; hipMalloc(&stamps, 8192 * 2 * 8);
    for (int blocks : {1, 32, 256}) for (int threads : {64, 256, 768}) {
        run<0, 0>(tab, nvox, stamps, blocks, threads); run<1, 0>(tab, nvox, stamps, blocks, threads);
        run<0, 1>(tab, nvox, stamps, blocks, threads); run<1, 1>(tab, nvox, stamps, blocks, threads);
    }
    return 0;
}
