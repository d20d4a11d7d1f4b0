// atomic_probe2.hip -- measurement tooling (not part of libnsr.so): what bounds the grid scatter's fp32 atomics on MI355X?
//   A. line-request rate against the number of blocks issuing (is 21 G lines/s a per-CU or a chip-wide limit?)
//   B. cache-policy bits on the atomic (none / sc1 / nt): does any of them change the rate?
//   C. a streaming read next to the atomics: how much of the chip's read bandwidth is left while atomics drain?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_atomic_probe2 tools/atomic_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int POL>
__device__ inline void atom(float* p, float v) {
    if (POL == 0) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
// every half wave updates one random 128-byte voxel row per instruction (the scatter's pattern)
template <int POL>
__global__ void k_atomics(float* tab, unsigned nvox, int iters, long long* stamps) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63, wave = tid >> 6;
    const unsigned grp = lane >> 5, sub = lane & 31;
    if (threadIdx.x == 0 && stamps) stamps[2 * blockIdx.x] = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const unsigned v = hash(wave * 977u + it * 131071u + grp * 7919u) % nvox;
        atom<POL>(tab + (size_t)v * 32 + sub, 1.0f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && stamps) stamps[2 * blockIdx.x + 1] = wall_clock64();
}
// role by block parity: even blocks stream-read `bytes_per_block` (16 B per lane, 8 loads in flight), odd blocks issue atomics
// until they have done `iters` of them
__global__ void k_mixed(const float4* src, size_t n16, float* tab, unsigned nvox, int iters, int read_rounds, int roles, long long* stamps,
                        float* sink) {
    const unsigned lane = threadIdx.x & 63;
    const bool reader = (roles == 1) || (roles == 3 && (blockIdx.x & 1) == 0);
    const bool atomer = (roles == 2) || (roles == 3 && (blockIdx.x & 1) == 1);
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = wall_clock64();
    if (reader) {
        float acc = 0.f;
        size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (int r = 0; r < read_rounds; ++r) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = src[i % n16]; i += stride; }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
        }
        if (acc == 123.456f) sink[0] = acc;
    } else if (atomer) {
        const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const unsigned grp = lane >> 5, sub = lane & 31;
        for (int it = 0; it < iters; ++it) {
            const unsigned v = hash(wave * 977u + it * 131071u + grp * 7919u) % nvox;
            atom<0>(tab + (size_t)v * 32 + sub, 1.0f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}

static double span_us(const std::vector<long long>& st, int nb, int parity /* -1: all */) {
    long long lo = 0, hi = 0; bool first = true;
    for (int b = 0; b < nb; ++b) {
        if (parity >= 0 && (b & 1) != parity) continue;
        if (first) { lo = st[2 * b]; hi = st[2 * b + 1]; first = false; }
        lo = std::min(lo, st[2 * b]); hi = std::max(hi, st[2 * b + 1]);
    }
    return (hi - lo) / 100.0;      // wall_clock64: 100 MHz
}

int main() {
    const unsigned nvox = 178192;  // Replica fine grid
    float* tab; hipMalloc(&tab, (size_t)nvox * 32 * 4); hipMemset(tab, 0, (size_t)nvox * 32 * 4);
    long long* stamps; hipMalloc(&stamps, 8192 * 2 * 8);
    std::vector<long long> st(8192 * 2);
    printf("== A. atomics alone, 256-thread blocks, 256 instructions per wave (2 voxel rows = 4 lines each), by block count\n");
    for (int blocks : {32, 64, 128, 256, 512, 1024, 3072}) {
        const int iters = 256;
        hipLaunchKernelGGL(k_atomics<0>, dim3(blocks), dim3(256), 0, 0, tab, nvox, 8, (long long*)nullptr);
        hipLaunchKernelGGL(k_atomics<0>, dim3(blocks), dim3(256), 0, 0, tab, nvox, iters, stamps);
        hipDeviceSynchronize();
        hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
        const double us = span_us(st, blocks, -1);
        const double lines = (double)blocks * 4 * iters * 4;
        printf("   blocks %5d : %8.1f us  %7.2f G lines/s  (%6.1f lines/us per block)\n", blocks, us, lines / us / 1e3, lines / us / blocks);
    }
    printf("== B. policy bits (1024 blocks)\n");
    for (int pol = 0; pol < 3; ++pol) {
        const int blocks = 1024, iters = 256;
        if (pol == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(blocks), dim3(256), 0, 0, tab, nvox, iters, stamps);
        if (pol == 1) hipLaunchKernelGGL(k_atomics<1>, dim3(blocks), dim3(256), 0, 0, tab, nvox, iters, stamps);
        if (pol == 2) hipLaunchKernelGGL(k_atomics<2>, dim3(blocks), dim3(256), 0, 0, tab, nvox, iters, stamps);
        hipDeviceSynchronize();
        hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
        const double us = span_us(st, blocks, -1);
        printf("   %-6s : %8.1f us  %7.2f G lines/s\n", pol == 0 ? "plain" : (pol == 1 ? "sc1" : "nt"), us, (double)blocks * 4 * iters * 4 / us / 1e3);
    }
    printf("== C. streaming reads (even blocks) next to atomics (odd blocks), 512 blocks of 256 threads\n");
    const size_t nbytes = (size_t)1 << 30;
    float4* src; hipMalloc(&src, nbytes); hipMemset(src, 0, nbytes);
    float* sink; hipMalloc(&sink, 64);
    const int blocks = 512, rounds = 256, iters = 1024;      // reader block: 256 thr * 8 * 16 B * rounds
    for (int roles : {1, 2, 3}) {
        hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(256), 0, 0, src, nbytes / 16, tab, nvox, 8, 8, roles, stamps, sink);
        hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(256), 0, 0, src, nbytes / 16, tab, nvox, iters, rounds, roles, stamps, sink);
        hipDeviceSynchronize();
        hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
        const int nread = roles == 1 ? blocks : blocks / 2, natom = roles == 2 ? blocks : blocks / 2;
        const double rbytes = (double)nread * 256 * 8 * 16 * rounds, lines = (double)natom * 4 * iters * 4;
        if (roles == 1) printf("   readers only (all %d blocks)   : %8.1f us  %6.2f TB/s\n", blocks, span_us(st, blocks, -1), rbytes / span_us(st, blocks, -1) / 1e6);
        if (roles == 2) printf("   atomics only (all %d blocks)   : %8.1f us  %6.2f G lines/s\n", blocks, span_us(st, blocks, -1), lines / span_us(st, blocks, -1) / 1e3);
        if (roles == 3) printf("   mixed: readers %8.1f us  %6.2f TB/s | atomics %8.1f us  %6.2f G lines/s\n", span_us(st, blocks, 0),
                               rbytes / span_us(st, blocks, 0) / 1e6, span_us(st, blocks, 1), lines / span_us(st, blocks, 1) / 1e3);
    }
    return 0;
}
