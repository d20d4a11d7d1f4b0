// atomic_probe4.hip -- measurement tooling (not part of libnsr.so): is the chip's ~20.5 G lines/s of fp32 atomics (atomic_probe2) a
// MEMORY-side limit or a cross-XCD ownership limit?
//   A. one shared table, every wave updates random 128-byte voxel rows of it (the scatter's pattern; baseline)
//   B. eight slabs indexed by the block's XCC_ID: no line is ever touched from two XCDs
//   C. a slab per BLOCK (CU-private rows)
//   D. each of the above with the cache-policy / scope bits an atomic can carry (none = agent scope as hipcc emits it, sc0 = returning
//      form is not used here, sc1 = system scope, nt) and with a small working set (rows that fit one XCD's 4 MB L2)
//   E. the same row updates as PLAIN 128-byte stores (what the vector-memory path does without the read-modify-write)
// If B (or C) ran >= 2x A, XCD-private accumulation + a touched-row reduce would pay for large batches; if not, the rate is the
// memory side's and only fewer lines help.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_atomic_probe4 tools/atomic_probe4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

template <int POL>
__device__ inline void atom(float* p, float v) {
    if (POL == 0) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(p), "v"(v) : "memory");   // same line count per row at half the bytes? (rate only)
}
// SLAB 0: one table; 1: slab = XCC_ID; 2: slab = block
template <int POL, int SLAB>
__global__ void k(float* tab, unsigned nvox, int iters, long long* stamps, unsigned* xcc_seen) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63, wave = tid >> 6;
    const unsigned grp = lane >> 5, sub = lane & 31;
    const unsigned xcc = xcc_id();
    float* base = tab;
    if (SLAB == 1) base = tab + (size_t)xcc * nvox * 32;
    if (SLAB == 2) base = tab + (size_t)blockIdx.x * nvox * 32;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = wall_clock64(); if (xcc_seen) xcc_seen[blockIdx.x] = xcc; }
    for (int it = 0; it < iters; ++it) {
        const unsigned v = hash(wave * 977u + it * 131071u + grp * 7919u) % nvox;
        atom<POL>(base + (size_t)v * 32 + sub, 1.0f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}
static double span_us(const std::vector<long long>& st, int nb) {
    long long lo = st[0], hi = st[1];
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, st[2 * b]); hi = std::max(hi, st[2 * b + 1]); }
    return (hi - lo) / 100.0;      // wall_clock64: 100 MHz
}
static const char* pol_name(int p) { return p == 0 ? "atomic" : p == 1 ? "atomic sc1" : p == 2 ? "atomic nt" : p == 3 ? "plain store" : "pk_add_bf16"; }
static const char* slab_name(int s) { return s == 0 ? "one table" : s == 1 ? "slab per XCD" : "slab per block"; }

template <int POL, int SLAB>
void run(float* tab, size_t tab_rows, unsigned nvox, long long* stamps, int blocks, int threads, int iters = 256) {
    const size_t need = SLAB == 0 ? nvox : (SLAB == 1 ? 8ull * nvox : (size_t)blocks * nvox);
    if (need > tab_rows) { printf("   %-12s %-14s rows %7u blocks %4d: skipped (table too small)\n", pol_name(POL), slab_name(SLAB), nvox, blocks); return; }
    std::vector<long long> st(2 * blocks);
    hipLaunchKernelGGL((k<POL, SLAB>), dim3(blocks), dim3(threads), 0, 0, tab, nvox, 8, stamps, (unsigned*)nullptr);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<POL, SLAB>), dim3(blocks), dim3(threads), 0, 0, tab, nvox, iters, stamps, (unsigned*)nullptr);
        hipDeviceSynchronize();
        hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
        best = std::min(best, span_us(st, blocks));
    }
    const double instr = (double)blocks * (threads / 64) * iters, lines = instr * 4;
    printf("   %-12s %-14s rows %7u  blocks %4d x %2d waves : %9.1f us  %6.2f G lines/s  (%5.1f lines/us per block)\n", pol_name(POL), slab_name(SLAB),
           nvox, blocks, threads / 64, best, lines / best / 1e3, lines / best / blocks);
}

int main() {
    const size_t tab_rows = 8ull * 178192;                    // 8 slabs of the Replica fine grid = 182 MB
    float* tab; hipMalloc(&tab, tab_rows * 32 * 4); hipMemset(tab, 0, tab_rows * 32 * 4);
    long long* stamps; hipMalloc(&stamps, 8192 * 2 * 8);
    unsigned* seen; hipMalloc(&seen, 8192 * 4);
    {   // where do blocks land?
        hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(256), 0, 0, tab, 1024u, 1, stamps, seen);
        hipDeviceSynchronize();
        std::vector<unsigned> h(256); hipMemcpy(h.data(), seen, 1024, hipMemcpyDeviceToHost);
        int same = 0; unsigned hist[16] = {0};
        for (int b = 0; b < 256; ++b) { same += h[b] == (unsigned)(b & 7); hist[h[b] & 15]++; }
        printf("XCC_ID of block b == b %% 8 for %d of 256 blocks; blocks per XCC:", same);
        for (int x = 0; x < 8; ++x) printf(" %u", hist[x]);
        printf("\n");
    }
    printf("== 1. large working set (178 192 rows = 22.8 MB per slab; the Replica fine grid), 256 blocks x 12 waves, 256 row-pair updates per wave\n");
    run<0, 0>(tab, tab_rows, 178192, stamps, 256, 768); run<0, 1>(tab, tab_rows, 178192, stamps, 256, 768);
    run<1, 0>(tab, tab_rows, 178192, stamps, 256, 768); run<1, 1>(tab, tab_rows, 178192, stamps, 256, 768);
    run<2, 0>(tab, tab_rows, 178192, stamps, 256, 768); run<2, 1>(tab, tab_rows, 178192, stamps, 256, 768);
    run<3, 0>(tab, tab_rows, 178192, stamps, 256, 768); run<3, 1>(tab, tab_rows, 178192, stamps, 256, 768);
    run<4, 0>(tab, tab_rows, 178192, stamps, 256, 768);
    printf("== 2. working set that fits ONE XCD's L2 (8 192 rows = 1 MB per slab)\n");
    run<0, 0>(tab, tab_rows, 8192, stamps, 256, 768); run<0, 1>(tab, tab_rows, 8192, stamps, 256, 768);
    run<1, 1>(tab, tab_rows, 8192, stamps, 256, 768); run<2, 1>(tab, tab_rows, 8192, stamps, 256, 768);
    run<3, 0>(tab, tab_rows, 8192, stamps, 256, 768); run<3, 1>(tab, tab_rows, 8192, stamps, 256, 768);
    printf("== 3. rows private to a block (512 rows = 64 KB per block: what a CU-private write-back region would see)\n");
    run<0, 2>(tab, tab_rows, 512, stamps, 256, 768); run<1, 2>(tab, tab_rows, 512, stamps, 256, 768); run<3, 2>(tab, tab_rows, 512, stamps, 256, 768);
    run<0, 2>(tab, tab_rows, 4096, stamps, 256, 768);
    printf("== 4. one XCD alone (32 blocks, every 8th block id would be needed to pin them: here simply 32 blocks) and block-count scaling, one table\n");
    for (int blocks : {8, 32, 64, 128, 256, 512}) { run<0, 0>(tab, tab_rows, 178192, stamps, blocks, 768); run<0, 1>(tab, tab_rows, 178192, stamps, blocks, 768); }
    printf("== 5. four waves per block (fewer requests in flight per CU)\n");
    run<0, 0>(tab, tab_rows, 178192, stamps, 256, 256); run<0, 1>(tab, tab_rows, 178192, stamps, 256, 256); run<3, 0>(tab, tab_rows, 178192, stamps, 256, 256);
    return 0;
}
