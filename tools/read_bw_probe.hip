// read_bw_probe.hip -- streaming read bandwidth on MI355X as the split backward's dW kernel sees it: one block per CU, a few
// loader waves per block moving 1 KB pieces global -> LDS with global_load_lds_dwordx4, against plain dwordx4 loads from
// every wave; buffer sizes around the dW kernel's 210 MB (inside / outside the 256 MB memory-side cache).
//   hipcc --offload-arch=gfx950 -O3 tools/read_bw_probe.hip -o tools/_read_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline void dma16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
// every wave: plain loads, `unroll` in flight per lane
__global__ void plain(const f32x4 *src, size_t n16, float *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        s += a + b + c + d;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// `loaders` waves per block issue DMA pieces into a ring of `depth` KB per wave; nobody reads the LDS (pure fetch rate)
template <int DEPTH>
__global__ void dma(const char *src, size_t bytes, float *out, int loaders) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= loaders) return;
    const size_t per_block = bytes / gridDim.x, per_wave = per_block / loaders;
    const char *p = src + (size_t)blockIdx.x * per_block + (size_t)wave * per_wave + lane * 16;
    char *l = lds + wave * DEPTH * 1024;
    const size_t n = per_wave / 1024 / DEPTH * DEPTH;
    for (size_t k = 0; k < n; k += DEPTH) {
        for (int d = 0; d < DEPTH; ++d) dma16(p + (k + d) * 1024, l + d * 1024);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) out[blockIdx.x * 16 + wave] = ((float *)l)[0];
}
int main() {
    float *out; hipMalloc(&out, 1 << 22);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t mb : {64, 210, 420, 1024}) {
        const size_t bytes = mb << 20;
        char *buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
        auto time = [&](const char *name, auto launch) {
            launch(); hipDeviceSynchronize();
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
            printf("%5zu MB  %-44s %8.1f us  %6.2f TB/s\n", mb, name, best * 1e3, bytes / (best * 1e-3) / 1e12);
        };
        time("plain dwordx4, 1024 blocks x 256", [&] { hipLaunchKernelGGL(plain, dim3(1024), dim3(256), 0, 0, (const f32x4 *)buf, bytes / 16, out); });
        time("plain dwordx4, 2048 blocks x 512", [&] { hipLaunchKernelGGL(plain, dim3(2048), dim3(512), 0, 0, (const f32x4 *)buf, bytes / 16, out); });
        time("dma, 256 blocks, 2 loaders x 12 KB", [&] { hipLaunchKernelGGL(dma<12>, dim3(256), dim3(640), 65536, 0, buf, bytes, out, 2); });
        time("dma, 256 blocks, 2 loaders x 32 KB", [&] { hipLaunchKernelGGL(dma<32>, dim3(256), dim3(640), 65536, 0, buf, bytes, out, 2); });
        time("dma, 256 blocks, 4 loaders x 16 KB", [&] { hipLaunchKernelGGL(dma<16>, dim3(256), dim3(640), 65536, 0, buf, bytes, out, 4); });
        time("dma, 256 blocks, 8 loaders x 16 KB", [&] { hipLaunchKernelGGL(dma<16>, dim3(256), dim3(640), 131072, 0, buf, bytes, out, 8); });
        time("dma, 512 blocks, 4 loaders x 16 KB", [&] { hipLaunchKernelGGL(dma<16>, dim3(512), dim3(640), 65536, 0, buf, bytes, out, 4); });
        hipFree(buf);
    }
    return 0;
}
