// coexec_probe.hip -- do fp32 MFMAs and fp32 vector instructions of DIFFERENT waves on one SIMD overlap on MI355X?
// A block of 8 waves = 2 per SIMD: waves 0..3 run a chain-free v_mfma_f32_16x16x4_f32 stream, waves 4..7 a chain-free stream of another
// instruction class (v_fma_f32 / packed v_pk_fma_f32 / integer v_and_or / ds_read_b32 / s_* scalar); each side alone and both together.
// If the pair takes max(alone_a, alone_b) the pipes are separate; if it takes the sum they share the datapath (or the issue port).
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_probe.hip -o tools/_coexec_probe && tools/_coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// mode bit 0: MFMA waves run; bit 1: the other waves run; kind: what the other waves issue
template <int KIND, int MK = 0>
__global__ __launch_bounds__(512) void probe(float *out, int it_mfma, int it_other, int mode, int swap, int prio) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + 1e-3f * i;
    __syncthreads();
    float s = 0.f;
    const bool mfma_wave = (mode & 4) ? false : (swap ? wave >= 4 : wave < 4);      // swap: the YOUNGER half of the block issues the MFMAs; mode bit 2: every wave is an `other` wave
    if (!mfma_wave && prio == 1) __builtin_amdgcn_s_setprio(3);
    if (mfma_wave && prio == 2) __builtin_amdgcn_s_setprio(3);
    if (mfma_wave) {
        if (mode & 1) {
            f32x4 acc[8];
            for (int n = 0; n < 8; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float a = 1.0f + lane, b = 2.0f - lane;
            if (MK == 0) {
                for (int it = 0; it < it_mfma; ++it)
#pragma unroll
                    for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
            } else {                        // bf16 16x16x16 (the matrix unit proper: 16x the fp32 rate)
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                const s16x4 ab = {(short)(0x3f80 + lane), (short)0x3f81, (short)0x3f82, (short)0x3f83}, bb = {(short)0x4000, (short)(0x4001 + lane), (short)0x4002, (short)0x4003};
                for (int it = 0; it < it_mfma; ++it)
#pragma unroll
                    for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, acc[n], 0, 0, 0);
            }
            for (int n = 0; n < 8; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
        }
    } else if (mode & 2) {
        if (KIND == 0) {                    // 8 independent v_fma_f32 chains
            float v[8];
            for (int n = 0; n < 8; ++n) v[n] = 1.0f + lane + n;
            const float m = 1.0000001f, c = 1e-7f;
            for (int it = 0; it < it_other; ++it)
#pragma unroll
                for (int n = 0; n < 8; ++n) v[n] = __builtin_fmaf(v[n], m, c);
            for (int n = 0; n < 8; ++n) s += v[n];
        } else if (KIND == 1) {             // 8 independent v_pk_fma_f32 chains
            f32x2 v[8];
            for (int n = 0; n < 8; ++n) v[n] = f32x2{1.0f + lane + n, 2.0f + n};
            const f32x2 m{1.0000001f, 1.0000002f}, c{1e-7f, 2e-7f};
            for (int it = 0; it < it_other; ++it)
#pragma unroll
                for (int n = 0; n < 8; ++n) v[n] = __builtin_elementwise_fma(v[n], m, c);
            for (int n = 0; n < 8; ++n) s += v[n][0] + v[n][1];
        } else if (KIND == 2) {             // integer vector ops
            unsigned v[8];
            for (int n = 0; n < 8; ++n) v[n] = 12345u + lane * 7 + n;
            for (int it = 0; it < it_other; ++it)
#pragma unroll
                for (int n = 0; n < 8; ++n) v[n] = (v[n] & 0x7fffffffu) | (unsigned)(it + n);
            for (int n = 0; n < 8; ++n) s += (float)v[n];
        } else if (KIND == 3) {             // LDS reads
            float v = 0.f;
            int idx = lane;
            for (int it = 0; it < it_other; ++it) {
#pragma unroll
                for (int n = 0; n < 8; ++n) v += lds[(idx + 64 * n) & 4095];
                idx += 1;
            }
            s += v;
        } else {                            // scalar ALU
            int u = __builtin_amdgcn_readfirstlane(wave + 3);
            for (int it = 0; it < it_other; ++it) {
#pragma unroll
                for (int n = 0; n < 8; ++n) { u = u * 3 + n; asm volatile("" : "+s"(u)); }
            }
            s += (float)u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int MK = 0>
float time_ms(float *out, int it_mfma, int it_other, int mode, int swap = 0, int prio = 0) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<KIND, MK>), dim3(256), dim3(512), 0, 0, out, 10, 10, mode, swap, prio);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<KIND, MK>), dim3(256), dim3(512), 0, 0, out, it_mfma, it_other, mode, swap, prio);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int KIND>
void run(const char *name, float *out) {
    const int it_mfma = 4000;                       // 32 000 MFMAs per MFMA wave
    // calibrate the other side to about the MFMA side's duration
    const float tm = time_ms<KIND>(out, it_mfma, 0, 1);
    int it_other = 4000;
    float to = time_ms<KIND>(out, 0, it_other, 2);
    it_other = (int)(it_other * tm / to);
    to = time_ms<KIND>(out, 0, it_other, 2);
    const float tb = time_ms<KIND>(out, it_mfma, it_other, 3);
    const float tbs = time_ms<KIND>(out, it_mfma, it_other, 3, 1, 0), tbp = time_ms<KIND>(out, it_mfma, it_other, 3, 0, 1), tbq = time_ms<KIND>(out, it_mfma, it_other, 3, 0, 2),
                tbsp = time_ms<KIND>(out, it_mfma, it_other, 3, 1, 1);
    const float t2 = time_ms<KIND>(out, 0, it_other, 6);
    printf("%-14s two `other` waves per SIMD, no MFMA: %7.3f ms = %5.2f cycles per instruction per SIMD (one wave alone: %5.2f)\n", name, t2, t2 * 1e-3 * 2.4e9 / (it_other * 16.0), to * 1e-3 * 2.4e9 / (it_other * 8.0));
    printf("%-14s both, MFMA waves younger %7.3f | other waves at s_setprio 3 %7.3f | MFMA waves at s_setprio 3 %7.3f | younger MFMA + other at prio 3 %7.3f ms\n", name, tbs, tbp, tbq, tbsp);
    printf("%-14s MFMA alone %7.3f ms (%5.1f cycles per MFMA per SIMD at 2.4 GHz) | other alone %7.3f ms (%5.2f cycles per instruction) | both %7.3f ms = %4.2f x max, %4.2f x sum\n",
           name, tm, tm * 1e-3 * 2.4e9 / (it_mfma * 8.0), to, to * 1e-3 * 2.4e9 / (it_other * 8.0), tb, tb / (tm > to ? tm : to), tb / (tm + to));
}
// the same pairing with a bf16 MFMA stream (v_mfma_f32_16x16x16_bf16): does the matrix unit proper overlap with vector instructions?
template <int KIND>
void run_bf16(const char *name, float *out) {
    const int it_mfma = 16000;
    const float tm = time_ms<KIND, 1>(out, it_mfma, 0, 1);
    int it_other = 4000;
    float to = time_ms<KIND, 1>(out, 0, it_other, 2);
    it_other = (int)(it_other * tm / to);
    to = time_ms<KIND, 1>(out, 0, it_other, 2);
    const float tb = time_ms<KIND, 1>(out, it_mfma, it_other, 3), tbs = time_ms<KIND, 1>(out, it_mfma, it_other, 3, 1, 0);
    printf("bf16 16x16x16 + %-12s MFMA alone %7.3f ms (%5.1f cycles per MFMA per SIMD at 2.4 GHz) | other alone %7.3f ms | both %7.3f ms = %4.2f x max, %4.2f x sum | MFMA waves younger: %7.3f ms\n",
           name, tm, tm * 1e-3 * 2.4e9 / (it_mfma * 8.0), to, tb, tb / (tm > to ? tm : to), tb / (tm + to), tbs);
}
int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run_bf16<0>("v_fma_f32", out);
    run_bf16<1>("v_pk_fma_f32", out);
    run_bf16<2>("v_and_or_b32", out);
    run_bf16<3>("ds_read_b32", out);
    run<0>("v_fma_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<2>("v_and_or_b32", out);
    run<3>("ds_read_b32", out);
    run<4>("s_mul/add", out);
    return 0;
}
