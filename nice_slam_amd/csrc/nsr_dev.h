// nsr_dev.h -- device-side primitives for gfx950 (CDNA4).  Pure HIP; the only build target.
// (tests/emu/ shadows this header with a host re-implementation of the same names so that the
//  kernel sources can be executed lane-by-lane on a CPU in the unit tests; that shim is test
//  infrastructure and is never part of libnsr.so.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NSR_DEV __device__ __forceinline__
#define NSR_KERNEL __global__
#define NSR_BOUNDS(n) __launch_bounds__(n)

namespace nsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct F4 { float x, y, z, w; };   // plain 16-byte POD used for vector loads/stores

// D = A(16x4) * B(4x16) + C, exact fp32 fma chain.  Lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; holds D[i=4*(l>>4)+r][j=l&15] in element r.  (v_mfma_f32_16x16x4_f32)
NSR_DEV f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

NSR_DEV int tid() { return (int)threadIdx.x; }
NSR_DEV int nthreads() { return (int)blockDim.x; }
NSR_DEV int bid_x() { return (int)blockIdx.x; }
NSR_DEV int f2i_rn(float x) { return __float2int_rn(x); }      // round half to even (cvRound)
// max(x, 0) as ONE instruction: on the bit pattern, v_max_i32(x, 0) (negative floats are negative integers).  fmaxf / fmed3
// cost two (canonicalise + max), and an inline-asm v_max_f32 right after an MFMA escapes the compiler's hazard recogniser
// (no wait states inserted, stale accumulators read).
NSR_DEV float relu1(float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }
NSR_DEV int bid_y() { return (int)blockIdx.y; }
NSR_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }      // pin a wave-uniform value to an SGPR
NSR_DEV int nblk_x() { return (int)gridDim.x; }

NSR_DEV unsigned long long ballot64(bool p) { return __ballot(p); }            // bit l = lane l's predicate
NSR_DEV float shfl(float v, int src) { return __shfl(v, src, 64); }
NSR_DEV int shfl_i(int v, int src) { return __shfl(v, src, 64); }
NSR_DEV double shfl_d(double v, int src) { return __shfl(v, src, 64); }
NSR_DEV float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
NSR_DEV double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }
NSR_DEV float shfl_up(float v, int d) { return __shfl_up(v, (unsigned)d, 64); }
NSR_DEV float shfl_down(float v, int d) { return __shfl_down(v, (unsigned)d, 64); }

// make this wave's earlier LDS writes visible to its own later LDS reads (other lanes)
NSR_DEV void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// keep the instruction scheduler from hoisting the next operand stream above this point
NSR_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// the same at sites that A/B builds may switch off (tools/build_ts.sh -DNSR_X_NOFENCE_GEMV / -DNSR_X_NOFENCE_EMB)
NSR_DEV void sched_fence_gemv() {
#ifndef NSR_X_NOFENCE_GEMV
    __builtin_amdgcn_sched_barrier(0);
#endif
}
NSR_DEV void sched_fence_emb() {
#ifndef NSR_X_NOFENCE_EMB
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// keep a loaded value (and thereby its load) alive up to this point without doing anything with it
NSR_DEV void keep_alive(float v) { asm volatile("" ::"v"(v)); }
NSR_DEV void keep_alive_d(double v) { asm volatile("" ::"v"(v)); }
// an opaque copy of a per-lane integer: what is derived from it inside a loop body is recomputed there instead of being hoisted out of
// the loop and kept in registers across it (the forward's pass kernel at its 168-register cap: lane-dependent LDS / shuffle indices
// hoisted out of the tile loop were spilled and re-loaded from scratch inside it: 60 -> 28 B of scratch)
NSR_DEV int opaque_i(int v) { asm volatile("" : "+v"(v)); return v; }
// compiler-only memory clobber: stops loop-invariant code motion of loads across loop iterations
NSR_DEV void loop_fence() { asm volatile("" ::: "memory"); }
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), i.e. every barrier would wait for all outstanding global loads, stores and fire-and-forget
// atomics of the wave (grid scatter, gradient-image stores: microseconds each).  No barrier in these kernels hands
// global data from one wave to another, so only lgkmcnt is drained (cdna_hip_programming.md, "pipelining across barriers").
NSR_DEV void block_sync() { __syncthreads(); }

// Profiling stamps (tests/perf/ts_probe.py): compiled in only with -DNSR_TS (tools/build_ts.sh); the product build has none.
struct Dbg {
    long long *p;        // this wave's slot array (64 entries), or NULL
    mutable long long last = 0;
    // slot: the time of the LAST passage; slot + 16: the time spent between the previous stamp and this one, summed over all
    // passages (the per-phase totals of a wave over its tiles); slot + 32: number of passages
    NSR_DEV void stamp(int slot) const {
#ifdef NSR_TS
        const long long t = (long long)__builtin_amdgcn_s_memtime();
        if (p && (threadIdx.x & 63) == 0) {
            p[slot] = t;
            if (last) { p[16 + slot] += t - last; p[32 + slot] += 1; }
        }
        last = t;
#else
        (void)slot;
#endif
    }
    NSR_DEV void note(int slot, long long v) const {
#ifdef NSR_TS
        if (p && (threadIdx.x & 63) == 0) p[slot] = v;
#else
        (void)slot; (void)v;
#endif
    }
};

// Pull the 64-byte line of `p` towards the L2 without a destination register: a global -> LDS load (gfx950
// global_load_lds_dword) into a sink region nobody reads.  A plain load kept alive in a VGPR gets spilled by the register
// allocator of the backward kernel -- `s_waitcnt vmcnt(0)` + scratch store right behind every such load, i.e. the full memory
// latency eight times in a row (measured: 6.7k cycles per tile).
NSR_DEV void prefetch_line(const float *p, float *lds_sink) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                     (__attribute__((address_space(3))) void *)lds_sink, 4, 0, 0);
}
// 16 bytes per lane, global -> LDS without a destination register (gfx950 global_load_lds_dwordx4): lane l's bytes land at
// lds_base + 16 l, i.e. one wave instruction moves 1 KB; completion is tracked by the vector-memory counter (dma_wait).
NSR_DEV void dma16(const float *gsrc, float *lds_base, int /*lane*/) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}
// wait until at most N of this wave's vector-memory requests are outstanding (they complete in order)
template <int N> NSR_DEV void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS flag words between the waves of a block (producer / consumer hand-offs without a block barrier): a store that orders
// the wave's earlier LDS traffic before it, a load that orders the later traffic after it, and the pause of a polling loop.
// (The asm "memory" clobbers keep the compiler from moving LDS accesses across; LDS operations of one wave execute in order.)
// The words live in LDS and are accessed as such (ds_write_b32 / ds_read_b32): through a generic pointer the compiler emits
// flat accesses, whose completion is counted by vmcnt -- a polling loader would drain its own DMA queue on every poll.
typedef __attribute__((address_space(3))) int lds_int;
NSR_DEV void flag_store(int *p, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(volatile lds_int *)p = v;
}
NSR_DEV int flag_load(const int *p) {
    const int v = *(const volatile lds_int *)p;
    asm volatile("" ::: "memory");
    return v;
}
NSR_DEV void spin_pause() { __builtin_amdgcn_s_sleep(1); }
// Atomics with an explicit address space.  Through a generic pointer the compiler emits FLAT instructions (flat_atomic_add_f32,
// flat_load, flat_atomic_cmpswap): they occupy the LDS and the vector-memory queue at once, count on vmcnt AND lgkmcnt -- the dX
// kernel's scatter walk (nsr_kernels.h) waited for all of its outstanding grid atomics at every `s_waitcnt vmcnt(0) lgkmcnt(0)`
// behind a hot-table probe -- and a select between an LDS and a global target becomes ONE flat atomic on a selected pointer.
// The casts below pin the address space: global_atomic_add_f32 / ds_add_f32 / ds_cmpst_rtn_b32 / ds_read_b32.
typedef __attribute__((address_space(1))) float nsr_gfloat;
typedef __attribute__((address_space(1))) double nsr_gdouble;
typedef __attribute__((address_space(1))) unsigned nsr_guint;
typedef __attribute__((address_space(1))) unsigned long long nsr_gu64;
typedef __attribute__((address_space(3))) float nsr_lfloat;
NSR_DEV void atomic_add_global(float *p, float v) {
    __hip_atomic_fetch_add((nsr_gfloat *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same at a WAVE-UNIFORM base + a 32-bit per-lane byte offset: global_atomic_add_f32 voffset, vdata, s[base] -- the per-lane 64-bit
// address (a 64-bit shift + a 64-bit add per atomic) becomes one v_lshl_add_u32.  The array must be smaller than 4 GB.
NSR_DEV void atomic_add_global_off(float *base, unsigned byte_off, float v) {
    __hip_atomic_fetch_add((nsr_gfloat *)(reinterpret_cast<char *>(base) + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NSR_DEV void atomic_add_lds(float *p, float v) {
    __hip_atomic_fetch_add((nsr_lfloat *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
NSR_DEV void atomic_add_lds_i(int *p, int v) {
    __hip_atomic_fetch_add((lds_int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
NSR_DEV int atomic_fetch_add_lds_i(int *p, int v) {
    return __hip_atomic_fetch_add((lds_int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
NSR_DEV int atomic_cas_lds_i(int *p, int expect, int v) {      // returns the old value
    __hip_atomic_compare_exchange_strong((lds_int *)p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return expect;
}
// plain LDS accesses through a pointer whose provenance the compiler has lost (table pointers handed around in structs)
NSR_DEV int lds_load_i(const int *p) { return *(const volatile lds_int *)p; }
NSR_DEV float lds_load_f(const float *p) { return *(const nsr_lfloat *)p; }
NSR_DEV void atomic_add_global_d(double *p, double v) {
    __hip_atomic_fetch_add((nsr_gdouble *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NSR_DEV unsigned long long atomic_fetch_add_global_u64(unsigned long long *p, unsigned long long v) {
    return __hip_atomic_fetch_add((nsr_gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// max of non-negative floats (their bit patterns order like unsigned integers)
NSR_DEV void atomic_max_pos(float *p, float v) {
    __hip_atomic_fetch_max((nsr_guint *)reinterpret_cast<unsigned *>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// returning flavours (the launch-wide hand-off of the window kernel: the value is back before the wave goes on)
NSR_DEV unsigned atomic_fetch_max_u32(unsigned *p, unsigned v) {
    return __hip_atomic_fetch_max((nsr_guint *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NSR_DEV unsigned atomic_exchange_u32(unsigned *p, unsigned v) {
    return __hip_atomic_exchange((nsr_guint *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NSR_DEV void keep_alive_u(unsigned v) { asm volatile("" ::"v"(v)); }

// One byte at a WAVE-UNIFORM address through the scalar cache (s_load_dword of the aligned word): counted by lgkmcnt, not by the
// in-order vector-memory counter -- a vector load issued behind a wave's scatter atomics could only be waited for together
// with all of them.  The array must not be written by the launch that reads it this way.
NSR_DEV unsigned uniform_load_u8(const unsigned char *p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(a & ~3ull)), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long base = ((unsigned long long)hi << 32) | lo;
    unsigned w;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(base) : "memory");
    return (w >> (8u * (__builtin_amdgcn_readfirstlane((unsigned)a) & 3u))) & 255u;
}

NSR_DEV char *lds_base() {
    extern __shared__ __attribute__((aligned(16))) char nsr_lds_[];
    return nsr_lds_;
}

// Operand streams (packed weights / flat parameter blob) are read with buffer loads: the 128-bit
// descriptor and the per-load offset live in SGPRs, the lane offset is ONE shared VGPR -- the
// 64-bit per-row address pairs a plain pointer walk needs would otherwise eat >150 VGPRs.
struct Stream { __amdgpu_buffer_rsrc_t rsrc; };
NSR_DEV Stream make_stream(const float *base) {
    Stream s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    return s;
}
// value at base[lane_off + const_off]   (offsets in floats; const_off is wave-uniform)
NSR_DEV float stream_ld(const Stream &s, int lane_off, int const_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, lane_off * 4, const_off * 4, 0));
}

NSR_DEV void stream_st(const Stream &s, int lane_off, int const_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), s.rsrc, lane_off * 4, const_off * 4, 0);
}

NSR_DEV F4 ld4(const float *p) { float4 v = *reinterpret_cast<const float4 *>(p); return F4{v.x, v.y, v.z, v.w}; }
NSR_DEV void st4(float *p, F4 v) { *reinterpret_cast<float4 *>(p) = make_float4(v.x, v.y, v.z, v.w); }

}  // namespace nsr

#define NSR_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, (hipStream_t)(stream), __VA_ARGS__)
