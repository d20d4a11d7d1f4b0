// tests/emu/nsr_dev.h -- TEST INFRASTRUCTURE.  Host re-implementation of the device primitives in
// nice_slam_amd/csrc/nsr_dev.h so that the *same kernel sources* (nsr_kernels.h, nsr_api.cpp) can be
// executed on a CPU, lane by lane, in the unit tests that run without a GPU.
//
// Every GPU thread is a user-level fiber; the 64 fibers of a wave rendezvous at each cross-lane
// primitive (MFMA, shuffles, wave fences), all fibers of a block at block_sync().  The MFMA lane
// layout implemented here is the documented v_mfma_f32_16x16x4_f32 layout; the first run on real
// hardware validates that assumption (tests/test_hip_parity.py).
// Never linked into libnsr.so and never reachable from the nice_slam_amd package.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define NSR_DEV inline
#define NSR_KERNEL
#define NSR_BOUNDS(n)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace nsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct F4 { float x, y, z, w; };

namespace emu {

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    int tid = 0;
    bool done = false;
    int wait = 0;                  // 0 runnable, 1 wave rendezvous, 2 block rendezvous
    unsigned long wait_gen = 0;
    unsigned long ncoll = 0;       // collectives executed by this lane (selects the exchange buffer)
};

struct Slot { unsigned char b[16]; };

struct Block {
    int nthreads = 0;
    dim3 bid, gdim;
    char *lds = nullptr;
    Fiber *fibers = nullptr;
    Fiber *cur = nullptr;
    void *sched_sp = nullptr;
    int wave_arrived[32];
    unsigned long wave_gen[32];
    int block_arrived = 0;
    unsigned long block_gen = 0;
    Slot xch[32][2][64];
    const std::function<void()> *body = nullptr;
};

extern thread_local Block *B;
void yield_to_scheduler();
void launch(dim3 grid, dim3 block, int lds_bytes, const std::function<void()> &body);

inline void wave_sync() {
    Block *b = B;
    Fiber *f = b->cur;
    const int w = f->tid >> 6;
    const unsigned long my = b->wave_gen[w];
    if (++b->wave_arrived[w] == 64) {
        b->wave_arrived[w] = 0;
        b->wave_gen[w] = my + 1;
    } else {
        f->wait = 1;
        f->wait_gen = my;
        yield_to_scheduler();
    }
}
inline void block_sync_impl() {
    Block *b = B;
    Fiber *f = b->cur;
    const unsigned long my = b->block_gen;
    if (++b->block_arrived == b->nthreads) {
        b->block_arrived = 0;
        b->block_gen = my + 1;
    } else {
        f->wait = 2;
        f->wait_gen = my;
        yield_to_scheduler();
    }
}
// deposit `n` bytes, rendezvous, return the wave's slot array for reading
inline const Slot *exchange(const void *src, int n) {
    Block *b = B;
    Fiber *f = b->cur;
    const int w = f->tid >> 6, lane = f->tid & 63;
    Slot *buf = b->xch[w][f->ncoll & 1];
    f->ncoll++;
    std::memcpy(buf[lane].b, src, n);
    wave_sync();
    return buf;
}
template <typename T>
inline T slot_get(const Slot *buf, int lane, int off = 0) {
    T v;
    std::memcpy(&v, buf[lane].b + off, sizeof(T));
    return v;
}

}  // namespace emu

NSR_DEV f32x4 mfma16(float a, float b, f32x4 c) {
    float ab[2] = {a, b};
    const emu::Slot *buf = emu::exchange(ab, 8);
    const int lane = emu::B->cur->tid & 63;
    const int j = lane & 15, g = lane >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float s = c[r];
        for (int k = 0; k < 4; ++k) {
            const float A = emu::slot_get<float>(buf, i + 16 * k, 0);     // A[i][k] held by lane i + 16k
            const float Bv = emu::slot_get<float>(buf, j + 16 * k, 4);    // B[k][j] held by lane j + 16k
            s = std::fmaf(A, Bv, s);
        }
        d[r] = s;
    }
    return d;
}

NSR_DEV int tid() { return emu::B->cur->tid; }
NSR_DEV int nthreads() { return emu::B->nthreads; }
NSR_DEV int bid_x() { return (int)emu::B->bid.x; }
NSR_DEV int f2i_rn(float x) { return (int)lrintf(x); }
NSR_DEV float relu1(float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }
NSR_DEV int bid_y() { return (int)emu::B->bid.y; }
NSR_DEV int uniform(int v) { return v; }
NSR_DEV int nblk_x() { return (int)emu::B->gdim.x; }

template <typename T>
inline T shfl_any(T v, int src) {
    const emu::Slot *buf = emu::exchange(&v, sizeof(T));
    return emu::slot_get<T>(buf, src & 63);
}
NSR_DEV unsigned long long ballot64(bool p) {
    int v = p ? 1 : 0;
    const emu::Slot *buf = emu::exchange(&v, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (emu::slot_get<int>(buf, l)) m |= 1ull << l;
    return m;
}
NSR_DEV int flag_load(const int *p) { return shfl_any(*p, 0); }
NSR_DEV float shfl(float v, int src) { return shfl_any(v, src); }
NSR_DEV int shfl_i(int v, int src) { return shfl_any(v, src); }
NSR_DEV double shfl_d(double v, int src) { return shfl_any(v, src); }
NSR_DEV float shfl_xor(float v, int m) { return shfl_any(v, (emu::B->cur->tid & 63) ^ m); }
NSR_DEV double shfl_xor_d(double v, int m) { return shfl_any(v, (emu::B->cur->tid & 63) ^ m); }
NSR_DEV float shfl_up(float v, int d) {
    const int lane = emu::B->cur->tid & 63;
    return shfl_any(v, lane >= d ? lane - d : lane);
}
NSR_DEV float shfl_down(float v, int d) {
    const int lane = emu::B->cur->tid & 63;
    return shfl_any(v, lane + d < 64 ? lane + d : lane);
}

struct Dbg { long long *p; NSR_DEV void stamp(int) const {} NSR_DEV void note(int, long long) const {} };
NSR_DEV void wave_fence() { emu::wave_sync(); }
NSR_DEV void sched_fence() {}
NSR_DEV void sched_fence_gemv() {}
NSR_DEV void sched_fence_emb() {}
NSR_DEV void keep_alive(float) {}
NSR_DEV void keep_alive_d(double) {}
NSR_DEV void loop_fence() {}
NSR_DEV int opaque_i(int v) { return v; }
NSR_DEV void block_sync() { emu::block_sync_impl(); }

NSR_DEV void prefetch_line(const float *, float *) {}
NSR_DEV void dma16(const float *gsrc, float *lds_base, int lane) { std::memcpy(lds_base + lane * 4, gsrc, 16); }
template <int N> NSR_DEV void dma_wait() {}
// (a wave's lanes run in lock step on the device: everything the wave did before the flag store has been done by ALL its lanes --
//  the fibers rendezvous here, else the first lane to arrive would publish the flag ahead of its siblings' copies)
NSR_DEV void flag_store(int *p, int v) { emu::wave_sync(); *p = v; }
NSR_DEV int flag_load(const int *p);                  // (below shfl_any: every lane of the wave sees lane 0's reading)
NSR_DEV void spin_pause() { emu::wave_sync(); }       // a polling wave lets the block's other waves run
NSR_DEV void atomic_add_global(float *p, float v) {
    // blocks may run on different OS threads: real atomic read-modify-write
    uint32_t *u = reinterpret_cast<uint32_t *>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    do {
        float f;
        std::memcpy(&f, &old, 4);
        f += v;
        std::memcpy(&nw, &f, 4);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
NSR_DEV void atomic_add_lds(float *p, float v) { *p += v; }
NSR_DEV void atomic_add_lds_i(int *p, int v) { *p += v; }
NSR_DEV int atomic_fetch_add_lds_i(int *p, int v) { const int o = *p; *p += v; return o; }
NSR_DEV int atomic_cas_lds_i(int *p, int expect, int v) { const int o = *p; if (o == expect) *p = v; return o; }
NSR_DEV int lds_load_i(const int *p) { return *p; }
NSR_DEV float lds_load_f(const float *p) { return *p; }
NSR_DEV void atomic_add_global_d(double *p, double v) { *p += v; }
NSR_DEV void atomic_add_global_off(float *base, unsigned byte_off, float v) { atomic_add_global(reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off), v); }
NSR_DEV unsigned long long atomic_fetch_add_global_u64(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
NSR_DEV void atomic_max_pos(float *p, float v) {
    uint32_t *u = reinterpret_cast<uint32_t *>(p);
    uint32_t nw, old = __atomic_load_n(u, __ATOMIC_RELAXED);
    std::memcpy(&nw, &v, 4);
    while (old < nw && !__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

NSR_DEV unsigned atomic_fetch_max_u32(unsigned *p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {}
    return old;
}
NSR_DEV unsigned atomic_exchange_u32(unsigned *p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_ACQ_REL); }
NSR_DEV void keep_alive_u(unsigned) {}

NSR_DEV unsigned uniform_load_u8(const unsigned char *p) { return *p; }
NSR_DEV char *lds_base() { return emu::B->lds; }

struct Stream { const float *base; };
NSR_DEV void stream_st(const Stream &s, int lane_off, int const_off, float v) { const_cast<float *>(s.base)[lane_off + const_off] = v; }
NSR_DEV Stream make_stream(const float *base) { return Stream{base}; }
NSR_DEV float stream_ld(const Stream &s, int lane_off, int const_off) { return s.base[lane_off + const_off]; }

NSR_DEV F4 ld4(const float *p) { return F4{p[0], p[1], p[2], p[3]}; }
NSR_DEV void st4(float *p, F4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }

using ::fmaf;
using ::fmaxf;
using ::fminf;
using ::floorf;
using ::rintf;
using ::expf;
using ::sqrtf;

}  // namespace nsr

#define NSR_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    ::nsr::emu::launch(grid, block, lds_bytes, [=]() { kernel(__VA_ARGS__); })
