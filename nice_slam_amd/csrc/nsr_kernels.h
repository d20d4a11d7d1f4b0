// nsr_kernels.h -- fused render kernels for the NICE-SLAM hot path on gfx950.
//
// One wave = one tile of 16 sample points.  Lane l = (pt = l & 15, g = l >> 4).  Every activation
// vector of a point (features c, embedding e, hidden h, and their gradients) is held "CL":
// for k-tile T the lane owns channels 16T + 4g + r, r = 0..3, as one f32x4.  With that layout
//   * the trilinear gather of a channels-last voxel (32 ch = 128 B) is two 16-B loads per corner per
//     lane and lands directly in MFMA operand position -- no LDS transposition,
//   * y = W x is a chain of v_mfma_f32_16x16x4_f32 with A = packed weights staged in LDS (one 16-byte read per lane
//     feeds four MFMAs) and B = the register that already holds x, output again CL,
//   * dx = W^T dy reads the same packed stream through a transposed index, B = the dy registers,
//   * dW = dy^T x contracts over points: each wave stages its tile's operands in LDS ("channel rows": one 16-byte read =
//     the operand of four k-steps) and ONE owner wave per 16x16 weight-block pair contracts over all tiles of the block;
//     results go to a per-block image of the flat gradient blob in global memory (L2), summed by reduce_partials_kernel.
// References: Renderer.render_batch_ray (src/utils/Renderer.py:63-198), eval_points (:23-61),
// NICE/MLP/MLP_no_xyz forward (src/conv_onet/models/decoder.py:168-203,254-274,312-342),
// raw2outputs_nerf_color (src/common.py:204-245), ATen grid_sampler_3d (GridSampler.h).
#pragma once
#include "nsr_dev.h"
#include "nsr_layout.h"
#include "../../include/nsr.h"

#ifndef NSR_BWD_TILES
#define NSR_BWD_TILES 6     // tiles per ray group of the backward kernel (see nsr_api.cpp)
#endif
#ifndef NSR_BWD_WAVES
#define NSR_BWD_WAVES 6     // waves per backward block; a group's tiles are processed NSR_BWD_WAVES at a time
#endif

namespace nsr {

template <int NT>
struct Act { f32x4 t[NT]; };

NSR_DEV f32x4 f4zero() { f32x4 v = {0.f, 0.f, 0.f, 0.f}; return v; }
NSR_DEV f32x4 to_v(F4 a) { f32x4 v = {a.x, a.y, a.z, a.w}; return v; }
NSR_DEV F4 to_F4(f32x4 v) { return F4{v[0], v[1], v[2], v[3]}; }
template <int NT> NSR_DEV void act_zero(Act<NT> &a) {
#pragma unroll
    for (int T = 0; T < NT; ++T) a.t[T] = f4zero();
}

// NaN-propagating min/max (torch.max / torch.min semantics, Renderer.py:102)
NSR_DEV double tmax(double a, double b) { return (a > b || a != a) ? a : b; }
NSR_DEV double tmin(double a, double b) { return (a < b || a != a) ? a : b; }

// ------------------------------------------------------------------------------------------------
// kernel parameter blocks
// ------------------------------------------------------------------------------------------------
struct GridDev {
    const float *feat;
    float *dfeat;
    int Z, Y, X;
    double lo[3];
    double inv[3];      // 1 / (hi - lo)
};

struct DecDev {
    const float *params;
    const float *packed;
    float *dparams;
};

struct RenderParams {
    int stage, n_samples, n_surface, S;     // S = n_samples + n_surface
    long long n_rays;
    int rays_per_block, tiles_per_block;    // tiles = ceil(rays_per_block * S / 16)
    long long n_groups;
    const float *rays_o, *rays_d, *gt_depth, *gt_max;
    double blo[3], bhi[3];
    float t_uniform[NSR_MAX_SAMPLES];
    double t_surface[NSR_MAX_SAMPLES];
    GridDev grid[4];
    DecDev dec[4];
    double *depth, *var;
    float *rgb, *raw;
    double *zvals;            // [N][S] saved sample depths (optional)
    // backward only
    const double *d_depth, *d_var, *g_depth;
    const float *d_rgb;
    float *d_rays_o, *d_rays_d;
    float *partials;          // [3 passes][gridDim.x][max param count]
    int partial_stride;       // floats between two blocks' partial images
    // eval_points only
    const double *points;
    long long n_points;
    float *out_points;
};

// ------------------------------------------------------------------------------------------------
// parameter packing: flat blob -> MFMA operand stream, four k-steps per lane contiguous (one 16-byte read each)
//   packed[m.pk + ((T*2 + Tp)*64 + lane)*4 + r] = W[16*Tp + (lane&15)][kbeg + 16T + 4(lane>>4) + r]
// ------------------------------------------------------------------------------------------------
template <int KIND>
NSR_KERNEL void pack_kernel(const float *__restrict__ flat, float *__restrict__ packed) {
    const int idx = bid_x() * nthreads() + tid();
    if (idx >= packed_total(KIND)) return;
    float v = 0.f;
#pragma unroll
    for (int id = 0; id < nmat_of(KIND); ++id) {
        const Mat m = mat_of(KIND, id);
        const int rel = idx - m.pk;
        if (rel >= 0 && rel < m.nt * 512) {
            const int r = rel & 3, lane = (rel >> 2) & 63, Tp = (rel >> 8) & 1, T = rel >> 9;
            const int o = 16 * Tp + (lane & 15);
            const int k = 16 * T + 4 * (lane >> 4) + r;
            if (k < m.kcols) v = flat[m.off + o * m.stride + m.kbeg + k];
        }
    }
    packed[idx] = v;
}

// stage the small per-decoder tables into LDS (all threads of the block)
template <int KIND>
NSR_DEV void load_aux(float *aux, const float *__restrict__ flat) {
    for (int idx = tid(); idx < AUX_FLOATS; idx += nthreads()) {
        float v = 0.f;
        if (idx < AUX_V) {
            const int i = idx >> 5, o = idx & 31;
            v = flat[bias_off(KIND, i) + o];
        } else if (idx < AUX_WO) {
            if (is_xyz(KIND)) { const int i = (idx - AUX_V) >> 5, o = idx & 31; v = flat[fcb_off(KIND, i) + o]; }
        } else if (idx < AUX_BO) {
            const int n = (idx - AUX_WO) >> 5, k = idx & 31;
            if (n < nout_of(KIND)) v = flat[wo_off(KIND) + n * 32 + k];
        } else if (idx < AUX_BM) {
            const int n = idx - AUX_BO;
            if (n < nout_of(KIND)) v = flat[bo_off(KIND) + n];
        } else if (is_xyz(KIND)) {
            const int rel = idx - AUX_BM, ch = (rel >> 4) * 4 + (rel & 3), d = (rel >> 2) & 3;   // [4-channel group][xyz.][4]
            if (ch < kE && d < 3) v = flat[B_off(KIND) + d * kE + ch];
        }
        aux[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// sample placement along the rays of one block  (Renderer.py:88-170)
// ------------------------------------------------------------------------------------------------
NSR_DEV double ray_far_bb(const RenderParams &P, long long ray) {
    double far = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double o = (double)P.rays_o[ray * 3 + a], d = (double)P.rays_d[ray * 3 + a];
        const double t0 = (P.blo[a] - o) / d, t1 = (P.bhi[a] - o) / d;
        const double m = tmax(t0, t1);
        far = (a == 0) ? m : tmin(far, m);
    }
    return far + 0.01;
}

// Fills zbuf[rays_per_block][S] (sorted ascending per ray).  ztmp is scratch of the same size.
NSR_DEV void compute_z(const RenderParams &P, long long ray0, double *ztmp, double *zbuf) {
    const int S = P.S, npts = P.rays_per_block * S;
    const bool guided = (P.gt_depth != nullptr) && (P.stage != NSR_STAGE_COARSE);
    for (int t = tid(); t < npts; t += nthreads()) {
        const int r = t / S, k = t - r * S;
        const long long ray = ray0 + r;
        double z = 0.0;
        if (ray < P.n_rays) {
            const double far_bb = ray_far_bb(P, ray);
            if (!guided) {
                const float tk = P.t_uniform[k];
                const float near_part = 0.01f * (1.f - tk);
                z = (double)near_part + far_bb * (double)tk;
            } else {
                const float g = P.gt_depth[ray];
                const float gmax = P.gt_max[0];
                if (k < P.n_samples) {
                    const float tk = P.t_uniform[k];
                    const double cap = (double)(gmax * 1.2f);
                    const double far = tmin(tmax(far_bb, 0.0), cap);
                    const float near = g * 0.01f;
                    z = (double)(near * (1.f - tk)) + far * (double)tk;
                } else {
                    const double s = P.t_surface[k - P.n_samples];
                    if (g > 0.f) {
                        const double e0 = (double)(0.95f * g), e1 = (double)(1.05f * g);
                        z = e0 * (1.0 - s) + e1 * s;
                    } else {
                        z = 0.001 * (1.0 - s) + (double)gmax * s;
                    }
                }
            }
        }
        ztmp[t] = z;
    }
    block_sync();
    if (P.n_surface > 0 && guided) {
        // rank sort of the S candidates of each ray (torch.sort, Renderer.py:168-170)
        for (int t = tid(); t < npts; t += nthreads()) {
            const int r = t / S, k = t - r * S;
            const double v = ztmp[t];
            int rank = 0;
#pragma unroll 16
            for (int j = 0; j < S; ++j) {      // unrolled: 16 LDS reads in flight instead of one latency per compare
                const double u = ztmp[r * S + j];
                rank += (u < v || (u == v && j < k)) ? 1 : 0;
            }
            zbuf[r * S + rank] = v;
        }
    } else {
        for (int t = tid(); t < npts; t += nthreads()) zbuf[t] = ztmp[t];
    }
    block_sync();
}

// ------------------------------------------------------------------------------------------------
// trilinear lookup  (decoder.py:168-175, common.py:269-284, ATen GridSampler.h:27-85)
// ------------------------------------------------------------------------------------------------
struct Lvl {
    int vox;            // linear voxel index of the (z0,y0,x0) corner
    int sx, sy, sz;     // voxel-index step to the +1 neighbour (0 when the axis has one cell)
    float fx, fy, fz;   // weight of the +1 neighbour
    float gx, gy, gz;   // weight of the base corner ((i0+1) - u, the ATen form)
    float mx, my, mz;   // d u / d g_normalised, 0 when the coordinate was clipped
};

NSR_DEV void axis_setup(double p, double lo, double inv, int n, int &i0, float &w0, float &w1, float &mult) {
    const double gn = ((p - lo) * inv) * 2.0 - 1.0;      // fp64 normalisation, then one rounding
    const float gf = (float)gn;
    const float nm1 = (float)(n - 1);
    float u = ((gf + 1.f) / 2.f) * nm1;
    mult = (u <= 0.f || u >= nm1) ? 0.f : nm1 / 2.f;
    u = fminf(nm1, fmaxf(u, 0.f));
    int i = (int)floorf(u);
    const int lim = n > 1 ? n - 2 : 0;
    if (i > lim) i = lim;
    i0 = i;
    w0 = ((float)i + 1.f) - u;
    w1 = u - (float)i;
    if (n == 1) { w0 = 1.f; w1 = 0.f; }
}

NSR_DEV Lvl make_level(const GridDev &G, double px, double py, double pz) {
    Lvl L;
    int x0, y0, z0;
    axis_setup(px, G.lo[0], G.inv[0], G.X, x0, L.gx, L.fx, L.mx);
    axis_setup(py, G.lo[1], G.inv[1], G.Y, y0, L.gy, L.fy, L.my);
    axis_setup(pz, G.lo[2], G.inv[2], G.Z, z0, L.gz, L.fz, L.mz);
    L.vox = (z0 * G.Y + y0) * G.X + x0;
    L.sx = G.X > 1 ? 1 : 0;
    L.sy = G.Y > 1 ? G.X : 0;
    L.sz = G.Z > 1 ? G.X * G.Y : 0;
    return L;
}

NSR_DEV float corner_w(const Lvl &L, int c) {       // c = dz*4 + dy*2 + dx ; ((wx*wy)*wz)
    const float wx = (c & 1) ? L.fx : L.gx, wy = (c & 2) ? L.fy : L.gy, wz = (c & 4) ? L.fz : L.gz;
    return (wx * wy) * wz;
}
NSR_DEV int corner_vox(const Lvl &L, int c) {
    return L.vox + ((c & 1) ? L.sx : 0) + ((c & 2) ? L.sy : 0) + ((c & 4) ? L.sz : 0);
}

// gather: lane (pt,g) accumulates channels 4g..4g+3 and 16+4g..16+4g+3 of its point
NSR_DEV Act<2> gather_feat(const GridDev &G, const Lvl &L, int g) {
    Act<2> c;
    c.t[0] = f4zero();
    c.t[1] = f4zero();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = corner_w(L, k);
        const float *src = G.feat + (long long)corner_vox(L, k) * kC + 4 * g;
        const F4 a = ld4(src), b = ld4(src + 16);
        c.t[0][0] = fmaf(a.x, w, c.t[0][0]); c.t[0][1] = fmaf(a.y, w, c.t[0][1]);
        c.t[0][2] = fmaf(a.z, w, c.t[0][2]); c.t[0][3] = fmaf(a.w, w, c.t[0][3]);
        c.t[1][0] = fmaf(b.x, w, c.t[1][0]); c.t[1][1] = fmaf(b.y, w, c.t[1][1]);
        c.t[1][2] = fmaf(b.z, w, c.t[1][2]); c.t[1][3] = fmaf(b.w, w, c.t[1][3]);
    }
    return c;
}

// per-wave transposition buffers: Tx[pt][kTxS]
NSR_DEV void tx_store(float *Tx, const Act<2> &v, int pt, int g) {
    st4(Tx + pt * kTxS + 4 * g, to_F4(v.t[0]));
    st4(Tx + pt * kTxS + 16 + 4 * g, to_F4(v.t[1]));
}
// backward of the gather, part 1: coordinate gradient (d value / d u per axis, ATen
// grid_sampler_3d_backward), returned already reduced over g.
NSR_DEV void coord_grad(const GridDev &G, const Lvl &L, int g, const Act<2> &dc, float &dux, float &duy, float &duz) {
    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float *src = G.feat + (long long)corner_vox(L, k) * kC + 4 * g;
        const F4 a = ld4(src), b = ld4(src + 16);
        float dot = a.x * dc.t[0][0];
        dot = fmaf(a.y, dc.t[0][1], dot); dot = fmaf(a.z, dc.t[0][2], dot); dot = fmaf(a.w, dc.t[0][3], dot);
        dot = fmaf(b.x, dc.t[1][0], dot); dot = fmaf(b.y, dc.t[1][1], dot);
        dot = fmaf(b.z, dc.t[1][2], dot); dot = fmaf(b.w, dc.t[1][3], dot);
        const float wx = (k & 1) ? L.fx : L.gx, wy = (k & 2) ? L.fy : L.gy, wz = (k & 4) ? L.fz : L.gz;
        ax = fmaf((k & 1) ? dot : -dot, wy * wz, ax);
        ay = fmaf((k & 2) ? dot : -dot, wx * wz, ay);
        az = fmaf((k & 4) ? dot : -dot, wx * wy, az);
    }
    ax += shfl_xor(ax, 16); ax += shfl_xor(ax, 32);
    ay += shfl_xor(ay, 16); ay += shfl_xor(ay, 32);
    az += shfl_xor(az, 16); az += shfl_xor(az, 32);
    dux = ax * L.mx; duy = ay * L.my; duz = az * L.mz;
}

// backward of the gather, part 2: scatter-add dc into the grid gradient.
// Measured on MI355X (tools/atomic_probe.hip): a global f32 atomic costs one request per touched 64-byte
// line (~21 G lines/s chip-wide) no matter how many of its 16 dwords an instruction updates, and lanes of
// one instruction that hit the SAME dword serialise.  So the tile is re-laid out through LDS to
// "lane = channel": each half-wave owns one corner index k and walks the tile's 16 points (consecutive
// samples of a ray, i.e. spatially sorted), summing weighted dc while the corner voxel stays the same and
// issuing ONE 32-lane atomic (two full 64-byte lines) per run of equal voxels.
//   Tx  : [16][kTxS] floats  dc of the tile, point-major
//   tab : [16][8] ints (corner voxel or -1) followed by [16][8] floats (corner weight)
NSR_DEV void scatter_merged(const GridDev &G, const Lvl &L, int lane, const Act<2> &dc, bool active, float *Tx, float *tab) {
    const int pt = lane & 15, g = lane >> 4;
    int *vt = reinterpret_cast<int *>(tab);
    float *wt = tab + 128;
    tx_store(Tx, dc, pt, g);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int k = 2 * g + c;
        vt[pt * 8 + k] = active ? corner_vox(L, k) : -1;
        wt[pt * 8 + k] = corner_w(L, k);
    }
    wave_fence();
    const int h = lane >> 5, ch = lane & 31;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int k = 2 * q + h;
        int v[16];
        float w[16], x[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {        // all LDS reads of the round in flight at once
            v[p] = vt[p * 8 + k];
            w[p] = wt[p * 8 + k];
            x[p] = Tx[p * kTxS + ch];
        }
        float acc = 0.f;
        int cur = -1;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (v[p] != cur) {
                if (cur >= 0) atomic_add_global(G.dfeat + (long long)cur * kC + ch, acc);
                acc = 0.f;
                cur = v[p];
            }
            acc = fmaf(x[p], w[p], acc);
        }
        if (cur >= 0) atomic_add_global(G.dfeat + (long long)cur * kC + ch, acc);
    }
    wave_fence();
}

// ------------------------------------------------------------------------------------------------
// MFMA building blocks
// ------------------------------------------------------------------------------------------------
// acc[Tp] += W(slice) * x          (A from the packed stream staged in LDS, B = x registers)
// One ds_read_b128 per (k-tile, output tile) feeds four MFMAs; a lane's 16 bytes are consecutive across the wave, so
// the read is conflict-free at full LDS bandwidth.
template <int NT>
NSR_DEV void gemv_fwd(f32x4 (&acc)[2], const Act<NT> &x, const float *pk, int lane) {
    // explicit operand ring, depth kD k-tiles (2 reads = 8 registers each): bounds the reads in flight (the scheduler
    // would otherwise hoist all 2*NT reads of the slice and spill) while covering the LDS latency behind 8 MFMAs
    constexpr int kD = NT < 2 ? NT : 2;
    f32x4 ra[kD], rb[kD];
#pragma unroll
    for (int T = 0; T < kD; ++T) { ra[T] = to_v(ld4(pk + (T * 128 + lane) * 4)); rb[T] = to_v(ld4(pk + (T * 128 + 64 + lane) * 4)); }
#pragma unroll
    for (int T = 0; T < NT; ++T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[0] = mfma16(ra[T % kD][r], x.t[T][r], acc[0]);
            acc[1] = mfma16(rb[T % kD][r], x.t[T][r], acc[1]);
        }
        if (T + kD < NT) {
            ra[T % kD] = to_v(ld4(pk + ((T + kD) * 128 + lane) * 4));
            rb[T % kD] = to_v(ld4(pk + ((T + kD) * 128 + 64 + lane) * 4));
        }
    }
    sched_fence();
}

// dx[Tk] += W(slice)^T * dy       (B = dy registers; A = W[16To+4g+r][16Tk+i])
// read from the same packed stream (`w` = LDS base of the slice): element W[o][k] of a slice sits at
//   ((Tk*2 + (o>>4))*64 + (o&15) + 16*((k>>2)&3))*4 + (k&3)    -- scalar reads, 4-way bank conflict, fine next to
// a 64-cycle MFMA pair; columns beyond kcols are zero in the packed stream.
template <int NTK>
NSR_DEV void gemv_bwd(f32x4 (&dx)[NTK], const Act<2> &dy, const float *w, int i, int g) {
    const int lo = (4 * g + 16 * (i >> 2)) * 4 + (i & 3);
    constexpr int kD = 4, NS = 8;                     // step q = (To, r); NTK reads + NTK MFMAs per step
    float ra[kD][NTK];
#pragma unroll
    for (int q = 0; q < kD; ++q)
#pragma unroll
        for (int Tk = 0; Tk < NTK; ++Tk) ra[q][Tk] = w[Tk * 512 + (q >> 2) * 256 + (q & 3) * 4 + lo];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
        for (int Tk = 0; Tk < NTK; ++Tk) dx[Tk] = mfma16(ra[q % kD][Tk], dy.t[q >> 2][q & 3], dx[Tk]);
        if (q + kD < NS) {
#pragma unroll
            for (int Tk = 0; Tk < NTK; ++Tk) ra[q % kD][Tk] = w[Tk * 512 + ((q + kD) >> 2) * 256 + ((q + kD) & 3) * 4 + lo];
        }
    }
    sched_fence();
}

// cooperative copy of a decoder's packed operand stream into LDS (caller provides the barriers)
template <int KIND>
NSR_DEV void load_packed(float *wl, const float *__restrict__ packed) {
    for (int t = tid(); t < packed_total(KIND) / 4; t += nthreads()) st4(wl + 4 * t, ld4(packed + 4 * t));
}

NSR_DEV float red_g(float v) { v += shfl_xor(v, 16); v += shfl_xor(v, 32); return v; }

// sin / cos for Fourier arguments |x| up to a few thousand: two-constant Cody-Waite reduction by pi (exact products
// through fma) to r in [-pi/2, pi/2], one odd minimax polynomial of degree 9 there (3.3e-9 in exact arithmetic, ~1.5e-7
// evaluated in fp32: the same class as libm sinf), sign from the parity of the quotient; branch-free, 13 VALU operations.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
NSR_DEV f32x4 splat(float v) { f32x4 r = {v, v, v, v}; return r; }
NSR_DEV f32x4 vfma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
// Four arguments at a time: every step is a <4 x float> operation, i.e. two packed-fp32 instructions (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32), half the VALU issue slots of the scalar form.
NSR_DEV f32x4 sin_poly4(f32x4 r, u32x4 sign) {
    const f32x4 r2 = r * r;
    f32x4 p = vfma(r2, splat(2.59048850e-06f), splat(-1.98008978e-04f));
    p = vfma(p, r2, splat(8.33289982e-03f));
    p = vfma(p, r2, splat(-1.66666476e-01f));
    const f32x4 s = vfma(p * r2, r, r);
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, s) ^ sign);
}
// quotient by the "1.5 * 2^23" trick: t = x/pi + 12582912 has round(x/pi) in its low mantissa bits (|x| < 1e7), so
// its parity is bit 0 of the float and no float->int conversion / compare / select is needed for the sign
NSR_DEV f32x4 sin_acc4(f32x4 x) {
    const f32x4 t = vfma(x, splat(0.318309886183790672f), splat(12582912.f));
    const f32x4 k = t - splat(12582912.f);
    f32x4 r = vfma(k, splat(-3.14159274101257324f), x);
    r = vfma(k, splat(8.74227765734758577e-08f), r);
    return sin_poly4(r, __builtin_bit_cast(u32x4, t) << 31);
}
// cos(x) = -(-1)^k sin(r) with x = (k + 1/2) pi + r
NSR_DEV f32x4 cos_acc4(f32x4 x) {
    const f32x4 t = vfma(x, splat(0.318309886183790672f), splat(-0.5f)) + splat(12582912.f);   // (12582912 - 0.5 is not representable)
    const f32x4 k = t - splat(12582912.f);
    f32x4 r = vfma(k, splat(-3.14159274101257324f), x);
    r = vfma(k, splat(8.74227765734758577e-08f), r);
    r = (r - splat(1.57079637050628662f)) + splat(4.37113882867379289e-08f);
    return sin_poly4(r, (__builtin_bit_cast(u32x4, t) << 31) ^ 0x80000000u);
}
// Fourier matrix rows of four consecutive channels (group = channel / 4): Bx[4], By[4], Bz[4]
struct B4 { f32x4 x, y, z; };
NSR_DEV B4 load_b4(const float *aux, int group) {
    const float *b = aux + AUX_BM + group * 16;
    return B4{to_v(ld4(b)), to_v(ld4(b + 4)), to_v(ld4(b + 8))};
}
NSR_DEV F4 load_b1(const float *aux, int ch) {        // (Bx, By, Bz) of one channel
    const float *b = aux + AUX_BM + (ch >> 2) * 16 + (ch & 3);
    return F4{b[0], b[4], b[8], 0.f};
}

// ------------------------------------------------------------------------------------------------
// Parameter gradients, owner-computes.
// Measured (tools/lds_atomic_probe.hip): ds_add_f32 costs ~195 cycles per wave instruction per CU, so the
// per-tile dW blocks are NOT summed with LDS atomics.  Instead every layer is a lock-step phase of the block:
//   1. each wave stages the operands of its tile (dH, dY, layer input, features, point positions) in LDS,
//   2. block barrier,
//   3. each 16x16 block pair of the layer's dW has ONE owner wave, which contracts over the 16 points of
//      EVERY tile of the block in one MFMA chain (K = 16 x tiles) and adds the result into the block's image of
//      the flat gradient blob (global memory, L2-resident; exclusive owner => plain loads / stores, no atomics),
//   4. block barrier.
// Per-wave staging region (floats): P[3][16] | DO[16][4] | A0 | A1 | X0 | C[cdim/32]; a tile is 16 rows x 32
// channels in the "channel rows" layout below (conflict-free scalar stores, one conflict-free 16-byte operand read).
// ------------------------------------------------------------------------------------------------
constexpr int kStP = 0, kStDO = 64, kStA0 = 128, kStA1 = 128 + 512, kStX0 = 128 + 1024, kStC = 128 + 1536;
constexpr int stg_floats(int kind) { return 128 + 512 * (3 + cdim_of(kind) / 32); }

// Tile layout "channel rows": 32 rows (channels) x 16 floats (points), 512 floats.  Channel ch = 16T + 4g + r lives in
// row rho = 16T + 4r + g (g and r swapped, so that the four lane groups g of one store instruction hit four different
// bank quarters); inside a row the four 4-point groups are XOR-permuted by (ch & 3) (so that the 16 lanes of one b128
// operand read cover all 64 banks).  Stores: 8 conflict-free ds_write_b32 per tile; operand reads: ONE conflict-free
// ds_read_b128 per (tile, 16-channel k-tile) = the MFMA operand of 4 k-steps, k-step q <-> point 4g + q.
NSR_DEV int st_row(int ch) { return (ch & ~15) + ((ch & 3) << 2) + ((ch >> 2) & 3); }
NSR_DEV void st_store(float *T, const Act<2> &v, int pt, int g) {
    float *B = T + g * 16 + (pt & 3);
    const int pg = pt >> 2;
#pragma unroll
    for (int Tt = 0; Tt < 2; ++Tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) B[(16 * Tt + 4 * r) * 16 + ((pg ^ r) << 2)] = v.t[Tt][r];
}
// element q = T[point 4g+q][channel 16*Tt + i]: MFMA operand "lane = channel i of k-tile Tt, k-step q = point 4g+q"
NSR_DEV f32x4 st_load_cm(const float *T, int Tt, int i, int g) {
    return to_v(ld4(T + st_row(16 * Tt + i) * 16 + ((g ^ (i & 3)) << 2)));
}
// points 4*grp .. 4*grp+3 of one channel
NSR_DEV F4 st_row4(const float *T, int ch, int grp) { return ld4(T + st_row(ch) * 16 + ((grp ^ (ch & 3)) << 2)); }

struct Own {
    Stream img;          // this block's image of the flat parameter-gradient blob (global partial buffer, L2 resident);
                         // addressed through a buffer descriptor: 32-bit offsets, no 64-bit address registers
    const float *stg;    // staging regions of all waves
    int stride, nw, wave, lane;
    bool first;          // first ray group of this block: store instead of accumulate (no zero-fill needed)
    float *small;        // per-wave LDS accumulators of the output layer: [wave][wo[4][32] | bo[4]]
};
// Owner of the t-th task of a phase (tasks listed heaviest first).  With six waves on four SIMDs (waves w and w+4 share
// one) the first four tasks go to one wave per SIMD, starting with the two waves that have their SIMD to themselves.
NSR_DEV bool mine(const Own &O, int t) {
    const int r = t % O.nw;
    const int w = O.nw == 6 ? ((0x541032 >> (4 * r)) & 0xF) : r;
    return w == O.wave;
}
NSR_DEV void img_add(const Own &O, int lane_off, int const_off, float v) {
    if (!O.first) v += stream_ld(O.img, lane_off, const_off);
    stream_st(O.img, lane_off, const_off, v);
}

// img[W slice, k-tile Tk] += sum over the block's tiles of A^T X,  X staged at x_off (sub-tile x_sub)
NSR_DEV void own_pair(const Own &O, const Mat m, int Tk, int a_off, int x_off, int x_sub) {
    const int i = O.lane & 15, g = O.lane >> 4;
    f32x4 d0 = f4zero(), d1 = f4zero();
    // the image values this task accumulates into (later ray groups of the block): requested now, consumed after the
    // MFMA chain, so the L2 round trip hides behind it
    const int k = 16 * Tk + i;
    const bool live = k < m.kcols;
    const int lo = 4 * g * m.stride + i;
    const int co = m.off + m.kbeg + 16 * Tk;
    if (!O.first && live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d0[r] = stream_ld(O.img, lo, co + r * m.stride);
            d1[r] = stream_ld(O.img, lo, co + (16 + r) * m.stride);
        }
    }
    // software pipeline over the block's tiles, two tiles per trip with ping-pong operand sets (no register rotation):
    // the LDS reads (and, for the embedding, the sines) of the next tile are issued before the 8 MFMAs of this one
    struct Ops { f32x4 a0, a1, x; };
    auto fetch = [&](int t) {
        const float *S = O.stg + (t < O.nw ? t : O.nw - 1) * O.stride;
        Ops o;
        o.a0 = st_load_cm(S + a_off, 0, i, g);
        o.a1 = st_load_cm(S + a_off, 1, i, g);
        o.x = st_load_cm(S + x_off, x_sub, i, g);
        return o;
    };
    auto fmas = [&](const Ops &o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            d0 = mfma16(o.a0[q], o.x[q], d0);
            d1 = mfma16(o.a1[q], o.x[q], d1);
        }
    };
    Ops A = fetch(0);
    for (int t = 0; t < O.nw; t += 2) {
        const Ops B = fetch(t + 1);
        fmas(A);
        A = fetch(t + 2);
        if (t + 1 < O.nw) fmas(B);
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            stream_st(O.img, lo, co + r * m.stride, d0[r]);
            stream_st(O.img, lo, co + (16 + r) * m.stride, d1[r]);
        }
    }
}
// The two weight blocks that read the Fourier embedding, W0 (layer 0) and W3e (embedding columns of layer 3), in one
// task: the 16 embedding channels of k-tile Tk are recomputed ONCE per tile (decoder.py:26-30) and contracted with
// dY0 (staged at a0_off) and dY3 (staged at a3_off):   img[W0 / W3e slice, k-tile Tk] += sum over tiles of dY^T E
NSR_DEV void own_embed_pair(const Own &O, const Mat m0, const Mat m3, int Tk, int a0_off, int a3_off, const float *aux) {
    const int i = O.lane & 15, g = O.lane >> 4;
    f32x4 d00 = f4zero(), d01 = f4zero(), d30 = f4zero(), d31 = f4zero();
    const int k = 16 * Tk + i;
    const bool live = k < m0.kcols;
    const int lo0 = 4 * g * m0.stride + i, lo3 = 4 * g * m3.stride + i;
    const int co0 = m0.off + m0.kbeg + 16 * Tk, co3 = m3.off + m3.kbeg + 16 * Tk;
    if (!O.first && live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d00[r] = stream_ld(O.img, lo0, co0 + r * m0.stride);
            d01[r] = stream_ld(O.img, lo0, co0 + (16 + r) * m0.stride);
            d30[r] = stream_ld(O.img, lo3, co3 + r * m3.stride);
            d31[r] = stream_ld(O.img, lo3, co3 + (16 + r) * m3.stride);
        }
    }
    const F4 b = load_b1(aux, 16 * Tk + i);
    struct Ops { f32x4 p0, p1, q0, q1, x; };
    auto fetch = [&](int t) {
        const float *S = O.stg + (t < O.nw ? t : O.nw - 1) * O.stride;
        Ops o;
        o.p0 = st_load_cm(S + a0_off, 0, i, g); o.p1 = st_load_cm(S + a0_off, 1, i, g);
        o.q0 = st_load_cm(S + a3_off, 0, i, g); o.q1 = st_load_cm(S + a3_off, 1, i, g);
        const f32x4 px = to_v(ld4(S + kStP + 4 * g)), py = to_v(ld4(S + kStP + 16 + 4 * g)), pz = to_v(ld4(S + kStP + 32 + 4 * g));
        o.x = sin_acc4(vfma(pz, splat(b.z), vfma(py, splat(b.y), px * splat(b.x))));       // points 4g .. 4g+3
        return o;
    };
    auto fmas = [&](const Ops &o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            d00 = mfma16(o.p0[q], o.x[q], d00);
            d01 = mfma16(o.p1[q], o.x[q], d01);
            d30 = mfma16(o.q0[q], o.x[q], d30);
            d31 = mfma16(o.q1[q], o.x[q], d31);
        }
    };
    Ops A = fetch(0);
    for (int t = 0; t < O.nw; t += 2) {
        const Ops B = fetch(t + 1);
        fmas(A);
        A = fetch(t + 2);
        if (t + 1 < O.nw) fmas(B);
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            stream_st(O.img, lo0, co0 + r * m0.stride, d00[r]);
            stream_st(O.img, lo0, co0 + (16 + r) * m0.stride, d01[r]);
            stream_st(O.img, lo3, co3 + r * m3.stride, d30[r]);
            stream_st(O.img, lo3, co3 + (16 + r) * m3.stride, d31[r]);
        }
    }
}
// img[off + ch] += sum over tiles and points of A[p][ch]    (bias gradients)
NSR_DEV void own_colsum(const Own &O, int off, int a_off) {
    const int ch = O.lane & 31, half = O.lane >> 5;
    float s = 0.f;
    for (int t = 0; t < O.nw; ++t) {
        const float *T = O.stg + t * O.stride + a_off;
        const F4 u = st_row4(T, ch, 2 * half), v = st_row4(T, ch, 2 * half + 1);      // points 8*half .. 8*half+7
        s += u.x; s += u.y; s += u.z; s += u.w; s += v.x; s += v.y; s += v.z; s += v.w;
    }
    s += shfl_xor(s, 32);
    if (half == 0) img_add(O, ch, off, s);
}
// output layer, wave-local (no block barrier, no atomics): slot[n*32 + ch] += sum_p d_out[p][n] * h4[p][ch],
// slot[128 + n] += sum_p d_out[p][n] over the 16 points of THIS wave's tile (DO and X0 = h4 staged in its own region);
// `slot` is this wave's private 132-float accumulator; the block sums the slots into the gradient image once per
// ray group (bwd_pass).
template <int NOUT>
NSR_DEV void out_layer_local(const Own &O, const float *S) {
    const int ch = O.lane & 31, half = O.lane >> 5;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    const F4 h0 = st_row4(S + kStX0, ch, 2 * half), h1 = st_row4(S + kStX0, ch, 2 * half + 1);
    const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int p = half * 8 + q;
        const F4 d = ld4(S + kStDO + p * 4);
        const float h = hv[q];
        s[0] = fmaf(d.x, h, s[0]); sb[0] += d.x;
        if (NOUT > 1) { s[1] = fmaf(d.y, h, s[1]); s[2] = fmaf(d.z, h, s[2]); s[3] = fmaf(d.w, h, s[3]); sb[1] += d.y; sb[2] += d.z; sb[3] += d.w; }
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        const float v = s[n] + shfl_xor(s[n], 32);
        const float bsum = sb[n] + shfl_xor(sb[n], 32);
        float *slot = O.small + O.wave * 132;
        if (half == 0) slot[n * 32 + ch] += v;
        if (O.lane == 0) slot[128 + n] += bsum;
    }
}
// Fourier matrix: img[B + d*93 + ch] += sum darg[p][ch] * p[p][d]  for the 16 channels of k-tile Tk
// (darg of the whole tile staged as [16][96] starting at kStA0)
NSR_DEV void own_dB(const Own &O, int Tk, int boff) {
    const int j = O.lane & 15, pg = O.lane >> 4, ch = 16 * Tk + j;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int t = 0; t < O.nw; ++t) {
        const float *S = O.stg + t * O.stride;
        const f32x4 px = to_v(ld4(S + kStP + 4 * pg)), py = to_v(ld4(S + kStP + 16 + 4 * pg)), pz = to_v(ld4(S + kStP + 32 + 4 * pg));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = S[kStA0 + (4 * pg + q) * 96 + ch];
            sx = fmaf(v, px[q], sx); sy = fmaf(v, py[q], sy); sz = fmaf(v, pz[q], sz);
        }
    }
    sx = red_g(sx); sy = red_g(sy); sz = red_g(sz);
    if (pg == 0 && ch < kE) {
        img_add(O, j, boff + 16 * Tk, sx);
        img_add(O, j, boff + kE + 16 * Tk, sy);
        img_add(O, j, boff + 2 * kE + 16 * Tk, sz);
    }
}


NSR_DEV unsigned relu_mask(f32x4 (&acc)[2]) {
    unsigned m = 0;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (acc[T][r] > 0.f) m |= 1u << (T * 4 + r); else acc[T][r] = 0.f;
        }
    return m;
}
NSR_DEV void relu_plain(f32x4 (&acc)[2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[T][r] = relu1(acc[T][r]);
}
NSR_DEV Act<2> apply_mask(const Act<2> &d, unsigned m) {
    Act<2> o;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.t[T][r] = (m >> (T * 4 + r)) & 1u ? d.t[T][r] : 0.f;
    return o;
}

NSR_DEV void embed(Act<kET> &e, const float *aux, float px, float py, float pz, int g) {
#pragma unroll
    for (int T = 0; T < kET; ++T) {
        const B4 b = load_b4(aux, 4 * T + g);
        const f32x4 arg = vfma(splat(pz), b.z, vfma(splat(py), b.y, splat(px) * b.x));     // decoder.py:29
        e.t[T] = sin_acc4(arg);                                                            // decoder.py:30
        sched_fence();                  // bound the ILP the scheduler extracts from 24 independent sines
    }
}

// ------------------------------------------------------------------------------------------------
// decoder forward for one tile.  KEEP keeps what the backward needs (h_i and relu masks).
// ------------------------------------------------------------------------------------------------
template <int KIND>
struct Kept {
    Act<2> h[5];
    unsigned mask[5];
};

// MLP (decoder.py:177-203): h_i = relu(W_i x_i + b_i) + (U_i c + v_i), x_3 = [e | h_2]
template <int KIND, bool KEEP>
NSR_DEV void mlp_xyz_fwd(const float *pk, const float *aux, float px, float py, float pz,
                         const Act<cdim_of(KIND) / 16> &c, int lane, float (&out)[nout_of(KIND)], Kept<KIND> *kept) {
    constexpr int CD = cdim_of(KIND), NOUT = nout_of(KIND), NTC = CD / 16;
    const int g = lane >> 4;
    Act<kET> e;
    embed(e, aux, px, py, pz, g);
    Act<2> h;
    act_zero(h);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        f32x4 acc[2];
        acc[0] = to_v(ld4(aux + AUX_B + i * 32 + 4 * g));
        acc[1] = to_v(ld4(aux + AUX_B + i * 32 + 16 + 4 * g));
        if (i == 0) {
            gemv_fwd<kET>(acc, e, pk + xyz_mat(CD, XW0).pk, lane);
        } else if (i == 3) {
            gemv_fwd<kET>(acc, e, pk + xyz_mat(CD, XW3E).pk, lane);
            gemv_fwd<2>(acc, h, pk + xyz_mat(CD, XW3H).pk, lane);
        } else {
            gemv_fwd<2>(acc, h, pk + xyz_mat(CD, i == 1 ? XW1 : (i == 2 ? XW2 : XW4)).pk, lane);
        }
        unsigned m = 0;
        if (KEEP) m = relu_mask(acc); else relu_plain(acc);
        acc[0] += to_v(ld4(aux + AUX_V + i * 32 + 4 * g));
        acc[1] += to_v(ld4(aux + AUX_V + i * 32 + 16 + 4 * g));
        gemv_fwd<NTC>(acc, c, pk + xyz_mat(CD, i == 0 ? XU0 : (i == 1 ? XU1 : (i == 2 ? XU2 : (i == 3 ? XU3 : XU4)))).pk, lane);
        h.t[0] = acc[0];
        h.t[1] = acc[1];
        if (KEEP) { kept->h[i] = h; kept->mask[i] = m; }
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        const F4 w0 = ld4(aux + AUX_WO + n * 32 + 4 * g), w1 = ld4(aux + AUX_WO + n * 32 + 16 + 4 * g);
        float s = w0.x * h.t[0][0];
        s = fmaf(w0.y, h.t[0][1], s); s = fmaf(w0.z, h.t[0][2], s); s = fmaf(w0.w, h.t[0][3], s);
        s = fmaf(w1.x, h.t[1][0], s); s = fmaf(w1.y, h.t[1][1], s); s = fmaf(w1.z, h.t[1][2], s); s = fmaf(w1.w, h.t[1][3], s);
        out[n] = red_g(s) + aux[AUX_BO + n];
    }
}

// MLP_no_xyz (decoder.py:262-274): h = c; h = relu(W_i h + b_i); after i == 2: h = [c | h]
template <bool KEEP>
NSR_DEV void mlp_nox_fwd(const float *pk, const float *aux, const Act<2> &c, int lane, float (&out)[1], Kept<0> *kept) {
    const int g = lane >> 4;
    Act<2> h = c;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        f32x4 acc[2];
        acc[0] = to_v(ld4(aux + AUX_B + i * 32 + 4 * g));
        acc[1] = to_v(ld4(aux + AUX_B + i * 32 + 16 + 4 * g));
        if (i == 3) {
            gemv_fwd<2>(acc, c, pk + nox_mat(NW3C).pk, lane);
            gemv_fwd<2>(acc, h, pk + nox_mat(NW3H).pk, lane);
        } else {
            gemv_fwd<2>(acc, h, pk + nox_mat(i == 0 ? NW0 : (i == 1 ? NW1 : (i == 2 ? NW2 : NW4))).pk, lane);
        }
        unsigned m = 0;
        if (KEEP) m = relu_mask(acc); else relu_plain(acc);
        h.t[0] = acc[0];
        h.t[1] = acc[1];
        if (KEEP) { kept->h[i] = h; kept->mask[i] = m; }
    }
    const F4 w0 = ld4(aux + AUX_WO + 4 * g), w1 = ld4(aux + AUX_WO + 16 + 4 * g);
    float s = w0.x * h.t[0][0];
    s = fmaf(w0.y, h.t[0][1], s); s = fmaf(w0.z, h.t[0][2], s); s = fmaf(w0.w, h.t[0][3], s);
    s = fmaf(w1.x, h.t[1][0], s); s = fmaf(w1.y, h.t[1][1], s); s = fmaf(w1.z, h.t[1][2], s); s = fmaf(w1.w, h.t[1][3], s);
    out[0] = red_g(s) + aux[AUX_BO];
}

// ------------------------------------------------------------------------------------------------
template <int STAGE>
NSR_DEV void load_stage_aux(const RenderParams &P, float *aux) {
    if (STAGE == NSR_STAGE_COARSE) {
        load_aux<NSR_COARSE>(aux, P.dec[NSR_COARSE].params);
    } else {
        load_aux<NSR_MIDDLE>(aux, P.dec[NSR_MIDDLE].params);
        if (STAGE >= NSR_STAGE_FINE) load_aux<NSR_FINE>(aux + AUX_FLOATS, P.dec[NSR_FINE].params);
        if (STAGE == NSR_STAGE_COLOR) load_aux<NSR_COLOR>(aux + 2 * AUX_FLOATS, P.dec[NSR_COLOR].params);
    }
}

// ------------------------------------------------------------------------------------------------
// compositor (common.py:231-244, occupancy branch), one wave per ray, lane = sample
// ------------------------------------------------------------------------------------------------
struct Comp { float alpha, T, w, t; };

NSR_DEV Comp comp_weights(float occ, bool active, int lane) {
    Comp c;
    c.alpha = active ? 1.f / (1.f + expf(-(10.f * occ))) : 0.f;
    c.t = active ? (1.f - c.alpha) + 1e-10f : 1.f;
    float v = c.t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = shfl_up(v, d);
        if (lane >= d) v *= o;
    }
    float T = shfl_up(v, 1);
    if (lane == 0) T = 1.f;
    c.T = T;
    c.w = c.alpha * T;
    return c;
}
NSR_DEV float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += shfl_xor(v, d);
    return v;
}
NSR_DEV double wave_sum_d(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += shfl_xor_d(v, d);
    return v;
}

// NICE.forward for the tile of this wave with the packed weights staged in LDS, one decoder after the other
// (block-wide barriers inside: EVERY wave of the block must call it).  On entry `wl` holds the first decoder of the
// stage (coarse or middle); on exit the last one.  Returns (r,g,b,occ) before the out-of-bound override.
template <int STAGE>
NSR_DEV F4 decode_tile_lds(const RenderParams &P, const float *aux, float *wl, double px, double py, double pz, int lane) {
    const int g = lane >> 4;
    F4 raw = F4{0.f, 0.f, 0.f, 0.f};
    if (STAGE == NSR_STAGE_COARSE) {
        const Lvl L = make_level(P.grid[NSR_COARSE], px, py, pz);
        const Act<2> c = gather_feat(P.grid[NSR_COARSE], L, g);
        float o[1];
        mlp_nox_fwd<false>(wl, aux, c, lane, o, nullptr);
        raw.w = o[0];
    } else {
        const float fx = (float)px, fy = (float)py, fz = (float)pz;     // decoder.py:189
        const Lvl Lm = make_level(P.grid[NSR_MIDDLE], px, py, pz);
        const Act<2> cm = gather_feat(P.grid[NSR_MIDDLE], Lm, g);
        float om[1];
        mlp_xyz_fwd<NSR_MIDDLE, false>(wl, aux, fx, fy, fz, cm, lane, om, nullptr);
        float occ = om[0];
        if (STAGE >= NSR_STAGE_FINE) {
            const Lvl Lf = make_level(P.grid[NSR_FINE], px, py, pz);
            const Act<2> cf = gather_feat(P.grid[NSR_FINE], Lf, g);
            block_sync();                                               // everyone is done with the middle weights
            load_packed<NSR_FINE>(wl, P.dec[NSR_FINE].packed);
            block_sync();
            Act<4> cc;
            cc.t[0] = cf.t[0]; cc.t[1] = cf.t[1]; cc.t[2] = cm.t[0]; cc.t[3] = cm.t[1];    // decoder.py:182-187
            float of[1];
            mlp_xyz_fwd<NSR_FINE, false>(wl, aux + AUX_FLOATS, fx, fy, fz, cc, lane, of, nullptr);
            occ = of[0] + om[0];                                                            // decoder.py:333,341
        }
        if (STAGE == NSR_STAGE_COLOR) {
            const Lvl Lc = make_level(P.grid[NSR_COLOR], px, py, pz);
            const Act<2> ccol = gather_feat(P.grid[NSR_COLOR], Lc, g);
            block_sync();
            load_packed<NSR_COLOR>(wl, P.dec[NSR_COLOR].packed);
            block_sync();
            float oc[4];
            mlp_xyz_fwd<NSR_COLOR, false>(wl, aux + 2 * AUX_FLOATS, fx, fy, fz, ccol, lane, oc, nullptr);
            raw.x = oc[0]; raw.y = oc[1]; raw.z = oc[2];
        }
        raw.w = occ;
    }
    return raw;
}

// ------------------------------------------------------------------------------------------------
// forward kernel
// LDS: aux[3*AUX] | ztmp[npts] f64 | zbuf[npts] f64 | rawbuf[npts] F4 | wl: packed weights of the decoder in flight
// The three decoders of a stage are evaluated one after the other; before each one the block stages that
// decoder's packed operand stream (61-82 KB) in LDS so that every MFMA operand is an LDS read (~100 cycles)
// instead of an L2 round trip.  The gathers of the next decoder's features are issued before the barrier.
// ------------------------------------------------------------------------------------------------
template <int STAGE>
NSR_KERNEL NSR_BOUNDS(768) void render_fwd_kernel(const RenderParams P) {
    char *lds = lds_base();
    const int npts = P.rays_per_block * P.S;
    float *aux = reinterpret_cast<float *>(lds);
    double *ztmp = reinterpret_cast<double *>(lds + sizeof(float) * (3 * AUX_FLOATS + (3 * AUX_FLOATS & 1)));
    double *zbuf = ztmp + npts;
    F4 *rawbuf = reinterpret_cast<F4 *>(zbuf + npts);
    float *wl = reinterpret_cast<float *>(rawbuf + npts);
    const int lane = tid() & 63, wave = tid() >> 6, nwaves = nthreads() >> 6, g = lane >> 4;
    const int S = P.S;

    load_stage_aux<STAGE>(P, aux);
    if (STAGE == NSR_STAGE_COARSE) load_packed<NSR_COARSE>(wl, P.dec[NSR_COARSE].packed);     // only one decoder: staged once
    if (STAGE == NSR_STAGE_MIDDLE) load_packed<NSR_MIDDLE>(wl, P.dec[NSR_MIDDLE].packed);
    for (long long grp = bid_x(); grp < P.n_groups; grp += nblk_x()) {
        loop_fence();
        const long long ray0 = grp * P.rays_per_block;
        if (STAGE > NSR_STAGE_MIDDLE) load_packed<NSR_MIDDLE>(wl, P.dec[NSR_MIDDLE].packed);   // fine / colour overwrote it
        compute_z(P, ray0, ztmp, zbuf);            // ends with block_sync (also covers the aux / weight staging)
        const int pidx = wave * kTile + (lane & 15);
        const int r = pidx / S, k = pidx - r * S;
        const long long ray = ray0 + r;
        const bool active = (pidx < npts) && (ray < P.n_rays);
        const long long rr = active ? ray : 0;
        const double z = active ? zbuf[pidx] : 0.0;
        // pts = o + d*z in fp64 (Renderer.py:172-174)
        const double px = (double)P.rays_o[rr * 3 + 0] + (double)P.rays_d[rr * 3 + 0] * z;
        const double py = (double)P.rays_o[rr * 3 + 1] + (double)P.rays_d[rr * 3 + 1] * z;
        const double pz = (double)P.rays_o[rr * 3 + 2] + (double)P.rays_d[rr * 3 + 2] * z;
        const bool inside = (px > P.blo[0]) && (px < P.bhi[0]) && (py > P.blo[1]) && (py < P.bhi[1]) &&
                            (pz > P.blo[2]) && (pz < P.bhi[2]);
        F4 raw = decode_tile_lds<STAGE>(P, aux, wl, px, py, pz, lane);
        if (!inside) raw.w = 100.f;                                         // Renderer.py:57
        if (active && g == 0) {
            rawbuf[pidx] = raw;
            if (P.raw) st4(P.raw + (ray * S + k) * 4, raw);
            if (P.zvals) P.zvals[ray * S + k] = z;
        }
        block_sync();
        for (int rq = wave; rq < P.rays_per_block; rq += nwaves) {
            const long long rayq = ray0 + rq;
            if (rayq >= P.n_rays) break;
            const bool act = lane < S;
            const F4 rw = act ? rawbuf[rq * S + lane] : F4{0.f, 0.f, 0.f, 0.f};
            const double zq = act ? zbuf[rq * S + lane] : 0.0;
            const Comp c = comp_weights(rw.w, act, lane);
            const float cr = wave_sum(c.w * rw.x), cg = wave_sum(c.w * rw.y), cb = wave_sum(c.w * rw.z);
            const double depth = wave_sum_d((double)c.w * zq);
            const double dz = zq - depth;
            const double var = wave_sum_d(((double)c.w * dz) * dz);
            if (lane == 0) {
                P.depth[rayq] = depth;
                P.var[rayq] = var;
                P.rgb[rayq * 3 + 0] = cr; P.rgb[rayq * 3 + 1] = cg; P.rgb[rayq * 3 + 2] = cb;
            }
        }
        block_sync();
    }
}

// Renderer.eval_points forward over a flat list of points (Renderer.py:23-61).  Same decoder phases as the render
// kernel (packed weights staged in LDS per decoder); a block takes groups of `nwaves` tiles.
template <int STAGE>
NSR_KERNEL NSR_BOUNDS(768) void eval_points_kernel(const RenderParams P) {
    float *aux = reinterpret_cast<float *>(lds_base());
    float *wl = aux + 3 * AUX_FLOATS + ((3 * AUX_FLOATS) & 3 ? 4 - ((3 * AUX_FLOATS) & 3) : 0);
    const int lane = tid() & 63, wave = tid() >> 6, nwaves = nthreads() >> 6;
    load_stage_aux<STAGE>(P, aux);
    if (STAGE == NSR_STAGE_COARSE) load_packed<NSR_COARSE>(wl, P.dec[NSR_COARSE].packed);
    if (STAGE == NSR_STAGE_MIDDLE) load_packed<NSR_MIDDLE>(wl, P.dec[NSR_MIDDLE].packed);
    const long long ntiles = (P.n_points + kTile - 1) / kTile;
    const long long ngroups = (ntiles + nwaves - 1) / nwaves;
    for (long long grp = bid_x(); grp < ngroups; grp += nblk_x()) {
        loop_fence();
        if (STAGE > NSR_STAGE_MIDDLE) load_packed<NSR_MIDDLE>(wl, P.dec[NSR_MIDDLE].packed);   // fine / colour overwrote it
        block_sync();
        const long long pi = (grp * nwaves + wave) * kTile + (lane & 15);
        const bool active = pi < P.n_points;
        const long long pp = active ? pi : 0;
        const double px = P.points[pp * 3 + 0], py = P.points[pp * 3 + 1], pz = P.points[pp * 3 + 2];
        F4 raw = decode_tile_lds<STAGE>(P, aux, wl, px, py, pz, lane);
        const bool inside = (px > P.blo[0]) && (px < P.bhi[0]) && (py > P.blo[1]) && (py < P.bhi[1]) &&
                            (pz > P.blo[2]) && (pz < P.bhi[2]);
        if (!inside) raw.w = 100.f;                                         // Renderer.py:57
        if (active && (lane >> 4) == 0) st4(P.out_points + pi * 4, raw);
        block_sync();                                                       // before the next group re-stages the weights
    }
}

// ------------------------------------------------------------------------------------------------
// backward of one decoder for one tile
// ------------------------------------------------------------------------------------------------
struct BwdFlags { bool grid, params, rays; };

// one layer of the xyz-decoder backward (i = 4..0), instantiated per layer so that every register
// array index is a compile-time constant
template <int KIND>
struct XyzBwd {
    static constexpr int CD = cdim_of(KIND), NTC = CD / 16;
    const float *wl;     // packed operand stream of this decoder, staged in LDS
    const float *aux;
    Own O;
    float *S;            // this wave's staging region
    const Kept<KIND> &K;
    BwdFlags F;
    int lane;
    Act<2> &dc;
    Act<2> &dh;
    Act<2> dY3, dY0;

    template <int I>
    NSR_DEV void layer() {
        const int i16 = lane & 15, g = lane >> 4;
        constexpr int uid = I == 0 ? XU0 : (I == 1 ? XU1 : (I == 2 ? XU2 : (I == 3 ? XU3 : XU4)));
        constexpr int hid = I == 1 ? XW1 : (I == 2 ? XW2 : (I == 3 ? XW3H : XW4));
        const Mat mu = xyz_mat(CD, uid);
        if (F.params) st_store(S + kStA0, dh, i16, g);                      // dH_i: gradient of (U_i c + v_i) is dh itself
        if (F.grid || F.rays) gemv_bwd<2>(dc.t, dh, wl + mu.pk, i16, g);      // first 32 feature columns only
        const Act<2> dY = apply_mask(dh, K.mask[I]);
        if (I == 3) dY3 = dY;
        if (I == 0) dY0 = dY;
        if (F.params) {
            st_store(S + kStA1, dY, i16, g);
            if (I > 0) st_store(S + kStX0, K.h[I > 0 ? I - 1 : 0], i16, g);
            else st_store(S + kStX0, dY3, i16, g);                           // layer 0: the W0 / W3e task needs dY3 too
            block_sync();
            int t = 0;                                   // tasks of this phase, heaviest first (see mine())
            if (I == 0) {
#pragma unroll
                for (int Tk = 0; Tk < kET; ++Tk, ++t)
                    if (mine(O, t)) own_embed_pair(O, xyz_mat(CD, XW0), xyz_mat(CD, XW3E), Tk, kStA1, kStX0, aux);
            }
            if (I > 0) {
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk, ++t)
                    if (mine(O, t)) own_pair(O, xyz_mat(CD, hid), Tk, kStA1, kStX0, Tk);
            }
#pragma unroll
            for (int Tk = 0; Tk < NTC; ++Tk, ++t)
                if (mine(O, t)) own_pair(O, mu, Tk, kStA0, kStC + (Tk >> 1) * 512, Tk & 1);
            if (mine(O, t)) own_colsum(O, fcb_off(KIND, I), kStA0);
            if (mine(O, t + 1)) own_colsum(O, bias_off(KIND, I), kStA1);
            block_sync();
        }
        if (I > 0) {
            Act<2> nd;
            act_zero(nd);
            gemv_bwd<2>(nd.t, dY, wl + xyz_mat(CD, hid).pk, i16, g);
            dh = nd;
        }
    }
};

// xyz decoder.  c: features (CL).  d_out: gradient of the decoder outputs of this lane's point.
// dc: gradient w.r.t. the first 32 feature channels (the decoder's own grid).  dp: gradient w.r.t.
// the fp32 world position through the embedding (already reduced over g).
// With F.params every wave of the block must call this function (it contains block barriers).
// Where this lane's d raw comes from: the compositor backward (a few waves of the block) fills `draw` in LDS while the
// other waves already re-run the decoder forward; `sync` = this is the first use of `draw` in the ray group, so a block
// barrier has to separate the two.
struct DrawRef {
    const F4 *draw;
    int pidx;
    bool active, inside, sync;
};
NSR_DEV F4 draw_fetch(const DrawRef &R) {
    if (R.sync) block_sync();
    F4 dr = R.active ? R.draw[R.pidx] : F4{0.f, 0.f, 0.f, 0.f};
    if (!R.inside) dr.w = 0.f;                                 // Renderer.py:57 cuts the occupancy gradient
    return dr;
}

template <int KIND>
NSR_DEV void mlp_xyz_bwd(const float *pk, const float *aux, const Own &O, float *S,
                         float px, float py, float pz, const Act<cdim_of(KIND) / 16> &c,
                         const DrawRef &R, BwdFlags F, int lane, Act<2> &dc, float (&dp)[3]) {
    constexpr int CD = cdim_of(KIND), NOUT = nout_of(KIND), NTC = CD / 16;
    const int i16 = lane & 15, g = lane >> 4;
    Kept<KIND> K;
    float out[NOUT];
    mlp_xyz_fwd<KIND, true>(pk, aux, px, py, pz, c, lane, out, &K);
    (void)out;
    const F4 dr = draw_fetch(R);
    float d_out[NOUT];
    if (NOUT == 1) { d_out[0] = dr.w; }
    else { d_out[0] = dr.x; d_out[NOUT > 1 ? 1 : 0] = dr.y; d_out[NOUT > 2 ? 2 : 0] = dr.z; d_out[NOUT > 3 ? 3 : 0] = 0.f; }   // decoder.py:341 overwrites the 4th colour output

    // output layer
    Act<2> dh;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        f32x4 v = f4zero();
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            const F4 w = ld4(aux + AUX_WO + n * 32 + 16 * T + 4 * g);
            v[0] = fmaf(w.x, d_out[n], v[0]); v[1] = fmaf(w.y, d_out[n], v[1]);
            v[2] = fmaf(w.z, d_out[n], v[2]); v[3] = fmaf(w.w, d_out[n], v[3]);
        }
        dh.t[T] = v;
    }
    if (F.params) {
        if (g == 0) {
            S[kStP + i16] = px; S[kStP + 16 + i16] = py; S[kStP + 32 + i16] = pz;       // [xyz][16 points]
            st4(S + kStDO + i16 * 4, F4{d_out[0], NOUT > 1 ? d_out[NOUT > 1 ? 1 : 0] : 0.f, NOUT > 2 ? d_out[NOUT > 2 ? 2 : 0] : 0.f,
                                        NOUT > 3 ? d_out[NOUT > 3 ? 3 : 0] : 0.f});
        }
#pragma unroll
        for (int q = 0; q < NTC / 2; ++q) {
            Act<2> cq;
            cq.t[0] = c.t[2 * q]; cq.t[1] = c.t[2 * q + 1];
            st_store(S + kStC + q * 512, cq, i16, g);
        }
        st_store(S + kStX0, K.h[4], i16, g);
        wave_fence();
        out_layer_local<NOUT>(O, S);
        wave_fence();
    }

    act_zero(dc);
    XyzBwd<KIND> X{pk, aux, O, S, K, F, lane, dc, dh};
    act_zero(X.dY3);
    act_zero(X.dY0);
    X.template layer<4>();
    X.template layer<3>();
    X.template layer<2>();
    X.template layer<1>();
    X.template layer<0>();
    const Act<2> dY3 = X.dY3, dY0 = X.dY0;

    // ---- embedding: dE = W0^T dY0 + W3e^T dY3 ; d arg = dE * cos(arg)
    dp[0] = dp[1] = dp[2] = 0.f;
    const bool need_dB = F.params;
    if (F.rays || need_dB) {
        const Mat m0 = xyz_mat(CD, XW0), m3 = xyz_mat(CD, XW3E);
        const int lo = (4 * g + 16 * (i16 >> 2)) * 4 + (i16 & 3);       // packed-stream position of W[.][16Tk+i16], see gemv_bwd
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int Tk = 0; Tk < kET; ++Tk) {
            f32x4 dE = f4zero(), dE2 = f4zero();
#pragma unroll
            for (int To = 0; To < 2; ++To)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a0 = pk[m0.pk + Tk * 512 + To * 256 + r * 4 + lo];
                    const float a3 = pk[m3.pk + Tk * 512 + To * 256 + r * 4 + lo];
                    dE = mfma16(a0, dY0.t[To][r], dE);
                    dE2 = mfma16(a3, dY3.t[To][r], dE2);
                }
            sched_fence();
            dE += dE2;
            const B4 b = load_b4(aux, 4 * Tk + g);
            const f32x4 darg = dE * cos_acc4(vfma(splat(pz), b.z, vfma(splat(py), b.y, splat(px) * b.x)));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ax = fmaf(darg[r], b.x[r], ax); ay = fmaf(darg[r], b.y[r], ay); az = fmaf(darg[r], b.z[r], az);
            }
            if (need_dB) st4(S + kStA0 + i16 * 96 + 16 * Tk + 4 * g, to_F4(darg));     // [16][96] over A0|A1|X0
        }
        dp[0] = red_g(ax); dp[1] = red_g(ay); dp[2] = red_g(az);
    }
    if (need_dB) {
        block_sync();
#pragma unroll
        for (int Tk = 0; Tk < kET; ++Tk)
            if (mine(O, Tk)) own_dB(O, Tk, B_off(KIND));
        block_sync();
    }
}

struct NoxBwd {
    const float *wl;     // packed operand stream of the coarse decoder, staged in LDS
    Own O;
    float *S;
    const Act<2> &c;
    const Kept<0> &K;
    BwdFlags F;
    int lane;
    Act<2> &dc;
    Act<2> &dh;

    template <int I>
    NSR_DEV void layer() {
        const int i16 = lane & 15, g = lane >> 4;
        const Act<2> dY = apply_mask(dh, K.mask[I]);
        const Mat mh = nox_mat(I == 0 ? NW0 : (I == 1 ? NW1 : (I == 2 ? NW2 : (I == 3 ? NW3H : NW4))));
        if (F.params) {
            st_store(S + kStA1, dY, i16, g);
            st_store(S + kStX0, I == 0 ? c : K.h[I > 0 ? I - 1 : 0], i16, g);
            block_sync();
            int t = 0;
            if (I == 3) {
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk, ++t)
                    if (mine(O, t)) own_pair(O, nox_mat(NW3C), Tk, kStA1, kStC, Tk);
            }
#pragma unroll
            for (int Tk = 0; Tk < 2; ++Tk, ++t)
                if (mine(O, t)) own_pair(O, mh, Tk, kStA1, kStX0, Tk);
            if (mine(O, t)) own_colsum(O, nox_b(I), kStA1);
            block_sync();
        }
        if (I == 3) gemv_bwd<2>(dc.t, dY, wl + nox_mat(NW3C).pk, i16, g);
        if (I == 0) {
            gemv_bwd<2>(dc.t, dY, wl + mh.pk, i16, g);
        } else {
            Act<2> nd;
            act_zero(nd);
            gemv_bwd<2>(nd.t, dY, wl + mh.pk, i16, g);
            dh = nd;
        }
    }
};

// coarse decoder backward (MLP_no_xyz)
NSR_DEV void mlp_nox_bwd(const float *pk, const float *aux, const Own &O, float *S,
                         const Act<2> &c, const DrawRef &R, BwdFlags F, int lane, Act<2> &dc) {
    const int i16 = lane & 15, g = lane >> 4;
    Kept<0> K;
    float out[1];
    mlp_nox_fwd<true>(pk, aux, c, lane, out, &K);
    (void)out;
    const float d_out = draw_fetch(R).w;
    Act<2> dh;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const F4 w = ld4(aux + AUX_WO + 16 * T + 4 * g);
        f32x4 v = {w.x * d_out, w.y * d_out, w.z * d_out, w.w * d_out};
        dh.t[T] = v;
    }
    if (F.params) {
        if (g == 0) st4(S + kStDO + i16 * 4, F4{d_out, 0.f, 0.f, 0.f});
        st_store(S + kStC, c, i16, g);
        st_store(S + kStX0, K.h[4], i16, g);
        wave_fence();
        out_layer_local<1>(O, S);
        wave_fence();
    }
    act_zero(dc);
    NoxBwd X{pk, O, S, c, K, F, lane, dc, dh};
    X.layer<4>();
    X.layer<3>();
    X.layer<2>();
    X.layer<1>();
    X.layer<0>();
}

// ------------------------------------------------------------------------------------------------
// backward kernel.  grid = (blocks, passes); pass p handles one decoder:
//   coarse stage: p0 = coarse.   otherwise: p0 = middle, p1 = fine, p2 = color.
// LDS: aux[AUX] | packed weights of the decoder | ztmp f64[npts] | zbuf f64[npts] | draw F4[npts] | dpb f64[npts*3]
//      | per-wave staging regions (stg_floats(KIND) each)
// The per-block image of the parameter gradients lives in the global partial buffer (stays in L2; exclusive owner
// per element, first ray group stores, later groups accumulate), summed over blocks by reduce_partials_kernel.
// ------------------------------------------------------------------------------------------------
// PARAMS is the compile-time twin of "this decoder's dparams != NULL": the pass without parameter gradients (tracking,
// decoders the optimiser does not step) needs neither the kept activations nor the staging code and compiles without
// register spills
template <int KIND, bool PARAMS>
NSR_DEV void bwd_pass(const RenderParams &P) {
    constexpr int NPAR = param_total(KIND);
    char *lds = lds_base();
    const int npts = P.rays_per_block * P.S, S = P.S;
    const int lane = tid() & 63, wave = tid() >> 6, nwaves = nthreads() >> 6;
    float *aux = reinterpret_cast<float *>(lds);
    float *wl = aux + AUX_FLOATS;                          // this decoder's packed operand stream
    constexpr int head = (AUX_FLOATS + packed_total(KIND) + 3) & ~3;
    double *ztmp = reinterpret_cast<double *>(aux + head);
    double *zbuf = ztmp + npts;
    F4 *draw = reinterpret_cast<F4 *>(zbuf + npts);
    double *dpb = reinterpret_cast<double *>(draw + npts);
    float *small = reinterpret_cast<float *>(dpb + 3 * npts);              // [waves][132] floats (wo[4][32] | bo[4])
    const int stg_off = (head * 4 + npts * (8 + 8 + 16 + 24) + nwaves * 132 * 4 + 15) & ~15;
    float *stg = reinterpret_cast<float *>(lds + stg_off);
    float *Sw = stg + wave * stg_floats(KIND);             // this wave's staging region (also Tx / tab of the scatter)

    const GridDev &G = P.grid[KIND];
    const DecDev &D = P.dec[KIND];
    BwdFlags F;
    F.grid = G.dfeat != nullptr;
    F.params = PARAMS;
    F.rays = P.d_rays_o != nullptr;
    if (!F.grid && !F.params && !F.rays) return;

    load_aux<KIND>(aux, D.params);
    load_packed<KIND>(wl, D.packed);                       // visible after the first barrier inside compute_z
    for (int t = tid(); t < 132 * nwaves; t += nthreads()) small[t] = 0.f;
    float *img = F.params ? P.partials + ((long long)bid_y() * nblk_x() + bid_x()) * P.partial_stride : nullptr;
    (void)NPAR;

    for (long long grp = bid_x(); grp < P.n_groups; grp += nblk_x()) {
        loop_fence();
        const bool first_grp = grp == (long long)bid_x();
        Own O{make_stream(img), stg, stg_floats(KIND), nwaves, wave, lane, first_grp, small};
        const long long ray0 = grp * P.rays_per_block;
        if (P.zvals) {            // sample depths saved by the forward pass: one coalesced load instead of re-deriving them
            for (int t = tid(); t < npts; t += nthreads())
                zbuf[t] = (ray0 + t / S < P.n_rays) ? P.zvals[ray0 * S + t] : 0.0;
            block_sync();
        } else {
            compute_z(P, ray0, ztmp, zbuf);
        }
        // ---- tile set-up first: the feature gathers (L2 / Infinity-Cache latency) fly while the compositor runs.
        // A ray group holds tiles_per_block tiles; the block's waves take them nwaves at a time (sub-rounds).
        const int g = lane >> 4;
        int pidx;
        bool active, inside;
        double px, py, pz;
        Lvl L;
        Act<2> c, cm;
        auto tile_setup = [&](int sub) {
            pidx = (sub * nwaves + wave) * kTile + (lane & 15);
            const long long ray_t = ray0 + pidx / S;
            active = (pidx < npts) && (ray_t < P.n_rays);
            const long long rr = active ? ray_t : 0;
            const double zt = active ? zbuf[pidx] : 0.0;
            px = (double)P.rays_o[rr * 3 + 0] + (double)P.rays_d[rr * 3 + 0] * zt;
            py = (double)P.rays_o[rr * 3 + 1] + (double)P.rays_d[rr * 3 + 1] * zt;
            pz = (double)P.rays_o[rr * 3 + 2] + (double)P.rays_d[rr * 3 + 2] * zt;
            inside = (px > P.blo[0]) && (px < P.bhi[0]) && (py > P.blo[1]) && (py < P.bhi[1]) &&
                     (pz > P.blo[2]) && (pz < P.bhi[2]);
            L = make_level(G, px, py, pz);
            c = gather_feat(G, L, g);
            if (KIND == NSR_FINE) {
                const Lvl Lm = make_level(P.grid[NSR_MIDDLE], px, py, pz);
                cm = gather_feat(P.grid[NSR_MIDDLE], Lm, g);
            }
        };
        tile_setup(0);
        // ---- compositor backward: d raw per sample (common.py:231-244 differentiated)
        // (rays go to waves 2, 3, ... first: with six waves on four SIMDs those two do not share their SIMD, and the
        // other waves meanwhile start on the decoder forward re-run -- the barrier sits inside mlp_*_bwd, see DrawRef)
        for (int r = nwaves >= 4 ? (wave + nwaves - 2) % nwaves : wave; r < P.rays_per_block; r += nwaves) {
            const long long ray = ray0 + r;
            if (ray >= P.n_rays) break;
            const bool act = lane < S;
            const F4 rw = act ? ld4(P.raw + (ray * S + lane) * 4) : F4{0.f, 0.f, 0.f, 0.f};
            const double z = act ? zbuf[r * S + lane] : 0.0;
            const Comp c = comp_weights(rw.w, act, lane);
            const double gD = P.d_depth ? P.d_depth[ray] : 0.0;
            const double gV = P.d_var ? P.d_var[ray] : 0.0;
            float gr = 0.f, gg = 0.f, gb = 0.f;
            if (P.d_rgb) { gr = P.d_rgb[ray * 3 + 0]; gg = P.d_rgb[ray * 3 + 1]; gb = P.d_rgb[ray * 3 + 2]; }
            const double dz = z - P.g_depth[ray];
            const double s1 = wave_sum_d((double)c.w * dz);
            const float Gz = (float)(gD * z + gV * (dz * dz - 2.0 * s1 * z));
            const float Gw = Gz + fmaf(gb, rw.z, fmaf(gg, rw.y, gr * rw.x));
            float v = act ? Gw * c.w : 0.f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float o = shfl_down(v, d);
                if (lane + d < 64) v += o;
            }
            float suffix = shfl_down(v, 1);
            if (lane == 63) suffix = 0.f;
            const float dalpha = Gw * c.T - suffix / c.t;
            const float docc = 10.f * (dalpha * ((1.f - c.alpha) * c.alpha));
            if (act) draw[r * S + lane] = F4{c.w * gr, c.w * gg, c.w * gb, docc};
        }
        for (int sub = 0;;) {   // ---- decoder backward for the tile of this wave
            O.first = first_grp && sub == 0;                       // layer images: stored by the block's first sub-round
            const DrawRef R{draw, pidx, active, inside, sub == 0};
            Act<2> dc;
            float dpe[3] = {0.f, 0.f, 0.f};
            if (KIND == NSR_COARSE) {
                mlp_nox_bwd(wl, aux, O, Sw, c, R, F, lane, dc);
            } else if (KIND == NSR_MIDDLE) {
                mlp_xyz_bwd<NSR_MIDDLE>(wl, aux, O, Sw, (float)px, (float)py, (float)pz, c, R, F, lane, dc, dpe);
            } else if (KIND == NSR_FINE) {
                Act<4> cc;
                cc.t[0] = c.t[0]; cc.t[1] = c.t[1]; cc.t[2] = cm.t[0]; cc.t[3] = cm.t[1];
                mlp_xyz_bwd<NSR_FINE>(wl, aux, O, Sw, (float)px, (float)py, (float)pz, cc, R, F, lane, dc, dpe);
            } else {
                mlp_xyz_bwd<NSR_COLOR>(wl, aux, O, Sw, (float)px, (float)py, (float)pz, c, R, F, lane, dc, dpe);
            }
            float dux = 0.f, duy = 0.f, duz = 0.f;
            if (F.rays) coord_grad(G, L, g, dc, dux, duy, duz);
            if (F.grid) scatter_merged(G, L, lane, dc, active, Sw + kStA0, Sw + kStA0 + kTile * kTxS);
            if (F.rays && active && g == 0) {
                // d p = d u * (n-1)/2 * 2/(hi-lo)  (+ embedding part), fp64 like autograd through Renderer.py:172
                dpb[pidx * 3 + 0] = (double)dux * (2.0 * G.inv[0]) + (double)dpe[0];
                dpb[pidx * 3 + 1] = (double)duy * (2.0 * G.inv[1]) + (double)dpe[1];
                dpb[pidx * 3 + 2] = (double)duz * (2.0 * G.inv[2]) + (double)dpe[2];
            }
            if (++sub * nwaves >= P.tiles_per_block) break;
            wave_fence();                                      // the scatter is done with this wave's staging region
            tile_setup(sub);
        }
        block_sync();
        if (F.rays) {
            for (int t = tid(); t < P.rays_per_block * 6; t += nthreads()) {
                const int r = t / 6, q = t - r * 6, a = q % 3;
                const long long ray = ray0 + r;
                if (ray >= P.n_rays) continue;
                double s = 0.0;
                for (int k = 0; k < S; ++k) {
                    const double d = dpb[(r * S + k) * 3 + a];
                    s += (q < 3) ? d : d * zbuf[r * S + k];
                }
                atomic_add_global((q < 3 ? P.d_rays_o : P.d_rays_d) + ray * 3 + a, (float)s);
            }
        }
        block_sync();
    }
    if (F.params) {              // output-layer gradients: the per-wave slots were accumulated over all ray groups of the block
        constexpr int NO = nout_of(KIND);
        const Stream st = make_stream(img);
        const int t = tid();
        if (t < NO * 32 || (t >= 128 && t < 128 + NO)) {
            float v = 0.f;
            for (int w = 0; w < nwaves; ++w) v += small[w * 132 + t];
            if (t < 128) stream_st(st, t, wo_off(KIND), v); else stream_st(st, t - 128, bo_off(KIND), v);
        }
    }
}

template <int STAGE>
NSR_KERNEL NSR_BOUNDS(64 * NSR_BWD_WAVES) void render_bwd_kernel(const RenderParams P) {
    if (STAGE == NSR_STAGE_COARSE) {
        if (P.dec[NSR_COARSE].dparams) bwd_pass<NSR_COARSE, true>(P); else bwd_pass<NSR_COARSE, false>(P);
    } else {
        const int pass = bid_y();
        if (pass == 0) {
            if (P.dec[NSR_MIDDLE].dparams) bwd_pass<NSR_MIDDLE, true>(P); else bwd_pass<NSR_MIDDLE, false>(P);
        } else if (pass == 1) {
            if (STAGE >= NSR_STAGE_FINE) { if (P.dec[NSR_FINE].dparams) bwd_pass<NSR_FINE, true>(P); else bwd_pass<NSR_FINE, false>(P); }
        } else {
            if (STAGE == NSR_STAGE_COLOR) { if (P.dec[NSR_COLOR].dparams) bwd_pass<NSR_COLOR, true>(P); else bwd_pass<NSR_COLOR, false>(P); }
        }
    }
}

// sum the per-block partial parameter gradients:  dparams[t] += sum_b partials[b][t], one grid row (blockIdx.y) per
// decoder pass of the stage.  block = 64 parameters x (blockDim/64) slices of the partial list (coalesced 256-byte
// rows, split serial sum)
struct ReduceJob {
    const float *partials;   // [nblocks][stride] of this pass
    float *dparams;          // flat gradient blob of the decoder (accumulated into)
    int n;                   // its parameter count (0: nothing to do for this row)
};
struct ReduceParams {
    ReduceJob job[3];
    int nblocks, stride;
    int overwrite;           // 1: dparams = sum (no caller-side zero fill needed), 0: dparams += sum
};
NSR_KERNEL void reduce_partials_kernel(const ReduceParams R) {
    const ReduceJob J = R.job[bid_y()];
    float *red = reinterpret_cast<float *>(lds_base());
    const int lane = tid() & 63, slice = tid() >> 6, nslice = nthreads() >> 6;
    const int t = bid_x() * 64 + lane;
    if (bid_x() * 64 >= J.n) return;                     // whole block beyond this decoder's blob (uniform)
    float s = 0.f;
    if (t < J.n)
        for (int b = slice; b < R.nblocks; b += nslice) s += J.partials[(long long)b * R.stride + t];
    red[tid()] = s;
    block_sync();
    if (slice == 0 && t < J.n) {
        for (int k = 1; k < nslice; ++k) s += red[k * 64 + lane];
        J.dparams[t] = R.overwrite ? s : J.dparams[t] + s;
    }
}

// ------------------------------------------------------------------------------------------------
// masked Adam on a channels-last feature grid (Mapper.py:368-379,394-401,504,511-519 fused; torch.optim.Adam
// single-tensor formulas).  One thread = 4 channels of one voxel (16-byte accesses, a voxel = 8 threads = one
// 128-byte line per array); HBM-bound: 896 B per updated voxel.
// ------------------------------------------------------------------------------------------------
struct AdamParams {
    float *p;
    const float *g;
    float *m, *v;
    const unsigned char *mask;
    long long n_vox;
    float step, b1, b2, eps, rs2;     // step = lr / (1 - b1^t), rs2 = sqrt(1 - b2^t)
};

NSR_KERNEL void masked_adam_kernel(const AdamParams A) {
    const long long t = (long long)bid_x() * nthreads() + tid();
    const long long vox = t >> 3;
    if (vox >= A.n_vox) return;
    if (A.mask && A.mask[vox] == 0) return;
    const long long o = vox * kC + (t & 7) * 4;
    const F4 g = ld4(A.g + o);
    F4 m = ld4(A.m + o), v = ld4(A.v + o), p = ld4(A.p + o);
    const float step = A.step, rs2 = A.rs2, omb1 = 1.f - A.b1, omb2 = 1.f - A.b2;
    // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2);
    // denom = exp_avg_sq.sqrt() / sqrt(bias2) + eps; param.addcdiv_(exp_avg, denom, value=-step)
#define NSR_ADAM1(c)                                                    \
    m.c = m.c + omb1 * (g.c - m.c);                                      \
    v.c = v.c * A.b2 + (omb2 * g.c) * g.c;                               \
    p.c = p.c - step * (m.c / (sqrtf(v.c) / rs2 + A.eps));
    NSR_ADAM1(x) NSR_ADAM1(y) NSR_ADAM1(z) NSR_ADAM1(w)
#undef NSR_ADAM1
    st4(A.m + o, m);
    st4(A.v + o, v);
    st4(A.p + o, p);
}

// ------------------------------------------------------------------------------------------------
// get_samples after the index draw (common.py:74-134)
// ------------------------------------------------------------------------------------------------
struct SampleParams {
    const long long *indices;
    long long n;
    int H0, W0, crop_w, W_full;
    float fx, fy, cx, cy;
    const float *c2w;
    int c2w_stride;
    const float *depth, *color;
    float *rays_o, *rays_d, *out_depth, *out_color;
};

NSR_KERNEL void get_samples_kernel(const SampleParams P) {
    const long long t = (long long)bid_x() * nthreads() + tid();
    if (t >= P.n) return;
    const long long idx = P.indices[t];
    const int row = (int)(idx / P.crop_w) + P.H0, col = (int)(idx % P.crop_w) + P.W0;
    const long long pix = (long long)row * P.W_full + col;
    P.out_depth[t] = P.depth[pix];
    P.out_color[t * 3 + 0] = P.color[pix * 3 + 0];
    P.out_color[t * 3 + 1] = P.color[pix * 3 + 1];
    P.out_color[t * 3 + 2] = P.color[pix * 3 + 2];
    const float dx = ((float)col - P.cx) / P.fx, dy = -(((float)row - P.cy) / P.fy), dzv = -1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float *R = P.c2w + a * P.c2w_stride;
        // torch.sum(dirs * c2w[:3,:3], -1): products, then left-to-right sum (common.py:87)
        P.rays_d[t * 3 + a] = (dx * R[0] + dy * R[1]) + dzv * R[2];
        P.rays_o[t * 3 + a] = R[3];
    }
}

// ------------------------------------------------------------------------------------------------
// bounding-box pre-filter of the callers (Mapper.py:471-481, Tracker.py:95-104): keep a ray iff its exit distance from
// the scene bound, t = min over axes of max((lo - o)/d, (hi - o)/d) in fp64, is >= its depth.  One thread per ray;
// also the maximum depth over the KEPT rays (the batch-global scalar of render_batch_ray, Renderer.py:109,144) so that
// neither the boolean-mask compaction nor its host sync is needed.
// ------------------------------------------------------------------------------------------------
struct AabbParams {
    const float *rays_o, *rays_d, *gt_depth;
    long long n;
    double lo[3], hi[3];
    unsigned char *keep;
    float *kept_max;               // optional; caller-initialised (0): max of gt_depth over kept rays
};

NSR_KERNEL void aabb_keep_kernel(const AabbParams P) {
    const long long r = (long long)bid_x() * nthreads() + tid();
    if (r >= P.n) return;
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double o = (double)P.rays_o[r * 3 + a], d = (double)P.rays_d[r * 3 + a];
        const double t0 = (P.lo[a] - o) / d, t1 = (P.hi[a] - o) / d;
        const double m = tmax(t0, t1);                            // torch.max(t, dim=2) / torch.min(., dim=1): NaN-propagating, like ray_far_bb
        t = (a == 0) ? m : tmin(t, m);
    }
    const float gd = P.gt_depth[r];
    const bool k = t >= (double)gd;
    P.keep[r] = k ? 1 : 0;
    if (k && P.kept_max && gd > 0.f) atomic_max_pos(P.kept_max, gd);
}

// ------------------------------------------------------------------------------------------------
// frustum feature selection (Mapper.get_mask_from_c2w, Mapper.py:93-164): one thread per voxel of a
// [Z][Y][X] grid.  phase 0: project the voxel centre, bilinear depth lookup (cv2.remap INTER_LINEAR semantics: 1/32-pixel
// fixed-point coordinates, zero border), store it, block maximum -> ws[n_vox + block].  phase 1: max over the block
// maxima (= np.max(depths), the fill value of zero-depth pixels, :147-148), depth test, near-camera sphere, mask byte.
// HBM-trivial (one depth gather + 5 B per voxel); exists to keep the per-frame mask on the device and off cv2/numpy.
// ------------------------------------------------------------------------------------------------
struct FrustumParams {
    float w2c[12];                 // rows 0..2 of inv(c2w), fp32 (Mapper.py:120)
    float cam_o[3];                // c2w[:3,3]
    double fx, fy, cx, cy;
    int H, W;
    const float *depth;            // [H][W]
    const float *xs, *ys, *zs;     // voxel-centre coordinates per axis (torch.linspace over the bound, :111-113)
    int nx, ny, nz, nblocks;
    long long n_vox;
    float *ws;                     // [n_vox] remapped depths | [nblocks] block maxima
    unsigned char *mask;           // [Z][Y][X]
};

struct FrustumProj {
    float u, v, px, py, pz;
    double zc;                     // camera z + 1e-5 (:129)
};

NSR_DEV FrustumProj frustum_project(const FrustumParams &P, long long vox) {
    FrustumProj R;
    const int ix = (int)(vox % P.nx), iy = (int)((vox / P.nx) % P.ny), iz = (int)(vox / ((long long)P.nx * P.ny));
    R.px = P.xs[ix]; R.py = P.ys[iy]; R.pz = P.zs[iz];
    float cam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)    // w2c @ [p,1] in fp32, sequential sum (oracle/frustum_oracle.py header)
        cam[r] = ((P.w2c[r * 4 + 0] * R.px + P.w2c[r * 4 + 1] * R.py) + P.w2c[r * 4 + 2] * R.pz) + P.w2c[r * 4 + 3];
    const double X = (double)(cam[0] * -1.f), Y = (double)cam[1], Z = (double)cam[2];
    const double uh = (P.fx * X + 0.0 * Y) + P.cx * Z;      // K @ cam_cord, fp64 (:126-128)
    const double vh = (0.0 * X + P.fy * Y) + P.cy * Z;
    R.zc = Z + 1e-5;
    R.u = (float)(uh / R.zc);
    R.v = (float)(vh / R.zc);
    return R;
}

NSR_DEV float frustum_pixel(const FrustumParams &P, int y, int x) {
    return (y >= 0 && y < P.H && x >= 0 && x < P.W) ? P.depth[(long long)y * P.W + x] : 0.f;
}

NSR_DEV float frustum_remap(const FrustumParams &P, float u, float v) {
    const float fu = u * 32.f, fv = v * 32.f;
    if (!(fabsf(fu) < 1.0e9f) || !(fabsf(fv) < 1.0e9f)) return 0.f;      // far outside / NaN: every tap is border
    const int sx = f2i_rn(fu), sy = f2i_rn(fv);
    const int x0 = sx >> 5, y0 = sy >> 5;
    const float ax = (float)(sx & 31) / 32.f, ay = (float)(sy & 31) / 32.f;
    float out = frustum_pixel(P, y0, x0) * ((1.f - ay) * (1.f - ax));
    out = out + frustum_pixel(P, y0, x0 + 1) * ((1.f - ay) * ax);
    out = out + frustum_pixel(P, y0 + 1, x0) * (ay * (1.f - ax));
    out = out + frustum_pixel(P, y0 + 1, x0 + 1) * (ay * ax);
    return out;
}

template <int PHASE>
NSR_KERNEL void frustum_mask_kernel(const FrustumParams P) {
    float *red = reinterpret_cast<float *>(lds_base());
    const long long vox = (long long)bid_x() * nthreads() + tid();
    const bool live = vox < P.n_vox;
    if (PHASE == 0) {
        float d = -INFINITY;
        if (live) {
            const FrustumProj R = frustum_project(P, vox);
            d = frustum_remap(P, R.u, R.v);
            P.ws[vox] = d;
        }
        red[tid()] = d;
        block_sync();
        if (tid() == 0) {
            float m = red[0];
            for (int k = 1; k < nthreads(); ++k) m = fmaxf(m, red[k]);
            P.ws[P.n_vox + bid_x()] = m;
        }
    } else {
        float m = -INFINITY;
        for (int k = tid(); k < P.nblocks; k += nthreads()) m = fmaxf(m, P.ws[P.n_vox + k]);
        red[tid()] = m;
        block_sync();
        if (!live) return;
        float dmax = red[0];
        for (int k = 1; k < nthreads(); ++k) dmax = fmaxf(dmax, red[k]);
        const FrustumProj R = frustum_project(P, vox);
        float d = P.ws[vox];
        if (d == 0.f) d = dmax;
        bool in = (R.u < (float)P.W) && (R.u > 0.f) && (R.v < (float)P.H) && (R.v > 0.f);
        const double zn = -R.zc;
        in = in && (0.0 <= zn) && (zn <= (double)(d + 0.5f));
        const float dx = R.px - P.cam_o[0], dy = R.py - P.cam_o[1], dz = R.pz - P.cam_o[2];
        const float dist = (dx * dx + dy * dy) + dz * dz;
        P.mask[vox] = (in || dist < 0.25f) ? 1 : 0;
    }
}

}  // namespace nsr
