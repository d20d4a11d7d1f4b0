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
//   * dW = dy^T x contracts over points: the forward saves every layer's input as [16 points][16 channels] slot tiles, which
//     read as dwords q * 64 + lane ARE the operand of a contraction over points (nsr_bwd2.h: the dW kernel).
// References: Renderer.render_batch_ray (src/utils/Renderer.py:63-198), eval_points (:23-61),
// NICE/MLP/MLP_no_xyz forward (src/conv_onet/models/decoder.py:168-203,254-274,312-342),
// raw2outputs_nerf_color (src/common.py:204-245), ATen grid_sampler_3d (GridSampler.h).
#pragma once
#include "nsr_dev.h"
#include "nsr_layout.h"
#include "../../include/nsr.h"

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
    const unsigned char *gmask;   // backward, optional: [Z][Y][X] bytes, 0 = nobody consumes this voxel's gradient (nsr_render_args.grad_voxel_mask)
    int Z, Y, X;
    double lo[3];
    double ext[3];      // hi - lo (the divisor of normalize_3d_coordinate, common.py:281-283)
    double inv[3];      // 1 / (hi - lo): only scales the coordinate GRADIENT (d p = d u * (n-1)/2 * 2/(hi-lo))
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
    float *acts;              // saved decoder activations (optional, see ActSink): the backward then loads h_i / relu masks
    int acts_masks_only;      // bit 0: only the relu masks are saved (no parameter gradients will be asked for); bit 1 + p: the same for decoder pass p alone
    long long n_points_total; // n_rays * S
    long long act_tiles;      // 16-point tiles of the sample-point list (n_points_total rounded up): tile stride of `acts` / `dy`
    // backward only
    const double *d_depth, *d_var, *g_depth;
    const double *g_scale;    // optional device scalar multiplying d_depth / d_var / d_rgb (nsr_bwd_args.grad_scale)
    const float *d_rgb;
    float *d_rays_o, *d_rays_d;
    float *partials;          // [passes][gridDim.x][max param count]
    int partial_stride;       // floats between two blocks' partial images
    // fused mapping loss (Mapper.py:487-493), optional
    const float *loss_depth;  // [N] sensor depth of the loss (also set where the SAMPLING is unguided: coarse mapper, Mapper.py:484-489)
    const float *gt_color;    // [N][3]
    const unsigned char *keep; // [N] bounding-box mask of the callers' pre-filter, or NULL (all rays count)
    double *loss;             // forward: += sum over rays of the loss terms
    double *dl_depth;         // forward: d loss / d depth per ray  [N]      (handed to the backward as d_depth / d_rgb)
    float *dl_rgb;            // forward: d loss / d rgb per ray    [N][3]
    float w_color;            // weight of the colour term (colour stage only)
    long long *dbg;           // profiling stamps (NSR_TS builds only), else NULL
    // split backward (nsr_bwd2.h): scratch behind the saved activations in `acts`
    float *dy;                // [passes][tiles][kDySlots][16][16]  dY_i of every decoder pass (same slot form as `acts`)
    float *draw;              // [n_points][4]  d raw per sample (compositor backward)
    double *pd;               // [n_points][4]  sample position (fp64) and depth: px, py, pz, z
    float *pf;                // [n_points][4]  the position rounded to fp32 (embedding argument)
    float *dbpart;            // [passes][dx blocks][288]  per-block partial sums of d embedder._B
    int dw_beg[4];            // dW kernel: blocks [dw_beg[p], dw_beg[p + 1]) = partial images of decoder pass p
    int dx_beg[4];            // dX kernel: blocks [dx_beg[p], dx_beg[p + 1]) work on decoder pass p (dealt by the passes' tile cost)
    int draw_scaled;          // 1: `draw` already carries nsr_bwd_args.grad_scale (comp_bwd_kernel ran); 0: the forward wrote it
    int xflags;               // measurement switches (NSR_X environment variable; 0 in normal operation)
    int skip_masked;          // 1 (with keep): rays with keep == 0 are not part of the batch (nsr_render_args.skip_masked)
    int lds_grid_floats;      // > 0 (coarse stage): the gradient grid (this many floats) is accumulated in the dX block's LDS
    float hot_z[4];           // dX kernel, per grid: samples with z below it use the block's hot-voxel table (0: no table)
    int hot_slots;            // dX kernel: slots of that table (the launch sizes it to the LDS the block has left)
    unsigned s_magic;         // ceil(2^32 / S), 0 for S = 1: floor(x / S) = (x * s_magic) >> 32 for every x < 2^25 (S <= 64), see ray_of_point
    int pass_beg[4];          // pass kernels (nsr_fwd2.h): blocks [pass_beg[p], pass_beg[p + 1]) of the launch work on decoder pass p
    // eval_points only
    const double *points;
    long long n_points;
    float *out_points;
};

// Does 16-point tile `tile` of the sample list hold a sample of a ray that is part of the batch?  (nsr_render_args.skip_masked:
// rays the callers' bounding-box pre-filter rejected are removed, like the reference's compaction, src/Mapper.py:471-481.)
// Wave-uniform; the mask bytes come through the scalar cache.  Forward passes, dX, dW and the compositor apply the same test.
NSR_DEV bool ray_live(const RenderParams &P, long long ray) { return !P.skip_masked || uniform_load_u8(P.keep + ray) != 0u; }
// floor(x / S) for a sample-point index x < 2^25: one multiplication by s_magic = ceil(2^32 / S) (exact while x e < 2^32, e = S s_magic - 2^32
// < S <= 64); S = 1 (s_magic = 2^32 does not fit the field and is passed as 0): the index itself
NSR_DEV unsigned ray_of_point(const RenderParams &P, unsigned x) {
    return P.s_magic ? (unsigned)(((unsigned long long)x * P.s_magic) >> 32) : x;
}
NSR_DEV bool tile_live(const RenderParams &P, long long tile) {
    if (!P.skip_masked) return true;
    // (32-bit: a call holds fewer than 2^25 sample points, nsr_api.cpp; the 64-bit form was two software divisions, ~700 scalar
    //  instructions per claimed tile)
    // Round 6: scalar throughout -- the tile index is pinned to an SGPR and the two divisions by S are multiplications by ceil(2^32 / S)
    // (exact for x < 2^26 when S <= 64: x e < 2^32 with e = S ceil(2^32 / S) - 2^32 < S; a call holds < 2^25 points); as per-lane 32-bit divisions they were ~30
    // vector instructions per claimed tile in each of the pass, dX and dW kernels.
    const unsigned p0 = (unsigned)uniform((int)tile) * kTile, np = (unsigned)P.n_points_total, pe = p0 + kTile < np ? p0 + kTile : np;
    const unsigned r0 = ray_of_point(P, p0), r1 = ray_of_point(P, pe - 1);
    bool live = false;
    for (unsigned r = r0; r <= r1; ++r) live = live || uniform_load_u8(P.keep + r) != 0u;
    return live;
}

// ------------------------------------------------------------------------------------------------
// parameter packing: flat blob -> [aux table | MFMA operand stream]  (what the render kernels copy into LDS, verbatim)
//   aux table (AUX_FLOATS): biases, fc_c biases, output layer, Fourier matrix in the order the kernels read them
//   stream, four k-steps per lane contiguous (one 16-byte read each):
//   stream[m.pk + ((T*2 + Tp)*64 + lane)*4 + r] = W[16*Tp + (lane&15)][kbeg + 16T + 4(lane>>4) + r]
// ------------------------------------------------------------------------------------------------
template <int KIND>
NSR_DEV float aux_value(const float *__restrict__ flat, int idx) {
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
    return v;
}

template <int KIND>
NSR_KERNEL void pack_kernel(const float *__restrict__ flat, float *__restrict__ packed) {
    const int gidx = bid_x() * nthreads() + tid();
    if (gidx >= packed_buf_total(KIND)) return;
    if (gidx < AUX_FLOATS) { packed[gidx] = aux_value<KIND>(flat, gidx); return; }
    const int idx = gidx - AUX_FLOATS;
    float v = 0.f;
    if (idx < packed_total(KIND)) {
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
    } else {
        // transposed stream (nsr_layout.h): T[((Tk*2 + To)*64 + lane)*4 + r] = W[16 To + 4 (lane>>4) + r][kbeg + 16 Tk + (lane&15)]
        const int idt = idx - packed_total(KIND);
#pragma unroll
        for (int id = 0; id < nmatT_of(KIND); ++id) {
            const Mat m = matT_of(KIND, id);
            const int rel = idt - m.pk;
            if (rel >= 0 && rel < m.nt * 512) {
                const int r = rel & 3, lane = (rel >> 2) & 63, To = (rel >> 8) & 1, Tk = rel >> 9;
                const int o = 16 * To + 4 * (lane >> 4) + r;
                const int k = 16 * Tk + (lane & 15);
                if (k < m.kcols) v = flat[m.off + o * m.stride + m.kbeg + k];
            }
        }
    }
    packed[gidx] = v;
}

// cooperative global -> LDS copy of NF4 16-byte words; B loads in flight per thread before the first store (a fresh block's
// first loads take several thousand cycles: the fewer dependent rounds, the shorter its prologue)
template <int NF4, int B = 4>
NSR_DEV void copy_f4(float *dst, const float *__restrict__ src) {
    const int nt = nthreads();
    int t = tid();
    for (; t + (B - 1) * nt < NF4; t += B * nt) {
        F4 v[B];
#pragma unroll
        for (int k = 0; k < B; ++k) v[k] = ld4(src + 4 * (t + k * nt));
#pragma unroll
        for (int k = 0; k < B; ++k) st4(dst + 4 * (t + k * nt), v[k]);
    }
    for (; t < NF4; t += nt) st4(dst + 4 * t, ld4(src + 4 * t));
}
// the same as global -> LDS DMA (no register round trip, every piece of the block in flight at once); the caller waits (dma_wait<0>)
// before its barrier.  A wave instruction moves the 1 KB [64 lanes][16 bytes] piece that starts at its first lane's element.
template <int NF4>
NSR_DEV void copy_f4_dma(float *dst, const float *__restrict__ src) {
    static_assert(NF4 % 64 == 0, "whole 1 KB pieces");
    const int nt = nthreads(), lane = tid() & 63;
    for (int t = tid(); t < NF4; t += nt) dma16(src + 4 * t, dst + 4 * (t - lane), lane);
}
// stage the small per-decoder tables into LDS (all threads of the block); `packed` = the decoder's packed buffer
template <int KIND>
NSR_DEV void load_aux(float *aux, const float *__restrict__ packed) { copy_f4<AUX_FLOATS / 4>(aux, packed); }

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
    int par;            // parity of the base corner's voxel coordinates: (x0 & 1) | (y0 & 1) << 1 | (z0 & 1) << 2
};

// a / b for a divisor whose correctly rounded reciprocal `rb` is known: one residual correction gives the correctly
// rounded quotient (Markstein), i.e. the value of the IEEE division the reference performs, for 3 instead of ~35 operations
NSR_DEV double div_by(double a, double b, double rb) {
    const double q = a * rb;
    return fma(fma(-q, b, a), rb, q);
}

NSR_DEV void axis_setup(double p, double lo, double ext, double inv, int n, int &i0, float &w0, float &w1, float &mult) {
    const double gn = div_by(p - lo, ext, inv) * 2.0 - 1.0;      // the reference's fp64 operations in its order, then one rounding
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
    axis_setup(px, G.lo[0], G.ext[0], G.inv[0], G.X, x0, L.gx, L.fx, L.mx);
    axis_setup(py, G.lo[1], G.ext[1], G.inv[1], G.Y, y0, L.gy, L.fy, L.my);
    axis_setup(pz, G.lo[2], G.ext[2], G.inv[2], G.Z, z0, L.gz, L.fz, L.mz);
    L.vox = (z0 * G.Y + y0) * G.X + x0;
    L.par = (x0 & 1) | ((y0 & 1) << 1) | ((z0 & 1) << 2);
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

// the same in two halves, so that the 16 loads can be issued long before their values are needed
struct GatherRaw { F4 a[8], b[8]; };
NSR_DEV void gather_issue(GatherRaw &R, const GridDev &G, const Lvl &L, int g) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float *src = G.feat + (long long)corner_vox(L, k) * kC + 4 * g;
        R.a[k] = ld4(src);
        R.b[k] = ld4(src + 16);
    }
}
NSR_DEV Act<2> gather_finish(const GatherRaw &R, const Lvl &L) {
    Act<2> c;
    c.t[0] = f4zero();
    c.t[1] = f4zero();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = corner_w(L, k);
        const F4 a = R.a[k], b = R.b[k];
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
// "lane = channel": each half-wave owns one PARITY CLASS of voxels ((x & 1) | (y & 1) << 1 | (z & 1) << 2: the eight corners
// of a cell fall into the eight classes, corner k into class k ^ parity(base corner)) and walks the tile's 16 points
// (consecutive samples of a ray, i.e. spatially sorted), summing weighted dc while the class's voxel stays the same and
// issuing ONE 32-lane atomic (two full 64-byte lines) per run of equal voxels.  A voxel keeps its class from cell to cell, so
// the corners two consecutive cells share (4 of 8 when the ray steps one cell along one axis) meet in the same walk and
// merge -- walking by corner INDEX, as the first two generations did, only merged samples inside one cell (613 k instead
// of 396 k voxel updates per 1000 colour-stage rays of the bench scene).
//   Tx  : [16][kTxS] floats  dc of the tile, point-major
//   tab : [16][8] ints (voxel of the class or -1) followed by [16][8] floats (its weight)
// Hot voxels.  The rays of a keyframe all leave from its camera centre: the few voxels around it take an update from every ray
// (~2 000 per 1000-ray batch of the bench scene), and memory-side atomics on one line serialise -- a third of the dX kernel's
// time in the middle stage (profiles/r03_dx_experiments.txt, the run with the voxels spread by a hash).  A dX block therefore
// keeps a small direct-mapped table of voxel rows in LDS: updates of samples closer to their ray's origin than `hot_z` (two
// cells) go there when their voxel owns or can claim its slot, everything else straight to memory, and the block adds its
// table to memory once at the end.
constexpr int kHotBit = 1 << 30, kHotRow = kC + 1;        // a slot: one tag + one 32-channel row; the slot count is the launch's (RenderParams.hot_slots)
// The table is addressed by its OFFSET from the block's LDS base (floats), a wave-uniform integer (-1: no table): a struct of
// generic pointers handed down to the walk was spilled to scratch by the register allocator and re-loaded inside the hot
// branch -- scratch loads count on vmcnt, so every probe waited for ALL of the wave's outstanding grid atomics.
struct HotTab { int off, slots; };                                // tags [slots] | rows [slots][32]
NSR_DEV int *hot_tag(const HotTab &H) { return reinterpret_cast<int *>(lds_base()) + H.off; }
NSR_DEV float *hot_val(const HotTab &H) { return reinterpret_cast<float *>(lds_base()) + H.off + H.slots; }
NSR_DEV int hot_slot(const HotTab &H, int vox) {                  // multiplicative hash, scaled to [0, slots) by the high half of a product
    return (int)(((unsigned long long)((unsigned)vox * 2654435761u) * (unsigned)H.slots) >> 32);
}
NSR_DEV void hot_init(const HotTab &H) {
    for (int i = tid(); i < H.slots * kC; i += nthreads()) hot_val(H)[i] = 0.f;
    for (int i = tid(); i < H.slots; i += nthreads()) hot_tag(H)[i] = -1;
}
NSR_DEV void hot_flush(const HotTab &H, const GridDev &G) {       // after a block barrier: one half wave per occupied slot
    const int ch = tid() & 31;
    for (int s = tid() >> 5; s < H.slots; s += nthreads() >> 5) {
        const int v = hot_tag(H)[s];
        if (v >= 0) atomic_add_global(G.dfeat + (long long)v * kC + ch, hot_val(H)[s * kC + ch]);
    }
}

// part 1 (the wave that owns the tile): dc point-major into Tx, the class table into tab.
// Round 6: the table carries the walk's DECISIONS, made here once per (point, class) by the lane that owns the entry -- two entries per
// lane, all 128 of a tile in parallel -- instead of being re-derived by every one of the walk's 64 sequential steps:
//   vt[p][cls] >= 0  the run of this class ENDS at point p and is to be added: bits 0..24 = the voxel, or, with kHotBit, bits 0..23 =
//                    the float offset of the voxel's row in the block's hot table (the slot is found or claimed HERE);
//             <  0   nothing to emit at this point (inside a run, or the corner is inactive / masked);
//   wt[p][cls]       the corner's weight, 0 for an inactive / masked corner.  The walk needs no "run continues" flag: it restarts its sum
//                    behind every emission, and what lies between two emitted runs adds exact zeros.
// Before: per step a neighbour comparison for "same", one for "end", a sign test and their conjunction, per emission the hot-bit test,
// a multiplicative hash, the tag read, a compare-and-swap on a free slot and two LDS round trips in the dependent chain -- ~11
// instructions per step and 11 / 35 per plain / hot emission against 6 and 9 / 9 now (the kernel's length follows its instruction count).
// `live`: bit c = the lane's corner 2 g + c takes part (consumed-gradient mask, GridDev.gmask; 3 = both)
NSR_DEV void scatter_stage(const Lvl &L, int lane, const Act<2> &dc, bool active, float *Tx, float *tab, HotTab hot = HotTab{-1, 0},
                           bool hot_pt = false, unsigned live = 3u) {
    const int pt = lane & 15, g = lane >> 4;
    int *vt = reinterpret_cast<int *>(tab);
    int *wt = vt + 128;
    tx_store(Tx, dc, pt, g);
    int raw[2], cls[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int k = 2 * g + c;
        cls[c] = k ^ L.par;
        raw[c] = (active && ((live >> c) & 1u)) ? corner_vox(L, k) : -1;
        vt[pt * 8 + cls[c]] = raw[c];
    }
    wave_fence();
    int next[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) next[c] = vt[(pt < 15 ? pt + 1 : pt) * 8 + cls[c]];
    wave_fence();                         // (every lane has its neighbour before any entry is rewritten)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bool emit = raw[c] >= 0 && (pt == 15 || raw[c] != next[c]);
        int word = emit ? raw[c] : -1;
        if (emit && hot.off >= 0 && hot_pt) {
            const int slot = hot_slot(hot, raw[c]);
            int tg = lds_load_i(hot_tag(hot) + slot);
            if (tg == -1) { const int old = atomic_cas_lds_i(hot_tag(hot) + slot, -1, raw[c]); tg = old == -1 ? raw[c] : old; }
            if (tg == raw[c]) word = kHotBit | (slot * kC);
        }
        vt[pt * 8 + cls[c]] = word;
        wt[pt * 8 + cls[c]] = raw[c] >= 0 ? __builtin_bit_cast(int, corner_w(L, 2 * g + c)) : 0;
    }
}
// part 2 (after a wave fence): the walks and the atomics.  Lane = (class of the pair 2 q + h, channel); per step one fma and the
// emission test; the only divergent code is the emission itself (an LDS atomic on a hot row or one memory atomic).
NSR_DEV void scatter_walk(const GridDev &G, int lane, const float *Tx, const float *tab,
                          int lds_grid = -1,           // >= 0: the whole gradient grid sits in LDS at this offset (floats; small grids, nsr_bwd2.h)
                          HotTab hot = HotTab{-1, 0}) {
    // (h, ch and what derives from them are formed per ROUND from an opaque copy of the lane index: hoisted out of the tile loop they are
    // registers the kernel does not have -- at a tighter register cap the allocator spilled them and re-loaded them from scratch at every
    // emission, i.e. behind `s_waitcnt vmcnt(0)`: a wait for all of the wave's outstanding atomics)
    // the table's LDS offset (floats) as ONE opaque per-lane register: the staging regions sit beyond the 64 KB an LDS instruction's
    // immediate offset reaches, and with the region's constant folded into every read's literal the compiler formed each of the 32
    // addresses of a round with a v_add of its own (128 per tile); off an opaque base the reads carry p * 32 bytes as their immediates
    const int tab_off = (int)(tab - reinterpret_cast<const float *>(lds_base()));
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int ln = opaque_i(lane), h = ln >> 5, ch = ln & 31;
        const int hv_off = hot.off + hot.slots + ch;          // this lane's channel of hot row 0 (floats from the LDS base)
        const int *vk = reinterpret_cast<const int *>(lds_base()) + opaque_i(tab_off + 2 * q + h);
        int v[16], w[16];
        float x[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {        // all LDS reads of the round in flight at once
            v[p] = vk[p * 8];
            w[p] = vk[p * 8 + 128];
            x[p] = Tx[p * kTxS + ch];
        }
        sched_fence();                        // (the scheduler would sink each dc read to its use: one LDS round trip per step)
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            // (the same fma chain per run as a sequential walk; a run's first term is fma(x, w, 0) = x * w)
            s = fmaf(x[p], __builtin_bit_cast(float, w[p]), s);
            if (__builtin_expect(v[p] >= 0, 0)) {      // (the emission out of line: a step that emits nothing -- 60 % of them -- takes no branch)
                if (lds_grid >= 0) atomic_add_lds(reinterpret_cast<float *>(lds_base()) + lds_grid + v[p] * kC + ch, s);
                else if (v[p] & kHotBit) atomic_add_lds(reinterpret_cast<float *>(lds_base()) + hv_off + (v[p] & (kHotBit - 1)), s);
                else {
#if defined(NSR_X_SCATTER_STORE)         // A/B build: plain stores to the same addresses (what the vector-memory path costs without the atomic)
                    G.dfeat[(long long)v[p] * kC + ch] = s;
#elif defined(NSR_X_SCATTER_HALF)        // A/B build: one of the two 64-byte lines of every voxel row
                    if (ch < 16) atomic_add_global_off(G.dfeat, ((unsigned)v[p] << 7) + ((unsigned)ch << 2), s);
#else
                    atomic_add_global_off(G.dfeat, ((unsigned)v[p] << 7) + ((unsigned)ch << 2), s);      // (a grid is < 2^25 voxels = 4 GB: nsr_api.cpp)
#endif
                }
                s = 0.f;                          // the next run of this half's class starts here
            }
        }
    }
}
// the consumed-gradient mask bytes of the lane's two corners (requested early, consumed by scatter_stage)
NSR_DEV unsigned gmask_bits(const GridDev &G, const Lvl &L, int g) {
    unsigned live = 3u;
    if (G.gmask) live = (G.gmask[corner_vox(L, 2 * g)] != 0 ? 1u : 0u) | (G.gmask[corner_vox(L, 2 * g + 1)] != 0 ? 2u : 0u);
    return live;
}
NSR_DEV void scatter_merged(const GridDev &G, const Lvl &L, int lane, const Act<2> &dc, bool active, float *Tx, float *tab,
                            int lds_grid = -1, HotTab hot = HotTab{-1, 0}, bool hot_pt = false, unsigned live = 3u) {
    scatter_stage(L, lane, dc, active, Tx, tab, hot, hot_pt, live);
    wave_fence();
    scatter_walk(G, lane, Tx, tab, lds_grid, hot);
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
    sched_fence_gemv();
}

// cooperative copy of a decoder's packed operand stream into LDS (caller provides the barriers)
template <int KIND>
NSR_DEV void load_packed(float *wl, const float *__restrict__ packed) { copy_f4<packed_total(KIND) / 4>(wl, packed + AUX_FLOATS); }

NSR_DEV float red_g(float v) { v += shfl_xor(v, 16); v += shfl_xor(v, 32); return v; }

// sin / cos for Fourier arguments |x| up to a few thousand: two-constant Cody-Waite reduction by pi (exact products
// through fma) to r in [-pi/2, pi/2], one odd minimax polynomial of degree 9 there (3.3e-9 in exact arithmetic, ~1.5e-7
// evaluated in fp32: the same class as libm sinf), sign from the parity of the quotient; branch-free, 13 VALU operations.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
NSR_DEV f32x4 splat(float v) { f32x4 r = {v, v, v, v}; return r; }
NSR_DEV f32x4 vfma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
// Four arguments at a time: every step is a <4 x float> operation, i.e. two packed-fp32 instructions (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32), half the VALU issue slots of the scalar form.
// `t`: the float whose bit 0 is the parity of the quotient (the cosine passes its complement).  The sign is applied as a FACTOR to r --
// sgn = as_float(0x3f800000 | t << 31) is one v_lshl_or_b32 per value, r * sgn one packed multiply per pair, and fma(p r^2, r sgn, r sgn) =
// sgn (r + r^3 p) to the bit -- where shifting the parity up and xor-ing it into the result was two to three unpacked integer
// instructions per value (round 6: vector instructions exclude the other waves' MFMAs).
NSR_DEV f32x4 sin_poly4(f32x4 r, u32x4 t) {
    const f32x4 sgn = __builtin_bit_cast(f32x4, (t << 31) | 0x3f800000u);
    const f32x4 r2 = r * r, rs = r * sgn;
    f32x4 p = vfma(r2, splat(2.59048850e-06f), splat(-1.98008978e-04f));
    p = vfma(p, r2, splat(8.33289982e-03f));
    p = vfma(p, r2, splat(-1.66666476e-01f));
    return vfma(p * r2, rs, rs);
}
// quotient by the "1.5 * 2^23" trick: t = x/pi + 12582912 has round(x/pi) in its low mantissa bits (|x| < 1e7), so
// its parity is bit 0 of the float and no float->int conversion / compare / select is needed for the sign
#if defined(NSR_X_LIBM_SIN)
// CPU emulator A/B only (tests/perf/parity_causes.py): correctly rounded sines -- how much of a parity miss is the kernels' own sine?
// NSR_X_LIBM_SIN = 1: every sine / cosine; 2: the forward's embedding only; 3: the dW kernel's re-evaluation only; 4: the dX kernel's cosines only
}  // namespace nsr
#include <cmath>
namespace nsr {
NSR_DEV f32x4 sin_libm4(f32x4 x) { f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)std::sin((double)x[i]); return r; }
NSR_DEV f32x4 cos_libm4(f32x4 x) { f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)std::cos((double)x[i]); return r; }
#endif
NSR_DEV f32x4 sin_acc4(f32x4 x) {
    const f32x4 t = vfma(x, splat(0.318309886183790672f), splat(12582912.f));
    const f32x4 k = t - splat(12582912.f);
    f32x4 r = vfma(k, splat(-3.14159274101257324f), x);
    r = vfma(k, splat(8.74227765734758577e-08f), r);
    return sin_poly4(r, __builtin_bit_cast(u32x4, t));
}
// cos(x) = -(-1)^k sin(r) with x = (k + 1/2) pi + r
NSR_DEV f32x4 cos_acc4(f32x4 x) {
    const f32x4 t = vfma(x, splat(0.318309886183790672f), splat(-0.5f)) + splat(12582912.f);   // (12582912 - 0.5 is not representable)
    const f32x4 k = t - splat(12582912.f);
    f32x4 r = vfma(k, splat(-3.14159274101257324f), x);
    r = vfma(k, splat(8.74227765734758577e-08f), r);
    r = (r - splat(1.57079637050628662f)) + splat(4.37113882867379289e-08f);
    return sin_poly4(r, ~__builtin_bit_cast(u32x4, t));
}
// the three places a Fourier feature is evaluated: the forward's embedding, the dW kernel's re-evaluation of it, the dX kernel's cosines
#if defined(NSR_X_LIBM_SIN)
NSR_DEV f32x4 sin_fwd4(f32x4 x) { return (NSR_X_LIBM_SIN == 1 || NSR_X_LIBM_SIN == 2) ? sin_libm4(x) : sin_acc4(x); }
NSR_DEV f32x4 sin_dw4(f32x4 x) { return (NSR_X_LIBM_SIN == 1 || NSR_X_LIBM_SIN == 3) ? sin_libm4(x) : sin_acc4(x); }
NSR_DEV f32x4 cos_dx4(f32x4 x) { return (NSR_X_LIBM_SIN == 1 || NSR_X_LIBM_SIN == 4) ? cos_libm4(x) : cos_acc4(x); }
#else
// The FORWARD's embedding sine in fp64 (round 6).  At ScanNet-size bounds (|p.B| ~ 1e3 rad) every gradient tensor of the path is a
// heavily cancelling sum: the fp32 reference itself sits 1e-2 from an fp64 evaluation, and matching IT to 1e-4 means matching the
// rounding of its per-element values.  Measured with these sources under the CPU emulator (tests/perf/parity_causes.py ->
// profiles/r06_parity_causes.json, scannet/fine, 5000 rays): with the packed fp32 sine above (abs err 1.5e-7, i.e. one ulp off a
// correctly rounded sine for a good share of the arguments) 31 of 50 gradient tensors miss 1e-4 (max 1.82e-4 -- the GPU's numbers);
// with a correctly rounded sine in the forward's embedding ALONE none does (max 7.9e-5), and the dW kernel's re-evaluation and the
// dX kernel's cosines do not matter at all.  fp64 vector operations issue at the unpacked fp32 rate on gfx950: ~16 operations per
// sine instead of ~6, in the forward's pass kernel only.  x -> k = round(x / pi) through the 1.5 * 2^52 trick (parity = sign),
// r = x - k pi (one fma: |k| < 2^21, the product's error k * 1.2e-16 is far below fp32), odd minimax polynomial of degree 11 on
// [-pi/2, pi/2] (max error 1.7e-11 abs = 3e-4 ulp of the fp32 result: 0.02 % of random arguments round differently from
// libm's double sine rounded once), ONE rounding to fp32.
NSR_DEV float sin_f64(float x) {
    const double xd = (double)x;
    const double t = fma(xd, 0.31830988618379067, 6755399441055744.0);
    const double k = t - 6755399441055744.0;
    const double r = fma(k, -3.141592653589793, xd);
    const double r2 = r * r;
    double p = fma(r2, -2.38466908183069e-08, 2.7522618644441596e-06);       // minimax on [0, (pi/2)^2] of (sin r - r) / r^3 in u = r^2
    p = fma(p, r2, -0.00019840804034260232);                                 // (weighted by r^3: max |error| of the sine 1.7e-11,
    p = fma(p, r2, 0.008333330495622904);                                    //  tests/test_oracle_modes.py pins it; one fma fewer and
    p = fma(p, r2, -0.16666666606465366);                                    //  3x fewer misrounded results than the degree-13 Taylor form)
    const float s = (float)fma(p * r2, r, r);
    const unsigned sign = (unsigned)__builtin_bit_cast(unsigned long long, t) << 31;
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) ^ sign);
}
NSR_DEV f32x4 sin_fwd4(f32x4 x) {
#if defined(NSR_X_FWD_SIN_F32)               // A/B build (tools/build_ts.sh): the packed fp32 sine in the forward, as until round 5 -- what the fp64 one costs
    return sin_acc4(x);
#else
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = sin_f64(x[i]);
    return r;
#endif
}
NSR_DEV f32x4 sin_dw4(f32x4 x) { return sin_acc4(x); }
NSR_DEV f32x4 cos_dx4(f32x4 x) { return cos_acc4(x); }
#endif
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

NSR_DEV float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

NSR_DEV unsigned relu_mask(f32x4 (&acc)[2]) {
    unsigned m = 0;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // max(x, 0) on the bit pattern (relu1), then "the result is not zero" as min(bits, 1) shifted into the mask: three instructions per
            // value (v_max_i32, v_min_u32, v_lshl_or_b32) where compare + two selects + or were four to five (round 6: vector instructions
            // exclude the other waves' MFMAs, profiles/r06_experiments.txt item 8)
            const float h = relu1(acc[T][r]);
            const unsigned hb = __builtin_bit_cast(unsigned, h);
            m |= (hb < 1u ? hb : 1u) << (T * 4 + r);
            acc[T][r] = h;
        }
    return m;
}
NSR_DEV void relu_plain(f32x4 (&acc)[2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[T][r] = relu1(acc[T][r]);
}
// d where bit `base + j` of `word` is set (j = 4 T + r), else 0: the bit sign-extended to a full mask (one v_bfe_i32) and one v_and per
// value -- the test / compare / select form was three
NSR_DEV Act<2> apply_mask(const Act<2> &d, unsigned word, int base) {
    Act<2> o;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = ((int)(word << (31 - (base + T * 4 + r)))) >> 31;
            const float f = d.t[T][r];          // (a bit_cast straight off a vector ELEMENT reads element 0 with this compiler)
            o.t[T][r] = __builtin_bit_cast(float, __builtin_bit_cast(int, f) & m);
        }
    return o;
}

NSR_DEV void embed(Act<kET> &e, const float *aux, float px, float py, float pz, int g) {
#pragma unroll
    for (int T = 0; T < kET; ++T) {
        const B4 b = load_b4(aux, 4 * T + g);
        const f32x4 arg = vfma(splat(pz), b.z, vfma(splat(py), b.y, splat(px) * b.x));     // decoder.py:29
        e.t[T] = sin_fwd4(arg);                                                            // decoder.py:30
        sched_fence_emb();                  // bound the ILP the scheduler extracts from 24 independent sines
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

// Saved activations (288 GB of HBM buy the backward its forward re-run): per decoder pass p (middle 0, fine 1, colour 2; the
// coarse decoder uses pass 0) and 16-point tile T of the flat sample-point list, 13 slots of 1 KB:
//     acts + ((p * n_tiles + T) * kActSlots + j) * 256 + ((gp & 15) * 4 + g) * 4        (T = gp >> 4)
//     j = 2 i + k: hidden state h_i (what the next layer reads), channels 16 k .. 16 k + 15;  j = 10 + k: the decoder's own
//     grid features c;  j = 12: the five relu masks of the lane
// A slot is a row-major [16 points][16 channels] matrix: a wave's 64 lanes (pt, g) write 1 KB contiguous per slot whatever
// the tile / ray-group geometry of the kernel, and the 12 KB the dW kernel (nsr_bwd2.h) streams per tile and decoder are
// ONE contiguous span (with slot-major arrays the same bytes were 12 scattered 1 KB pieces, 3 MB apart: 3 TB/s instead of
// the HBM rate).
constexpr int kActSlots = 13;
constexpr int kActC = 10, kActMask = 12;
struct ActSink {
    float *p;                // slot 0 of this lane, NULL for a lane without a point
    bool full;               // hidden states and features too (false: relu masks only)
    static constexpr long long stride = 256;        // floats between two slots
};
NSR_DEV ActSink act_sink(const RenderParams &P, int pass, long long gp, int g) {
    ActSink a;
    a.full = !(P.acts_masks_only & (1 | (2 << pass)));
    a.p = (P.acts && gp >= 0) ? P.acts + (((long long)pass * P.act_tiles + (gp >> 4)) * kActSlots) * 256 + ((gp & 15) * 4 + g) * 4 : nullptr;
    return a;
}
NSR_DEV int act_pass(int kind) { return kind == NSR_COARSE ? 0 : kind - NSR_MIDDLE; }

// MLP (decoder.py:177-203): h_i = relu(W_i x_i + b_i) + (U_i c + v_i), x_3 = [e | h_2]
// `save` (forward kernel only): non-NULL = write h_i and the relu masks to the lane's activation slots
template <int KIND, bool KEEP, bool SAVE = false>
NSR_DEV void mlp_xyz_fwd(const float *pk, const float *aux, float px, float py, float pz,
                         const Act<cdim_of(KIND) / 16> &c, int lane, float (&out)[nout_of(KIND)], Kept<KIND> *kept,
                         const ActSink *save = nullptr) {
    constexpr int CD = cdim_of(KIND), NOUT = nout_of(KIND), NTC = CD / 16;
    const int g = lane >> 4;
    Act<kET> e;
    embed(e, aux, px, py, pz, g);
    Act<2> h;
    act_zero(h);
    unsigned mpack0 = 0, mpack1 = 0;
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
        if (KEEP || SAVE) m = relu_mask(acc); else relu_plain(acc);
        acc[0] += to_v(ld4(aux + AUX_V + i * 32 + 4 * g));
        acc[1] += to_v(ld4(aux + AUX_V + i * 32 + 16 + 4 * g));
        gemv_fwd<NTC>(acc, c, pk + xyz_mat(CD, i == 0 ? XU0 : (i == 1 ? XU1 : (i == 2 ? XU2 : (i == 3 ? XU3 : XU4)))).pk, lane);
        h.t[0] = acc[0];
        h.t[1] = acc[1];
        if (KEEP) { kept->h[i] = h; kept->mask[i] = m; }
        if (SAVE) {
            if (save->p && save->full) { st4(save->p + (2 * i) * save->stride, to_F4(h.t[0])); st4(save->p + (2 * i + 1) * save->stride, to_F4(h.t[1])); }
            if (i < 4) mpack0 |= m << (8 * i); else mpack1 = m;
        }
    }
    if (SAVE && save->p) {
        st4(save->p + kActMask * save->stride, F4{__builtin_bit_cast(float, mpack0), __builtin_bit_cast(float, mpack1), 0.f, 0.f});
        if (save->full) {
            st4(save->p + kActC * save->stride, to_F4(c.t[0]));
            st4(save->p + (kActC + 1) * save->stride, to_F4(c.t[1]));
        }
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
template <bool KEEP, bool SAVE = false>
NSR_DEV void mlp_nox_fwd(const float *pk, const float *aux, const Act<2> &c, int lane, float (&out)[1], Kept<0> *kept,
                         const ActSink *save = nullptr) {
    const int g = lane >> 4;
    Act<2> h = c;
    unsigned mpack0 = 0, mpack1 = 0;
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
        if (KEEP || SAVE) m = relu_mask(acc); else relu_plain(acc);
        h.t[0] = acc[0];
        h.t[1] = acc[1];
        if (KEEP) { kept->h[i] = h; kept->mask[i] = m; }
        if (SAVE) {
            if (save->p && save->full) { st4(save->p + (2 * i) * save->stride, to_F4(h.t[0])); st4(save->p + (2 * i + 1) * save->stride, to_F4(h.t[1])); }
            if (i < 4) mpack0 |= m << (8 * i); else mpack1 = m;
        }
    }
    if (SAVE && save->p) {
        st4(save->p + kActMask * save->stride, F4{__builtin_bit_cast(float, mpack0), __builtin_bit_cast(float, mpack1), 0.f, 0.f});
        if (save->full) {
            st4(save->p + kActC * save->stride, to_F4(c.t[0]));
            st4(save->p + (kActC + 1) * save->stride, to_F4(c.t[1]));
        }
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
        load_aux<NSR_COARSE>(aux, P.dec[NSR_COARSE].packed);
    } else {
        load_aux<NSR_MIDDLE>(aux, P.dec[NSR_MIDDLE].packed);
        if (STAGE >= NSR_STAGE_FINE) load_aux<NSR_FINE>(aux + AUX_FLOATS, P.dec[NSR_FINE].packed);
        if (STAGE == NSR_STAGE_COLOR) load_aux<NSR_COLOR>(aux + 2 * AUX_FLOATS, P.dec[NSR_COLOR].packed);
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
template <int STAGE, bool SAVE = false>
NSR_DEV F4 decode_tile_lds(const RenderParams &P, const float *aux, float *wl, double px, double py, double pz, int lane,
                           long long gp = -1) {      // gp: global index of the lane's sample point (saved activations), -1: none
    const int g = lane >> 4;
    F4 raw = F4{0.f, 0.f, 0.f, 0.f};
    if (STAGE == NSR_STAGE_COARSE) {
        const Lvl L = make_level(P.grid[NSR_COARSE], px, py, pz);
        const Act<2> c = gather_feat(P.grid[NSR_COARSE], L, g);
        float o[1];
        const ActSink sc = act_sink(P, 0, gp, g);
        mlp_nox_fwd<false, SAVE>(wl, aux, c, lane, o, nullptr, &sc);
        raw.w = o[0];
    } else {
        const float fx = (float)px, fy = (float)py, fz = (float)pz;     // decoder.py:189
        const Lvl Lm = make_level(P.grid[NSR_MIDDLE], px, py, pz);
        const Act<2> cm = gather_feat(P.grid[NSR_MIDDLE], Lm, g);
        float om[1];
        const ActSink sm = act_sink(P, 0, gp, g);
        const Dbg dbg{P.dbg ? P.dbg + ((long long)bid_x() * 12 + (tid() >> 6)) * 64 : nullptr};
        dbg.stamp(10);                                                  // features gathered (issued), before the middle decoder
        mlp_xyz_fwd<NSR_MIDDLE, false, SAVE>(wl, aux, fx, fy, fz, cm, lane, om, nullptr, &sm);
        dbg.stamp(3);
        float occ = om[0];
        if (STAGE >= NSR_STAGE_FINE) {
            const Lvl Lf = make_level(P.grid[NSR_FINE], px, py, pz);
            const Act<2> cf = gather_feat(P.grid[NSR_FINE], Lf, g);
            block_sync();                                               // everyone is done with the middle weights
            load_packed<NSR_FINE>(wl, P.dec[NSR_FINE].packed);
            block_sync();
            dbg.stamp(4);
            Act<4> cc;
            cc.t[0] = cf.t[0]; cc.t[1] = cf.t[1]; cc.t[2] = cm.t[0]; cc.t[3] = cm.t[1];    // decoder.py:182-187
            float of[1];
            const ActSink sf = act_sink(P, 1, gp, g);
            mlp_xyz_fwd<NSR_FINE, false, SAVE>(wl, aux + AUX_FLOATS, fx, fy, fz, cc, lane, of, nullptr, &sf);
            occ = of[0] + om[0];                                                            // decoder.py:333,341
            dbg.stamp(5);
        }
        if (STAGE == NSR_STAGE_COLOR) {
            const Lvl Lc = make_level(P.grid[NSR_COLOR], px, py, pz);
            const Act<2> ccol = gather_feat(P.grid[NSR_COLOR], Lc, g);
            block_sync();
            load_packed<NSR_COLOR>(wl, P.dec[NSR_COLOR].packed);
            block_sync();
            dbg.stamp(11);
            float oc[4];
            const ActSink sc = act_sink(P, 2, gp, g);
            mlp_xyz_fwd<NSR_COLOR, false, SAVE>(wl, aux + 2 * AUX_FLOATS, fx, fy, fz, ccol, lane, oc, nullptr, &sc);
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
template <int STAGE, bool SAVE = false>
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
    const Dbg dbg{P.dbg ? P.dbg + ((long long)bid_x() * 12 + wave) * 64 : nullptr};
    dbg.stamp(0);

    load_stage_aux<STAGE>(P, aux);
    if (STAGE == NSR_STAGE_COARSE) load_packed<NSR_COARSE>(wl, P.dec[NSR_COARSE].packed);     // only one decoder: staged once
    if (STAGE == NSR_STAGE_MIDDLE) load_packed<NSR_MIDDLE>(wl, P.dec[NSR_MIDDLE].packed);
    double loss_acc = 0.0;                         // lane 0 of every wave: loss terms of the rays it composited
    for (long long grp = bid_x(); grp < P.n_groups; grp += nblk_x()) {
        loop_fence();
        const long long ray0 = grp * P.rays_per_block;
        if (STAGE > NSR_STAGE_MIDDLE) load_packed<NSR_MIDDLE>(wl, P.dec[NSR_MIDDLE].packed);   // fine / colour overwrote it
        dbg.stamp(1);
        compute_z(P, ray0, ztmp, zbuf);            // ends with block_sync (also covers the aux / weight staging)
        dbg.stamp(2);
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
        F4 raw = decode_tile_lds<STAGE, SAVE>(P, aux, wl, px, py, pz, lane, active ? ray * S + k : -1);
        dbg.stamp(6);
        if (!inside) raw.w = 100.f;                                         // Renderer.py:57
        if (active && g == 0) {
            rawbuf[pidx] = raw;
            if (P.raw) st4(P.raw + (ray * S + k) * 4, raw);
            if (P.zvals) P.zvals[ray * S + k] = z;
        }
        block_sync();
        dbg.stamp(7);
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
            if (P.loss) {
                // Mapper.py:487-493 on the rays the pre-filter keeps, and its derivative w.r.t. this ray's outputs:
                // d|gt - depth| = sign(depth - gt) where gt > 0, w_color * sign(rgb - gt_rgb) in the colour stage
                // (every lane of the wave evaluates the same scalars; lane 0 publishes them)
                const bool kp = !P.keep || P.keep[rayq];
                const float gd = P.loss_depth ? P.loss_depth[rayq] : 0.f;
                double gD = 0.0, lterm = 0.0;
                float g3[3] = {0.f, 0.f, 0.f};
                if (kp && gd > 0.f) {
                    const double df = depth - (double)gd;
                    lterm += fabs(df);
                    gD = df > 0.0 ? 1.0 : (df < 0.0 ? -1.0 : 0.0);
                }
                if (kp && STAGE == NSR_STAGE_COLOR && P.gt_color) {
                    const float e[3] = {cr - P.gt_color[rayq * 3 + 0], cg - P.gt_color[rayq * 3 + 1], cb - P.gt_color[rayq * 3 + 2]};
                    lterm += (double)(P.w_color * ((fabsf(e[0]) + fabsf(e[1])) + fabsf(e[2])));
#pragma unroll
                    for (int q = 0; q < 3; ++q) g3[q] = e[q] > 0.f ? P.w_color : (e[q] < 0.f ? -P.w_color : 0.f);
                }
                if (lane == 0) {
                    loss_acc += lterm;
                    if (P.dl_depth) P.dl_depth[rayq] = gD;
                    if (P.dl_rgb) { P.dl_rgb[rayq * 3 + 0] = g3[0]; P.dl_rgb[rayq * 3 + 1] = g3[1]; P.dl_rgb[rayq * 3 + 2] = g3[2]; }
                }
                if (P.draw) {
                    // ... and, for the split backward (nsr_bwd2.h), what its compositor-backward kernel would compute from
                    // them: d raw per sample, the sample's position and bound test -- one launch fewer per iteration.  Same
                    // expressions as comp_bwd_kernel (d var = 0).
                    const double dzc = zq - depth;
                    const double s1 = wave_sum_d((double)c.w * dzc);
                    const float Gz = (float)(gD * zq + 0.0 * (dzc * dzc - 2.0 * s1 * zq));
                    const float Gw = Gz + fmaf(g3[2], rw.z, fmaf(g3[1], rw.y, g3[0] * rw.x));
                    float v = act ? Gw * c.w : 0.f;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const float o_ = shfl_down(v, d);
                        if (lane + d < 64) v += o_;
                    }
                    float suffix = shfl_down(v, 1);
                    if (lane == 63) suffix = 0.f;
                    const float dalpha = Gw * c.T - suffix / c.t;
                    float docc = 10.f * (dalpha * ((1.f - c.alpha) * c.alpha));
                    const double qx = (double)P.rays_o[rayq * 3 + 0] + (double)P.rays_d[rayq * 3 + 0] * zq;
                    const double qy = (double)P.rays_o[rayq * 3 + 1] + (double)P.rays_d[rayq * 3 + 1] * zq;
                    const double qz = (double)P.rays_o[rayq * 3 + 2] + (double)P.rays_d[rayq * 3 + 2] * zq;
                    const bool ins = (qx > P.blo[0]) && (qx < P.bhi[0]) && (qy > P.blo[1]) && (qy < P.bhi[1]) && (qz > P.blo[2]) && (qz < P.bhi[2]);
                    if (!ins) docc = 0.f;
                    if (act) {
                        const long long gq = rayq * S + lane;
                        st4(P.draw + gq * 4, F4{c.w * g3[0], c.w * g3[1], c.w * g3[2], docc});
                        double *pq = P.pd + gq * 4;
                        pq[0] = qx; pq[1] = qy; pq[2] = qz; pq[3] = zq;
                        st4(P.pf + gq * 4, F4{(float)qx, (float)qy, (float)qz, 0.f});
                    }
                }
            }
        }
        dbg.stamp(8);
        block_sync();
    }
    if (P.loss && lane == 0 && loss_acc != 0.0) atomic_add_global_d(P.loss, loss_acc);
    dbg.stamp(9);
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

// The same for all grids of a stage with the step counts on the device (capturable iterations): adam_tick_kernel bumps
// each grid's counter and forms the two bias-correction scalars in fp64 (what torch.optim.Adam does on the host); the update
// kernel takes the grid from blockIdx.y.
struct AdamMulti {
    float *p[4];
    float *g[4];
    float *m[4], *v[4];
    const unsigned char *mask[4];
    long long n_vox[4];
    int *step[4];
    float lr[4];
    int n;
    float b1, b2, eps;
    float omb1, omb2;         // 1 - beta (flat_adam_kernel: formed in fp64 by the host, like torch's python scalars)
    double b1d, b2d;          // the betas as the caller's doubles: the bias corrections 1 - beta^t are formed from THESE (with
                              // 0.999f instead of 0.999 the first hundreds of steps are 6e-6 off torch's step size)
    int zero_grad;
    float *scal;              // [4][2]: step_size, bias2_sqrt
};
NSR_KERNEL void adam_tick_kernel(const AdamMulti A) {
    const int i = tid();
    if (i >= A.n) return;
    const int t = A.step[i][0] + 1;
    A.step[i][0] = t;
    A.scal[2 * i + 0] = (float)((double)A.lr[i] / (1.0 - pow(A.b1d, (double)t)));
    A.scal[2 * i + 1] = (float)sqrt(1.0 - pow(A.b2d, (double)t));
}
NSR_KERNEL void masked_adam_multi_kernel(const AdamMulti A) {
    const int i = bid_y();
    const long long t = (long long)bid_x() * nthreads() + tid();
    const long long vox = t >> 3;
    if (vox >= A.n_vox[i]) return;
    if (A.mask[i] && A.mask[i][vox] == 0) return;
    const long long o = vox * kC + (t & 7) * 4;
    const F4 g = ld4(A.g[i] + o);
    F4 m = ld4(A.m[i] + o), v = ld4(A.v[i] + o), p = ld4(A.p[i] + o);
    const float step = A.scal[2 * i], rs2 = A.scal[2 * i + 1], omb1 = 1.f - A.b1, omb2 = 1.f - A.b2;
#define NSR_ADAM1(c)                                                    \
    m.c = m.c + omb1 * (g.c - m.c);                                      \
    v.c = v.c * A.b2 + (omb2 * g.c) * g.c;                               \
    p.c = p.c - step * (m.c / (sqrtf(v.c) / rs2 + A.eps));
    NSR_ADAM1(x) NSR_ADAM1(y) NSR_ADAM1(z) NSR_ADAM1(w)
#undef NSR_ADAM1
    st4(A.m[i] + o, m);
    st4(A.v[i] + o, v);
    st4(A.p[i] + o, p);
    if (A.zero_grad) st4(A.g[i] + o, F4{0.f, 0.f, 0.f, 0.f});
}

// Flat spans (decoder parameter blobs, pose tensors): the same arithmetic, one thread per element; span = blockIdx.y.  AdamMulti's
// n_vox holds the element count, mask is unused.
NSR_KERNEL void flat_adam_kernel(const AdamMulti A) {
    const int i = bid_y();
    const long long e = (long long)bid_x() * nthreads() + tid();
    if (e >= A.n_vox[i]) return;
    const float g = A.g[i][e];
    float m = A.m[i][e], v = A.v[i][e], p = A.p[i][e];
    const float step = A.scal[2 * i], rs2 = A.scal[2 * i + 1], omb1 = A.omb1, omb2 = A.omb2;
    m = m + omb1 * (g - m);
    v = v * A.b2 + (omb2 * g) * g;
    p = p - step * (m / (sqrtf(v) / rs2 + A.eps));
    A.m[i][e] = m;
    A.v[i][e] = v;
    A.p[i][e] = p;
    if (A.zero_grad) A.g[i][e] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU gradient exchange: gather the frustum-selected voxel rows (32 floats) of up to 4 grid gradients
// and up to 4 flat spans (decoder-gradient blob, pose gradients, loss) into ONE contiguous buffer for a single all-reduce,
// and scatter the sums back.  One thread per float.
// ------------------------------------------------------------------------------------------------
struct PackParams {
    float *grid[4];
    const long long *rows[4];
    long long n_rows[4];
    float *span[4];
    long long span_n[4];
    int n_grids, n_spans, unpack;
    float *packed;
    long long total;
};
NSR_KERNEL void pack_rows_kernel(const PackParams P) {
    long long t = (long long)bid_x() * nthreads() + tid();
    if (t >= P.total) return;
    const long long t0 = t;
    float *where = nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!where && i < P.n_grids) {
            const long long seg = P.n_rows[i] * kC;
            if (t < seg) where = P.grid[i] + P.rows[i][t / kC] * kC + (t % kC);
            else t -= seg;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!where && i < P.n_spans) {
            if (t < P.span_n[i]) where = P.span[i] + t;
            else t -= P.span_n[i];
        }
    }
    if (!where) return;
    if (P.unpack) *where = P.packed[t0]; else P.packed[t0] = *where;
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
// The mapper's sampling loop in one launch (Mapper.py:437-468: get_samples per keyframe of the window, torch.cat) together
// with its bounding-box pre-filter (:471-481) as a byte mask + the kept rays' maximum depth.  grid.y = frame.
// ------------------------------------------------------------------------------------------------
#define NSR_MAX_WINDOW 32
#define NSR_MAX_PEERS 15
// philox4x32-10 (Salmon et al., SC'11; the generator torch.randint runs on the device): 128-bit counter, 64-bit key
NSR_DEV unsigned philox_word(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

struct WindowParams {
    const long long *indices;          // [K][n]; NULL: the kernel draws them (rng) and writes them to indices_out
    long long *indices_out;
    unsigned long long *rng;           // [seed, calls so far, blocks done (internal), kept-max bits (internal)]: one uniform draw in
    unsigned crop_pixels;              //   [0, crop_h * crop_w) per ray from philox(counter = (ray, call), key = seed); the last block
    long long n;                       //   advances `calls`.  Also the launch's last-block hand-off when `hdr` is set (below)
    int K, H0, W0, crop_w, W_full;
    float fx, fy, cx, cy;
    const float *depth[NSR_MAX_WINDOW], *color[NSR_MAX_WINDOW], *c2w[NSR_MAX_WINDOW];
    int c2w_stride[NSR_MAX_WINDOW];
    float *rays_o, *rays_d, *out_depth, *out_color;      // [K*n] concatenated in frame order
    double lo[3], hi[3];
    unsigned char *keep;               // optional
    float *kept_max;                   // optional, caller-zeroed (ignored when hdr is set)
    // The fused iteration's zero fill inside this launch (nsr_get_samples_window_fused): x-blocks >= sample_bx of every grid row
    // store zeros over zero[0 .. zero_n) while the first sample_bx place the rays (the fill is bandwidth, the sampling a latency
    // chain: side by side they cost the longer of the two instead of two launches); the iteration's header {loss (fp64) = 0,
    // kept max, 0} is written by the LAST block to finish, from maxima the blocks accumulate in rng[3] -- nothing has to be
    // zeroed before the launch.
    float *hdr;                        // [4] or NULL
    float *zero;                       // 16-byte aligned
    long long zero_n;                  // floats
    int sample_bx;                     // x-blocks that sample; the rest (if any) fill
    // Multi-GPU without a collective for the batch-global depth cap (nsr_get_samples_window_sharded): every rank draws its pixels from
    // philox(counter = (ray, call), key = ITS seed) and all ranks make the same calls, so a rank can re-draw every other rank's pixels:
    // x-blocks [sample_bx, sample_bx * (1 + n_peers)) repeat the draw, the depth gather and the pre-filter test of peer p's rays with
    // peer_seed[p] -- no output, only their kept maximum joins the hand-off word.  The header's maximum is then the maximum over the
    // UNION batch, computed redundantly (and bit-identically) on every rank: Renderer.py:109,144 without an all-reduce.
    int n_peers;
    unsigned long long peer_seed[NSR_MAX_PEERS];
};

NSR_KERNEL void get_samples_window_kernel(const WindowParams P) {
    const int draw_bx = P.sample_bx * (1 + P.n_peers);       // x-blocks that draw rays: this rank's, then the peers' (maximum only)
    const int peer = bid_x() < draw_bx ? bid_x() / P.sample_bx : 0;      // 0: this rank's own rays
    const bool own = peer == 0;
    const long long i = (long long)(bid_x() - peer * P.sample_bx) * nthreads() + tid();
    const int k = bid_y();
    float vmax = 0.f;                  // this thread's candidate for the kept rays' maximum depth
    if (bid_x() >= draw_bx) {
        // a fill block: its share of the span, then done (it takes no part in the hand-off below: a thousand blocks counting
        // themselves on one word were 10 us of same-address atomics)
        const long long fb = nblk_x() - draw_bx, chunk = (long long)k * fb + (bid_x() - draw_bx), nchunk = fb * P.K;
        const long long n4 = P.zero_n >> 2;
        for (long long q = chunk * nthreads() + tid(); q < n4; q += nchunk * nthreads()) st4(P.zero + q * 4, F4{0.f, 0.f, 0.f, 0.f});
        if (chunk == 0 && tid() < (int)(P.zero_n & 3)) P.zero[n4 * 4 + tid()] = 0.f;
        return;
    }
    if (i < P.n) {
    const long long t = (long long)k * P.n + i;
    // the frame's pose is requested first, with the draw's state: behind the index store below the compiler may not move the loads
    float Rm[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float *R = P.c2w[k] + a * P.c2w_stride[k];
        Rm[a][0] = R[0]; Rm[a][1] = R[1]; Rm[a][2] = R[2]; Rm[a][3] = R[3];
    }
    long long idx;
    if (P.indices) {
        idx = P.indices[t];
    } else {
        const unsigned long long seed = own ? P.rng[0] : P.peer_seed[peer - 1], call = P.rng[1];
        const unsigned r = philox_word((unsigned)t, (unsigned)(t >> 32), (unsigned)call, (unsigned)(call >> 32), (unsigned)seed, (unsigned)(seed >> 32));
        idx = (long long)(((unsigned long long)r * P.crop_pixels) >> 32);                  // uniform up to 2^-32 * crop_pixels
        if (own) P.indices_out[t] = idx;
    }
    const int row = (int)(idx / P.crop_w) + P.H0, col = (int)(idx % P.crop_w) + P.W0;
    const long long pix = (long long)row * P.W_full + col;
    const float gd = P.depth[k][pix];
    if (own) {
        const float c0 = P.color[k][pix * 3 + 0], c1 = P.color[k][pix * 3 + 1], c2 = P.color[k][pix * 3 + 2];    // (all four in flight together)
        P.out_depth[t] = gd;
        P.out_color[t * 3 + 0] = c0;
        P.out_color[t * 3 + 1] = c1;
        P.out_color[t * 3 + 2] = c2;
    }
    const float dx = ((float)col - P.cx) / P.fx, dy = -(((float)row - P.cy) / P.fy), dzv = -1.f;
    double tb = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float *R = Rm[a];
        const float d = (dx * R[0] + dy * R[1]) + dzv * R[2];       // common.py:87: products, then left-to-right sum
        const float o = R[3];
        if (own) { P.rays_d[t * 3 + a] = d; P.rays_o[t * 3 + a] = o; }
        const double t0 = (P.lo[a] - (double)o) / (double)d, t1 = (P.hi[a] - (double)o) / (double)d;
        const double m = tmax(t0, t1);
        tb = (a == 0) ? m : tmin(tb, m);
    }
    const bool kp = tb >= (double)gd;
    if (P.keep && own) P.keep[t] = kp ? 1 : 0;
    if (kp && gd > 0.f) vmax = gd;
    }
    if (P.hdr) {
        // one returning atomic per wave (it has returned before the barrier below, i.e. before this block counts itself done)
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const float o = shfl_xor(vmax, m); vmax = o > vmax ? o : vmax; }
        if ((tid() & 63) == 0 && vmax > 0.f) keep_alive_u(atomic_fetch_max_u32(reinterpret_cast<unsigned *>(P.rng + 3), __builtin_bit_cast(unsigned, vmax)));
    } else if (P.kept_max && vmax > 0.f) {
        atomic_max_pos(P.kept_max, vmax);
    }
    if (!P.indices || P.hdr) {
        // the last block to get here (every thread of every block has read the call counter by then) advances it
        block_sync();
        if (tid() == 0) {
            const unsigned long long total = (unsigned long long)draw_bx * P.K;              // the sampling blocks (peers' included)
            if (atomic_fetch_add_global_u64(P.rng + 2, 1ull) == total - 1) {
                P.rng[2] = 0ull;
                if (!P.indices) P.rng[1] = P.rng[1] + 1ull;
                if (P.hdr) {
                    const unsigned mx = atomic_exchange_u32(reinterpret_cast<unsigned *>(P.rng + 3), 0u);
                    st4(P.hdr, F4{0.f, 0.f, __builtin_bit_cast(float, mx), 0.f});
                }
            }
        }
    }
}

// gradient of the window's poses from the ray gradients (autograd of common.py:74-88): per frame k
//   d c2w[k][a][j] = sum_i d_rays_d[i][a] * dirs[i][j] (j < 3),   d c2w[k][a][3] = sum_i d_rays_o[i][a]
// one block per frame, out [K][12] (rows 0..2 of the pose)
struct PoseGradParams {
    const long long *indices;
    long long n;
    int H0, W0, crop_w;
    float fx, fy, cx, cy;
    const float *d_rays_o, *d_rays_d;
    float *out;
    int out_stride;
};
NSR_KERNEL void pose_grad_kernel(const PoseGradParams P) {
    float *red = reinterpret_cast<float *>(lds_base());          // [12][nthreads]
    const int k = bid_x(), nt = nthreads();
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.f;
    for (long long i = tid(); i < P.n; i += nt) {
        const long long t = (long long)k * P.n + i;
        const long long idx = P.indices[t];
        const int row = (int)(idx / P.crop_w) + P.H0, col = (int)(idx % P.crop_w) + P.W0;
        const float dir[3] = {((float)col - P.cx) / P.fx, -(((float)row - P.cy) / P.fy), -1.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float gd = P.d_rays_d[t * 3 + a];
            acc[a * 4 + 0] = fmaf(gd, dir[0], acc[a * 4 + 0]);
            acc[a * 4 + 1] = fmaf(gd, dir[1], acc[a * 4 + 1]);
            acc[a * 4 + 2] = fmaf(gd, dir[2], acc[a * 4 + 2]);
            acc[a * 4 + 3] += P.d_rays_o[t * 3 + a];
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) red[q * nt + tid()] = acc[q];
    block_sync();
    if (tid() < 12) {
        float s = 0.f;
        for (int j = 0; j < nt; ++j) s += red[tid() * nt + j];
        P.out[k * P.out_stride + tid()] = s;
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

// ------------------------------------------------------------------------------------------------
// The tracker's loss on rendered outputs (Tracker.optimize_cam_in_batch, src/Tracker.py:108-124), one block:
//   tmp_i  = |gt_i - depth_i| / sqrt(var_i + 1e-10)                 (fp64; var is detached, :109)
//   mask_i = keep_i & (gt_i > 0) [& (tmp_i < 10 * median(tmp over the kept rays))   -- handle_dynamic, :110-112]
//   loss  += sum_mask tmp_i [+ w_color * sum_mask |gt_rgb_i - rgb_i|                -- use_color_in_tracking, :119-122]
// plus d loss / d depth and d loss / d rgb per ray, i.e. what autograd hands render_batch_ray's backward.  `keep` is the
// bounding-box pre-filter as a mask (the reference compacts the batch, :92-104; the median runs over the compacted batch,
// zero-depth rays included).  torch.median: the LOWER middle element, NaN if any element is NaN -- found here by an 8-pass
// radix select over the bit patterns of the non-negative doubles (they order like unsigned integers), no sort, any N.
// ------------------------------------------------------------------------------------------------
struct TrackLossParams {
    long long n;
    const float *gt_depth, *gt_color, *rgb;
    const unsigned char *keep;       // or NULL: every ray counts
    const double *depth, *var;
    int handle_dynamic, use_color;
    float w_color;
    double *loss;                    // += the loss
    double *dl_depth;                // [n]
    float *dl_rgb;                   // [n][3] (written when use_color)
};

NSR_DEV double track_tmp(const TrackLossParams &P, long long i) {
    return fabs((double)P.gt_depth[i] - P.depth[i]) / sqrt(P.var[i] + 1e-10);
}

// LDS: hist int[256] | ctl int[4] | pre u64[2] | red f64[nthreads] | keys u64[key_cap]   (key_cap >= n: the kept rays' tmp bit
// patterns are computed once and cached; key_cap == 0: recomputed in every pass)
NSR_KERNEL void tracking_loss_kernel(const TrackLossParams P, const int key_cap) {
    int *hist = reinterpret_cast<int *>(lds_base());
    int *ctl = hist + 256;                                        // n_kept, n_nan, k, pad
    unsigned long long *pre = reinterpret_cast<unsigned long long *>(ctl + 4);
    double *red = reinterpret_cast<double *>(pre + 2);            // [nthreads] loss partials
    const int t = tid(), nt = nthreads();
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(red + nt);
    const bool cached = key_cap > 0;
    constexpr unsigned long long kSkip = ~0ull;                    // not kept: above every finite / inf / NaN pattern of a non-negative double
    double thr = 0.0;
    bool use_thr = false;
    if (P.handle_dynamic && cached && P.n <= nt) {
        // The tracker's own batch (200 rays): one ray per thread, everything stays in registers -- one round of loads (the colour
        // term's included), one sqrt / divide, the rank of the thread's key by counting (two keys per 16-byte LDS read), then mask,
        // derivative and loss from the same registers.  Same expressions and the same summation order as the general path below.
        const long long i = t;
        const bool in = i < P.n;
        const bool k = in && (!P.keep || P.keep[i]);
        const float gd = in ? P.gt_depth[i] : 0.f;
        const double dep = in ? P.depth[i] : 0.0, vr = in ? P.var[i] : 0.0;
        float gc[3] = {0.f, 0.f, 0.f}, rc[3] = {0.f, 0.f, 0.f};
        if (P.use_color && in) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { gc[a] = P.gt_color[i * 3 + a]; rc[a] = P.rgb[i * 3 + a]; }
        }
        const double diff = (double)gd - dep, rs = sqrt(vr + 1e-10);
        const double v = fabs(diff) / rs;
        const unsigned long long ki = k ? __builtin_bit_cast(unsigned long long, v) : kSkip;
        if (t < 4) ctl[t] = 0;
        if (t == 0) pre[0] = 0ull;
        keys[t] = in ? ki : kSkip;                              // (key_cap >= n; the slots up to nt exist: see the launch)
        block_sync();
        if (k) atomic_add_lds_i(ctl + 0, 1);
        if (k && v != v) atomic_add_lds_i(ctl + 1, 1);
        int rank = 0;
        const int n2 = ((int)P.n + 1) & ~1;
        for (int j = 0; j < n2; j += 2) {
            const unsigned long long k0 = keys[j], k1 = keys[j + 1];
            rank += (k0 < ki || (k0 == ki && j < t)) ? 1 : 0;
            rank += (k1 < ki || (k1 == ki && j + 1 < t)) ? 1 : 0;
        }
        block_sync();
        const int n_kept = ctl[0], n_nan = ctl[1];
        thr = __builtin_nan("");
        if (n_kept > 0 && n_nan == 0) {
            if (k && rank == (n_kept - 1) / 2) pre[0] = ki;
            block_sync();
            thr = 10.0 * __builtin_bit_cast(double, pre[0]);
        }
        bool m = k && gd > 0.f;
        m = m && (v < thr);
        double part = 0.0, cpart = 0.0;
        if (in) {
            P.dl_depth[i] = m ? (diff > 0.0 ? -1.0 : (diff < 0.0 ? 1.0 : 0.0)) / rs : 0.0;
            if (m) part += v;
            if (P.use_color) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float dc = gc[a] - rc[a];
                    P.dl_rgb[i * 3 + a] = m ? (dc > 0.f ? -P.w_color : (dc < 0.f ? P.w_color : 0.f)) : 0.f;
                    if (m) cpart += (double)fabsf(dc);
                }
            }
        }
        const double mine = wave_sum_d(part + (double)P.w_color * cpart);
        if ((t & 63) == 0) red[t >> 6] = mine;
        block_sync();
        if (t == 0) {
            double s = 0.0;
            for (int w = 0; w < (nt >> 6); ++w) s += red[w];
            atomic_add_global_d(P.loss, s);
        }
        return;
    }
    if (P.handle_dynamic) {
        if (t < 4) ctl[t] = 0;
        if (t == 0) { pre[0] = 0ull; pre[1] = 0ull; }
        block_sync();
        int kept = 0, nan = 0;
        for (long long i = t; i < P.n; i += nt) {
            const bool k = !P.keep || P.keep[i];
            const double v = k ? track_tmp(P, i) : 0.0;
            if (cached) keys[i] = k ? __builtin_bit_cast(unsigned long long, v) : kSkip;
            kept += k ? 1 : 0;
            nan += (k && v != v) ? 1 : 0;
        }
        if (kept) atomic_add_lds_i(ctl + 0, kept);
        if (nan) atomic_add_lds_i(ctl + 1, nan);
        block_sync();
        const int n_kept = ctl[0], n_nan = ctl[1];
        use_thr = true;
        thr = __builtin_nan("");                                   // NaN median (or an empty batch): nothing passes `tmp < 10 * median`
        if (n_kept > 0 && n_nan == 0) {
            const int kth = (n_kept - 1) / 2;                      // torch.median: the lower middle element (0-based rank)
            if (cached && P.n <= 1024) {
                // small batch: every thread ranks its own key against all others (ties broken by index)
                for (long long i = t; i < P.n; i += nt) {
                    const unsigned long long ki = keys[i];
                    if (ki == kSkip) continue;
                    int rank = 0;
                    for (int j = 0; j < (int)P.n; ++j) {
                        const unsigned long long kj = keys[j];
                        rank += (kj < ki || (kj == ki && j < (int)i)) ? 1 : 0;
                    }
                    if (rank == kth) pre[0] = ki;
                }
                block_sync();
            } else {
                // radix select, 8 bits per pass from the top: histogram of the digit among the keys matching the prefix so far
                if (t == 0) ctl[2] = kth;
                for (int pass = 0; pass < 8; ++pass) {
                    const int shift = 56 - 8 * pass;
                    for (int b = t; b < 256; b += nt) hist[b] = 0;
                    block_sync();
                    const unsigned long long prefix = pre[0], pmask = pre[1];
                    const int k = ctl[2];
                    for (long long i = t; i < P.n; i += nt) {
                        unsigned long long key;
                        if (cached) { key = keys[i]; if (key == kSkip) continue; }
                        else { if (P.keep && !P.keep[i]) continue; key = __builtin_bit_cast(unsigned long long, track_tmp(P, i)); }
                        if ((key & pmask) == prefix) atomic_add_lds_i(hist + (int)((key >> shift) & 255ull), 1);
                    }
                    block_sync();
                    if (t < 256) {                                 // the digit whose bin holds rank k: bins below sum to <= k < including it
                        int below = 0;
                        for (int j = 0; j < t; ++j) below += hist[j];
                        const int mine = hist[t];
                        if (below <= k && k < below + mine) {
                            ctl[2] = k - below;
                            pre[0] = prefix | ((unsigned long long)t << shift);
                            pre[1] = pmask | (255ull << shift);
                        }
                    }
                    block_sync();
                }
            }
            thr = 10.0 * __builtin_bit_cast(double, pre[0]);
        }
    }
    double part = 0.0, cpart = 0.0;
    for (long long i = t; i < P.n; i += nt) {
        const float gd = P.gt_depth[i];
        const double diff = (double)gd - P.depth[i], rs = sqrt(P.var[i] + 1e-10);
        const double v = (cached && i < key_cap && (!P.keep || P.keep[i])) ? __builtin_bit_cast(double, keys[i]) : fabs(diff) / rs;
        bool m = (!P.keep || P.keep[i]) && gd > 0.f;
        if (use_thr) m = m && (v < thr);
        // d |x| = sign(x) with sign(0) = 0 (torch.abs backward)
        P.dl_depth[i] = m ? (diff > 0.0 ? -1.0 : (diff < 0.0 ? 1.0 : 0.0)) / rs : 0.0;
        if (m) part += v;
        if (P.use_color) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float dc = P.gt_color[i * 3 + a] - P.rgb[i * 3 + a];
                P.dl_rgb[i * 3 + a] = m ? (dc > 0.f ? -P.w_color : (dc < 0.f ? P.w_color : 0.f)) : 0.f;
                if (m) cpart += (double)fabsf(dc);
            }
        }
    }
    const double mine = wave_sum_d(part + (double)P.w_color * cpart);      // wave sums, then one LDS round over the waves
    if ((t & 63) == 0) red[t >> 6] = mine;
    block_sync();
    if (t == 0) {
        double s = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) s += red[w];
        atomic_add_global_d(P.loss, s);
    }
}

// ------------------------------------------------------------------------------------------------
// get_camera_from_tensor / quad2rotation (src/common.py:137-176): [quaternion (w, x, y, z) | translation] -> 3x4 [R | T],
// R = I - two_s * (...) with two_s = 2 / |q|^2 (no normalisation of q: the optimiser moves all four components).  One
// thread per camera; `d_rt` != NULL: the backward (d cam from d [R | T]) instead.  Replaces ~25 ATen launches forward and
// ~45 backward per tracking / BA iteration (Tracker.py:87, Mapper.py:447-451).
// ------------------------------------------------------------------------------------------------
struct CamParams {
    const float *cam;     // [B][7]
    long long n;
    float *rt;            // forward:  [B][3][4]
    const float *d_rt;    // backward: [B][3][4]
    float *d_cam;         // backward: [B][7]
};
NSR_KERNEL void camera_from_tensor_kernel(const CamParams P) {
    const long long b = (long long)bid_x() * nthreads() + tid();
    if (b >= P.n) return;
    const float *c = P.cam + b * 7;
    const float qr = c[0], qi = c[1], qj = c[2], qk = c[3];
    const float nn = ((qr * qr + qi * qi) + qj * qj) + qk * qk;
    const float s = 2.0f / nn;
    // R = [[1 - s*a00, s*a01, s*a02], ...] with the bracket terms of common.py:152-160
    const float a[9] = {qj * qj + qk * qk, qi * qj - qk * qr, qi * qk + qj * qr,
                        qi * qj + qk * qr, qi * qi + qk * qk, qj * qk - qi * qr,
                        qi * qk - qj * qr, qj * qk + qi * qr, qi * qi + qj * qj};
    if (!P.d_rt) {
        float *o = P.rt + b * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int k = 0; k < 3; ++k) o[r * 4 + k] = (r == k) ? 1.0f - s * a[r * 3 + k] : s * a[r * 3 + k];
            o[r * 4 + 3] = c[4 + r];
        }
        return;
    }
    const float *g = P.d_rt + b * 12;
    // d R_rk = sgn_rk * (ds * a_rk + s * da_rk),  sgn = -1 on the diagonal, +1 off it;  ds/dq_m = -s^2 q_m / ... = -(s*s/2)*2 q_m / 2
    float ga[9], gs = 0.f;                                       // gradient w.r.t. the bracket terms and w.r.t. s
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sg = (r == k) ? -g[r * 4 + k] : g[r * 4 + k];
            ga[r * 3 + k] = sg * s;
            gs += sg * a[r * 3 + k];
        }
    // s = 2 / nn  ->  ds/dq_m = -2 / nn^2 * 2 q_m = -(s / nn) * 2 q_m
    const float gn = -gs * s / nn;                               // gradient w.r.t. nn
    float dqr = gn * 2.f * qr, dqi = gn * 2.f * qi, dqj = gn * 2.f * qj, dqk = gn * 2.f * qk;
    // a00 = qj^2 + qk^2
    dqj += ga[0] * 2.f * qj; dqk += ga[0] * 2.f * qk;
    // a01 = qi qj - qk qr ; a10 = qi qj + qk qr
    dqi += (ga[1] + ga[3]) * qj; dqj += (ga[1] + ga[3]) * qi; dqk += (ga[3] - ga[1]) * qr; dqr += (ga[3] - ga[1]) * qk;
    // a02 = qi qk + qj qr ; a20 = qi qk - qj qr
    dqi += (ga[2] + ga[6]) * qk; dqk += (ga[2] + ga[6]) * qi; dqj += (ga[2] - ga[6]) * qr; dqr += (ga[2] - ga[6]) * qj;
    // a11 = qi^2 + qk^2
    dqi += ga[4] * 2.f * qi; dqk += ga[4] * 2.f * qk;
    // a12 = qj qk - qi qr ; a21 = qj qk + qi qr
    dqj += (ga[5] + ga[7]) * qk; dqk += (ga[5] + ga[7]) * qj; dqi += (ga[7] - ga[5]) * qr; dqr += (ga[7] - ga[5]) * qi;
    // a22 = qi^2 + qj^2
    dqi += ga[8] * 2.f * qi; dqj += ga[8] * 2.f * qj;
    float *o = P.d_cam + b * 7;
    o[0] = dqr; o[1] = dqi; o[2] = dqj; o[3] = dqk;
    o[4] = g[3]; o[5] = g[7]; o[6] = g[11];
}

}  // namespace nsr

#include "nsr_bwd2.h"
#include "nsr_fwd2.h"
