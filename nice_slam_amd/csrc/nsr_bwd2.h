// nsr_bwd2.h -- split backward of the render path over saved activations (third generation), included by nsr_kernels.h.
//
// The second-generation kernel (nsr_bwd.h, removed in round 4) kept every parameter-gradient accumulator of a decoder in the
// registers of ONE wave, which pinned it to one wave per SIMD with spills.  Over the decoders' activations that the forward saves
// (nsr_render_args.acts: hidden states, relu masks, grid features), the backward is four launches instead:
//   comp_bwd_kernel        one wave per ray: d raw per sample from the output gradients (compositor backward), the sample's
//                          fp64 position and its bound test                                  (common.py:231-244 differentiated)
//   render_bwd_dx_kernel   "dX": per 16-point tile dh = W^T dY through the five layers (masks from the forward), dc += U^T dh,
//                          embedding backward, grid scatter, ray gradients; writes dY_i for the dW kernel.  No parameter-gradient
//                          accumulators (only the 18 partial sums of d embedder._B): forward-like register budget, 3 waves/SIMD.
//                          W^T operands come from a second, transposed operand stream (nsr_layout.h) so that one conflict-free
//                          16-byte LDS read feeds four MFMAs (the second generation read W^T through scalar LDS reads with
//                          4-way bank conflicts).
//   render_bwd_dw_kernel   "dW": split-K GEMM over the sample points, dW_i = sum_p dY_i[p]^T x_i[p].  Two loader waves per
//                          block stream 16-point tiles of every operand (dY_i, h_i, c, positions, d raw) through a six-slot LDS
//                          ring with global->LDS DMA (no staging registers); the slot layout of the saved activations IS the
//                          MFMA operand layout of a contraction over points; 8 compute waves own disjoint 16x16 output tiles
//                          (<= 10 each) and run free of barriers (LDS flag words), so there is no cross-wave reduction;
//                          one partial image of the flat gradient blob per block.
//                          The fc_c weights are not contracted directly: dH_i = W_{i+1}^T dY_{i+1}, hence
//                          dU_i = W_{i+1}^T (sum_p dY_{i+1}[p]^T c[p]) =: W_{i+1}^T G_{i+1}  (and dv_i = W_{i+1}^T db_{i+1}):
//                          the kernel accumulates G_j (stored in the image where dU_{j-1} lives), the finalize kernel applies
//                          the 32x32 matrices -- 224 instead of 240 MFMAs per tile, and no dH in memory.
//   bwd_finalize_kernel    sums the partial images (and the dX kernel's d _B partials), applies W^T to G / db, writes dparams.
// References: autograd of Renderer.render_batch_ray (src/utils/Renderer.py:63-198), MLP / MLP_no_xyz (src/conv_onet/models/
// decoder.py:177-203, 262-274), src/Mapper.py:503, src/Tracker.py:125.
#pragma once

namespace nsr {

constexpr int kDySlots = 10;                  // dY_i k-tile T at slot 2 i + T (same [n_points][16] form as the activation slots)
constexpr int kDxMaxWaves = 12;
// per-wave LDS staging of the dX kernel (floats): Tx[16][kTxS] | tab[256] (grid scatter) | P[3][16] | DP[3][16]
constexpr int kDxTx = 0, kDxTab = kTile * kTxS, kDxP = kDxTab + 256, kDxDP = kDxP + 48, kDxStg = kDxDP + 48;
static_assert(kDxStg % 4 == 0, "staging regions must stay 16-byte aligned");
constexpr int kDbPart = 288;                  // d _B partial image of a dX block: [3][96]
// ------------------------------------------------------------------------------------------------
// compositor backward, one wave per ray (lane = sample): d raw, fp64 sample position, bound test
// ------------------------------------------------------------------------------------------------
NSR_KERNEL void comp_bwd_kernel(const RenderParams P) {
    const int lane = tid() & 63, wave = tid() >> 6, nw = nthreads() >> 6;
    const long long ray = (long long)bid_x() * nw + wave;
    if (ray >= P.n_rays) return;
    const int S = P.S;
    if (!ray_live(P, ray)) {             // removed by the pre-filter: its samples inside a tile shared with a kept ray carry no gradient
        if (lane < S) st4(P.draw + (ray * S + lane) * 4, F4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const bool act = lane < S;
    const long long gp = ray * S + (act ? lane : 0);
    const F4 rw = act ? ld4(P.raw + gp * 4) : F4{0.f, 0.f, 0.f, 0.f};
    const double sc = P.g_scale ? P.g_scale[0] : 1.0;            // incoming gradient of a fused loss node
    const double gD = P.d_depth ? P.d_depth[ray] * sc : 0.0, gV = P.d_var ? P.d_var[ray] * sc : 0.0, dep = P.g_depth[ray];
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (P.d_rgb) { gr = P.d_rgb[ray * 3 + 0] * (float)sc; gg = P.d_rgb[ray * 3 + 1] * (float)sc; gb = P.d_rgb[ray * 3 + 2] * (float)sc; }
    const double z = act ? P.zvals[gp] : 0.0;
    const Comp cw = comp_weights(rw.w, act, lane);
    const double dz = z - dep;
    const double s1 = wave_sum_d((double)cw.w * dz);
    const float Gz = (float)(gD * z + gV * (dz * dz - 2.0 * s1 * z));
    const float Gw = Gz + fmaf(gb, rw.z, fmaf(gg, rw.y, gr * rw.x));
    float v = act ? Gw * cw.w : 0.f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = shfl_down(v, d);
        if (lane + d < 64) v += o;
    }
    float suffix = shfl_down(v, 1);
    if (lane == 63) suffix = 0.f;
    const float dalpha = Gw * cw.T - suffix / cw.t;
    float docc = 10.f * (dalpha * ((1.f - cw.alpha) * cw.alpha));
    // pts = o + d z in fp64 (Renderer.py:172-174); outside the un-enlarged bound the occupancy is overridden and its
    // gradient cut (Renderer.py:57)
    const double px = (double)P.rays_o[ray * 3 + 0] + (double)P.rays_d[ray * 3 + 0] * z;
    const double py = (double)P.rays_o[ray * 3 + 1] + (double)P.rays_d[ray * 3 + 1] * z;
    const double pz = (double)P.rays_o[ray * 3 + 2] + (double)P.rays_d[ray * 3 + 2] * z;
    const bool inside = (px > P.blo[0]) && (px < P.bhi[0]) && (py > P.blo[1]) && (py < P.bhi[1]) && (pz > P.blo[2]) && (pz < P.bhi[2]);
    if (!inside) docc = 0.f;
    if (act) {
        st4(P.draw + gp * 4, F4{cw.w * gr, cw.w * gg, cw.w * gb, docc});
        double *q = P.pd + gp * 4;
        q[0] = px; q[1] = py; q[2] = pz; q[3] = z;
        st4(P.pf + gp * 4, F4{(float)px, (float)py, (float)pz, 0.f});       // float32(p): what the embedding sees (decoder.py:189)
    }
}

// ------------------------------------------------------------------------------------------------
// dX kernel
// ------------------------------------------------------------------------------------------------
// dx[Tk] += W(slice)^T dy: A = transposed stream (one 16-byte LDS read per (Tk, To) = four k-steps), B = the dy registers;
// result CL (lane (pt, g) holds input channels 16 Tk + 4 g + r).  k-steps alternate between the NTK accumulators.
template <int NTK>
NSR_DEV void gemv_t(f32x4 (&dx)[NTK], const Act<2> &dy, const float *wt, int lane) {
#pragma unroll
    for (int To = 0; To < 2; ++To) {
        f32x4 a[NTK];
#pragma unroll
        for (int Tk = 0; Tk < NTK; ++Tk) a[Tk] = to_v(ld4(wt + ((Tk * 2 + To) * 64 + lane) * 4));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int Tk = 0; Tk < NTK; ++Tk) dx[Tk] = mfma16(a[Tk][r], dy.t[To][r], dx[Tk]);
    }
    sched_fence_gemv();
}

// What a tile's layers need first: d raw and the relu masks.  Requested one tile ahead and handed over RAW -- anything computed on
// a loaded value at request time (zeroing the rows beyond the last point, the incoming-gradient scale) makes the compiler wait for
// the load right behind its issue, i.e. at the top of the previous tile instead of a whole tile later.
struct DxIn {
    F4 dr;
    unsigned m0, m1;
};
NSR_DEV DxIn dx_load(const RenderParams &P, const float *acts_pass, long long tile, int pt, int g) {
    DxIn I;
    const long long gp = tile * kTile + pt;
    const long long q = gp < P.n_points_total ? gp : 0;
    I.dr = ld4(P.draw + q * 4);
    const float *mp = acts_pass + ((q >> 4) * kActSlots + kActMask) * 256 + ((q & 15) * 4 + g) * 4;
    I.m0 = __builtin_bit_cast(unsigned, mp[0]);
    I.m1 = __builtin_bit_cast(unsigned, mp[1]);
    return I;
}
NSR_DEV void dx_keep(const DxIn &I) {           // force the loads behind `I` to have landed (see dx_pass)
    keep_alive(I.dr.w); keep_alive(__builtin_bit_cast(float, I.m1));
}

// block `bi` of the `nbp` blocks of this decoder pass; `gb`: the block's index in the launch (d _B partial image, stamp slots)
template <int KIND, bool PARAMS, bool RAYS>
NSR_DEV void dx_pass(const RenderParams &P, int bi, int nbp, int gb) {
    constexpr bool XYZ = KIND != NSR_COARSE;
    constexpr int NOUT = nout_of(KIND);
    char *lds = lds_base();
    const int lane = tid() & 63, wave = uniform(tid() >> 6), nw = nthreads() >> 6;
    const int pt = lane & 15, g = lane >> 4;
    float *aux = reinterpret_cast<float *>(lds);
    float *wt = aux + AUX_FLOATS;                            // transposed operand stream of this decoder
    float *stg = wt + packedT_total(KIND);
    float *Sw = stg + wave * kDxStg;
    // A grid small enough for LDS (the coarse grid: 616 voxels = 79 KB at Replica) takes every block's contributions there
    // (LDS atomics) and goes to memory once per block: all 32 k samples of a 1024-ray batch hit those 616 voxels, and
    // same-line memory-side atomics serialise (coarse dX kernel 67 -> see profiles/r03*_c0).
    const int tail_off = uniform((int)(stg + nw * kDxStg - reinterpret_cast<float *>(lds))); // what follows the waves' staging (floats from the LDS base)
    const int gl_off = (KIND == NSR_COARSE && P.lds_grid_floats > 0) ? tail_off : -1;
    float *gl = gl_off >= 0 ? reinterpret_cast<float *>(lds) + gl_off : nullptr;
    const bool use_hot = KIND != NSR_COARSE && P.hot_z[KIND] > 0.f && P.grid[KIND].dfeat != nullptr;
    const HotTab hot{use_hot ? tail_off : -1, P.hot_slots};
    // Tiles: block i of the n of a pass owns the contiguous range [T i / n, T (i + 1) / n) and its waves draw from it through
    // an LDS counter (a wave whose tile was cheap takes the next one: no rounds); NSR_X bit 7: the static deal tile = block *
    // waves + wave, + blocks * waves, ... of the first version (measurement).
    const bool dyn = !(P.xflags & 128);
    int *tcnt = reinterpret_cast<int *>(stg + nw * kDxStg + (KIND == NSR_COARSE ? P.lds_grid_floats : P.hot_slots * kHotRow));
    const GridDev &G = P.grid[KIND];
    const DecDev &D = P.dec[KIND];
    const bool do_grid = G.dfeat != nullptr;
    if (!do_grid && !PARAMS && !RAYS) return;
    const Dbg dbg{P.dbg ? P.dbg + ((long long)gb * kDxMaxWaves + wave) * 64 : nullptr};
    dbg.stamp(0);
    constexpr long long sstride = 256;                       // floats between two slots of a tile
    const float *acts_pass = P.acts + (long long)act_pass(KIND) * P.act_tiles * kActSlots * 256;
    const long long ntiles = (P.n_points_total + kTile - 1) / kTile;
    const long long t0 = dyn ? ntiles * bi / nbp : 0, tend = dyn ? ntiles * (bi + 1) / nbp : ntiles;
    // A wave's FIRST tile is dealt statically (range start + wave; the counter starts behind them) and its inputs are requested here,
    // in front of the operand staging: they land under the 64 KB copy instead of being waited for behind the barrier (round 6: ~2 us
    // of every launch).  A first tile without a ray of the batch (pre-filter) falls back to the counter.
    long long tile = dyn ? t0 + wave : (long long)bi * nw + wave;
    const bool pre = tile < tend && (!dyn || tile_live(P, tile));
    DxIn cur;
    if (pre) cur = dx_load(P, acts_pass, tile, pt, g);
    copy_f4<AUX_FLOATS / 4>(aux, D.packed);
#if defined(NSR_X_DX_STAGE_COPY)             // A/B build: the transposed stream through registers (rounds 2-5)
    copy_f4<packedT_total(KIND) / 4>(wt, D.packed + AUX_FLOATS + packed_total(KIND));
#else
    copy_f4_dma<packedT_total(KIND) / 4>(wt, D.packed + AUX_FLOATS + packed_total(KIND));
#endif
    if (gl) for (int i = tid(); i < P.lds_grid_floats; i += nthreads()) gl[i] = 0.f;
    if (use_hot) hot_init(hot);
    if (tid() == 0) tcnt[0] = nw;
    dma_wait<0>();
    block_sync();
    dbg.stamp(1);
    float *dys = P.dy + (long long)act_pass(KIND) * P.act_tiles * kDySlots * 256;
    const long long tstep = (long long)nbp * nw;
    const bool need_dc = do_grid || RAYS;
    float aB[kET][3];                                        // d _B partial sums of lane (channel j, point group)
#pragma unroll
    for (int k = 0; k < kET; ++k) { aB[k][0] = 0.f; aB[k][1] = 0.f; aB[k][2] = 0.f; }

    auto claim = [&]() -> long long {                        // the block's next unclaimed tile that holds a ray of the batch (wave-uniform)
        for (;;) {
            int k = 0;
            if (lane == 0) k = atomic_fetch_add_lds_i(tcnt, 1);
            const long long t = t0 + shfl_i(k, 0);
            if (t >= tend || tile_live(P, t)) return t;
        }
    };
    // d raw written by the forward's loss epilogue: the incoming gradient applies here (one uniform scalar, read once)
    const float dr_scale = (!P.draw_scaled && P.g_scale) ? (float)P.g_scale[0] : 1.f;
    if (!pre && dyn) {
        tile = claim();
        if (tile < tend) cur = dx_load(P, acts_pass, tile, pt, g);
    }
    if (tile < tend) dx_keep(cur);                                                     // (waited for HERE: otherwise the loop header
    while (tile < tend) {                                                              //  carries an `s_waitcnt vmcnt(0)` for them, which
                                                                                       //  every later iteration spends on its atomics)
        loop_fence();
        // The vector-memory counter is in order: a load behind this wave's scatter atomics waits for all of them.  So the next
        // tile's first inputs (d raw, masks) are requested now and waited for (dx_keep) before this tile's atomics are issued;
        // this tile's positions are requested now as well and first used behind the layers, by when the previous tile's
        // atomics have drained.
        const long long nxt = dyn ? claim() : tile + tstep;
        const bool has_next = nxt < tend;
        DxIn nx = cur;
        if (has_next) nx = dx_load(P, acts_pass, nxt, pt, g);
        const long long gp = tile * kTile + pt;
        const bool active = gp < P.n_points_total;
        const double *pp = P.pd + (active ? gp : 0) * 4;
        const double cpx = pp[0], cpy = pp[1], cpz = pp[2], cz = pp[3];
        dbg.stamp(2);
        const unsigned mw0 = active ? cur.m0 : 0u, mw1 = active ? cur.m1 : 0u;       // relu masks: byte i of (m0, m1) = layer i
        const float dsc = active ? dr_scale : 0.f;           // rows beyond the last point carry no gradient
        float d_out[4] = {0.f, 0.f, 0.f, 0.f};
        if (NOUT == 1) d_out[0] = cur.dr.w * dsc;
        else { d_out[0] = cur.dr.x * dsc; d_out[1] = cur.dr.y * dsc; d_out[2] = cur.dr.z * dsc; }      // decoder.py:341 overwrites the 4th colour output
        // output layer
        Act<2> dh;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            f32x4 v = f4zero();
#pragma unroll
            for (int n = 0; n < (NOUT == 1 ? 1 : 3); ++n) {
                const F4 w = ld4(aux + AUX_WO + n * 32 + 16 * T + 4 * g);
                v[0] = fmaf(w.x, d_out[n], v[0]); v[1] = fmaf(w.y, d_out[n], v[1]);
                v[2] = fmaf(w.z, d_out[n], v[2]); v[3] = fmaf(w.w, d_out[n], v[3]);
            }
            dh.t[T] = v;
        }
        Act<2> dc, dY0, dY3;
        act_zero(dc); act_zero(dY0); act_zero(dY3);
        float *dyp = dys + tile * (kDySlots * 256) + (pt * 4 + g) * 4;
#pragma unroll
        for (int I = 4; I >= 0; --I) {
            if (XYZ && need_dc) gemv_t<2>(dc.t, dh, wt + xyzT_u(I), lane);         // gradient of (U_i c + v_i) is dh itself
            const Act<2> dY = apply_mask(dh, I < 4 ? mw0 : mw1, I < 4 ? 8 * I : 0);
            if (PARAMS && active && !(P.xflags & 2)) {
                st4(dyp + (2 * I) * sstride, to_F4(dY.t[0]));
                st4(dyp + (2 * I + 1) * sstride, to_F4(dY.t[1]));
            }
            if (I == 3) dY3 = dY;
            if (I == 0) dY0 = dY;
            if (!XYZ) {
                if (I == 3 && need_dc) gemv_t<2>(dc.t, dY, wt + nox_mat(NW3C).pk, lane);
                if (I == 0 && need_dc) gemv_t<2>(dc.t, dY, wt + nox_mat(NW0).pk, lane);
            }
            if (I > 0) {
                Act<2> nd;
                act_zero(nd);
                if (XYZ) gemv_t<2>(nd.t, dY, wt + xyzT_wh(I - 1), lane);
                else gemv_t<2>(nd.t, dY, wt + nox_mat(I == 1 ? NW1 : (I == 2 ? NW2 : (I == 3 ? NW3H : NW4))).pk, lane);
                dh = nd;
            }
        }
        // ---- embedding backward: dE = W0^T dY0 + W3e^T dY3, d arg = dE cos(arg); d p (rays) and d _B (parameters)
        dbg.stamp(3);
        const float px = (float)cpx, py = (float)cpy, pz = (float)cpz;       // decoder.py:189
        float dpe[3] = {0.f, 0.f, 0.f};
        if (XYZ && PARAMS && !(P.xflags & 4)) {
            // "lane = channel" form: swapping the MFMA operands (A = dY registers, B = transposed stream) yields
            // dE[point 4 g + r][channel 16 Tk + j] in lane (j, g) -- the layout the contraction over points for d _B needs
            if (g == 0) { Sw[kDxP + pt] = px; Sw[kDxP + 16 + pt] = py; Sw[kDxP + 32 + pt] = pz; }
            wave_fence();
            const f32x4 qx = to_v(ld4(Sw + kDxP + 4 * g)), qy = to_v(ld4(Sw + kDxP + 16 + 4 * g)), qz = to_v(ld4(Sw + kDxP + 32 + 4 * g));
            f32x4 sx = f4zero(), sy = f4zero(), sz = f4zero();
#pragma unroll
            for (int Tk = 0; Tk < kET; ++Tk) {
                f32x4 e0 = f4zero(), e3 = f4zero();
#pragma unroll
                for (int To = 0; To < 2; ++To) {
                    const f32x4 a0 = to_v(ld4(wt + xyzT_w0() + ((Tk * 2 + To) * 64 + lane) * 4));
                    const f32x4 a3 = to_v(ld4(wt + xyzT_w3e() + ((Tk * 2 + To) * 64 + lane) * 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        e0 = mfma16(dY0.t[To][r], a0[r], e0);
                        e3 = mfma16(dY3.t[To][r], a3[r], e3);
                    }
                }
                sched_fence_emb();
                const F4 b = load_b1(aux, 16 * Tk + pt);
                const f32x4 darg = (e0 + e3) * cos_dx4(vfma(qz, splat(b.z), vfma(qy, splat(b.y), qx * splat(b.x))));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    aB[Tk][0] = fmaf(darg[r], qx[r], aB[Tk][0]); aB[Tk][1] = fmaf(darg[r], qy[r], aB[Tk][1]);
                    aB[Tk][2] = fmaf(darg[r], qz[r], aB[Tk][2]);
                }
                if (RAYS) { sx = vfma(darg, splat(b.x), sx); sy = vfma(darg, splat(b.y), sy); sz = vfma(darg, splat(b.z), sz); }
            }
            if (RAYS) {
                // d p[point][d] = sum over channels: across the 16 lanes of the row, then over to the CL lane of the point
#pragma unroll
                for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sx[r] += shfl_xor(sx[r], m); sy[r] += shfl_xor(sy[r], m); sz[r] += shfl_xor(sz[r], m); }
                if (pt == 0) { st4(Sw + kDxDP + 4 * g, to_F4(sx)); st4(Sw + kDxDP + 16 + 4 * g, to_F4(sy)); st4(Sw + kDxDP + 32 + 4 * g, to_F4(sz)); }
                wave_fence();
                dpe[0] = Sw[kDxDP + pt]; dpe[1] = Sw[kDxDP + 16 + pt]; dpe[2] = Sw[kDxDP + 32 + pt];
            }
            wave_fence();
        } else if (XYZ && RAYS) {
            f32x4 dE[kET];
#pragma unroll
            for (int Tk = 0; Tk < kET; ++Tk) dE[Tk] = f4zero();
            gemv_t<kET>(dE, dY0, wt + xyzT_w0(), lane);
            gemv_t<kET>(dE, dY3, wt + xyzT_w3e(), lane);
            float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
            for (int Tk = 0; Tk < kET; ++Tk) {
                const B4 b = load_b4(aux, 4 * Tk + g);
                const f32x4 darg = dE[Tk] * cos_dx4(vfma(splat(pz), b.z, vfma(splat(py), b.y, splat(px) * b.x)));
#pragma unroll
                for (int r = 0; r < 4; ++r) { ax = fmaf(darg[r], b.x[r], ax); ay = fmaf(darg[r], b.y[r], ay); az = fmaf(darg[r], b.z[r], az); }
                sched_fence_emb();
            }
            dpe[0] = red_g(ax); dpe[1] = red_g(ay); dpe[2] = red_g(az);
        }
        // ---- grid: coordinate gradient, scatter; ray gradients
        dbg.stamp(4);
        Lvl L;
        if (need_dc) L = make_level(G, cpx, cpy, cpz);
        // consumed-gradient mask (opt-in, nsr_render_args.grad_voxel_mask): two byte loads per lane, requested here, first used by the
        // scatter's staging -- like every load of the iteration BEFORE this tile's atomics
        const unsigned live = (do_grid && G.gmask) ? gmask_bits(G, L, g) : 3u;
        float dux = 0.f, duy = 0.f, duz = 0.f;
        if (RAYS) coord_grad(G, L, g, dc, dux, duy, duz);
        dx_keep(nx);
        // (every load of this iteration is consumed on EVERY path before the atomics: a destination register still pending at the
        // loop header costs an `s_waitcnt vmcnt(0)` there, i.e. a wait for the whole tile's atomics)
        keep_alive_d(cpx); keep_alive_d(cpy); keep_alive_d(cpz); keep_alive_d(cz);
        dbg.stamp(5);
        if (do_grid && !(P.xflags & 1))
            scatter_merged(G, L, lane, dc, active, Sw + kDxTx, Sw + kDxTab, gl_off, hot, (float)cz < P.hot_z[KIND], live);
        dbg.stamp(6);
        if (RAYS) {
            // d p = d u * (n-1)/2 * 2/(hi-lo) (+ embedding part), fp64 like autograd through Renderer.py:172;
            // d rays_o += d p, d rays_d += d p * z
            double v[6];
            const bool mine = active && g == 0;
            v[0] = mine ? (double)dux * (2.0 * G.inv[0]) + (double)dpe[0] : 0.0;
            v[1] = mine ? (double)duy * (2.0 * G.inv[1]) + (double)dpe[1] : 0.0;
            v[2] = mine ? (double)duz * (2.0 * G.inv[2]) + (double)dpe[2] : 0.0;
            v[3] = v[0] * cz; v[4] = v[1] * cz; v[5] = v[2] * cz;
            if ((P.S & (kTile - 1)) == 0) {                      // a tile never straddles two rays: one atomic per component
#pragma unroll
                for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                    for (int q = 0; q < 6; ++q) v[q] += shfl_xor_d(v[q], m);
                if (lane == 0 && active) {
                    const long long ray = (long long)((unsigned)gp / (unsigned)P.S);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        atomic_add_global(P.d_rays_o + ray * 3 + q, (float)v[q]);
                        atomic_add_global(P.d_rays_d + ray * 3 + q, (float)v[3 + q]);
                    }
                }
            } else if (mine) {
                const long long ray = (long long)((unsigned)gp / (unsigned)P.S);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    atomic_add_global(P.d_rays_o + ray * 3 + q, (float)v[q]);
                    atomic_add_global(P.d_rays_d + ray * 3 + q, (float)v[3 + q]);
                }
            }
        }
        cur = nx;
        tile = nxt;
        dbg.stamp(7);
    }
    dbg.stamp(8);
    // The block's end: ONE barrier.  A wave leaves its d _B sums (over its lane groups) in its OWN staging region as soon as it runs out
    // of tiles -- no barrier in front of that, the region is the wave's --, and behind the barrier the partial image is summed and stored
    // BEFORE the hot table's flush: its store does not queue behind the flush's atomics (round 6; before: barrier, flush, barrier, sums,
    // barrier, store -- 8 us from the last wave's last tile to the end of the block in the stamps of tests/perf/ts_dx.py).
    if (XYZ && PARAMS) {
#pragma unroll
        for (int k = 0; k < kET; ++k)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v = aB[k][d];
                v += shfl_xor(v, 16); v += shfl_xor(v, 32);
                if (g == 0) Sw[d * 96 + 16 * k + pt] = v;
            }
    }
    if (use_hot || (gl && do_grid) || (XYZ && PARAMS)) block_sync();
    dbg.stamp(10);
    if (XYZ && PARAMS) {
        float *part = P.dbpart + (long long)gb * kDbPart;
        for (int t = tid(); t < kDbPart; t += nthreads()) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += stg[w * kDxStg + t];
            part[t] = s;
        }
    }
    if (use_hot) hot_flush(hot, G);
    dbg.stamp(11);
    if (gl && do_grid) {
        for (int i = tid(); i < P.lds_grid_floats; i += nthreads()) {
            const float v = gl[i];
            if (v != 0.f) atomic_add_global(G.dfeat + i, v);
        }
    }
    dbg.stamp(9);
}

// grid = the blocks of all decoder passes of the stage: [dx_beg[p], dx_beg[p + 1]) work on pass p (the host deals them by a tile's cost: a
// pass without parameter or ray gradients skips the embedding backward, round 6); a block stages ONE decoder's transposed stream and walks
// the tiles of its contiguous share of the pass.
template <int STAGE, bool RAYS>
NSR_KERNEL NSR_BOUNDS(64 * kDxMaxWaves) void render_bwd_dx_kernel(const RenderParams P) {
    const int b = bid_x();
    if (STAGE == NSR_STAGE_COARSE) {
        if (P.dec[NSR_COARSE].dparams) dx_pass<NSR_COARSE, true, RAYS>(P, b, nblk_x(), b); else dx_pass<NSR_COARSE, false, RAYS>(P, b, nblk_x(), b);
    } else {
        if (b < P.dx_beg[1]) {
            const int nbp = P.dx_beg[1];
            if (P.dec[NSR_MIDDLE].dparams) dx_pass<NSR_MIDDLE, true, RAYS>(P, b, nbp, b); else dx_pass<NSR_MIDDLE, false, RAYS>(P, b, nbp, b);
        } else if (b < P.dx_beg[2]) {
            const int bi = b - P.dx_beg[1], nbp = P.dx_beg[2] - P.dx_beg[1];
            if (STAGE >= NSR_STAGE_FINE) { if (P.dec[NSR_FINE].dparams) dx_pass<NSR_FINE, true, RAYS>(P, bi, nbp, b); else dx_pass<NSR_FINE, false, RAYS>(P, bi, nbp, b); }
        } else {
            const int bi = b - P.dx_beg[2], nbp = P.dx_beg[3] - P.dx_beg[2];
            if (STAGE == NSR_STAGE_COLOR) { if (P.dec[NSR_COLOR].dparams) dx_pass<NSR_COLOR, true, RAYS>(P, bi, nbp, b); else dx_pass<NSR_COLOR, false, RAYS>(P, bi, nbp, b); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dW kernel: split-K GEMM over the sample points; loader waves stream the operands through an LDS ring, compute waves run free
// ------------------------------------------------------------------------------------------------
// A slot tile is a row-major [16 points][16 channels] matrix (1 KB): lane (i = channel, g) reads the dwords q * 64 + lane,
// q = 0..3, i.e. X[point 4 q + g][channel i] -- the "lane = channel" operand form of a contraction over points, conflict-free
// in LDS, no transposition anywhere.  Block = 8 compute waves, which own disjoint output tiles (roles below), + 2 loader
// waves, which copy the 16-point tiles of the block with global -> LDS DMA into a ring of kDwRing slots (22 / 24 pieces of
// 1 KB per tile, tiles alternating between the loaders, one tile in flight each: a landed tile is published at once).  There is NO block barrier: a loader
// publishes "my m-th tile has landed" in an LDS word, every compute wave publishes "I am done with tile k", the loaders
// reuse a slot when the slowest compute wave has left it.  Compute waves therefore drift apart, and on every SIMD the LDS
// reads / sines of one wave run under the MFMAs of the other.
// (Measured alternatives, profiles/r03_dw_variants.txt: every compute wave issuing its share of the DMA pieces and ONE
// barrier per tile or per pair of tiles, 8 or 16 waves, 3..6 slots, roles balanced per SIMD, second-half waves in
// anti-phase: 70-75 us per 1000 colour-stage rays for 29 us of MFMA work whatever the arrangement -- the time was the SUM
// of the DMA cadence (35 us), the LDS read phases and the MFMAs: lock-step phases behind the barriers, the youngest waves
// of a SIMD starving at issue.  Operands straight from global memory into registers, no LDS: 159 us, 240-256 VGPRs.)
constexpr int kDwCompute = 8, kDwLoaders = 2, kDwWaves = kDwCompute + kDwLoaders, kDwRing = 6;
constexpr long long kDwDbgOff = 768ll * kDxMaxWaves * 64;      // this kernel's stamp slots sit behind the dX kernel's (tests/perf/ts_d*.py)
template <int KIND>
struct DwLay {
    static constexpr int NC = KIND == NSR_FINE ? 4 : 2;          // feature operand tiles ([c_fine | c_mid] for the fine decoder)
    static constexpr int oDY = 0, oH = 10 * 256, oCM = 22 * 256;  // LDS slot: dY (10 KB) | h_0..h_4, c (12 KB) | fine: c_mid (2 KB)
    static constexpr int oPF = (20 + NC) * 256, oDR = oPF + 64;   // [16][4] fp32 positions | [16][4] d raw
    static constexpr int kSlot = oDR + 64;
    static constexpr int NOPS = 20 + NC + 2;                      // DMA pieces per tile
};
constexpr int dw_lds_bytes(int kind) { return (kDwRing * (kind == NSR_FINE ? DwLay<NSR_FINE>::kSlot : DwLay<NSR_MIDDLE>::kSlot) + 16) * 4; }
struct DwSrc {                 // where one 16-point tile's operands live (LDS slot)
    const float *dy;           // [kDySlots][16][16]   dY_i
    const float *act;          // [12][16][16]         h_i, c
    const float *cm;           // fine decoder: the middle decoder's features, addressed like activation slots kActC, kActC + 1
    const float *pf;           // [16][4] float positions
    const float *dr;           // [16][4] d raw
};
template <int KIND>
NSR_DEV DwSrc dw_src(const float *slot) {
    typedef DwLay<KIND> Y;
    DwSrc s;
    s.dy = slot + Y::oDY; s.act = slot + Y::oH; s.cm = slot + Y::oCM - kActC * 256; s.pf = slot + Y::oPF; s.dr = slot + Y::oDR;
    return s;
}
// request tile `tile`'s operands into ring slot `slot` (one loader wave: NOPS pieces)
template <int KIND>
NSR_DEV void dw_issue(const RenderParams &P, long long tile, float *slot, int lane) {
    typedef DwLay<KIND> Y;
    if (P.xflags & 16) tile &= 3;                                 // measurement: operands from cache-resident tiles
    // a tile's dY (10 KB) and its hidden states + features (12 KB) are contiguous spans in memory: piece n = the n-th KB
    const float *dt = P.dy + ((long long)act_pass(KIND) * P.act_tiles + tile) * (kDySlots * 256);
    const float *at = P.acts + ((long long)act_pass(KIND) * P.act_tiles + tile) * (kActSlots * 256);
#pragma unroll
    for (int n = 0; n < 10; ++n) dma16(dt + n * 256 + lane * 4, slot + Y::oDY + n * 256, lane);
#pragma unroll
    for (int n = 0; n < 12; ++n) dma16(at + n * 256 + lane * 4, slot + Y::oH + n * 256, lane);
    if (KIND == NSR_FINE) {                                       // the middle decoder's features of the same tile (pass 0)
        const float *cm = P.acts + (tile * kActSlots + kActC) * 256;
        dma16(cm + lane * 4, slot + Y::oCM, lane);
        dma16(cm + 256 + lane * 4, slot + Y::oCM + 256, lane);
    }
    if (lane < 16) {
        dma16(P.pf + tile * 64 + lane * 4, slot + Y::oPF, lane);
        dma16(P.draw + tile * 64 + lane * 4, slot + Y::oDR, lane);
    }
}
NSR_DEV f32x4 gop(const float *tile, int lane) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = tile[q * 64 + lane];
    return v;
}
// rows of the ragged last tile beyond the last point: element q of lane (i, g) is point 4 q + g
NSR_DEV void rag(f32x4 &v, int g, int nvalid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (4 * q + g >= nvalid) v[q] = 0.f;
}
NSR_DEV void img_tile(float *img, const Mat m, int To, int Tk, const f32x4 acc, int i, int g) {
    if (16 * Tk + i < m.kcols) {
#pragma unroll
        for (int r = 0; r < 4; ++r) img[m.off + (16 * To + 4 * g + r) * m.stride + m.kbeg + 16 * Tk + i] = acc[r];
    }
}
NSR_DEV float red_g4(float v) { v += shfl_xor(v, 16); v += shfl_xor(v, 32); return v; }

// What one wave of an xyz-decoder block owns.  16x16 output tiles in units: W0 / W3e against embedding k-tile k ("We_k": 4
// tiles, one sine evaluation), a row tile of a layer's hidden-state block ("Wh(j, row)": 2 tiles) and of its feature block
// ("G(j, row)": c_dim / 16 tiles).  Waves w and w + 4 share a SIMD, so the units are dealt such that every SIMD carries the
// same number of MFMAs (14 tiles per 16 points, 18 for the fine decoder):
//   w = 0..3:  We_w + Wh(j, row) + G(j, row)   with j = 1 (w < 2) or 2, row = w & 1
//   w = 4, 5:  We_w + G(3, w - 4)              w = 4 also the bias sums of layer 0
//   w = 6, 7:  Wh(4, row) + G(4, row) + Wh(3, row),  row = w - 6;  w = 6 also the output layer (VALU sums)
// The bias sums of layer j go with its Wh unit.  Within a tile the k-steps of ALL the wave's output tiles are issued
// round-robin (a 16x16x4 fp32 MFMA has a 40-cycle dependent latency against a 32-cycle issue interval).
template <int KIND, int WAVE>
struct DwXyzWave {
    typedef DwLay<KIND> Y;
    static constexpr int CD = cdim_of(KIND), NC = Y::NC, NOUT = nout_of(KIND), NO = NOUT == 1 ? 1 : 3;
    static constexpr bool kWe = WAVE < 6;
    static constexpr int WJ = WAVE < 2 ? 1 : (WAVE < 4 ? 2 : (WAVE < 6 ? 0 : 4));        // layer of the first Wh unit (0: none)
    static constexpr int GJ = WAVE < 2 ? 1 : (WAVE < 4 ? 2 : (WAVE < 6 ? 3 : 4));        // layer of the G unit
    static constexpr int W2J = WAVE >= 6 ? 3 : 0;                                         // layer of the second Wh unit
    static constexpr int ROW = WAVE < 4 ? (WAVE & 1) : (WAVE < 6 ? WAVE - 4 : WAVE - 6);
    static constexpr bool kOut = WAVE == 6, kB0 = WAVE == 4;    // (round 6: the output layer's VALU sums moved from wave 5 -- an embedding k-tile, i.e. sines,
                                                                // + a feature block: the block's slowest role, tests/perf/ts_dw.py -- to wave 6, which evaluates no sine)
    f32x4 we[4];               // [W0 To 0, W0 To 1, W3e To 0, W3e To 1] x embedding k-tile WAVE
    f32x4 wh[2], w2[2], gg[4]; // row ROW of layers WJ / W2J (hidden-state k-tiles), of layer GJ (feature k-tiles)
    float vb, vb2, vb0[2];     // bias sums: layer WJ, layer W2J (row ROW); layer 0 (kB0)
    float wo[3][2], bo[3], go[3][4];
    float bx, by, bz;          // Fourier matrix column of channel 16 WAVE + i
    float dscale;              // incoming gradient still to be applied to d raw (see RenderParams.draw_scaled)

    NSR_DEV void init(const RenderParams &P, int i) {
        dscale = (!P.draw_scaled && P.g_scale) ? (float)P.g_scale[0] : 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { we[k] = f4zero(); gg[k] = f4zero(); }
        wh[0] = wh[1] = w2[0] = w2[1] = f4zero();
        vb = vb2 = vb0[0] = vb0[1] = 0.f;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            bo[n] = 0.f; wo[n][0] = wo[n][1] = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) go[n][b] = 0.f;
        }
        bx = by = bz = 0.f;
        if (kWe) {
            const F4 b = load_b1(P.dec[KIND].packed, 16 * WAVE + i);     // aux table of the packed buffer (global)
            bx = b.x; by = b.y; bz = b.z;
        }
    }
    // operands of one tile as loaded ("lane = channel" form); nothing here is computed on, so that fetch() only issues loads
    struct Ops { f32x4 y[4], a1, a2, ag, h1[2], h2[2], c[NC], h4[2]; F4 pos[4]; float dn[NO][4]; };
    NSR_DEV void fetch(const DwSrc s, int lane, Ops &o) {
        const int g = lane >> 4;
        if (kWe) {
            o.y[0] = gop(s.dy + 0 * 256, lane); o.y[1] = gop(s.dy + 1 * 256, lane);
            o.y[2] = gop(s.dy + 6 * 256, lane); o.y[3] = gop(s.dy + 7 * 256, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) o.pos[q] = ld4(s.pf + (4 * q + g) * 4);
        }
        if (WJ) {
            o.a1 = gop(s.dy + (2 * WJ + ROW) * 256, lane);
            o.h1[0] = gop(s.act + (2 * (WJ - 1)) * 256, lane); o.h1[1] = gop(s.act + (2 * (WJ - 1) + 1) * 256, lane);
        }
        if (W2J) {
            o.a2 = gop(s.dy + (2 * W2J + ROW) * 256, lane);
            o.h2[0] = gop(s.act + (2 * (W2J - 1)) * 256, lane); o.h2[1] = gop(s.act + (2 * (W2J - 1) + 1) * 256, lane);
        }
        if (GJ != WJ) o.ag = gop(s.dy + (2 * GJ + ROW) * 256, lane);
#pragma unroll
        for (int Tc = 0; Tc < NC; ++Tc) o.c[Tc] = gop((Tc < 2 ? s.act : s.cm) + (kActC + (Tc & 1)) * 256, lane);
        if (kOut) {
            o.h4[0] = gop(s.act + 8 * 256, lane); o.h4[1] = gop(s.act + 9 * 256, lane);
#pragma unroll
            for (int n = 0; n < NO; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) o.dn[n][q] = s.dr[(4 * q + g) * 4 + (NOUT == 1 ? 3 : n)] * dscale;
        }
    }
    // the tile's sums; nvalid < 16: the ragged last tile (rows beyond the last point hold whatever the padding holds)
    NSR_DEV void consume(Ops &o, int lane, int nvalid) {
        const int g = lane >> 4;
        if (nvalid < kTile) {
            if (kWe) { rag(o.y[0], g, nvalid); rag(o.y[1], g, nvalid); rag(o.y[2], g, nvalid); rag(o.y[3], g, nvalid); }
            if (WJ) { rag(o.a1, g, nvalid); rag(o.h1[0], g, nvalid); rag(o.h1[1], g, nvalid); }
            if (W2J) { rag(o.a2, g, nvalid); rag(o.h2[0], g, nvalid); rag(o.h2[1], g, nvalid); }
            if (GJ != WJ) rag(o.ag, g, nvalid);
#pragma unroll
            for (int Tc = 0; Tc < NC; ++Tc) rag(o.c[Tc], g, nvalid);
            if (kOut) {
                rag(o.h4[0], g, nvalid); rag(o.h4[1], g, nvalid);
#pragma unroll
                for (int n = 0; n < NO; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (4 * q + g >= nvalid) o.dn[n][q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (kWe && 4 * q + g >= nvalid) o.pos[q] = F4{0.f, 0.f, 0.f, 0.f};
        }
        const f32x4 ag = (GJ == WJ) ? o.a1 : o.ag;
        f32x4 e = f4zero();
        if (kWe) {
            f32x4 qx, qy, qz;
#pragma unroll
            for (int q = 0; q < 4; ++q) { qx[q] = o.pos[q].x; qy[q] = o.pos[q].y; qz[q] = o.pos[q].z; }
#if defined(NSR_X_DW_NOSIN)              // A/B build (tools/build_ts.sh): what the sines cost this kernel (wrong numbers, timing only)
            e = vfma(qz, splat(bz), vfma(qy, splat(by), qx * splat(bx)));
#else
            e = sin_dw4(vfma(qz, splat(bz), vfma(qy, splat(by), qx * splat(bx))));     // decoder.py:29-30
#endif
            if (kB0) { vb0[0] += sum4(o.y[0]); vb0[1] += sum4(o.y[1]); }
        }
        if (WJ) vb += sum4(o.a1);
        if (W2J) vb2 += sum4(o.a2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (kWe) {
#pragma unroll
                for (int n = 0; n < 4; ++n) we[n] = mfma16(o.y[n][q], e[q], we[n]);
            }
            if (WJ) { wh[0] = mfma16(o.a1[q], o.h1[0][q], wh[0]); wh[1] = mfma16(o.a1[q], o.h1[1][q], wh[1]); }
#pragma unroll
            for (int Tc = 0; Tc < NC; ++Tc) gg[Tc] = mfma16(ag[q], o.c[Tc][q], gg[Tc]);
            if (W2J) { w2[0] = mfma16(o.a2[q], o.h2[0][q], w2[0]); w2[1] = mfma16(o.a2[q], o.h2[1][q], w2[1]); }
        }
        if (kOut) {
            // output layer: d Wo[n][k] = sum_p d_out[p][n] h4[p][k], d bo, and g_out[n][c] = sum_p d_out[p][n] c[p][c] (-> dU_4)
#pragma unroll
            for (int n = 0; n < NO; ++n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wo[n][0] = fmaf(o.dn[n][q], o.h4[0][q], wo[n][0]); wo[n][1] = fmaf(o.dn[n][q], o.h4[1][q], wo[n][1]);
                    bo[n] += o.dn[n][q];
                }
#pragma unroll
                for (int Tc = 0; Tc < NC; ++Tc)
#pragma unroll
                    for (int q = 0; q < 4; ++q) go[n][Tc] = fmaf(o.dn[n][q], o.c[Tc][q], go[n][Tc]);
            }
        }
    }
    static constexpr int hid_of(int j) { return j == 1 ? XW1 : (j == 2 ? XW2 : (j == 3 ? XW3H : XW4)); }
    // ---- flush.  Round 5: the fc_c gradients leave the block finished.  dU_{j-1} = W_j^T G_j and dv_{j-1} = W_j^T db_j used to be
    // formed by the finalize kernel from the SUMMED G / db (its 45 fc_c blocks per decoder fetched 16 of every 64 bytes and were its
    // longest chain); the product is linear, so every block applies W_j^T to its own partial G / db here -- 16 + 8 MFMAs per wave,
    // once per block -- and finalize is a plain full-line sum over the images.  Row tile ROW of G_j (this wave's accumulators) is a
    // K-slice of the product: D[k][c] = sum_o W_j[16 ROW + o][k] G[o][c].  With the K index ordered o = 4 g + s the B operand of
    // k-step s is the accumulator element s the lane already holds (no shuffle); the A operand W_j[16 ROW + 4 g + s][16 Tk + i] comes
    // straight from the parameter blob.  The two row tiles of a layer meet through LDS (the ring is free by then: block barrier).
    static constexpr int wt_off(int j) { return xyz_w(CD, j) + (j == 3 ? kE : 0); }      // pts_linears.j weight, hidden-state columns
    struct WtA { float a[2][4]; };
    NSR_DEV static WtA wt_load(const float *params, int j, int row, int i, int g) {
        WtA A;
#pragma unroll
        for (int Tk = 0; Tk < 2; ++Tk)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) A.a[Tk][sI] = params[wt_off(j) + (16 * row + 4 * g + sI) * xyz_in(j) + 16 * Tk + i];
        return A;
    }
    // LDS scratch of the flush (floats from the ring's base): [layer j - 1][Tk][Tc][lane][4] products of row tile 1 | [j - 1][32] dv
    static constexpr int kCombB = 4 * 2 * NC * 256;
    // dv partial of layer j from this wave's bias sums `db` (every lane (i, *) holds db[16 row + i]): column 0 of the same product
    NSR_DEV static void dv_part(const WtA &A, float db, int lane, f32x4 (&out)[2]) {
        const int i = lane & 15, g = lane >> 4;
        out[0] = f4zero(); out[1] = f4zero();
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            const float bsel = shfl(db, 4 * g + sI);              // db[16 row + 4 g + s], wanted in column 0 only
            const float bv = i == 0 ? bsel : 0.f;
            out[0] = mfma16(A.a[0][sI], bv, out[0]);
            out[1] = mfma16(A.a[1][sI], bv, out[1]);
        }
    }
    f32x4 pu[2][4];            // W_GJ^T G_GJ, this wave's row-tile share: [Tk][Tc]
    f32x4 pv[2], pv2[2];       // W^T db shares of the layers whose bias sums this wave owns (WJ, W2J)
    NSR_DEV void flush_products(const float *params, float *comb, int lane) {
        const int i = lane & 15, g = lane >> 4;
        const WtA A = wt_load(params, GJ, ROW, i, g);
#pragma unroll
        for (int Tk = 0; Tk < 2; ++Tk)
#pragma unroll
            for (int Tc = 0; Tc < NC; ++Tc) {
                f32x4 acc = f4zero();
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) acc = mfma16(A.a[Tk][sI], gg[Tc][sI], acc);
                pu[Tk][Tc] = acc;
            }
        if (WJ) {                                                   // (WJ == GJ where it is set: the same operand)
            const float db = red_g4(vb);
            dv_part(A, db, lane, pv);
        }
        if (W2J) {
            const WtA A2 = wt_load(params, W2J, ROW, i, g);
            const float db2 = red_g4(vb2);
            dv_part(A2, db2, lane, pv2);
        }
        if (ROW == 1) {                                             // hand the share to the wave of row tile 0
#pragma unroll
            for (int Tk = 0; Tk < 2; ++Tk)
#pragma unroll
                for (int Tc = 0; Tc < NC; ++Tc) st4(comb + (((GJ - 1) * 2 + Tk) * NC + Tc) * 256 + lane * 4, to_F4(pu[Tk][Tc]));
            if (i == 0) {
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk) {
                    if (WJ) st4(comb + kCombB + (WJ - 1) * 32 + 16 * Tk + 4 * g, to_F4(pv[Tk]));
                    if (W2J) st4(comb + kCombB + (W2J - 1) * 32 + 16 * Tk + 4 * g, to_F4(pv2[Tk]));
                }
            }
        }
    }
    NSR_DEV void flush(float *img, const float *params, const float *comb, int lane) {
        const int i = lane & 15, g = lane >> 4;
        if (kWe) {
            img_tile(img, xyz_mat(CD, XW0), 0, WAVE, we[0], i, g); img_tile(img, xyz_mat(CD, XW0), 1, WAVE, we[1], i, g);
            img_tile(img, xyz_mat(CD, XW3E), 0, WAVE, we[2], i, g); img_tile(img, xyz_mat(CD, XW3E), 1, WAVE, we[3], i, g);
        }
        if (kB0) {
#pragma unroll
            for (int T = 0; T < 2; ++T) { const float v = red_g4(vb0[T]); if (g == 0) img[bias_off(KIND, 0) + 16 * T + i] = v; }
        }
        if (WJ) {
            img_tile(img, xyz_mat(CD, hid_of(WJ)), ROW, 0, wh[0], i, g); img_tile(img, xyz_mat(CD, hid_of(WJ)), ROW, 1, wh[1], i, g);
            const float v = red_g4(vb);
            if (g == 0) img[bias_off(KIND, WJ) + 16 * ROW + i] = v;
        }
        if (W2J) {
            img_tile(img, xyz_mat(CD, hid_of(W2J)), ROW, 0, w2[0], i, g); img_tile(img, xyz_mat(CD, hid_of(W2J)), ROW, 1, w2[1], i, g);
            const float v = red_g4(vb2);
            if (g == 0) img[bias_off(KIND, W2J) + 16 * ROW + i] = v;
        }
        if (ROW == 0) {
            // dU_{GJ-1} = this share + row tile 1's: fc_c.(GJ-1).weight[k = 16 Tk + 4 g + r][c = 16 Tc + i]
#pragma unroll
            for (int Tk = 0; Tk < 2; ++Tk)
#pragma unroll
                for (int Tc = 0; Tc < NC; ++Tc) {
                    const f32x4 o = pu[Tk][Tc] + to_v(ld4(comb + (((GJ - 1) * 2 + Tk) * NC + Tc) * 256 + lane * 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) img[xyz_fcw(CD, GJ - 1) + (16 * Tk + 4 * g + r) * CD + 16 * Tc + i] = o[r];
                }
            if (i == 0) {
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk) {
                    if (WJ) {
                        const f32x4 o = pv[Tk] + to_v(ld4(comb + kCombB + (WJ - 1) * 32 + 16 * Tk + 4 * g));
#pragma unroll
                        for (int r = 0; r < 4; ++r) img[xyz_fcb(CD, WJ - 1) + 16 * Tk + 4 * g + r] = o[r];
                    }
                    if (W2J) {
                        const f32x4 o = pv2[Tk] + to_v(ld4(comb + kCombB + (W2J - 1) * 32 + 16 * Tk + 4 * g));
#pragma unroll
                        for (int r = 0; r < 4; ++r) img[xyz_fcb(CD, W2J - 1) + 16 * Tk + 4 * g + r] = o[r];
                    }
                }
            }
        }
        if (kOut) {
            float gsum[NO][NC], bsum[NO];
#pragma unroll
            for (int n = 0; n < NO; ++n) {
#pragma unroll
                for (int T = 0; T < 2; ++T) { const float v = red_g4(wo[n][T]); if (g == 0) img[wo_off(KIND) + n * 32 + 16 * T + i] = v; }
                // every lane of a point group summed the same d_out: one lane per group
                bsum[n] = red_g4(bo[n]);
                if (lane == 0) img[bo_off(KIND) + n] = bsum[n];
#pragma unroll
                for (int Tc = 0; Tc < NC; ++Tc) gsum[n][Tc] = red_g4(go[n][Tc]);       // g_out[n][c = 16 Tc + i], in every lane group
            }
            if (NOUT == 4 && lane < 32) img[wo_off(KIND) + 3 * 32 + lane] = 0.f;      // decoder.py:341: the 4th colour output is discarded
            if (NOUT == 4 && lane == 0) img[bo_off(KIND) + 3] = 0.f;
            // fc_c.4: dU_4 = Wo^T g_out, dv_4 = Wo^T d bo (the output layer has <= 3 live rows: plain fma, lane group g takes rows 8 g .. 8 g + 7)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 8 * g + kk;
                float wk[NO];
#pragma unroll
                for (int n = 0; n < NO; ++n) wk[n] = params[wo_off(KIND) + n * 32 + k];
#pragma unroll
                for (int Tc = 0; Tc < NC; ++Tc) {
                    float v = 0.f;
#pragma unroll
                    for (int n = 0; n < NO; ++n) v = fmaf(wk[n], gsum[n][Tc], v);
                    img[xyz_fcw(CD, 4) + k * CD + 16 * Tc + i] = v;
                }
                if (i == 0) {
                    float v = 0.f;
#pragma unroll
                    for (int n = 0; n < NO; ++n) v = fmaf(wk[n], bsum[n], v);
                    img[xyz_fcb(CD, 4) + k] = v;
                }
            }
        }
    }
};

// MLP_no_xyz: wave m < 6 owns the four 16x16 tiles of matrix m (NW0, NW1, NW2, NW3C, NW3H, NW4) and, where it reads a
// layer's dY first, its bias sums; wave 6 the output layer.
template <int WAVE>
struct DwNoxWave {
    static constexpr int LJ = WAVE == NW3C ? 3 : (WAVE == NW3H ? 3 : (WAVE == NW4 ? 4 : WAVE));       // layer whose dY this matrix contracts
    static constexpr bool kMat = WAVE < 6, kBias = kMat && WAVE != NW3H, kOut = WAVE == 6;
    f32x4 w[2][2];
    float vb[2], wo[2], bo, dscale;
    NSR_DEV void init(const RenderParams &P, int) {
        dscale = (!P.draw_scaled && P.g_scale) ? (float)P.g_scale[0] : 1.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) { vb[a] = 0.f; wo[a] = 0.f; w[a][0] = f4zero(); w[a][1] = f4zero(); }
        bo = 0.f;
    }
    struct Ops { f32x4 a[2], x[2]; float dn[4]; };
    NSR_DEV void fetch(const DwSrc s, int lane, Ops &o) {
        const int g = lane >> 4;
        if (kMat) {
            o.a[0] = gop(s.dy + (2 * LJ) * 256, lane); o.a[1] = gop(s.dy + (2 * LJ + 1) * 256, lane);
#pragma unroll
            for (int Tk = 0; Tk < 2; ++Tk)      // input of the matrix: the features (NW0, NW3C) or the previous hidden state
                o.x[Tk] = gop(s.act + ((WAVE == NW0 || WAVE == NW3C) ? kActC + Tk : 2 * (LJ - 1) + Tk) * 256, lane);
        }
        if (kOut) {
            o.x[0] = gop(s.act + 8 * 256, lane); o.x[1] = gop(s.act + 9 * 256, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) o.dn[q] = s.dr[(4 * q + g) * 4 + 3] * dscale;
        }
    }
    NSR_DEV void consume(Ops &o, int lane, int nvalid) {
        const int g = lane >> 4;
        if (nvalid < kTile) {
            if (kMat) { rag(o.a[0], g, nvalid); rag(o.a[1], g, nvalid); }
            if (kMat || kOut) { rag(o.x[0], g, nvalid); rag(o.x[1], g, nvalid); }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (kOut && 4 * q + g >= nvalid) o.dn[q] = 0.f;
        }
        if (kMat) {
            if (kBias) { vb[0] += sum4(o.a[0]); vb[1] += sum4(o.a[1]); }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int To = 0; To < 2; ++To)
#pragma unroll
                    for (int Tk = 0; Tk < 2; ++Tk) w[To][Tk] = mfma16(o.a[To][q], o.x[Tk][q], w[To][Tk]);
        }
        if (kOut) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { wo[0] = fmaf(o.dn[q], o.x[0][q], wo[0]); wo[1] = fmaf(o.dn[q], o.x[1][q], wo[1]); bo += o.dn[q]; }
        }
    }
    NSR_DEV void flush_products(const float *, float *, int) {}
    NSR_DEV void flush(float *img, const float *, const float *, int lane) {
        const int i = lane & 15, g = lane >> 4;
        if (kMat) {
            const Mat m = nox_mat(WAVE);
#pragma unroll
            for (int To = 0; To < 2; ++To)
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk) img_tile(img, m, To, Tk, w[To][Tk], i, g);
            if (kBias) {
#pragma unroll
                for (int T = 0; T < 2; ++T) { const float v = red_g4(vb[T]); if (g == 0) img[nox_b(LJ) + 16 * T + i] = v; }
            }
        }
        if (kOut) {
#pragma unroll
            for (int T = 0; T < 2; ++T) { const float v = red_g4(wo[T]); if (g == 0) img[nox_wo() + 16 * T + i] = v; }
            const float vbo = red_g4(bo);
            if (lane == 0) img[nox_bo()] = vbo;
        }
    }
};

// Which of the 64 tiles bi + (64 c + i) * step, i = 0..63, of a block's sequence hold a ray of the batch (bit i; see tile_live):
// one vector load per lane and a ballot, once per 64 tiles, so that neither the compute waves' tile loop nor the loaders' DMA
// queue waits for a mask byte per tile.
NSR_DEV unsigned long long dw_live_mask(const RenderParams &P, long long bi, long long step, int chunk, long long ntiles, int lane) {
    if (!P.skip_masked) return ~0ull;
    const long long t = bi + ((long long)chunk * 64 + lane) * step;
    bool live = false;
    if (t < ntiles) {
        const unsigned p0 = (unsigned)t * kTile, np = (unsigned)P.n_points_total, pe = p0 + kTile < np ? p0 + kTile : np;   // (< 2^25 points per call)
        const unsigned r0 = ray_of_point(P, p0), r1 = ray_of_point(P, pe - 1);
        for (unsigned r = r0; r <= r1; ++r) live = live || P.keep[r] != 0;
    }
    return ballot64(live);
}

// control words behind the ring: landed[j] = tiles loader j has landed, prog[w] = tiles compute wave w is done with
template <int KIND, class W>
NSR_DEV void dw_compute(const RenderParams &P, W &Wv, float *ring, int *ctl, float *img, int wave, int lane, int bi, int nbp) {
    typedef DwLay<KIND> Y;
    Wv.init(P, lane & 15);
    const long long ntiles = (P.n_points_total + kTile - 1) / kTile;
    const long long ntl = (P.xflags & 64) ? 0 : ntiles;             // measurement: no tiles (prologue + flush only)
    const long long last = ntiles - 1, step = nbp;
    const int ragged = (int)(P.n_points_total & (kTile - 1));
    // (Requesting tile k + 1's operands into a second register set before tile k's MFMAs -- the two waves of a SIMD wait for
    // the same slot flags, so their flag / read / wait phases coincide -- needs 168 VGPRs + 64-320 B of scratch at the three
    // waves per SIMD a 10-wave block implies: 135 us instead of 65, profiles/r03_dw_variants.txt.)
    typename W::Ops ops;
    int k = 0;
    unsigned long long lm = ~0ull;
    const Dbg dbg{P.dbg ? P.dbg + kDwDbgOff + ((long long)bid_x() * kDxMaxWaves + wave) * 64 : nullptr};   // (-DNSR_TS builds: tests/perf/ts_dw.py)
    dbg.stamp(0);
    for (long long t = bi; t < ntl; t += step, ++k) {
        loop_fence();
        dbg.stamp(1);
        if ((k & 63) == 0) lm = dw_live_mask(P, bi, step, k >> 6, ntiles, lane);
        while (flag_load(ctl + (k % kDwLoaders)) <= (k / kDwLoaders)) spin_pause();   // the loader of this tile has landed it
        dbg.stamp(2);
        if ((lm >> (k & 63)) & 1ull) {                                       // (else: the loader left the slot empty)
            Wv.fetch(dw_src<KIND>(ring + (k % kDwRing) * Y::kSlot), lane, ops);
#ifdef NSR_TS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (stamped builds: the operand reads on their own)
            dbg.stamp(9);
#endif
            Wv.consume(ops, lane, (t == last && ragged) ? ragged : kTile);
        }
        dbg.stamp(3);
        flag_store(ctl + kDwLoaders + wave, k + 1);                          // (release: the slot's reads have returned)
        dbg.stamp(4);
    }
    dbg.stamp(5);
    // flush: every wave of the block is done with the ring (barrier), the row-tile pairs exchange their W^T products through it
    // (barrier), the images are written.  The loader waves take part in both barriers (dw_loader).
    block_sync();
    dbg.stamp(6);
    Wv.flush_products(P.dec[KIND].params, ring, lane);
    block_sync();
    dbg.stamp(7);
    Wv.flush(img, P.dec[KIND].params, ring, lane);
    dbg.stamp(8);
}

template <int KIND>
NSR_DEV void dw_loader(const RenderParams &P, float *ring, int *ctl, int j, int lane, int bi, int nbp) {
    typedef DwLay<KIND> Y;
    const long long ntiles = (P.n_points_total + kTile - 1) / kTile;
    const long long step = nbp, ntl = (P.xflags & 64) ? 0 : ntiles;
    int m = 0;                                                               // this loader's m-th tile is the block's tile 2 m + j
    unsigned long long lm = ~0ull;
    int chunk = -1;
    const Dbg dbg{P.dbg ? P.dbg + kDwDbgOff + ((long long)bid_x() * kDxMaxWaves + kDwCompute + j) * 64 : nullptr};
    dbg.stamp(0);
    for (long long t = bi + j * step; t < ntl; t += kDwLoaders * step, ++m) {
        loop_fence();
        dbg.stamp(1);
        const int k = kDwLoaders * m + j;
        if ((k >> 6) != chunk) { chunk = k >> 6; lm = dw_live_mask(P, bi, step, chunk, ntiles, lane); }
        // The previous tile is waited for and published FIRST (it was requested a whole iteration ago: it has landed), then the ring slot
        // of this one.  (Until round 5 the order was "slot, issue, then wait for the previous tile and publish it" -- two tiles in
        // flight per loader, but a tile that had landed stayed unpublished for as long as its loader waited for the slot of the NEXT
        // one, i.e. for the slowest compute wave: the faster waves stalled on tiles that sat in LDS; tests/perf/ts_dw.py.)
        if (m >= 1) { dma_wait<0>(); flag_store(ctl + j, m); }
        if (k >= kDwRing) {                                                  // the slot still holds tile k - kDwRing
            for (;;) {
                int lo = flag_load(ctl + kDwLoaders);
#pragma unroll
                for (int w = 1; w < kDwCompute; ++w) { const int v = flag_load(ctl + kDwLoaders + w); lo = v < lo ? v : lo; }
                if (lo > k - kDwRing) break;
                spin_pause();
            }
        }
        dbg.stamp(2);
        if ((lm >> (k & 63)) & 1ull) {
            dw_issue<KIND>(P, t, ring + (k % kDwRing) * Y::kSlot, lane);
            dbg.stamp(3);
        }
        dbg.stamp(4);
    }
    dma_wait<0>();
    flag_store(ctl + j, m);
    dbg.stamp(5);
    block_sync();                                                            // the compute waves' flush (dw_compute)
    block_sync();
}

template <int KIND>
NSR_DEV void dw_pass(const RenderParams &P, int bi, int nbp) {      // block bi of the nbp blocks of this decoder pass
    float *ring = reinterpret_cast<float *>(lds_base());
    int *ctl = reinterpret_cast<int *>(ring + kDwRing * DwLay<KIND>::kSlot);
    const int lane = tid() & 63, wave = uniform(tid() >> 6);
    if (tid() < 16) ctl[tid()] = 0;
    block_sync();
    float *img = P.partials + (long long)bid_x() * P.partial_stride;
    // one specialisation per compute wave: its role's output tiles and operand reads are compile-time constants
#define NSR_DW_CASE(WV)                                                                                         \
    case WV: {                                                                                                  \
        if constexpr (KIND == NSR_COARSE) { DwNoxWave<WV> W; dw_compute<KIND>(P, W, ring, ctl, img, WV, lane, bi, nbp); } \
        else { DwXyzWave<KIND, WV> W; dw_compute<KIND>(P, W, ring, ctl, img, WV, lane, bi, nbp); }                        \
        break;                                                                                                  \
    }
    switch (wave) {
        NSR_DW_CASE(0) NSR_DW_CASE(1) NSR_DW_CASE(2) NSR_DW_CASE(3) NSR_DW_CASE(4) NSR_DW_CASE(5) NSR_DW_CASE(6) NSR_DW_CASE(7)
        default: dw_loader<KIND>(P, ring, ctl, wave - kDwCompute, lane, bi, nbp); break;
    }
#undef NSR_DW_CASE
}

// grid = blocks of all decoder passes: [dw_beg[p], dw_beg[p + 1]) work on pass p and leave one partial image each (image =
// block index).  The host deals the blocks in proportion to a tile's measured cost (240 / 260 for the fine decoder, 224 for the others)
// and gives none to a decoder whose parameter gradients nobody asked for.
template <int STAGE>
NSR_KERNEL NSR_BOUNDS(64 * kDwWaves) void render_bwd_dw_kernel(const RenderParams P) {
    const int b = bid_x();
    if (STAGE == NSR_STAGE_COARSE) {
        dw_pass<NSR_COARSE>(P, b, nblk_x());
    } else {
        if (b < P.dw_beg[1]) dw_pass<NSR_MIDDLE>(P, b, P.dw_beg[1]);
        else if (b < P.dw_beg[2]) { if (STAGE >= NSR_STAGE_FINE) dw_pass<NSR_FINE>(P, b - P.dw_beg[1], P.dw_beg[2] - P.dw_beg[1]); }
        else { if (STAGE == NSR_STAGE_COLOR) dw_pass<NSR_COLOR>(P, b - P.dw_beg[2], P.dw_beg[3] - P.dw_beg[2]); }
    }
}

// ------------------------------------------------------------------------------------------------
// finalize: dparams (+)= sum of the partial images.  grid = (blocks, decoders with gradients), 1024 threads: block b owns the 64
// parameters [64 b, 64 b + 64) of the flat blob, its 16 waves take slices of the image list.  Every parameter is a plain sum over the
// dW blocks' images (round 5: the dW blocks apply W^T to their G / db shares themselves, see DwXyzWave::flush_products) except
// embedder._B, whose partials come from the dX blocks ([ndx][3][96]).
// ------------------------------------------------------------------------------------------------
struct FinalJob {
    const float *images;     // [nimg][stride]
    const float *dbpart;     // [ndx][kDbPart]
    float *dparams;
    int kind, nimg, ndx;
};
struct FinalParams {
    FinalJob job[3];
    int stride, overwrite;
};
NSR_DEV void final_store(const FinalParams &R, float *p, float v) { *p = R.overwrite ? v : *p + v; }
// sum of p[k * stride] over k = first, first + step, ... < n, in that order, with sixteen loads in flight at a time (the kernel
// is a chain of memory round trips: with four in flight the 255 images of a one-decoder stage took eight of them)
NSR_DEV float strided_sum(const float *p, long long stride, int first, int step, int n) {
    float s = 0.f;
    for (int k0 = first; k0 < n; k0 += 16 * step) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int k = k0 + j * step; v[j] = k < n ? p[(long long)k * stride] : 0.f; }
#pragma unroll
        for (int j = 0; j < 16; ++j) if (k0 + j * step < n) s += v[j];
    }
    return s;
}

NSR_KERNEL void bwd_finalize_kernel(const FinalParams R) {
    const FinalJob J = R.job[bid_y()];
    float *red = reinterpret_cast<float *>(lds_base());            // [nthreads]
    const int kind = J.kind, cd = cdim_of(kind);
    const int total = param_total(kind);
    const int t = tid(), nt = nthreads();
    const int lane = t & 63, slice = t >> 6, nslice = nt >> 6;
    const int e = bid_x() * 64 + lane;                             // parameter index in the flat blob
    if (bid_x() * 64 >= total) return;                             // (grid sized for the largest decoder of the stage)
    float s = 0.f;
    if (e < total) {
        const int rb = kind != NSR_COARSE ? e - xyz_B(cd) : -1;    // embedder._B[d][ch] sits at xyz_B + d * 93 + ch
        if (rb >= 0 && rb < 3 * kE) {
            const int d = rb / kE, ch = rb - d * kE;
            s = strided_sum(J.dbpart + d * 96 + ch, kDbPart, slice, nslice, J.ndx);
        } else {
            s = strided_sum(J.images + e, R.stride, slice, nslice, J.nimg);
        }
    }
    red[t] = s;
    block_sync();
    if (slice == 0 && e < total) {
        for (int k = 1; k < nslice; ++k) s += red[k * 64 + lane];
        final_store(R, J.dparams + e, s);
    }
}

}  // namespace nsr
