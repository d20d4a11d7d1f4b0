// nsr_fwd2.h -- the forward of a differentiated render call as three launches (second generation), included by nsr_kernels.h.
//
// render_fwd_kernel (nsr_kernels.h) evaluates the decoders of a stage one after the other inside every block: each block
// stages up to three operand streams (61-82 KB each) for twelve tiles of work, with two block barriers per decoder -- a wave
// timeline of the colour stage at 1000 rays (profiles/r03_fwd_timeline.txt) shows 40 us of decoder work (the matrix-core
// bound) in 87 us: the rest is staging, the skew of twelve waves meeting at six barriers, and a serial prologue / epilogue.
// When the call saves activations for the split backward (nsr_render_args.acts + zvals) the forward takes the split
// backward's shape instead:
//   fwd_sample_kernel        one wave per ray (lane = sample): sample placement (Renderer.py:88-170) and the fp64 sample
//                            positions (Renderer.py:172-174) -> zvals, pd, pf.  The same expressions as compute_z / the
//                            position code of render_fwd_kernel, operation by operation.
//   render_fwd_pass_kernel   grid = (blocks, decoder passes): a block stages ONE decoder's stream once and its waves walk the
//                            16-point tiles of the pass independently -- no barrier after the staging.  Feature gather,
//                            decoder (mlp_xyz_fwd / mlp_nox_fwd with the activation sink), the decoder's output per sample:
//                            occupancy parts into the (still unused) d raw array of the workspace, colours into `raw`.
//   fwd_composite_kernel     one wave per ray: occupancy = fine + middle (decoder.py:333,341), bound override
//                            (Renderer.py:57), raw, compositor (common.py:231-244), depth / variance / colour, and the fused
//                            mapping loss with d raw for the split backward (Mapper.py:487-493) -- the second half of
//                            render_fwd_kernel, unchanged.
// Small batches gain twice: the decoder passes of the tracker's 200 rays run side by side on different CUs instead of one
// after the other in 50 blocks.
#pragma once

namespace nsr {

// ------------------------------------------------------------------------------------------------
// (1) sample placement + positions, one wave per ray
// ------------------------------------------------------------------------------------------------
NSR_KERNEL void fwd_sample_kernel(const RenderParams P) {
    double *zs = reinterpret_cast<double *>(lds_base());             // [waves][64] rank-sort scratch
    const int lane = tid() & 63, wave = tid() >> 6, nw = nthreads() >> 6;
    const long long ray = (long long)bid_x() * nw + wave;
    const int S = P.S;
    double *zw = zs + wave * 64;
    if (ray >= P.n_rays) return;                                     // whole wave (no block barrier below)
    // (rays the pre-filter removed are placed too: a tile that straddles a removed and a kept ray is evaluated as a whole)
    const bool act = lane < S;
    const int k = act ? lane : 0;
    const bool guided = (P.gt_depth != nullptr) && (P.stage != NSR_STAGE_COARSE);
    double z = 0.0;
    {
        const double far_bb = ray_far_bb(P, ray);
        if (!guided) {
            const float tk = P.t_uniform[k];
            const float near_part = 0.01f * (1.f - tk);
            z = (double)near_part + far_bb * (double)tk;
        } else {
            const float gdep = P.gt_depth[ray];
            const float gmax = P.gt_max[0];
            if (k < P.n_samples) {
                const float tk = P.t_uniform[k];
                const double cap = (double)(gmax * 1.2f);
                const double far = tmin(tmax(far_bb, 0.0), cap);
                const float near = gdep * 0.01f;
                z = (double)(near * (1.f - tk)) + far * (double)tk;
            } else {
                const double s = P.t_surface[k - P.n_samples];
                if (gdep > 0.f) {
                    const double e0 = (double)(0.95f * gdep), e1 = (double)(1.05f * gdep);
                    z = e0 * (1.0 - s) + e1 * s;
                } else {
                    z = 0.001 * (1.0 - s) + (double)gmax * s;
                }
            }
        }
    }
    if (P.n_surface > 0 && guided) {
        // rank sort of the S candidates (torch.sort, Renderer.py:168-170): stable, like compute_z
        if (act) zw[lane] = z;
        wave_fence();
        int rank = 0;
#pragma unroll 16
        for (int j = 0; j < S; ++j) {          // unrolled: 16 LDS reads in flight instead of one latency per compare
            const double u = zw[j];
            rank += (u < z || (u == z && j < k)) ? 1 : 0;
        }
        wave_fence();
        if (act) zw[rank] = z;
        wave_fence();
        z = act ? zw[lane] : 0.0;
    }
    if (!act) return;
    const long long gq = ray * S + lane;
    // pts = o + d*z in fp64 (Renderer.py:172-174)
    const double qx = (double)P.rays_o[ray * 3 + 0] + (double)P.rays_d[ray * 3 + 0] * z;
    const double qy = (double)P.rays_o[ray * 3 + 1] + (double)P.rays_d[ray * 3 + 1] * z;
    const double qz = (double)P.rays_o[ray * 3 + 2] + (double)P.rays_d[ray * 3 + 2] * z;
    P.zvals[gq] = z;
    double *pq = P.pd + gq * 4;
    pq[0] = qx; pq[1] = qy; pq[2] = qz; pq[3] = z;
    st4(P.pf + gq * 4, F4{(float)qx, (float)qy, (float)qz, 0.f});
}

// ------------------------------------------------------------------------------------------------
// (2) one decoder pass over the tiles of the sample list
// ------------------------------------------------------------------------------------------------
// Blocks [pass_beg[p], pass_beg[p + 1]) of the launch belong to decoder pass p (more of them for the fine decoder: 288 instead
// of 240 MFMAs per tile and two feature gathers); block i of the n of a pass owns the contiguous tile range
// [T i / n, T (i + 1) / n) and its waves draw tiles from it through an LDS counter -- no rounds, a wave that gets a cheap
// tile (rays that left the bound) simply takes the next one.
template <int KIND, bool SAVE>
NSR_DEV void fwd_pass(const RenderParams &P, int bi, int nbp) {
    float *aux = reinterpret_cast<float *>(lds_base());
    float *wl = aux + AUX_FLOATS;
    int *cnt = reinterpret_cast<int *>(wl + packed_total(KIND));
    const int lane0 = tid() & 63, wave = tid() >> 6;
    const DecDev &D = P.dec[KIND];
    const Dbg dbg{P.dbg ? P.dbg + ((long long)bid_x() * 12 + wave) * 64 : nullptr};
    dbg.stamp(0);
    load_aux<KIND>(aux, D.packed);
    copy_f4_dma<packed_total(KIND) / 4>(wl, D.packed + AUX_FLOATS);      // (global -> LDS DMA, like the dX kernel's stream: round 6)
    if (tid() == 0) cnt[0] = 0;
    dma_wait<0>();
    block_sync();
    dbg.stamp(1);
    const long long ntiles = (P.n_points_total + kTile - 1) / kTile;
    const long long t0 = ntiles * bi / nbp, t1 = ntiles * (bi + 1) / nbp;
    for (;;) {
        int take = 0;
        if (lane0 == 0) take = atomic_fetch_add_lds_i(cnt, 1);
        const long long tile = t0 + shfl_i(take, 0);
        if (tile >= t1) break;
        if (!tile_live(P, tile)) continue;                           // every ray of the tile was removed by the pre-filter
        loop_fence();
        const int lane = opaque_i(lane0), pt = lane & 15, g = lane >> 4;      // (lane-dependent addresses formed per tile: see opaque_i)
        const long long gp = tile * kTile + pt;
        const bool active = gp < P.n_points_total;
        const double *pp = P.pd + (active ? gp : 0) * 4;
        const double px = pp[0], py = pp[1], pz = pp[2];
        dbg.stamp(2);
        const long long sp = (SAVE && active) ? gp : -1;
        float *scr = P.draw + gp * 4;                                // per-sample scratch of the passes (d raw comes later)
        if (KIND == NSR_COARSE) {
            const Lvl L = make_level(P.grid[NSR_COARSE], px, py, pz);
            const Act<2> c = gather_feat(P.grid[NSR_COARSE], L, g);
            float o[1];
            const ActSink sk = act_sink(P, 0, sp, g);
            mlp_nox_fwd<false, SAVE>(wl, aux, c, lane, o, nullptr, &sk);
            if (active && g == 0) scr[0] = o[0];
        } else {
            const float fx = (float)px, fy = (float)py, fz = (float)pz;     // decoder.py:189
            if (KIND == NSR_MIDDLE) {
                const Lvl Lm = make_level(P.grid[NSR_MIDDLE], px, py, pz);
                const Act<2> cm = gather_feat(P.grid[NSR_MIDDLE], Lm, g);
                float om[1];
                const ActSink sk = act_sink(P, 0, sp, g);
                mlp_xyz_fwd<NSR_MIDDLE, false, SAVE>(wl, aux, fx, fy, fz, cm, lane, om, nullptr, &sk);
                if (active && g == 0) scr[0] = om[0];
            } else if (KIND == NSR_FINE) {
                const Lvl Lf = make_level(P.grid[NSR_FINE], px, py, pz);
                const Act<2> cf = gather_feat(P.grid[NSR_FINE], Lf, g);
                const Lvl Lm = make_level(P.grid[NSR_MIDDLE], px, py, pz);
                const Act<2> cm = gather_feat(P.grid[NSR_MIDDLE], Lm, g);
                Act<4> cc;
                cc.t[0] = cf.t[0]; cc.t[1] = cf.t[1]; cc.t[2] = cm.t[0]; cc.t[3] = cm.t[1];    // decoder.py:182-187
                float of[1];
                const ActSink sk = act_sink(P, 1, sp, g);
                mlp_xyz_fwd<NSR_FINE, false, SAVE>(wl, aux, fx, fy, fz, cc, lane, of, nullptr, &sk);
                if (active && g == 0) scr[1] = of[0];
            } else {
                const Lvl Lc = make_level(P.grid[NSR_COLOR], px, py, pz);
                const Act<2> ccol = gather_feat(P.grid[NSR_COLOR], Lc, g);
                float oc[4];
                const ActSink sk = act_sink(P, 2, sp, g);
                mlp_xyz_fwd<NSR_COLOR, false, SAVE>(wl, aux, fx, fy, fz, ccol, lane, oc, nullptr, &sk);
                if (active && g == 0) { float *rw = P.raw + gp * 4; rw[0] = oc[0]; rw[1] = oc[1]; rw[2] = oc[2]; }
            }
        }
        dbg.stamp(3);
    }
    dbg.stamp(9);
}

template <int STAGE, bool SAVE>
NSR_KERNEL NSR_BOUNDS(768) void render_fwd_pass_kernel(const RenderParams P) {
    const int b = bid_x();
    if (STAGE == NSR_STAGE_COARSE) {
        fwd_pass<NSR_COARSE, SAVE>(P, b, nblk_x());
    } else {
        if (b < P.pass_beg[1]) fwd_pass<NSR_MIDDLE, SAVE>(P, b, P.pass_beg[1]);
        else if (b < P.pass_beg[2]) { if (STAGE >= NSR_STAGE_FINE) fwd_pass<NSR_FINE, SAVE>(P, b - P.pass_beg[1], P.pass_beg[2] - P.pass_beg[1]); }
        else { if (STAGE == NSR_STAGE_COLOR) fwd_pass<NSR_COLOR, SAVE>(P, b - P.pass_beg[2], P.pass_beg[3] - P.pass_beg[2]); }
    }
}

// ------------------------------------------------------------------------------------------------
// (3) raw, compositor, outputs, fused loss: one wave per ray
// ------------------------------------------------------------------------------------------------
template <int STAGE>
NSR_KERNEL void fwd_composite_kernel(const RenderParams P) {
    double *red = reinterpret_cast<double *>(lds_base());            // [waves] loss terms of the block
    const int lane = tid() & 63, wave = tid() >> 6, nw = nthreads() >> 6;
    const long long rayq = (long long)bid_x() * nw + wave;
    const int S = P.S;
    double lterm = 0.0;
    if (rayq < P.n_rays && !ray_live(P, rayq)) {                     // removed by the pre-filter: no sample was evaluated
        if (lane == 0) {
            P.depth[rayq] = 0.0; P.var[rayq] = 0.0;
            P.rgb[rayq * 3 + 0] = 0.f; P.rgb[rayq * 3 + 1] = 0.f; P.rgb[rayq * 3 + 2] = 0.f;
            if (P.loss && P.dl_depth) P.dl_depth[rayq] = 0.0;
            if (P.loss && P.dl_rgb) { P.dl_rgb[rayq * 3 + 0] = 0.f; P.dl_rgb[rayq * 3 + 1] = 0.f; P.dl_rgb[rayq * 3 + 2] = 0.f; }
        }
        // its samples inside a tile shared with a kept ray carry no gradient
        if (P.loss && P.dl_depth && lane < S) st4(P.draw + (rayq * S + lane) * 4, F4{0.f, 0.f, 0.f, 0.f});
    } else if (rayq < P.n_rays) {
        const bool act = lane < S;
        const long long gq = rayq * S + (act ? lane : 0);
        // the loss epilogue's inputs are requested HERE, with the sample data: behind the stores below the compiler may not move them
        // (the pointers could alias), and they were a second memory round trip in an 8 us kernel
        const bool kp = !P.loss || !P.keep || P.keep[rayq];
        const float gd = (P.loss && P.loss_depth) ? P.loss_depth[rayq] : 0.f;
        float gtc[3] = {0.f, 0.f, 0.f};
        if (P.loss && STAGE == NSR_STAGE_COLOR && P.gt_color) { gtc[0] = P.gt_color[rayq * 3 + 0]; gtc[1] = P.gt_color[rayq * 3 + 1]; gtc[2] = P.gt_color[rayq * 3 + 2]; }
        double qx = 0.0, qy = 0.0, qz = 0.0, zq = 0.0;
        F4 rw = F4{0.f, 0.f, 0.f, 0.f};
        bool ins = true;
        if (act) {
            const double *pq = P.pd + gq * 4;
            qx = pq[0]; qy = pq[1]; qz = pq[2]; zq = pq[3];
            const float *scr = P.draw + gq * 4;
            float occ = scr[0];
            if (STAGE >= NSR_STAGE_FINE) occ = scr[1] + scr[0];                             // decoder.py:333,341
            if (STAGE == NSR_STAGE_COLOR) { const float *rc = P.raw + gq * 4; rw.x = rc[0]; rw.y = rc[1]; rw.z = rc[2]; }
            ins = (qx > P.blo[0]) && (qx < P.bhi[0]) && (qy > P.blo[1]) && (qy < P.bhi[1]) && (qz > P.blo[2]) && (qz < P.bhi[2]);
            rw.w = ins ? occ : 100.f;                                                       // Renderer.py:57
            st4(P.raw + gq * 4, rw);
        }
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
            // Mapper.py:487-493 on the rays the pre-filter keeps, and its derivative w.r.t. this ray's outputs (see
            // render_fwd_kernel: the same expressions)
            double gD = 0.0;
            float g3[3] = {0.f, 0.f, 0.f};
            if (kp && gd > 0.f) {
                const double df = depth - (double)gd;
                lterm += fabs(df);
                gD = df > 0.0 ? 1.0 : (df < 0.0 ? -1.0 : 0.0);
            }
            if (kp && STAGE == NSR_STAGE_COLOR && P.gt_color) {
                const float e[3] = {cr - gtc[0], cg - gtc[1], cb - gtc[2]};
                lterm += (double)(P.w_color * ((fabsf(e[0]) + fabsf(e[1])) + fabsf(e[2])));
#pragma unroll
                for (int q = 0; q < 3; ++q) g3[q] = e[q] > 0.f ? P.w_color : (e[q] < 0.f ? -P.w_color : 0.f);
            }
            if (lane == 0) {
                if (P.dl_depth) P.dl_depth[rayq] = gD;
                if (P.dl_rgb) { P.dl_rgb[rayq * 3 + 0] = g3[0]; P.dl_rgb[rayq * 3 + 1] = g3[1]; P.dl_rgb[rayq * 3 + 2] = g3[2]; }
            }
            if (P.dl_depth) {
                // d raw per sample for the split backward (comp_bwd_kernel's expressions, d var = 0)
                const double s1 = wave_sum_d((double)c.w * dz);
                const float Gz = (float)(gD * zq + 0.0 * (dz * dz - 2.0 * s1 * zq));
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
                if (!ins) docc = 0.f;
                if (act) st4(P.draw + gq * 4, F4{c.w * g3[0], c.w * g3[1], c.w * g3[2], docc});
            }
        }
    }
    if (P.loss) {                                                    // one atomic per block
        if (lane == 0) red[wave] = lterm;
        block_sync();
        if (tid() == 0) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += red[w];
            if (s != 0.0) atomic_add_global_d(P.loss, s);
        }
    }
}

}  // namespace nsr
