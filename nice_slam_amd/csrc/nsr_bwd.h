// nsr_bwd.h -- backward of the render path (second generation), included by nsr_kernels.h.
//
// Block = 4 waves = one wave per SIMD with the whole 512-entry register file (VGPR + AGPR) of its SIMD.  A wave owns
// whole 16-point tiles and, for the decoder of its pass, the COMPLETE set of parameter-gradient accumulators: every
// 16x16 block of every dW lives in four registers per lane (MFMA accumulators, AGPRs) for the whole kernel and is only
// written out once, at the end of the block's last ray group (cross-wave sum through LDS -> one image per block in the
// global partial buffer -> reduce_partials_kernel).  Consequences against the first generation (owner-computes over
// the tiles of a block, one lock-step phase per layer):
//   * no block barrier inside a tile: a layer's dW = dY^T X only needs THIS wave's operands, re-laid out "lane =
//     channel" through its private LDS staging tile (8 scalar stores + one 16-byte read per 16-channel operand);
//   * no per-ray-group read-modify-write of the gradient image (was 189 MB of L2 writes per launch at 1000 rays);
//   * waves run independently; the only barriers are the three per ray group around the shared sample / d-raw buffers.
// Per-tile work of a wave (decoder with parameter gradients): forward re-run (kept hidden states + relu masks), output
// layer, layers 4..0 { stage dH, dY, layer input -> dc += U^T dH, dU += dH^T c, dW += dY^T x, bias sums, dh = W^T dY },
// embedding stage (dE, d arg, d p, dB; the Fourier-feature blocks W0 / W3e contract against sines recomputed in
// "lane = channel" form), grid scatter, ray gradients.
// References: autograd of Renderer.render_batch_ray (src/utils/Renderer.py:63-198, src/Mapper.py:503, src/Tracker.py:125).
#pragma once

#ifndef NSR_BWD_TILES
#define NSR_BWD_TILES 12    // tiles per ray group (4 rays of 48 samples): three per wave
#endif

namespace nsr {

constexpr int kBwdWaves = 4;

NSR_DEV float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// d0 / d1 += A0^T X, A1^T X over the 16 points of a tile: operands in "lane = channel, element q = point 4g+q" form
NSR_DEV void dw_pair(f32x4 &d0, f32x4 &d1, f32x4 a0, f32x4 a1, f32x4 x) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        d0 = mfma16(a0[q], x[q], d0);
        d1 = mfma16(a1[q], x[q], d1);
    }
}

// ------------------------------------------------------------------------------------------------
// accumulators of one wave for one decoder.  Weight tiles: acc[To][Tk][r] <-> dW[16To + 4g + r][16Tk + (lane & 15)].
// Vectors (bias-like): lane (i, g) holds the partial sum over its four points of channel 16T + i.
// Register budget (gfx950: 256 VGPR + 256 AGPR per lane at one wave per SIMD; every MFMA result lives in AGPRs): the 60
// weight tiles of a c_dim-32 decoder (240 registers) plus the MFMA accumulators of the activation chain do not fit, so
//   * the w3e tiles of the first lds_acc_ktiles(KIND) k-tiles are kept in the wave's LDS staging region and read /
//     written around their MFMAs (they are touched once per tile, next to the sine evaluation),
//   * the gradients of the fc_c biases are not accumulated at all: v_i = sum_p dH_i[p] and dH_{i-1}[p] = W_i^T dY_i[p], so
//     v_{i-1} = W_i^T b_i (and v_4 = Wo^T bo) -- five 32x32 matrix-vector products per block at flush time.
// ------------------------------------------------------------------------------------------------
//   * the c_dim-64 (fine) decoder has 80 weight tiles.  Its 40 fc_c tiles (dU_i = dH_i^T [c_fine | c_mid]) are
//     OWNER-COMPUTED inside the block: wave w owns feature k-tile w of every layer (2 tiles per layer, 10 in all) and
//     contracts over the tiles of all four waves, whose dH / c staging tiles it reads straight from their LDS regions.
//     That costs the fine pass one block barrier per layer (dH staged ping-pong, so one barrier suffices) and nothing
//     else: no extra registers, no memory traffic, no reduction at flush (exclusive owner).
constexpr int lds_acc_ktiles(int kind) { return kind == 0 ? 0 : (cdim_of(kind) == 32 ? kET : 0); }
constexpr bool owner_du(int kind) { return kind != 0 && cdim_of(kind) == 64; }
constexpr int st_lacc(int kind) { return stg_floats(kind); }          // LDS-resident accumulator tiles: behind the C region
constexpr int bwd_stg_floats(int kind) { return stg_floats(kind) + lds_acc_ktiles(kind) * 512 + (owner_du(kind) ? 512 : 0); }   // + second dH tile
template <int CD>
struct XyzAcc {
    static constexpr int NTC = CD / 16;
    f32x4 w0[2][kET], w3e[2][kET];
    f32x4 wh[4][2][2];            // W1, W2, W3h, W4
    f32x4 u[5][2][NTC];
    float b[5][2];
    float wo[4][2], bo[4];
    float B[kET][3];
};
struct NoxAcc {
    f32x4 w[6][2][2];             // NW0, NW1, NW2, NW3C, NW3H, NW4
    float b[5][2];
    float wo[2], bo;
};

template <int CD>
NSR_DEV void acc_zero(XyzAcc<CD> &A) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int k = 0; k < kET; ++k) { A.w0[o][k] = f4zero(); A.w3e[o][k] = f4zero(); }
#pragma unroll
        for (int m = 0; m < 4; ++m) { A.wh[m][o][0] = f4zero(); A.wh[m][o][1] = f4zero(); }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int k = 0; k < XyzAcc<CD>::NTC; ++k) A.u[i][o][k] = f4zero();
            A.b[i][o] = 0.f;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) A.wo[n][o] = 0.f;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) A.bo[n] = 0.f;
#pragma unroll
    for (int k = 0; k < kET; ++k) { A.B[k][0] = 0.f; A.B[k][1] = 0.f; A.B[k][2] = 0.f; }
}
NSR_DEV void acc_zero(NoxAcc &A) {
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int o = 0; o < 2; ++o) { A.w[m][o][0] = f4zero(); A.w[m][o][1] = f4zero(); }
#pragma unroll
    for (int i = 0; i < 5; ++i) { A.b[i][0] = 0.f; A.b[i][1] = 0.f; }
    A.wo[0] = A.wo[1] = 0.f;
    A.bo = 0.f;
}

// Visitors: f(acc tile, Mat, To, Tk, running tile index) / g(vector accumulator, flat offset of its channel 0, valid
// channels, running index).  Everything is unrolled, so the running indices are compile-time constants at each call.
template <int KIND, class F>
NSR_DEV void visit_tiles(XyzAcc<cdim_of(KIND)> &A, F &&f) {
    constexpr int CD = cdim_of(KIND), NTC = CD / 16;
    int t = 0;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int k = 0; k < kET; ++k) { f(A.w0[o][k], xyz_mat(CD, XW0), o, k, t); ++t; }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int k = 0; k < kET; ++k) { f(A.w3e[o][k], xyz_mat(CD, XW3E), o, k, t); ++t; }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < 2; ++k) { f(A.wh[m][o][k], xyz_mat(CD, m == 0 ? XW1 : (m == 1 ? XW2 : (m == 2 ? XW3H : XW4))), o, k, t); ++t; }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < NTC; ++k) { f(A.u[i][o][owner_du(KIND) ? 0 : k], xyz_mat(CD, i == 0 ? XU0 : (i == 1 ? XU1 : (i == 2 ? XU2 : (i == 3 ? XU3 : XU4)))), o, k, t); ++t; }
}
constexpr int xyz_ntiles(int cd) { return 4 * kET + 16 + 10 * (cd / 16); }
template <int KIND, class G>
NSR_DEV void visit_vecs(XyzAcc<cdim_of(KIND)> &A, G &&g) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            g(A.b[i][o], bias_off(KIND, i) + 16 * o, 16, t); ++t;
        }
#pragma unroll
    for (int n = 0; n < nout_of(KIND); ++n) {
        g(A.wo[n][0], wo_off(KIND) + n * 32, 16, t); ++t;
        g(A.wo[n][1], wo_off(KIND) + n * 32 + 16, 16, t); ++t;
        g(A.bo[n], bo_off(KIND) + n, 1, t); ++t;
    }
#pragma unroll
    for (int k = 0; k < kET; ++k)
#pragma unroll
        for (int d = 0; d < 3; ++d) { g(A.B[k][d], B_off(KIND) + d * kE + 16 * k, kE - 16 * k < 16 ? kE - 16 * k : 16, t); ++t; }
}
constexpr int xyz_nvecs(int kind) { return 10 + 3 * nout_of(kind) + 3 * kET; }

template <int KIND, class F>
NSR_DEV void visit_tiles(NoxAcc &A, F &&f) {
    int t = 0;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < 2; ++k) { f(A.w[m][o][k], nox_mat(m), o, k, t); ++t; }
}
template <int KIND, class G>
NSR_DEV void visit_vecs(NoxAcc &A, G &&g) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o) { g(A.b[i][o], nox_b(i) + 16 * o, 16, t); ++t; }
    g(A.wo[0], nox_wo(), 16, t); ++t;
    g(A.wo[1], nox_wo() + 16, 16, t); ++t;
    g(A.bo, nox_bo(), 1, t); ++t;
}
constexpr int ntiles_of(int kind) { return kind == 0 ? 24 : xyz_ntiles(cdim_of(kind)); }
constexpr int nvecs_of(int kind) { return kind == 0 ? 13 : xyz_nvecs(kind); }

// ------------------------------------------------------------------------------------------------
// end of the block: sum the four waves' accumulators through LDS (`red`: the region that held the packed weights) and
// write the block's image of the flat gradient blob (plain stores; every parameter exactly once).
// ------------------------------------------------------------------------------------------------
template <int KIND, class ACC>
NSR_DEV void flush_acc(ACC &A, float *wl, const float *aux, float *scratch, float *img, int wave, int lane, const Dbg dbg) {
    const Stream st = make_stream(img);
    const int j = lane & 15, g = lane >> 4;
    // ---- vectors first (the packed weights in `wl` are still needed): reduce over the four point groups of the lane's
    // channel, then over the waves through `scratch` (the staging regions; nobody stages any more)
    constexpr int NV = nvecs_of(KIND);
    constexpr int kSum = NV * kBwdWaves * 16;                            // bsum[5][32] | bosum[4] behind the vector slots
    static_assert(kSum + 164 <= kBwdWaves * stg_floats(0), "scratch too small");
    dbg.stamp(56);
    block_sync();                                                        // every wave is done with its staging region
    dbg.stamp(57);
    {
        float part[NV];                                                  // all vectors at once: the cross-lane adds pipeline
        visit_vecs<KIND>(A, [&](float &acc, int, int, int t) { part[t] = acc; });
#pragma unroll
        for (int t = 0; t < NV; ++t) part[t] += shfl_xor(part[t], 16);
#pragma unroll
        for (int t = 0; t < NV; ++t) part[t] += shfl_xor(part[t], 32);
        if (g == 0) {
#pragma unroll
            for (int t = 0; t < NV; ++t) scratch[(t * kBwdWaves + wave) * 16 + j] = part[t];
        }
    }
    block_sync();
    visit_vecs<KIND>(A, [&](float &, int off, int nvalid, int t) {
        if ((t % kBwdWaves) == wave && g == 0 && j < nvalid) {
            const float *src = scratch + t * kBwdWaves * 16 + j;
            const float v = (src[0] + src[16]) + (src[32] + src[48]);
            stream_st(st, j, off, v);
            if (KIND != NSR_COARSE) {
                if (t < 10) scratch[kSum + 16 * t + j] = v;                  // b[i][o]: visited first, t = 2 i + o
                if (off >= bo_off(KIND)) scratch[kSum + 160 + (off - bo_off(KIND))] = v;
            }
        }
    });
    if (KIND != NSR_COARSE) {
        // v_{i-1} = W_i^T b_i  (i = 1..4; W_3: the hidden-state columns),  v_4 = Wo^T bo      (see XyzAcc); W from the packed
        // stream in LDS: element W[o][k] of a slice sits at ((Tk*2 + (o>>4))*64 + (o&15) + 16*((k>>2)&3))*4 + (k&3)
        block_sync();
        const int t = tid();
        if (t < 160) {
            constexpr int KX = KIND == NSR_COARSE ? NSR_MIDDLE : KIND;
            constexpr int CD = cdim_of(KX);
            const int i = t >> 5, k = t & 31;
            float v = 0.f;
            if (i < 4) {
                const Mat m = xyz_mat(CD, i == 0 ? XW1 : (i == 1 ? XW2 : (i == 2 ? XW3H : XW4)));
                const float *w = wl + m.pk + (k >> 4) * 512 + 64 * ((k >> 2) & 3) + (k & 3);
                const float *bs = scratch + kSum + 32 * (i + 1);
                float wv[32], bv[32];
#pragma unroll
                for (int o = 0; o < 32; ++o) { wv[o] = w[(o >> 4) * 256 + (o & 15) * 4]; bv[o] = bs[o]; }     // 64 LDS reads in flight
#pragma unroll
                for (int o = 0; o < 32; ++o) v = fmaf(wv[o], bv[o], v);
            } else {
                for (int n = 0; n < nout_of(KX); ++n) v = fmaf(aux[AUX_WO + n * 32 + k], scratch[kSum + 160 + n], v);
            }
            stream_st(st, t & 31, fcb_off(KX, i), v);
        }
    }
    // ---- weight tiles: sum the four waves' accumulators through LDS (from `wl`, the region that held the packed weights, on) and
    // write the block's image of the flat gradient blob (plain stores; every parameter exactly once)
    // (everything from the packed weights to the end of the staging regions is free now: sample / d-raw buffers included)
    constexpr int RT = (packed_total(KIND) + kBwdWaves * bwd_stg_floats(KIND)) / (kBwdWaves * 256);   // tiles per round that fit
    constexpr int NT = ntiles_of(KIND), NR = (NT + RT - 1) / RT;
    static_assert(RT >= 1, "reduction buffer too small");
    float *red = wl;
    dbg.stamp(58);
    block_sync();                                                        // everyone is done with the packed weights
    dbg.stamp(59);
#pragma unroll
    for (int R = 0; R < NR; ++R) {
        constexpr int t_u = 4 * kET + 16;                                // first fc_c tile in visiting order
        visit_tiles<KIND>(A, [&](f32x4 &acc, const Mat, int, int, int t) {
            if (t / RT == R && !(owner_du(KIND) && t >= t_u)) st4(red + (((t % RT) * kBwdWaves + wave) * 64 + lane) * 4, to_F4(acc));
        });
        block_sync();
        visit_tiles<KIND>(A, [&](f32x4 &acc, const Mat m, int To, int Tk, int t) {
            const bool owned = owner_du(KIND) && t >= t_u;               // fine dU: wave Tk holds the complete sum already
            if (t / RT == R && (owned ? Tk == wave : (t % kBwdWaves) == wave)) {
                f32x4 s = acc;
                if (!owned) {
                    const float *src = red + ((t % RT) * kBwdWaves * 64 + lane) * 4;
                    s = to_v(ld4(src));
#pragma unroll
                    for (int w = 1; w < kBwdWaves; ++w) s += to_v(ld4(src + w * 256));
                }
                if (16 * Tk + j < m.kcols) {
                    const int lo = 4 * g * m.stride + j, co = m.off + m.kbeg + 16 * Tk + 16 * To * m.stride;
#pragma unroll
                    for (int r = 0; r < 4; ++r) stream_st(st, lo, co + r * m.stride, s[r]);
                }
            }
        });
        if (R + 1 < NR) block_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// xyz decoder, one tile
// ------------------------------------------------------------------------------------------------
template <int KIND, bool PARAMS>
struct XyzTile {
    static constexpr int CD = cdim_of(KIND), NTC = CD / 16;
    const float *wl;     // packed operand stream of this decoder (LDS)
    const float *aux;
    float *S;            // this wave's staging region
    XyzAcc<CD> &A;
    const Kept<KIND> &K;
    BwdFlags F;
    int lane;
    Act<2> &dc;
    Act<2> &dh;
    const float *stg;    // staging regions of all waves (owner-computed dU of the fine decoder)
    int wave;
    Act<2> dY3, dY0;

    template <int I>
    NSR_DEV void layer() {
        const int i16 = lane & 15, g = lane >> 4;
        constexpr bool kOwn = PARAMS && owner_du(KIND);
        constexpr int kA0 = kOwn ? (I & 1 ? kStA0 : stg_floats(KIND) + lds_acc_ktiles(KIND) * 512) : kStA0;      // dH tile, ping-pong
        constexpr int uid = I == 0 ? XU0 : (I == 1 ? XU1 : (I == 2 ? XU2 : (I == 3 ? XU3 : XU4)));
        constexpr int hid = I == 1 ? XW1 : (I == 2 ? XW2 : (I == 3 ? XW3H : XW4));
        constexpr int hacc = I == 1 ? 0 : (I == 2 ? 1 : (I == 3 ? 2 : 3));
        const Mat mu = xyz_mat(CD, uid);
        if (PARAMS) st_store(S + kA0, dh, i16, g);                           // dH_i: gradient of (U_i c + v_i) is dh itself
        if (F.grid || F.rays) gemv_bwd<2>(dc.t, dh, wl + mu.pk, i16, g);      // first 32 feature columns only
        const Act<2> dY = apply_mask(dh, K.mask[I]);
        if (I == 3) dY3 = dY;
        if (I == 0) dY0 = dY;
        if (PARAMS) {
            st_store(S + kStA1, dY, i16, g);
            if (I > 0) st_store(S + kStX0, K.h[I > 0 ? I - 1 : 0], i16, g);
            else st_store(S + kStX0, dY3, i16, g);                           // layer 0: the W0 / W3e blocks need dY3 too
            if (kOwn) block_sync(); else wave_fence();                       // owner-computed dU: every wave's dH tile is staged
            const f32x4 y0 = st_load_cm(S + kStA1, 0, i16, g), y1 = st_load_cm(S + kStA1, 1, i16, g);
            A.b[I][0] += sum4(y0); A.b[I][1] += sum4(y1);
            if (kOwn) {
                // this wave's two dU tiles of the layer (feature k-tile = wave): contract over the tiles of all four waves
#pragma unroll
                for (int w = 0; w < kBwdWaves; ++w) {
                    const float *So = stg + w * bwd_stg_floats(KIND);
                    const f32x4 a0 = st_load_cm(So + kA0, 0, i16, g), a1 = st_load_cm(So + kA0, 1, i16, g);
                    const f32x4 cc = st_load_cm(So + kStC + (wave >> 1) * 512, wave & 1, i16, g);
                    dw_pair(A.u[I][0][0], A.u[I][1][0], a0, a1, cc);
                }
            } else {
                const f32x4 a0 = st_load_cm(S + kA0, 0, i16, g), a1 = st_load_cm(S + kA0, 1, i16, g);
#pragma unroll
                for (int Tc = 0; Tc < NTC; ++Tc) {
                    const f32x4 cc = st_load_cm(S + kStC + (Tc >> 1) * 512, Tc & 1, i16, g);
                    dw_pair(A.u[I][0][Tc], A.u[I][1][Tc], a0, a1, cc);
                }
            }
            if (I > 0) {
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk) {
                    const f32x4 x = st_load_cm(S + kStX0, Tk, i16, g);
                    dw_pair(A.wh[hacc][0][Tk], A.wh[hacc][1][Tk], y0, y1, x);
                }
            } else {
                // W0 and W3e read the Fourier embedding: its 16 channels of k-tile Tk are recomputed for the lane's four
                // points (decoder.py:26-30) and contracted with dY0 and dY3
                const f32x4 q0 = st_load_cm(S + kStX0, 0, i16, g), q1 = st_load_cm(S + kStX0, 1, i16, g);
                const f32x4 px = to_v(ld4(S + kStP + 4 * g)), py = to_v(ld4(S + kStP + 16 + 4 * g)), pz = to_v(ld4(S + kStP + 32 + 4 * g));
#pragma unroll
                for (int Tk = 0; Tk < kET; ++Tk) {
                    const F4 b = load_b1(aux, 16 * Tk + i16);
                    const f32x4 e = sin_acc4(vfma(pz, splat(b.z), vfma(py, splat(b.y), px * splat(b.x))));
                    dw_pair(A.w0[0][Tk], A.w0[1][Tk], y0, y1, e);
                    if (Tk < lds_acc_ktiles(KIND)) {               // LDS-resident accumulator pair (see XyzAcc)
                        float *La = S + st_lacc(KIND) + Tk * 512 + lane * 4;
                        f32x4 t0 = to_v(ld4(La)), t1 = to_v(ld4(La + 256));
                        dw_pair(t0, t1, q0, q1, e);
                        st4(La, to_F4(t0)); st4(La + 256, to_F4(t1));
                    } else {
                        dw_pair(A.w3e[0][Tk], A.w3e[1][Tk], q0, q1, e);
                    }
                }
            }
            wave_fence();
        }
        if (I > 0) {
            Act<2> nd;
            act_zero(nd);
            gemv_bwd<2>(nd.t, dY, wl + xyz_mat(CD, hid).pk, i16, g);
            dh = nd;
        }
    }
};

// c: features (CL).  dr: d raw of this lane's point (occupancy gradient already cut outside the bound).
// dc: gradient w.r.t. the first 32 feature channels (the decoder's own grid).  dp: gradient w.r.t. the fp32 world position
// through the embedding (already reduced over g).
// With owner_du(KIND) && PARAMS every wave of the block must call this function (one block barrier per layer).
// `mid`: called once the hidden-state chain is done (before the embedding stage); the pass issues the NEXT tile's feature
// gather there, so that its latency hides behind the embedding stage.
template <int KIND, bool PARAMS, class Mid>
NSR_DEV void xyz_bwd_tile(const float *pk, const float *aux, float *S, XyzAcc<cdim_of(KIND)> &A,
                          float px, float py, float pz, const Act<cdim_of(KIND) / 16> &c,
                          const F4 dr, BwdFlags F, int lane, Act<2> &dc, float (&dp)[3], const Dbg dbg, int ts, Mid &&mid,
                          const float *stg, int wave) {
    constexpr int CD = cdim_of(KIND), NOUT = nout_of(KIND), NTC = CD / 16;
    const int i16 = lane & 15, g = lane >> 4;
    Kept<KIND> K;
    {
        float out[NOUT];
        mlp_xyz_fwd<KIND, true>(pk, aux, px, py, pz, c, lane, out, &K);
        (void)out;
    }
    dbg.stamp(ts + 1);
    float d_out[4];
    if (NOUT == 1) { d_out[0] = dr.w; d_out[1] = d_out[2] = d_out[3] = 0.f; }
    else { d_out[0] = dr.x; d_out[1] = dr.y; d_out[2] = dr.z; d_out[3] = 0.f; }   // decoder.py:341 overwrites the 4th colour output

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
    if (PARAMS) {
        // owner-computed dU: the other waves may still be reading this wave's C / dH tiles of the previous round
        if (owner_du(KIND)) block_sync();
        if (g == 0) {
            S[kStP + i16] = px; S[kStP + 16 + i16] = py; S[kStP + 32 + i16] = pz;       // [xyz][16 points]
#pragma unroll
            for (int n = 0; n < NOUT; ++n) S[kStDO + n * 16 + i16] = d_out[n];         // [output][16 points]
        }
#pragma unroll
        for (int q = 0; q < NTC / 2; ++q) {
            Act<2> cq;
            cq.t[0] = c.t[2 * q]; cq.t[1] = c.t[2 * q + 1];
            st_store(S + kStC + q * 512, cq, i16, g);
        }
        st_store(S + kStX0, K.h[4], i16, g);
        wave_fence();
        const f32x4 h0 = st_load_cm(S + kStX0, 0, i16, g), h1 = st_load_cm(S + kStX0, 1, i16, g);
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            const f32x4 d = to_v(ld4(S + kStDO + n * 16 + 4 * g));
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0 = fmaf(d[q], h0[q], s0); s1 = fmaf(d[q], h1[q], s1); }
            A.wo[n][0] += s0; A.wo[n][1] += s1; A.bo[n] += sum4(d);
        }
        wave_fence();
    }

    act_zero(dc);
    dbg.stamp(ts + 2);
    XyzTile<KIND, PARAMS> X{pk, aux, S, A, K, F, lane, dc, dh, stg, wave};
    act_zero(X.dY3);
    act_zero(X.dY0);
    X.template layer<4>(); dbg.stamp(ts + 3);
    X.template layer<3>(); dbg.stamp(ts + 4);
    X.template layer<2>(); dbg.stamp(ts + 5);
    X.template layer<1>(); dbg.stamp(ts + 6);
    X.template layer<0>(); dbg.stamp(ts + 7);
    const Act<2> dY3 = X.dY3, dY0 = X.dY0;
    mid();
    sched_fence();                                   // keep the loads issued by `mid` above the embedding stage

    // ---- embedding: dE = W0^T dY0 + W3e^T dY3 ; d arg = dE * cos(arg)
    dp[0] = dp[1] = dp[2] = 0.f;
    if (F.rays || PARAMS) {
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
            sched_fence_emb();
            dE += dE2;
            const B4 b = load_b4(aux, 4 * Tk + g);
            const f32x4 darg = dE * cos_acc4(vfma(splat(pz), b.z, vfma(splat(py), b.y, splat(px) * b.x)));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ax = fmaf(darg[r], b.x[r], ax); ay = fmaf(darg[r], b.y[r], ay); az = fmaf(darg[r], b.z[r], az);
            }
            if (PARAMS) st4(S + kStA0 + i16 * 96 + 16 * Tk + 4 * g, to_F4(darg));     // [16][96] over A0|A1|X0
        }
        dp[0] = red_g(ax); dp[1] = red_g(ay); dp[2] = red_g(az);
    }
    dbg.stamp(ts + 8);
    if (PARAMS) {
        // Fourier matrix: dB[d][ch] += sum_p darg[p][ch] * p[p][d]; lane (j, pg) takes channel 16Tk + j, points 4pg..4pg+3
        wave_fence();
        const f32x4 qx = to_v(ld4(S + kStP + 4 * g)), qy = to_v(ld4(S + kStP + 16 + 4 * g)), qz = to_v(ld4(S + kStP + 32 + 4 * g));
#pragma unroll
        for (int Tk = 0; Tk < kET; ++Tk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = S[kStA0 + (4 * g + q) * 96 + 16 * Tk + i16];
                A.B[Tk][0] = fmaf(v, qx[q], A.B[Tk][0]); A.B[Tk][1] = fmaf(v, qy[q], A.B[Tk][1]); A.B[Tk][2] = fmaf(v, qz[q], A.B[Tk][2]);
            }
        }
        wave_fence();
    }
    dbg.stamp(ts + 9);
}

// ------------------------------------------------------------------------------------------------
// coarse decoder (MLP_no_xyz), one tile
// ------------------------------------------------------------------------------------------------
template <bool PARAMS>
struct NoxTile {
    const float *wl;
    float *S;
    NoxAcc &A;
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
        constexpr int mid = I == 0 ? NW0 : (I == 1 ? NW1 : (I == 2 ? NW2 : (I == 3 ? NW3H : NW4)));
        const Mat mh = nox_mat(mid);
        if (PARAMS) {
            st_store(S + kStA1, dY, i16, g);
            st_store(S + kStX0, I == 0 ? c : K.h[I > 0 ? I - 1 : 0], i16, g);
            wave_fence();
            const f32x4 y0 = st_load_cm(S + kStA1, 0, i16, g), y1 = st_load_cm(S + kStA1, 1, i16, g);
            A.b[I][0] += sum4(y0); A.b[I][1] += sum4(y1);
            if (I == 3) {
#pragma unroll
                for (int Tk = 0; Tk < 2; ++Tk) dw_pair(A.w[NW3C][0][Tk], A.w[NW3C][1][Tk], y0, y1, st_load_cm(S + kStC, Tk, i16, g));
            }
#pragma unroll
            for (int Tk = 0; Tk < 2; ++Tk) dw_pair(A.w[mid][0][Tk], A.w[mid][1][Tk], y0, y1, st_load_cm(S + kStX0, Tk, i16, g));
            wave_fence();
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

template <bool PARAMS, class Mid>
NSR_DEV void nox_bwd_tile(const float *pk, const float *aux, float *S, NoxAcc &A, const Act<2> &c, const F4 dr,
                          BwdFlags F, int lane, Act<2> &dc, Mid &&mid) {
    const int i16 = lane & 15, g = lane >> 4;
    Kept<0> K;
    float out[1];
    mlp_nox_fwd<true>(pk, aux, c, lane, out, &K);
    (void)out;
    const float d_out = dr.w;
    Act<2> dh;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const F4 w = ld4(aux + AUX_WO + 16 * T + 4 * g);
        f32x4 v = {w.x * d_out, w.y * d_out, w.z * d_out, w.w * d_out};
        dh.t[T] = v;
    }
    if (PARAMS) {
        if (g == 0) S[kStDO + i16] = d_out;
        st_store(S + kStC, c, i16, g);
        st_store(S + kStX0, K.h[4], i16, g);
        wave_fence();
        const f32x4 h0 = st_load_cm(S + kStX0, 0, i16, g), h1 = st_load_cm(S + kStX0, 1, i16, g);
        const f32x4 d = to_v(ld4(S + kStDO + 4 * g));
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { s0 = fmaf(d[q], h0[q], s0); s1 = fmaf(d[q], h1[q], s1); }
        A.wo[0] += s0; A.wo[1] += s1; A.bo += sum4(d);
        wave_fence();
    }
    act_zero(dc);
    NoxTile<PARAMS> X{pk, S, A, c, K, F, lane, dc, dh};
    X.template layer<4>();
    X.template layer<3>();
    mid();
    sched_fence();
    X.template layer<2>();
    X.template layer<1>();
    X.template layer<0>();
}

template <int KIND> struct AccOf { typedef XyzAcc<cdim_of(KIND)> type; };
template <> struct AccOf<0> { typedef NoxAcc type; };

// ------------------------------------------------------------------------------------------------
// backward kernel.  grid = (blocks, passes); pass p handles one decoder:
//   coarse stage: p0 = coarse.   otherwise: p0 = middle, p1 = fine, p2 = color.
// LDS: aux[AUX] | packed weights of the decoder | ztmp f64[npts] | zbuf f64[npts] | draw F4[npts] | dpb f64[npts*3]
//      | rays f32[rays_per_block][6] | per-wave staging regions (bwd_stg_floats(KIND) each)
// PARAMS is the compile-time twin of "this decoder's dparams != NULL" (tracking and decoders the caller does not want
// gradients for run without accumulators and staging).
// Memory-operation order inside a wave: a tile's grid-scatter atomics are fire-and-forget, but the vector-memory counter
// is in-order, so any LOAD issued after them would wait for all of them.  Hence the next tile's feature gather is issued
// and consumed BEFORE the current tile's scatter; rays and sample depths of the group sit in LDS; the atomics then drain
// behind the next tile's forward re-run.  To keep that gather short, the `mid` hook (behind the hidden-state chain) already
// touches one word of each of its 8 (fine: 16) voxel lines: the lines travel to the L2 during the embedding stage at the
// price of 8 registers instead of the 64 a full early gather would hold.
// ------------------------------------------------------------------------------------------------
struct TileCtx {
    int pidx;
    bool active, inside;
    float px, py, pz;        // float32(p): what the embedding sees (decoder.py:189)
    Lvl L;
};

template <int KIND, bool PARAMS>
NSR_DEV void bwd_pass(const RenderParams &P) {
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
    float *rayb = reinterpret_cast<float *>(dpb + 3 * npts);
    const int stg_off = (head * 4 + npts * (8 + 8 + 16 + 24) + P.rays_per_block * 24 + 15) & ~15;
    float *stg = reinterpret_cast<float *>(lds + stg_off);
    float *Sw = stg + wave * bwd_stg_floats(KIND);         // this wave's staging region (also Tx / tab of the scatter)

    const GridDev &G = P.grid[KIND];
    const DecDev &D = P.dec[KIND];
    BwdFlags F;
    F.grid = G.dfeat != nullptr;
    F.params = PARAMS;
    F.rays = P.d_rays_o != nullptr;
    if (!F.grid && !F.params && !F.rays) return;

    const Dbg dbg{P.dbg ? P.dbg + (((long long)bid_y() * nblk_x() + bid_x()) * kBwdWaves + wave) * 64 : nullptr};
    dbg.stamp(0);
#ifdef NSR_TS
    dbg.note(61, KIND + 1);                                  // the probe's key: which pass this block served, how many blocks it had
    dbg.note(62, nblk_x());
#endif
    typename AccOf<KIND>::type A;
    if (PARAMS) {
        acc_zero(A);
        for (int t = lane; t < lds_acc_ktiles(KIND) * 512; t += 64) Sw[st_lacc(KIND) + t] = 0.f;      // this wave's LDS-resident tiles
    }
    const int g = lane >> 4;

    // sample point of a tile: position, bound test, trilinear cell of this pass's grid
    auto setup = [&](TileCtx &T, int tile) {
        T.pidx = tile * kTile + (lane & 15);
        const int r = T.pidx / S;
        T.active = (T.pidx < npts) && (r < P.rays_per_block) && (rayb[r * 6] == rayb[r * 6]);      // NaN origin marks a ray beyond n_rays
        const int rr = T.active ? r : 0;
        const double zt = T.active ? zbuf[T.pidx] : 0.0;
        const double ox = T.active ? (double)rayb[rr * 6 + 0] : 0.0, oy = T.active ? (double)rayb[rr * 6 + 1] : 0.0, oz = T.active ? (double)rayb[rr * 6 + 2] : 0.0;
        const double px = ox + (double)rayb[rr * 6 + 3] * zt, py = oy + (double)rayb[rr * 6 + 4] * zt, pz = oz + (double)rayb[rr * 6 + 5] * zt;
        T.inside = (px > P.blo[0]) && (px < P.bhi[0]) && (py > P.blo[1]) && (py < P.bhi[1]) && (pz > P.blo[2]) && (pz < P.bhi[2]);
        T.L = make_level(G, px, py, pz);
        T.px = (float)px; T.py = (float)py; T.pz = (float)pz;
        return make_level(P.grid[KIND == NSR_FINE ? NSR_MIDDLE : KIND], px, py, pz);     // fine: the cell of the middle grid too
    };

    for (long long grp = bid_x(); grp < P.n_groups; grp += nblk_x()) {
        loop_fence();
        const long long ray0 = grp * P.rays_per_block;
        // Every independent global load of the group is issued before anything waits (a short-lived block pays a TLB miss
        // on most first touches): rays and saved sample depths (one element per thread), the compositor inputs of this
        // wave's ray, and -- first group only -- the aux table + operand stream copy.
        const int t0 = tid();
        const bool has_ray = t0 < P.rays_per_block * 6, has_z = P.zvals && t0 < npts;
        float ray_v = __builtin_nanf("");                       // NaN origin: no such ray
        if (has_ray) {
            const long long ray = ray0 + t0 / 6;
            const int q = t0 % 6;
            if (ray < P.n_rays) ray_v = q < 3 ? P.rays_o[ray * 3 + q] : P.rays_d[ray * 3 + q - 3];
        }
        double z_v = 0.0;
        if (has_z && ray0 + t0 / S < P.n_rays) z_v = P.zvals[ray0 * S + t0];
        const long long cray = ray0 + wave;                      // the ray this wave composites (rays_per_block <= waves handled below)
        const bool c_ok = wave < P.rays_per_block && cray < P.n_rays;
        const bool c_act = lane < S;
        F4 c_rw = F4{0.f, 0.f, 0.f, 0.f};
        double c_gD = 0.0, c_gV = 0.0, c_dep = 0.0;
        float c_gr = 0.f, c_gg = 0.f, c_gb = 0.f;
        const double gsc = P.g_scale ? P.g_scale[0] : 1.0;      // incoming gradient of a fused loss node (nsr_bwd_args.grad_scale)
        if (c_ok) {
            if (c_act) c_rw = ld4(P.raw + (cray * S + lane) * 4);
            if (P.d_depth) c_gD = P.d_depth[cray] * gsc;
            if (P.d_var) c_gV = P.d_var[cray] * gsc;
            if (P.d_rgb) { c_gr = P.d_rgb[cray * 3 + 0] * (float)gsc; c_gg = P.d_rgb[cray * 3 + 1] * (float)gsc; c_gb = P.d_rgb[cray * 3 + 2] * (float)gsc; }
            c_dep = P.g_depth[cray];
        }
#ifdef NSR_X_EARLYWAIT
        if (has_ray) rayb[t0] = ray_v;
        if (has_z) zbuf[t0] = z_v;
        if (c_act) draw[lane] = F4{c_rw.x + (float)c_gD, c_rw.y + (float)c_gV, c_rw.z + c_gr + c_gg + c_gb, c_rw.w + (float)c_dep};
        dbg.stamp(60);
#endif
        if (grp == (long long)bid_x()) {
            copy_f4<(AUX_FLOATS + packed_total(KIND)) / 4>(aux, D.packed);      // visible after the barrier below (8 or 11 loads in
                                                                                 // flight per thread instead of 4: measured, no gain)
            dbg.stamp(1);
        }
        if (has_ray) rayb[t0] = ray_v;
        if (P.zvals) {
            if (has_z) zbuf[t0] = z_v;
            for (int t = t0 + nthreads(); t < npts; t += nthreads())           // (more sample points than threads: generic tail)
                zbuf[t] = (ray0 + t / S < P.n_rays) ? P.zvals[ray0 * S + t] : 0.0;
            block_sync();
        } else {
            block_sync();
            compute_z(P, ray0, ztmp, zbuf);
        }
        dbg.stamp(2);
        // ---- first tile: set up and issue its feature gather; the loads fly while the compositor runs
        TileCtx cur;
        GatherRaw raw;
        Act<2> c, cm;
        {
            const Lvl Lm = setup(cur, wave);
            if (KIND == NSR_FINE) { gather_issue(raw, P.grid[NSR_MIDDLE], Lm, g); cm = gather_finish(raw, Lm); }
            gather_issue(raw, G, cur.L, g);
        }
        // ---- compositor backward: d raw per sample (common.py:231-244 differentiated), one wave per ray
        for (int r = wave; r < P.rays_per_block; r += nwaves) {
            const long long ray = ray0 + r;
            if (ray >= P.n_rays) break;
            const bool act = lane < S;
            F4 rw = c_rw;
            double gD = c_gD, gV = c_gV, dep = c_dep;
            float gr = c_gr, gg = c_gg, gb = c_gb;
            if (r != wave) {                                    // further rays of this wave (more rays than waves per group)
                rw = act ? ld4(P.raw + (ray * S + lane) * 4) : F4{0.f, 0.f, 0.f, 0.f};
                gD = P.d_depth ? P.d_depth[ray] * gsc : 0.0;
                gV = P.d_var ? P.d_var[ray] * gsc : 0.0;
                gr = gg = gb = 0.f;
                if (P.d_rgb) { gr = P.d_rgb[ray * 3 + 0] * (float)gsc; gg = P.d_rgb[ray * 3 + 1] * (float)gsc; gb = P.d_rgb[ray * 3 + 2] * (float)gsc; }
                dep = P.g_depth[ray];
            }
            const double z = act ? zbuf[r * S + lane] : 0.0;
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
            const float docc = 10.f * (dalpha * ((1.f - cw.alpha) * cw.alpha));
            if (act) draw[r * S + lane] = F4{cw.w * gr, cw.w * gg, cw.w * gb, docc};
        }
        c = gather_finish(raw, cur.L);
        dbg.stamp(3);
        block_sync();                                           // d raw of the whole group is in LDS
        dbg.stamp(4);
        // ---- decoder backward, tile by tile.  With the owner-computed dU (fine pass) the tile function contains block
        // barriers: all four waves run the same number of rounds, tiles beyond the group are inactive (zero gradients in)
        constexpr bool kLockstep = PARAMS && owner_du(KIND);
        const int tile_end = kLockstep ? ((P.tiles_per_block + nwaves - 1) / nwaves) * nwaves : P.tiles_per_block;
        for (int tile = wave; tile < tile_end; tile += nwaves) {
            const int ts = 8 + 12 * ((tile / nwaves) & 3);
            dbg.stamp(ts);
            const bool has_next = tile + nwaves < tile_end;
            TileCtx nx;
            Lvl Lm_next;
            // Next tile: set up, and pull its voxel lines towards L1 / L2 while the embedding stage runs.
            // Big grids (fine, colour pass): through LDS-sink loads, no destination register -- plain loads get spilled there and
            // every spill waits for its load (measured: -7 % / -5 % kernel time against them); every lane touches its own 16 bytes
            // of all 8 voxels (measured best among 2 / 4 / 8 requests per lane).  The middle / coarse passes have registers to
            // spare: lane group g loads one word of each half of voxels 2g, 2g+1 and keeps them alive until the gather (5 %
            // faster there than the sink, or than no prefetch).
#ifndef NSR_PF_MODE
#define NSR_PF_MODE 0
#endif
            constexpr int kPfMode = NSR_PF_MODE ? NSR_PF_MODE : ((KIND == NSR_FINE || KIND == NSR_COLOR) ? 1 : 2);
            float warm[4] = {0.f, 0.f, 0.f, 0.f};
            auto touch = [&](const GridDev &GG, const Lvl &LL) {
                if (kPfMode == 1) {
                    float *sink = reinterpret_cast<float *>(ztmp);      // (scratch of the sample placement: dead during the tiles)
#pragma unroll
                    for (int k = 0; k < 8; ++k) prefetch_line(GG.feat + (long long)corner_vox(LL, k) * kC + 4 * g, sink);
                } else {
                    const int v0 = LL.vox + ((g & 1) ? LL.sy : 0) + ((g & 2) ? LL.sz : 0);
                    const float *p0 = GG.feat + (long long)v0 * kC, *p1 = p0 + (long long)LL.sx * kC;
                    warm[0] = p0[0]; warm[1] = p0[16]; warm[2] = p1[0]; warm[3] = p1[16];
                }
            };
            auto mid = [&]() {
                if (has_next) {
                    Lm_next = setup(nx, tile + nwaves);
                    touch(G, nx.L);
                    if constexpr (KIND == NSR_FINE) touch(P.grid[NSR_MIDDLE], Lm_next);
                }
            };
            F4 dr = cur.active ? draw[cur.pidx] : F4{0.f, 0.f, 0.f, 0.f};
            if (!cur.inside) dr.w = 0.f;                        // Renderer.py:57 cuts the occupancy gradient
            Act<2> dc;
            float dpe[3] = {0.f, 0.f, 0.f};
            if constexpr (KIND == NSR_COARSE) {
                nox_bwd_tile<PARAMS>(wl, aux, Sw, A, c, dr, F, lane, dc, mid);
            } else if constexpr (KIND == NSR_FINE) {
                Act<4> cc;
                cc.t[0] = c.t[0]; cc.t[1] = c.t[1]; cc.t[2] = cm.t[0]; cc.t[3] = cm.t[1];
                xyz_bwd_tile<NSR_FINE, PARAMS>(wl, aux, Sw, A, cur.px, cur.py, cur.pz, cc, dr, F, lane, dc, dpe, dbg, ts, mid, stg, wave);
            } else {
                xyz_bwd_tile<KIND, PARAMS>(wl, aux, Sw, A, cur.px, cur.py, cur.pz, c, dr, F, lane, dc, dpe, dbg, ts, mid, stg, wave);
            }
            // every load of the next tile is consumed before this tile's atomics are issued (see the header)
            Act<2> c_next, cm_next;
            if (has_next) {
                GatherRaw rawm;
                gather_issue(raw, G, nx.L, g);
                if (KIND == NSR_FINE) gather_issue(rawm, P.grid[NSR_MIDDLE], Lm_next, g);
                c_next = gather_finish(raw, nx.L);
                if (KIND == NSR_FINE) cm_next = gather_finish(rawm, Lm_next);
                if (kPfMode == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) keep_alive(warm[k]);
                }
            }
            float dux = 0.f, duy = 0.f, duz = 0.f;
            if (F.rays) coord_grad(G, cur.L, g, dc, dux, duy, duz);
#ifndef NSR_X_NOSCATTER
            if (F.grid) scatter_merged(G, cur.L, lane, dc, cur.active, Sw + kStA0, Sw + kStA0 + kTile * kTxS);
#endif
            if (F.rays && cur.active && g == 0) {
                // d p = d u * (n-1)/2 * 2/(hi-lo)  (+ embedding part), fp64 like autograd through Renderer.py:172
                dpb[cur.pidx * 3 + 0] = (double)dux * (2.0 * G.inv[0]) + (double)dpe[0];
                dpb[cur.pidx * 3 + 1] = (double)duy * (2.0 * G.inv[1]) + (double)dpe[1];
                dpb[cur.pidx * 3 + 2] = (double)duz * (2.0 * G.inv[2]) + (double)dpe[2];
            }
            dbg.stamp(ts + 10);
            if (has_next) { cur = nx; c = c_next; cm = cm_next; }
        }
        dbg.stamp(5);
        block_sync();
        dbg.stamp(6);
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
            block_sync();
        }
    }
    if (PARAMS) {
        float *img = P.partials + ((long long)bid_y() * nblk_x() + bid_x()) * P.partial_stride;
        if constexpr (KIND != NSR_COARSE) {                      // LDS-resident accumulator tiles join the others
#pragma unroll
            for (int Tk = 0; Tk < lds_acc_ktiles(KIND); ++Tk) {
                A.w3e[0][Tk] = to_v(ld4(Sw + st_lacc(KIND) + Tk * 512 + lane * 4));
                A.w3e[1][Tk] = to_v(ld4(Sw + st_lacc(KIND) + Tk * 512 + 256 + lane * 4));
            }
        }
        flush_acc<KIND>(A, wl, aux, stg, img, wave, lane, dbg);
    }
    dbg.stamp(7);
}

// grid = (blocks per pass, decoder passes of the stage): block (x, p) takes the ray groups x, x + gridDim.x, ... of pass p.
// This is the backward of calls WITHOUT an activation buffer (nsr_render_args.acts == NULL): it re-runs the decoder it
// differentiates.  With saved activations the split kernels of nsr_bwd2.h run instead.
template <int STAGE>
NSR_KERNEL NSR_BOUNDS(64 * kBwdWaves) void render_bwd_kernel(const RenderParams P) {
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

}  // namespace nsr
