// nsr_layout.h -- compile-time description of the decoder parameter blobs and operand streams.
//
// Flat parameter blob = reference named_parameters() order (src/conv_onet/models/decoder.py):
//   MLP (middle / fine / color, :124-159):  fc_c.{0..4}.{weight[32][cd],bias[32]}, embedder._B[3][93],
//        pts_linears.{0..4}.{weight,bias} (in: 93, 32, 32, 125, 32), output_linear.{weight[nout][32],bias}
//   MLP_no_xyz (coarse, :235-245):          pts_linears.{0..4} (in: 32, 32, 32, 64, 32), output_linear
//
// "Matrix" = a [32][kcols] column slice of one weight, the unit the MFMA loops work on.
// Channel <-> register convention ("CL"): a 16-point tile lives in one wave; lane l = (pt = l&15,
// g = l>>4) holds, for every k-tile T of 16 channels, the four channels 16T+4g+r (r = 0..3).
#pragma once

namespace nsr {

constexpr int kC = 32;         // feature channels per grid
constexpr int kH = 32;         // hidden width
constexpr int kE = 93;         // Fourier embedding size
constexpr int kET = 6;         // embedding k-tiles (96 padded channels)
constexpr int kTile = 16;      // points per wave tile
constexpr int kTxS = 36;       // row stride (floats) of the per-wave transposition buffers

struct Mat {
    int off;      // offset of row 0 / col 0 of the owning weight in the flat blob
    int stride;   // row stride of the owning weight
    int kbeg;     // first column of the slice
    int kcols;    // valid columns in the slice
    int nt;       // k-tiles (ceil(kcols/16))
    int pk;       // offset of the slice in the packed forward stream (floats)
};

constexpr int cdim_of(int kind) { return kind == 2 ? 64 : 32; }
constexpr int nout_of(int kind) { return kind == 3 ? 4 : 1; }
constexpr bool is_xyz(int kind) { return kind != 0; }

// ---- MLP with xyz embedding --------------------------------------------------------------------
constexpr int xyz_fcw(int cd, int i) { return i * (32 * cd + 32); }
constexpr int xyz_fcb(int cd, int i) { return xyz_fcw(cd, i) + 32 * cd; }
constexpr int xyz_B(int cd) { return 5 * (32 * cd + 32); }
constexpr int xyz_w(int cd, int i) {
    int o = xyz_B(cd) + 3 * kE;
    const int in[5] = {kE, 32, 32, kE + 32, 32};
    for (int j = 0; j < i; ++j) o += 32 * in[j] + 32;
    return o;
}
constexpr int xyz_in(int i) { return i == 0 ? kE : (i == 3 ? kE + 32 : 32); }
constexpr int xyz_b(int cd, int i) { return xyz_w(cd, i) + 32 * xyz_in(i); }
constexpr int xyz_wo(int cd) { return xyz_b(cd, 4) + 32; }
constexpr int xyz_bo(int cd, int nout) { return xyz_wo(cd) + 32 * nout; }
constexpr int xyz_total(int cd, int nout) { return xyz_bo(cd, nout) + nout; }

// matrix ids of an xyz decoder
enum { XW0 = 0, XU0 = 1, XW1 = 2, XU1 = 3, XW2 = 4, XU2 = 5, XW3E = 6, XW3H = 7, XU3 = 8, XW4 = 9, XU4 = 10, XNMAT = 11 };
constexpr int xyz_nt(int cd, int id) {
    return (id == XW0 || id == XW3E) ? kET : ((id == XU0 || id == XU1 || id == XU2 || id == XU3 || id == XU4) ? cd / 16 : 2);
}
constexpr int xyz_pk(int cd, int id) {
    int o = 0;
    for (int j = 0; j < id; ++j) o += xyz_nt(cd, j) * 512;
    return o;
}
constexpr Mat xyz_mat(int cd, int id) {
    switch (id) {
        case XW0: return Mat{xyz_w(cd, 0), kE, 0, kE, kET, xyz_pk(cd, id)};
        case XU0: return Mat{xyz_fcw(cd, 0), cd, 0, cd, cd / 16, xyz_pk(cd, id)};
        case XW1: return Mat{xyz_w(cd, 1), 32, 0, 32, 2, xyz_pk(cd, id)};
        case XU1: return Mat{xyz_fcw(cd, 1), cd, 0, cd, cd / 16, xyz_pk(cd, id)};
        case XW2: return Mat{xyz_w(cd, 2), 32, 0, 32, 2, xyz_pk(cd, id)};
        case XU2: return Mat{xyz_fcw(cd, 2), cd, 0, cd, cd / 16, xyz_pk(cd, id)};
        case XW3E: return Mat{xyz_w(cd, 3), kE + 32, 0, kE, kET, xyz_pk(cd, id)};
        case XW3H: return Mat{xyz_w(cd, 3), kE + 32, kE, 32, 2, xyz_pk(cd, id)};
        case XU3: return Mat{xyz_fcw(cd, 3), cd, 0, cd, cd / 16, xyz_pk(cd, id)};
        case XW4: return Mat{xyz_w(cd, 4), 32, 0, 32, 2, xyz_pk(cd, id)};
        default: return Mat{xyz_fcw(cd, 4), cd, 0, cd, cd / 16, xyz_pk(cd, id)};
    }
}
constexpr int xyz_packed_total(int cd) { return xyz_pk(cd, XNMAT); }

// ---- MLP_no_xyz (coarse) ------------------------------------------------------------------------
constexpr int nox_in(int i) { return i == 3 ? 64 : 32; }
constexpr int nox_w(int i) {
    int o = 0;
    for (int j = 0; j < i; ++j) o += 32 * nox_in(j) + 32;
    return o;
}
constexpr int nox_b(int i) { return nox_w(i) + 32 * nox_in(i); }
constexpr int nox_wo() { return nox_b(4) + 32; }
constexpr int nox_bo() { return nox_wo() + 32; }
constexpr int nox_total() { return nox_bo() + 1; }
enum { NW0 = 0, NW1 = 1, NW2 = 2, NW3C = 3, NW3H = 4, NW4 = 5, NNMAT = 6 };
constexpr Mat nox_mat(int id) {
    switch (id) {
        case NW0: return Mat{nox_w(0), 32, 0, 32, 2, 0};
        case NW1: return Mat{nox_w(1), 32, 0, 32, 2, 1024};
        case NW2: return Mat{nox_w(2), 32, 0, 32, 2, 2048};
        case NW3C: return Mat{nox_w(3), 64, 0, 32, 2, 3072};
        case NW3H: return Mat{nox_w(3), 64, 32, 32, 2, 4096};
        default: return Mat{nox_w(4), 32, 0, 32, 2, 5120};
    }
}
constexpr int nox_packed_total() { return 6144; }

// ---- per decoder-kind totals -------------------------------------------------------------------
constexpr int param_total(int kind) { return kind == 0 ? nox_total() : xyz_total(cdim_of(kind), nout_of(kind)); }
constexpr int packed_total(int kind) { return kind == 0 ? nox_packed_total() : xyz_packed_total(cdim_of(kind)); }
constexpr int nmat_of(int kind) { return kind == 0 ? (int)NNMAT : (int)XNMAT; }
constexpr Mat mat_of(int kind, int id) { return kind == 0 ? nox_mat(id) : xyz_mat(cdim_of(kind), id); }
constexpr int bias_off(int kind, int i) { return kind == 0 ? nox_b(i) : xyz_b(cdim_of(kind), i); }
constexpr int fcb_off(int kind, int i) { return xyz_fcb(cdim_of(kind), i); }
constexpr int wo_off(int kind) { return kind == 0 ? nox_wo() : xyz_wo(cdim_of(kind)); }
constexpr int bo_off(int kind) { return kind == 0 ? nox_bo() : xyz_bo(cdim_of(kind), nout_of(kind)); }
constexpr int B_off(int kind) { return xyz_B(cdim_of(kind)); }

static_assert(param_total(0) == 6337, "coarse decoder parameter count");
static_assert(param_total(1) == 15800, "middle decoder parameter count");
static_assert(param_total(2) == 20920, "fine decoder parameter count");
static_assert(param_total(3) == 15899, "color decoder parameter count");

// ---- per-decoder auxiliary table staged in LDS (floats) -----------------------------------------
constexpr int AUX_B = 0;       // b[5][32]   pts_linears biases
constexpr int AUX_V = 160;     // v[5][32]   fc_c biases (xyz only)
constexpr int AUX_WO = 320;    // wo[4][32]  output weights (rows >= nout are zero)
constexpr int AUX_BO = 448;    // bo[4]
constexpr int AUX_BM = 452;    // Bm[24][4][4]  Fourier matrix: per group of 4 channels Bx[4] | By[4] | Bz[4] | 0
constexpr int AUX_FLOATS = 452 + 96 * 4;   // 836
// ---- transposed operand stream (split backward, nsr_bwd2.h): dx = W^T dy -------------------------------------
// For a [32][32 or 96] column slice W: T[((Tk * 2 + To) * 64 + lane) * 4 + r] = W[16 To + 4 (lane >> 4) + r][kbeg + 16 Tk + (lane & 15)]
// (zero beyond kcols): one 16-byte LDS read per lane = the A operand (B operand in the "lane = channel" form) of four MFMA
// k-steps.  xyz decoders: U0..U4 (first 32 feature columns only: the decoder's own grid; the middle features of the fine
// decoder carry no gradient, decoder.py:185), W1, W2, W3h, W4, W0, W3e.  MLP_no_xyz: NW0..NW4 at their forward offsets.
constexpr int xyzT_u(int i) { return i * 1024; }
constexpr int xyzT_wh(int m) { return 5120 + m * 1024; }      // m: 0 W1, 1 W2, 2 W3h, 3 W4
constexpr int xyzT_w0() { return 9216; }
constexpr int xyzT_w3e() { return 12288; }
constexpr int xyzT_total() { return 15360; }
constexpr int packedT_total(int kind) { return kind == 0 ? nox_packed_total() : xyzT_total(); }
// source slice of entry `id` of the transposed stream (xyz: 0..4 U_i, 5..8 W1 W2 W3h W4, 9 W0, 10 W3e) and its offset
constexpr int xyzT_nmat() { return 11; }
constexpr Mat xyzT_mat(int cd, int id) {
    return id < 5 ? Mat{xyz_fcw(cd, id), cd, 0, 32, 2, xyzT_u(id)}
         : id == 5 ? Mat{xyz_w(cd, 1), 32, 0, 32, 2, xyzT_wh(0)}
         : id == 6 ? Mat{xyz_w(cd, 2), 32, 0, 32, 2, xyzT_wh(1)}
         : id == 7 ? Mat{xyz_w(cd, 3), kE + 32, kE, 32, 2, xyzT_wh(2)}
         : id == 8 ? Mat{xyz_w(cd, 4), 32, 0, 32, 2, xyzT_wh(3)}
         : id == 9 ? Mat{xyz_w(cd, 0), kE, 0, kE, kET, xyzT_w0()}
                   : Mat{xyz_w(cd, 3), kE + 32, 0, kE, kET, xyzT_w3e()};
}
constexpr int nmatT_of(int kind) { return kind == 0 ? (int)NNMAT : xyzT_nmat(); }
constexpr Mat matT_of(int kind, int id) { return kind == 0 ? nox_mat(id) : xyzT_mat(cdim_of(kind), id); }

// device buffer written by nsr_pack_params for one decoder: [aux table | forward operand stream | transposed stream]
constexpr int packed_buf_total(int kind) { return AUX_FLOATS + packed_total(kind) + packedT_total(kind); }
static_assert(AUX_FLOATS % 4 == 0, "the operand stream behind the aux table must stay 16-byte aligned");

}  // namespace nsr
