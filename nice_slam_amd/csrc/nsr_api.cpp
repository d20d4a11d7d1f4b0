// nsr_api.cpp -- the C ABI of libnsr.so (see include/nsr.h).  Compiled as HIP for gfx950.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "nsr_rt.h"
#include "nsr_kernels.h"

namespace {

thread_local std::string g_err;

int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}

constexpr int kMaxTiles = 12;            // 12 waves = 768 threads per block
constexpr int kDefaultBwdBlocks = 256;   // persistent-grid cap of the backward kernels: one block per CU (LDS-bound)
constexpr int kLdsLimit = 160 * 1024;

inline int round16(int bytes) { return (bytes + 15) & ~15; }

// forward: up to 12 tiles (768 threads, 3 waves/SIMD, <=168 VGPRs)
int rays_per_block(int S) {
    const int rb = (kMaxTiles * nsr::kTile) / S;
    return rb < 1 ? 1 : rb;
}

int stage_passes(int stage) { return stage == NSR_STAGE_COARSE ? 1 : 3; }

int max_param_count(int stage) {
    return stage == NSR_STAGE_COARSE ? nsr::param_total(0) : nsr::param_total(2);
}

// Decoder passes of one backward launch: a block serves ONE pass -- that decoder's operand stream sits in its LDS.
int bwd_passes(int stage) { return stage == NSR_STAGE_COARSE ? 1 : stage; }     // middle 1, fine 2, colour 3

// ---- split backward over saved activations (nsr_bwd2.h) -----------------------------------------------------------------
// layout of nsr_render_args.acts (floats): [passes][tiles][kActSlots][16][16] saved by the forward | [passes][tiles][kDySlots]
// [16][16] dY (dX kernel -> dW kernel) | [npad][4] d raw | [npad][4] fp32 positions | [npad][4] doubles: position + depth.  npad = 16 * tiles = the
// sample points rounded up to whole 16-point tiles (DMA pieces of 1 KB).
struct SplitLayout {
    long long npts, npad, stride;      // stride: floats between two slots
    long long o_dy, o_draw, o_pf, o_pd, total;
};
SplitLayout split_layout(int stage, long long n_rays, int S) {
    SplitLayout L;
    const int passes = bwd_passes(stage);
    L.npts = n_rays * S;
    L.npad = (L.npts + nsr::kTile - 1) / nsr::kTile * nsr::kTile;
    L.stride = L.npad * 16;
    L.o_dy = (long long)passes * nsr::kActSlots * L.stride;
    L.o_draw = L.o_dy + (long long)passes * nsr::kDySlots * L.stride;
    L.o_pf = L.o_draw + L.npad * 4;
    L.o_pd = L.o_pf + L.npad * 4;
    L.total = L.o_pd + L.npad * 8;
    return L;
}
// launch geometry: the dX kernel runs `nb` blocks of `waves` waves per decoder pass (one block per CU over all passes,
// fewer waves per block when the batch is small: every CU gets work); the dW kernel `nimg` blocks per pass, each of which
// leaves one partial image of the gradient blob.
struct SplitGeo { int nb, waves, nimg; };             // dX: blocks per pass, waves per block; dW: images per pass
int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}
SplitGeo split_geo(int stage, long long n_rays, int S, int max_blocks) {
    static const int dw_mult = env_int("NSR_DW_BLOCKS_PER_CU", 1);
    SplitGeo G;
    const int passes = bwd_passes(stage);
    const long long tiles = (n_rays * S + nsr::kTile - 1) / nsr::kTile;
    const int cap = max_blocks > 0 ? max_blocks : kDefaultBwdBlocks;
    int per_pass = cap / passes;
    if (per_pass < 1) per_pass = 1;
    long long w = (tiles + per_pass - 1) / per_pass;
    static const int wave_cap = env_int("NSR_DX_MAX_WAVES", nsr::kDxMaxWaves);      // (measurement: fewer waves per dX block)
    const int wcap = wave_cap < 1 ? 1 : (wave_cap > nsr::kDxMaxWaves ? nsr::kDxMaxWaves : wave_cap);
    G.waves = (int)(w < 1 ? 1 : (w > wcap ? wcap : w));
    const long long nb = (tiles + G.waves - 1) / G.waves;
    G.nb = (int)(nb < 1 ? 1 : (nb > per_pass ? per_pass : nb));
    long long ni = (long long)per_pass * (dw_mult < 1 ? 1 : dw_mult);
    if (ni > tiles) ni = tiles;
    G.nimg = (int)(ni < 1 ? 1 : ni);
    return G;
}

// validate + translate the public argument block
int build_params(const nsr_render_args *a, nsr::RenderParams &P, bool need_rays, bool bwd = false) {
    if (!a) return fail("nsr: null argument block");
    if (a->stage < 0 || a->stage > 3) return fail("nsr: stage out of range");
    std::memset(&P, 0, sizeof(P));
    P.stage = a->stage;
    P.n_samples = a->n_samples;
    const bool guided = a->gt_depth != nullptr && a->stage != NSR_STAGE_COARSE;
    P.n_surface = guided ? a->n_surface : 0;
    P.S = P.n_samples + P.n_surface;
    if (P.n_samples < 1 || P.n_surface < 0 || P.S > NSR_MAX_SAMPLES)
        return fail("nsr: n_samples + n_surface must be in [1, 64]");
    if (guided && !a->gt_max) return fail("nsr: gt_max is required when gt_depth is given");
    P.n_rays = a->n_rays;
    if (need_rays) {
        if (a->n_rays < 0) return fail("nsr: negative ray count");
        if (a->n_rays > 0 && (!a->rays_o || !a->rays_d)) return fail("nsr: null ray pointers");
    }
    P.s_magic = P.S > 1 ? (unsigned)(((1ull << 32) + (unsigned)P.S - 1) / (unsigned)P.S) : 0u;     // (S = 1: nsr_kernels.h::ray_of_point)
    P.rays_per_block = rays_per_block(P.S);
    if (!bwd) {
        // Small batches (the tracker's 200 rays): a forward block runs its decoders one after the other, so with few blocks the
        // launch takes one block's serial chain while most CUs idle.  Fewer rays per block -> one block per CU as long as
        // the batch allows (NSR_FWD_SMALL=0: always full 12-tile blocks).
        static const bool small_ok = [] { const char *e = getenv("NSR_FWD_SMALL"); return !(e && e[0] == '0'); }();
        if (small_ok) {
            long long want = (P.n_rays + kDefaultBwdBlocks - 1) / kDefaultBwdBlocks;
            if (want < 1) want = 1;
            if (want < P.rays_per_block) P.rays_per_block = (int)want;
        }
    }
    P.tiles_per_block = (P.rays_per_block * P.S + nsr::kTile - 1) / nsr::kTile;
    P.n_groups = (P.n_rays + P.rays_per_block - 1) / P.rays_per_block;
    P.rays_o = a->rays_o;
    P.rays_d = a->rays_d;
    P.acts = a->acts;
    P.acts_masks_only = a->acts_masks_only & 15;
    if (!bwd && a->acts && a->zvals) {
        // activation buffer: the forward leaves the sample positions (and, with the fused loss, d raw) for the split backward;
        // the three-launch forward (nsr_fwd2.h) keeps its per-sample scratch there
        const SplitLayout L = split_layout(P.stage, P.n_rays, P.S);
        P.draw = a->acts + L.o_draw;
        P.pf = a->acts + L.o_pf;
        P.pd = reinterpret_cast<double *>(a->acts + L.o_pd);
    }
    P.n_points_total = (long long)P.n_rays * P.S;
    P.act_tiles = split_layout(P.stage, P.n_rays, P.S).npad / nsr::kTile;
    P.gt_depth = guided ? a->gt_depth : nullptr;
    P.gt_max = a->gt_max;
    for (int i = 0; i < 3; ++i) { P.blo[i] = a->bound_lo[i]; P.bhi[i] = a->bound_hi[i]; }
    std::memcpy(P.t_uniform, a->t_uniform, sizeof(P.t_uniform));
    std::memcpy(P.t_surface, a->t_surface, sizeof(P.t_surface));
    const int first = a->stage == NSR_STAGE_COARSE ? NSR_COARSE : NSR_MIDDLE;
    const int last = a->stage == NSR_STAGE_COARSE ? NSR_COARSE : a->stage;
    for (int s = first; s <= last; ++s) {
        const nsr_grid &g = a->grid[s];
        if (!g.feat) return fail("nsr: missing feature grid for this stage");
        if (g.Z < 1 || g.Y < 1 || g.X < 1) return fail("nsr: bad grid shape");
        if ((long long)g.Z * g.Y * g.X >= (1ll << 25)) return fail("nsr: grid too large for 32-bit byte offsets (2^25 voxels = 4 GB)");
        nsr::GridDev &G = P.grid[s];
        G.feat = g.feat; G.dfeat = g.dfeat; G.gmask = a->grad_voxel_mask[s]; G.Z = g.Z; G.Y = g.Y; G.X = g.X;
        for (int i = 0; i < 3; ++i) {
            if (!(g.hi[i] > g.lo[i])) return fail("nsr: empty normalisation box");
            G.lo[i] = g.lo[i];
            G.ext[i] = g.hi[i] - g.lo[i];
            G.inv[i] = 1.0 / (g.hi[i] - g.lo[i]);
        }
        const nsr_decoder &d = a->dec[s];
        if (!d.params || !d.packed) return fail("nsr: missing decoder parameters / packed stream for this stage");
        P.dec[s].params = d.params; P.dec[s].packed = d.packed; P.dec[s].dparams = d.dparams;
    }
    P.depth = a->depth; P.var = a->var; P.rgb = a->rgb; P.raw = a->raw; P.zvals = a->zvals;
    P.gt_color = a->gt_color; P.keep = a->keep; P.loss = a->loss; P.w_color = a->w_color;
    P.skip_masked = (a->skip_masked && a->keep && a->acts && a->zvals && a->raw) ? 1 : 0;      // (the one-launch forward renders every ray)
    P.dl_depth = a->dl_depth; P.dl_rgb = a->dl_rgb; P.loss_depth = a->gt_depth;
    return 0;
}

int finish(const char *what) {
    if (const char *e = nsr::rt_check_last()) return fail(std::string(what) + ": " + e);
    return 0;
}

// raise the dynamic-LDS limit of a kernel once per (kernel, device, size): not a stream operation, but kept out of the
// steady state so that a captured hipGraph contains kernel launches only.  Keyed by the kernel's ADDRESS (all render
// kernels share one function-pointer type) and the current device.
template <typename K>
int launch_cfg(K kernel, int lds_bytes, const char *what) {
    if (lds_bytes > kLdsLimit) return fail(std::string(what) + ": LDS budget exceeded");
    if (lds_bytes <= 48 * 1024) return 0;
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> granted;
    const std::pair<const void *, int> key(reinterpret_cast<const void *>(kernel), nsr::rt_current_device());
    std::lock_guard<std::mutex> lock(mu);
    int &g = granted[key];
    if (lds_bytes > g) {
        if (const char *e = nsr::rt_allow_lds(kernel, lds_bytes)) return fail(std::string(what) + ": " + e);
        g = lds_bytes;
    }
    return 0;
}

int fwd_lds_bytes(int stage, int npts) {
    const int wl = stage == NSR_STAGE_COARSE ? nsr::packed_total(0) : (stage == NSR_STAGE_MIDDLE ? nsr::packed_total(1) : nsr::packed_total(2));
    return round16(3 * nsr::AUX_FLOATS * 4) + npts * (8 + 8 + 16) + wl * 4;
}

// the backward as comp_bwd -> dX -> dW -> finalize over the activations the forward saved (nsr_bwd2.h)
int render_bwd_split(const nsr_render_args *a, const nsr_bwd_args *b, nsr::RenderParams &P, bool any_params, void *stream) {
    const SplitLayout L = split_layout(P.stage, P.n_rays, P.S);
    if (L.stride * 4 >= (1ll << 31)) return fail("nsr_render_bwd: more than 2^25 sample points in one call (split the ray batch)");
    const SplitGeo G = split_geo(P.stage, P.n_rays, P.S, b->max_blocks);
    const int passes = bwd_passes(P.stage);
    P.dy = P.acts + L.o_dy;
    P.draw = P.acts + L.o_draw;
    P.pf = P.acts + L.o_pf;
    P.pd = reinterpret_cast<double *>(P.acts + L.o_pd);
    for (int s = 0; s < 4; ++s)
        if (P.dec[s].dparams && (P.acts_masks_only & (1 | (2 << (s == NSR_COARSE ? 0 : s - NSR_MIDDLE)))))
            return fail("nsr_render_bwd: the forward saved relu masks only for a decoder whose parameter gradients are asked for (acts_masks_only); they need the full activations");
    // (the fine decoder's input is [c_fine | c_mid], decoder.py:182-187: its dW reads the middle pass's saved features)
    if (P.dec[NSR_FINE].dparams && (P.acts_masks_only & (1 | 2)))
        return fail("nsr_render_bwd: the forward saved relu masks only for the middle decoder (acts_masks_only), whose features the fine decoder's parameter gradients read");
    static const int xflags = env_int("NSR_X", 0);
    P.xflags = xflags;
    if (any_params) {
        const long long need = (long long)passes * ((long long)G.nimg * P.partial_stride + (long long)G.nb * nsr::kDbPart);
        if (!b->workspace || b->workspace_floats < need) return fail("nsr_render_bwd: workspace too small");
        P.partials = b->workspace;
        P.dbpart = b->workspace + (long long)passes * G.nimg * P.partial_stride;
    }
    if (b->ev_start) nsr::rt_record(b->ev_start, stream);
    // the forward's loss epilogue already wrote d raw (for an incoming gradient of 1): used when the caller SAYS that it hands
    // back exactly the derivative arrays that forward produced, unmodified (nsr_bwd_args.loss_grads_from_forward); else the
    // compositor backward runs here on d_depth / d_var / d_rgb as given
    const bool draw_ready = b->loss_grads_from_forward && a->loss && a->dl_depth && b->d_depth == a->dl_depth && !b->d_var &&
                            (b->d_rgb == nullptr || b->d_rgb == a->dl_rgb) && (P.stage != NSR_STAGE_COLOR || b->d_rgb == a->dl_rgb || !a->gt_color);
    P.draw_scaled = draw_ready ? 0 : 1;
    if (!draw_ready) {
        const int tb = 256, rays_per_block = tb / 64;
        NSR_LAUNCH(nsr::comp_bwd_kernel, dim3((unsigned)((P.n_rays + rays_per_block - 1) / rays_per_block)), dim3(tb), 0, stream, P);
    }
    const bool rays = P.d_rays_o != nullptr;
    {
        int lds = 0;
        const int first = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : NSR_MIDDLE, last = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : P.stage;
        for (int kind = first; kind <= last; ++kind) {
            const int need = (nsr::AUX_FLOATS + nsr::packedT_total(kind) + G.waves * nsr::kDxStg + 4) * 4;   // + the tile counter
            lds = need > lds ? need : lds;
        }
        // hot-voxel table of a dX block: samples within `hot_cells` cells of their ray's origin (NSR_DX_HOT_CELLS, 0: off), as many
        // slots as the block's LDS has left (NSR_DX_HOT_SLOTS caps it; round 5: 64 slots / 2 cells -> what fits / 6 cells, measured)
        static const int hot_cells = env_int("NSR_DX_HOT_CELLS", 6), hot_cap = env_int("NSR_DX_HOT_SLOTS", 512);
        P.hot_slots = 0;
        if (P.stage != NSR_STAGE_COARSE) {
            int slots = (kLdsLimit - lds) / (nsr::kHotRow * 4);
            slots = slots > hot_cap ? hot_cap : slots;
            P.hot_slots = slots < 16 ? 16 : slots;
            lds += P.hot_slots * nsr::kHotRow * 4;
        }
        for (int s = 0; s < 4; ++s) {
            P.hot_z[s] = 0.f;
            if (s == NSR_COARSE || hot_cells <= 0 || !P.grid[s].dfeat) continue;
            double cell = 0.0;
            const int nn[3] = {P.grid[s].X, P.grid[s].Y, P.grid[s].Z};
            for (int ax = 0; ax < 3; ++ax) { const double c = nn[ax] > 1 ? P.grid[s].ext[ax] / (nn[ax] - 1) : 0.0; cell = c > cell ? c : cell; }
            P.hot_z[s] = (float)(hot_cells * cell);
        }
        P.lds_grid_floats = 0;
        if (P.stage == NSR_STAGE_COARSE && P.grid[NSR_COARSE].dfeat) {       // a coarse gradient grid that fits next to the rest
            const long long gf = (long long)P.grid[NSR_COARSE].X * P.grid[NSR_COARSE].Y * P.grid[NSR_COARSE].Z * nsr::kC;
            if (lds + gf * 4 <= kLdsLimit) { P.lds_grid_floats = (int)gf; lds += (int)gf * 4; }
        }
        // the passes * nb blocks dealt over the decoder passes by a tile's cost: a pass that owes neither parameter nor ray gradients skips
        // its embedding backward (96 of 240 MFMAs, 24 cosines per lane) -- with equal shares the other passes' blocks set the kernel's length
        // (`--stepped-grads-only`: dX<3> 95 us against 85 with every decoder's gradients); measured 10 / 8 / 7 / 6 / 5 to 10 for a full pass: 92.4 / 85.3 / 82.6 / 79.5 / 80.9 us
        static const int w_light = env_int("NSR_DX_LIGHT_WEIGHT", 6), w_mid = env_int("NSR_DX_MIDDLE_PCT", 100);
        {
            const long long tiles = (P.n_points_total + nsr::kTile - 1) / nsr::kTile;
            int wgt[3] = {0, 0, 0}, wsum = 0;
            for (int p = 0; p < passes; ++p) {
                const int s = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : NSR_MIDDLE + p;
                wgt[p] = (P.dec[s].dparams || rays || P.stage == NSR_STAGE_COARSE) ? 10 : (w_light < 1 ? 1 : (w_light > 10 ? 10 : w_light));
                wgt[p] *= (s == NSR_MIDDLE ? w_mid : 100);          // (the middle grid's cells are twice as long: fewer voxel runs per tile)
                wsum += wgt[p];
            }
            const int total = passes * G.nb;
            int used = 0;
            P.dx_beg[0] = 0;
            for (int p = 0; p < 3; ++p) {
                long long n = 0;
                if (p < passes) {
                    n = p == passes - 1 ? total - used : (long long)total * wgt[p] / wsum;
                    if (n < 1) n = 1;
                    if (n > tiles) n = tiles;
                }
                used += (int)n;
                P.dx_beg[p + 1] = P.dx_beg[p] + (int)n;
            }
        }
        const dim3 grid(P.dx_beg[3]), block(64 * G.waves);
#define NSR_DX(ST, RY)                                                                                  \
    if (int rc = launch_cfg(nsr::render_bwd_dx_kernel<ST, RY>, lds, "nsr_render_bwd(dx)")) return rc;   \
    NSR_LAUNCH((nsr::render_bwd_dx_kernel<ST, RY>), grid, block, lds, stream, P);
        switch (P.stage) {
            case 0: if (rays) { NSR_DX(0, true) } else { NSR_DX(0, false) } break;
            case 1: if (rays) { NSR_DX(1, true) } else { NSR_DX(1, false) } break;
            case 2: if (rays) { NSR_DX(2, true) } else { NSR_DX(2, false) } break;
            default: if (rays) { NSR_DX(3, true) } else { NSR_DX(3, false) } break;
        }
#undef NSR_DX
    }
    if (b->ev_dx_done) nsr::rt_record(b->ev_dx_done, stream);
    if (any_params) {
        const int lds = nsr::dw_lds_bytes(P.stage >= NSR_STAGE_FINE ? NSR_FINE : NSR_MIDDLE);
        // the passes * nimg blocks (= partial images, the workspace's size) dealt over the decoders that want parameter gradients,
        // in proportion to a tile's measured cost
        const long long tiles = (P.n_points_total + nsr::kTile - 1) / nsr::kTile;
        int wgt[3] = {0, 0, 0}, wsum = 0;
        for (int p = 0; p < passes; ++p) {
            const int s = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : NSR_MIDDLE + p;
            // the fine decoder's share (the others: 224): its 288 MFMAs per tile against 224 overstate it -- a tile's time has a part that does
            // not scale with the MFMA count (flags, operand reads, sines).  Measured (round 6, log item 22): colour stage 240, fine stage 260.
            static const int dw_fine = env_int("NSR_DW_FINE_WEIGHT", 0);
            const int fine_w = dw_fine > 0 ? dw_fine : (P.stage == NSR_STAGE_COLOR ? 240 : 260);
            wgt[p] = P.dec[s].dparams ? (s == NSR_FINE ? fine_w : 224) : 0;
            wsum += wgt[p];
        }
        const int total = passes * G.nimg;
        P.dw_beg[0] = 0;
        for (int p = 0; p < 3; ++p) {
            long long n = (p < passes && wgt[p]) ? (long long)total * wgt[p] / wsum : 0;
            if (p < passes && wgt[p] && n < 1) n = 1;
            if (n > tiles) n = tiles;
            P.dw_beg[p + 1] = P.dw_beg[p] + (int)n;
        }
        const dim3 grid(P.dw_beg[3]), block(64 * nsr::kDwWaves);
#define NSR_DW(ST)                                                                                  \
    if (int rc = launch_cfg(nsr::render_bwd_dw_kernel<ST>, lds, "nsr_render_bwd(dw)")) return rc;   \
    NSR_LAUNCH((nsr::render_bwd_dw_kernel<ST>), grid, block, lds, stream, P);
        switch (P.stage) {
            case 0: NSR_DW(0) break;
            case 1: NSR_DW(1) break;
            case 2: NSR_DW(2) break;
            default: NSR_DW(3) break;
        }
#undef NSR_DW
        if (b->ev_dw_done) nsr::rt_record(b->ev_dw_done, stream);
        const int first = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : NSR_MIDDLE, last = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : P.stage;
        nsr::FinalParams R;
        R.stride = P.partial_stride; R.overwrite = b->overwrite_dparams ? 1 : 0;
        int rows = 0, nblocks = 0;
        for (int s = first; s <= last; ++s) {
            if (!P.dec[s].dparams) continue;
            const int pass = P.stage == NSR_STAGE_COARSE ? 0 : s - NSR_MIDDLE;
            nsr::FinalJob &J = R.job[rows++];
            J.images = P.partials + (long long)P.dw_beg[pass] * P.partial_stride;
            J.dbpart = P.dbpart + (long long)P.dx_beg[pass] * nsr::kDbPart;
            J.dparams = P.dec[s].dparams;
            J.kind = s; J.nimg = P.dw_beg[pass + 1] - P.dw_beg[pass]; J.ndx = P.dx_beg[pass + 1] - P.dx_beg[pass];
            const int nb = (nsr::param_total(s) + 63) / 64;
            nblocks = nb > nblocks ? nb : nblocks;
        }
        for (int r = rows; r < 3; ++r) R.job[r] = nsr::FinalJob{nullptr, nullptr, nullptr, 0, 0, 0};
        // (512 threads = four resident blocks per CU instead of two was measured: 17.4 vs 15.1 us -- the kernel is a chain of round trips, not a queue of blocks)
        static const int fin_threads = env_int("NSR_FIN_THREADS", 1024);
        NSR_LAUNCH(nsr::bwd_finalize_kernel, dim3(nblocks, rows), dim3(fin_threads), fin_threads * 4, stream, R);
    }
    if (b->ev_stop) nsr::rt_record(b->ev_stop, stream);
    return finish("nsr_render_bwd(split)");
}

}  // namespace

extern "C" {

int nsr_version(void) { return NSR_VERSION; }
const char *nsr_last_error(void) { return g_err.c_str(); }

int64_t nsr_param_count(int slot) { return (slot < 0 || slot > 3) ? -1 : nsr::param_total(slot); }
int64_t nsr_packed_count(int slot) { return (slot < 0 || slot > 3) ? -1 : nsr::packed_buf_total(slot); }

int64_t nsr_acts_floats(int stage, int64_t n_rays, int n_samples_total) {
    if (stage < 0 || stage > 3 || n_rays < 0 || n_samples_total < 1 || n_samples_total > NSR_MAX_SAMPLES) return -1;
    return split_layout(stage, n_rays, n_samples_total).total;
}

int64_t nsr_bwd_workspace_floats(int stage, int64_t n_rays, int n_samples_total, int max_blocks) {
    if (stage < 0 || stage > 3 || n_samples_total < 1 || n_samples_total > NSR_MAX_SAMPLES) return -1;
    // partial images of the dW kernel + d _B partials of the dX kernel
    const SplitGeo G = split_geo(stage, n_rays, n_samples_total, max_blocks);
    const long long split = (long long)bwd_passes(stage) * ((long long)G.nimg * max_param_count(stage) + (long long)G.nb * nsr::kDbPart);
    return (int64_t)split;
}

int nsr_pack_params(int slot, const float *params, float *packed, void *stream) {
    if (slot < 0 || slot > 3) return fail("nsr_pack_params: slot out of range");
    if (!params || !packed) return fail("nsr_pack_params: null pointer");
    const int n = nsr::packed_buf_total(slot), tb = 256, nb = (n + tb - 1) / tb;
    switch (slot) {
        case 0: NSR_LAUNCH(nsr::pack_kernel<0>, dim3(nb), dim3(tb), 0, stream, params, packed); break;
        case 1: NSR_LAUNCH(nsr::pack_kernel<1>, dim3(nb), dim3(tb), 0, stream, params, packed); break;
        case 2: NSR_LAUNCH(nsr::pack_kernel<2>, dim3(nb), dim3(tb), 0, stream, params, packed); break;
        default: NSR_LAUNCH(nsr::pack_kernel<3>, dim3(nb), dim3(tb), 0, stream, params, packed); break;
    }
    return finish("nsr_pack_params");
}

int nsr_render_fwd(const nsr_render_args *a, void *stream) {
    nsr::RenderParams P;
    if (int rc = build_params(a, P, true)) return rc;
#ifdef NSR_TS
    if (const char *e = getenv("NSR_DBG_FWD_PTR")) P.dbg = reinterpret_cast<long long *>(strtoull(e, nullptr, 16));
#endif
    if (!a->depth || !a->var || !a->rgb) return fail("nsr_render_fwd: null output pointer");
    if (P.n_rays == 0) return 0;
    static const int fwd_split = env_int("NSR_FWD_SPLIT", 1);      // 0: always the one-launch kernel (measurement)
    if (a->acts && a->zvals && a->raw) {
        // a call that hands over an activation buffer (with zvals and raw: without them nsr_render_bwd refuses) will be differentiated: only the three-launch path saves into it, and the
        // tile / keep arithmetic of the kernels that read it is 32-bit (tile_live, dw_live_mask, split_layout)
        if (!fwd_split) return fail("nsr_render_fwd: NSR_FWD_SPLIT=0 (the one-launch kernel saves no activations) with an activation buffer: "
                                    "nsr_render_bwd would read an unwritten buffer");
        if (P.n_points_total > (1ll << 25) - 16) return fail("nsr_render_fwd: more than 2^25 sample points in one differentiated call (split the ray batch)");
    }
    if (fwd_split && P.acts && P.zvals && P.raw && P.draw) {
        // a differentiated call with an activation buffer: sample placement -> decoder passes -> compositor (nsr_fwd2.h)
        const int passes = bwd_passes(P.stage), rpb = 4;
        // blocks per decoder pass in proportion to the measured cost of a tile (the fine decoder: 288 MFMAs and two feature
        // gathers against 240 and one), one block per CU over all passes; waves per block from the largest tile share
        static const int w_fine_env = env_int("NSR_FWD_FINE_WEIGHT", 0);
        const int w_fine = w_fine_env > 0 ? w_fine_env : (P.stage == NSR_STAGE_FINE ? 12 : 14);     // (swept again in round 6: fine stage 12, colour stage 14)
        const long long tiles = (P.n_points_total + nsr::kTile - 1) / nsr::kTile;
        const int wsum = passes == 1 ? 10 : (passes == 2 ? 10 + w_fine : 20 + w_fine);
        long long most = 1;
        P.pass_beg[0] = 0;
        for (int p = 0; p < 3; ++p) {
            int nbp = 0;
            if (p < passes) {
                nbp = (int)((long long)kDefaultBwdBlocks * (p == 1 ? w_fine : 10) / wsum);
                if (nbp > tiles) nbp = (int)tiles;
                if (nbp < 1) nbp = 1;
                const long long share = (tiles + nbp - 1) / nbp;
                most = share > most ? share : most;
            }
            P.pass_beg[p + 1] = P.pass_beg[p] + nbp;
        }
        static const int fwd_cap = env_int("NSR_FWD_MAX_WAVES", nsr::kDxMaxWaves);   // (measurement: fewer waves per pass-kernel block)
        const int fcap = fwd_cap < 1 ? 1 : (fwd_cap > nsr::kDxMaxWaves ? nsr::kDxMaxWaves : fwd_cap);
        const int waves = (int)(most > fcap ? fcap : most);
        const dim3 rgrid((unsigned)((P.n_rays + rpb - 1) / rpb)), rblock(64 * rpb);
        NSR_LAUNCH(nsr::fwd_sample_kernel, rgrid, rblock, rpb * 64 * 8, stream, P);
        int lds = 0;
        const int first = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : NSR_MIDDLE, last = P.stage == NSR_STAGE_COARSE ? NSR_COARSE : P.stage;
        for (int kind = first; kind <= last; ++kind) {
            const int need = (nsr::AUX_FLOATS + nsr::packed_total(kind) + 4) * 4;
            lds = need > lds ? need : lds;
        }
        const dim3 grid(P.pass_beg[3]), block(64 * waves);
#define NSR_FWP(ST, SV)                                                                                      \
    if (int rc = launch_cfg(nsr::render_fwd_pass_kernel<ST, SV>, lds, "nsr_render_fwd(pass)")) return rc;    \
    if (a->ev_pass_start) nsr::rt_record(a->ev_pass_start, stream);                                          \
    NSR_LAUNCH((nsr::render_fwd_pass_kernel<ST, SV>), grid, block, lds, stream, P);                          \
    if (a->ev_pass_stop) nsr::rt_record(a->ev_pass_stop, stream);
        switch (P.stage) {
            case 0: NSR_FWP(0, true) NSR_LAUNCH(nsr::fwd_composite_kernel<0>, rgrid, rblock, rpb * 8, stream, P); break;
            case 1: NSR_FWP(1, true) NSR_LAUNCH(nsr::fwd_composite_kernel<1>, rgrid, rblock, rpb * 8, stream, P); break;
            case 2: NSR_FWP(2, true) NSR_LAUNCH(nsr::fwd_composite_kernel<2>, rgrid, rblock, rpb * 8, stream, P); break;
            default: NSR_FWP(3, true) NSR_LAUNCH(nsr::fwd_composite_kernel<3>, rgrid, rblock, rpb * 8, stream, P); break;
        }
#undef NSR_FWP
        return finish("nsr_render_fwd(split)");
    }
    const int npts = P.rays_per_block * P.S;
    const int lds = fwd_lds_bytes(P.stage, npts);
    const dim3 grid((unsigned)(P.n_groups < (1 << 20) ? P.n_groups : (1 << 20))), block(64 * P.tiles_per_block);
#define NSR_FWD(ST, SV)                                                                               \
    if (int rc = launch_cfg(nsr::render_fwd_kernel<ST, SV>, lds, "nsr_render_fwd")) return rc;         \
    NSR_LAUNCH((nsr::render_fwd_kernel<ST, SV>), grid, block, lds, stream, P);
    // (the one-launch kernel saves nothing: a call that will be differentiated passes acts + zvals + raw and took the branch above)
    P.acts = nullptr;
    switch (P.stage) {
        case 0: NSR_FWD(0, false) break;
        case 1: NSR_FWD(1, false) break;
        case 2: NSR_FWD(2, false) break;
        default: NSR_FWD(3, false) break;
    }
#undef NSR_FWD
    return finish("nsr_render_fwd");
}

int nsr_render_bwd(const nsr_render_args *a, const nsr_bwd_args *b, void *stream) {
    nsr::RenderParams P;
    if (int rc = build_params(a, P, true, true)) return rc;
    if (!b) return fail("nsr_render_bwd: null backward block");
    if (!a->raw) return fail("nsr_render_bwd: the forward pass must have saved `raw`");
    if (!b->d_depth && !b->d_var && !b->d_rgb) return fail("nsr_render_bwd: no output gradient given");
    if (!b->depth) return fail("nsr_render_bwd: forward depth is required");
    if ((b->d_rays_o == nullptr) != (b->d_rays_d == nullptr)) return fail("nsr_render_bwd: d_rays_o / d_rays_d must be given together");
    if (P.n_rays == 0) return 0;
    P.d_depth = b->d_depth; P.d_var = b->d_var; P.d_rgb = b->d_rgb; P.g_depth = b->depth; P.g_scale = b->grad_scale;
    P.d_rays_o = b->d_rays_o; P.d_rays_d = b->d_rays_d;
#ifdef NSR_TS
    if (const char *e = getenv("NSR_DBG_PTR")) P.dbg = reinterpret_cast<long long *>(strtoull(e, nullptr, 16));
#endif
    const int passes = bwd_passes(P.stage);
    bool any_params = false;
    for (int s = 0; s < 4; ++s) any_params |= P.dec[s].dparams != nullptr;
    P.partial_stride = max_param_count(P.stage);
    if (!P.acts || !a->zvals)
        return fail("nsr_render_bwd: the forward must have been given an activation buffer (nsr_render_args.acts, sized by nsr_acts_floats) and zvals");
    return render_bwd_split(a, b, P, any_params, stream);
}

int nsr_eval_points_fwd(const nsr_render_args *a, const double *points, int64_t n_points, float *out, void *stream) {
    nsr::RenderParams P;
    if (int rc = build_params(a, P, false)) return rc;
    if (n_points < 0 || (n_points > 0 && (!points || !out))) return fail("nsr_eval_points_fwd: bad points / out");
    if (n_points == 0) return 0;
    P.points = points; P.n_points = n_points; P.out_points = out;
    const int waves = kMaxTiles;
    const int wlf = P.stage == NSR_STAGE_COARSE ? nsr::packed_total(0) : (P.stage == NSR_STAGE_MIDDLE ? nsr::packed_total(1) : nsr::packed_total(2));
    const int lds = round16(3 * nsr::AUX_FLOATS * 4) + 16 + wlf * 4;
    const long long tiles = (n_points + nsr::kTile - 1) / nsr::kTile;
    long long nb = (tiles + waves - 1) / waves;
    if (nb > 2048) nb = 2048;
    const dim3 grid((unsigned)nb), block(64 * waves);
#define NSR_EVP(ST)                                                                              \
    case ST:                                                                                     \
        if (int rc = launch_cfg(nsr::eval_points_kernel<ST>, lds, "nsr_eval_points_fwd")) return rc; \
        NSR_LAUNCH(nsr::eval_points_kernel<ST>, grid, block, lds, stream, P);                     \
        break;
    switch (P.stage) { NSR_EVP(0) NSR_EVP(1) NSR_EVP(2) NSR_EVP(3) }
#undef NSR_EVP
    return finish("nsr_eval_points_fwd");
}

int nsr_masked_adam(float *p, const float *g, float *m, float *v, const uint8_t *voxel_mask, int64_t n_voxels,
                    float step_size, float beta1, float beta2, float eps, float bias2_sqrt, void *stream) {
    if (n_voxels < 0) return fail("nsr_masked_adam: negative voxel count");
    if (n_voxels == 0) return 0;
    if (!p || !g || !m || !v) return fail("nsr_masked_adam: null pointer");
    if (!(bias2_sqrt > 0.f)) return fail("nsr_masked_adam: bias2_sqrt must be positive (step >= 1)");
    nsr::AdamParams A;
    A.p = p; A.g = g; A.m = m; A.v = v; A.mask = voxel_mask; A.n_vox = n_voxels;
    A.step = step_size; A.b1 = beta1; A.b2 = beta2; A.eps = eps; A.rs2 = bias2_sqrt;
    const int tb = 256;
    const long long nthreads = n_voxels * 8;
    NSR_LAUNCH(nsr::masked_adam_kernel, dim3((unsigned)((nthreads + tb - 1) / tb)), dim3(tb), 0, stream, A);
    return finish("nsr_masked_adam");
}

int nsr_get_samples(const int64_t *indices, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                    int32_t W_full, float fx, float fy, float cx, float cy,
                    const float *c2w, int32_t c2w_stride, const float *depth, const float *color,
                    float *rays_o, float *rays_d, float *out_depth, float *out_color, void *stream) {
    if (n < 0 || H1 <= H0 || W1 <= W0 || W_full < W1) return fail("nsr_get_samples: bad crop");
    if (n == 0) return 0;
    if (!indices || !c2w || !depth || !color || !rays_o || !rays_d || !out_depth || !out_color)
        return fail("nsr_get_samples: null pointer");
    nsr::SampleParams S;
    S.indices = reinterpret_cast<const long long *>(indices);
    S.n = n; S.H0 = H0; S.W0 = W0; S.crop_w = W1 - W0; S.W_full = W_full;
    S.fx = fx; S.fy = fy; S.cx = cx; S.cy = cy;
    S.c2w = c2w; S.c2w_stride = c2w_stride; S.depth = depth; S.color = color;
    S.rays_o = rays_o; S.rays_d = rays_d; S.out_depth = out_depth; S.out_color = out_color;
    const int tb = 256;
    NSR_LAUNCH(nsr::get_samples_kernel, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, stream, S);
    return finish("nsr_get_samples");
}

int64_t nsr_frustum_workspace_floats(int64_t n_voxels) {
    return n_voxels < 0 ? -1 : n_voxels + (n_voxels + 255) / 256;
}

int nsr_frustum_mask(const float *w2c, const float *cam_center, double fx, double fy, double cx, double cy,
                     int32_t H, int32_t W, const float *depth, const float *xs, const float *ys, const float *zs,
                     int32_t nx, int32_t ny, int32_t nz, float *workspace, uint8_t *voxel_mask, void *stream) {
    if (nx < 0 || ny < 0 || nz < 0 || H <= 0 || W <= 0) return fail("nsr_frustum_mask: bad shape");
    const int64_t n_vox = (int64_t)nx * ny * nz;
    if (n_vox == 0) return 0;
    if (H > 32766 || W > 32766) return fail("nsr_frustum_mask: image larger than 32766 pixels per side");
    if (!w2c || !cam_center || !depth || !xs || !ys || !zs || !workspace || !voxel_mask)
        return fail("nsr_frustum_mask: null pointer");
    nsr::FrustumParams P;
    for (int i = 0; i < 12; ++i) P.w2c[i] = w2c[i];
    for (int i = 0; i < 3; ++i) P.cam_o[i] = cam_center[i];
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.H = H; P.W = W;
    P.depth = depth; P.xs = xs; P.ys = ys; P.zs = zs; P.nx = nx; P.ny = ny; P.nz = nz;
    const int tb = 256;
    P.nblocks = (int)((n_vox + tb - 1) / tb);
    P.n_vox = n_vox; P.ws = workspace; P.mask = voxel_mask;
    NSR_LAUNCH(nsr::frustum_mask_kernel<0>, dim3((unsigned)P.nblocks), dim3(tb), tb * sizeof(float), stream, P);
    NSR_LAUNCH(nsr::frustum_mask_kernel<1>, dim3((unsigned)P.nblocks), dim3(tb), tb * sizeof(float), stream, P);
    return finish("nsr_frustum_mask");
}

}  // extern "C"
namespace {
int window_launch(const int64_t *indices, int64_t *indices_out, uint64_t *rng, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                  int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                  float *rays_o, float *rays_d, float *out_depth, float *out_color,
                  const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream,
                  float *hdr = nullptr, float *zero = nullptr, int64_t zero_floats = 0, const uint64_t *peer_seeds = nullptr, int32_t n_peers = 0) {
    if (K < 0 || K > NSR_MAX_WINDOW) return fail("nsr_get_samples_window: K must be in [0, 32]");
    if (n < 0 || H1 <= H0 || W1 <= W0 || W_full < W1) return fail("nsr_get_samples_window: bad crop");
    if (K == 0 || n == 0) return 0;
    if ((!indices && (!indices_out || !rng)) || !frames || !rays_o || !rays_d || !out_depth || !out_color || !bound_lo || !bound_hi)
        return fail("nsr_get_samples_window: null pointer");
    if ((long long)(H1 - H0) * (W1 - W0) >= (1ll << 32)) return fail("nsr_get_samples_window: crop too large");
    nsr::WindowParams P;
    std::memset(&P, 0, sizeof(P));
    P.indices = reinterpret_cast<const long long *>(indices);
    P.indices_out = reinterpret_cast<long long *>(indices_out);
    P.rng = reinterpret_cast<unsigned long long *>(rng);
    P.crop_pixels = (unsigned)((H1 - H0) * (W1 - W0));
    P.n = n; P.K = K; P.H0 = H0; P.W0 = W0; P.crop_w = W1 - W0; P.W_full = W_full;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy;
    for (int k = 0; k < K; ++k) {
        if (!frames[k].depth || !frames[k].color || !frames[k].c2w) return fail("nsr_get_samples_window: null frame pointer");
        P.depth[k] = frames[k].depth; P.color[k] = frames[k].color; P.c2w[k] = frames[k].c2w; P.c2w_stride[k] = frames[k].c2w_stride;
    }
    P.rays_o = rays_o; P.rays_d = rays_d; P.out_depth = out_depth; P.out_color = out_color;
    for (int a = 0; a < 3; ++a) { P.lo[a] = bound_lo[a]; P.hi[a] = bound_hi[a]; }
    P.keep = keep; P.kept_max = kept_max;
    const int tb = 256;
    const long long sbx = (n + tb - 1) / tb;
    P.sample_bx = (int)sbx;
    long long fbx = 0;
    if (hdr) {
        // fill blocks beside the sampling blocks: 32 KB each (eight 16-byte stores per thread), at most ~2048 over the K grid rows
        if (zero_floats < 0 || (zero_floats > 0 && (!zero || (reinterpret_cast<uintptr_t>(zero) & 15)))) return fail("nsr_get_samples_window_fused: zero span must be 16-byte aligned");
        P.hdr = hdr; P.zero = zero; P.zero_n = zero_floats;
        long long want = (zero_floats * 4 + 32767) / 32768;
        if (want > 2048) want = 2048;
        fbx = (want + K - 1) / K;
    }
    if (n_peers) {
        if (n_peers < 0 || n_peers > NSR_MAX_PEERS || !peer_seeds) return fail("nsr_get_samples_window_sharded: 0..15 peers");
        if (indices || !hdr) return fail("nsr_get_samples_window_sharded: peers need the in-kernel pixel draw (indices == NULL) and the fused header");
        P.n_peers = n_peers;
        for (int p = 0; p < n_peers; ++p) P.peer_seed[p] = peer_seeds[p];
    }
    NSR_LAUNCH(nsr::get_samples_window_kernel, dim3((unsigned)(sbx * (1 + n_peers) + fbx), K), dim3(tb), 0, stream, P);
    return finish("nsr_get_samples_window");
}
}  // namespace
extern "C" {

int nsr_get_samples_window(const int64_t *indices, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                           int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                           float *rays_o, float *rays_d, float *out_depth, float *out_color,
                           const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream) {
    if (!indices) return fail("nsr_get_samples_window: null pointer");
    return window_launch(indices, nullptr, nullptr, K, n, H0, H1, W0, W1, W_full, fx, fy, cx, cy, frames, rays_o, rays_d, out_depth, out_color,
                         bound_lo, bound_hi, keep, kept_max, stream);
}

int nsr_get_samples_window_draw(int64_t *indices_out, uint64_t *rng_state, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0,
                                int32_t W1, int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                                float *rays_o, float *rays_d, float *out_depth, float *out_color,
                                const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream) {
    return window_launch(nullptr, indices_out, rng_state, K, n, H0, H1, W0, W1, W_full, fx, fy, cx, cy, frames, rays_o, rays_d, out_depth,
                         out_color, bound_lo, bound_hi, keep, kept_max, stream);
}

int nsr_get_samples_window_fused(const int64_t *indices, int64_t *indices_out, uint64_t *state, int32_t K, int64_t n, int32_t H0, int32_t H1,
                                 int32_t W0, int32_t W1, int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                                 float *rays_o, float *rays_d, float *out_depth, float *out_color,
                                 const double *bound_lo, const double *bound_hi, uint8_t *keep, float *header,
                                 float *zero, int64_t zero_floats, void *stream) {
    if (!state || !header) return fail("nsr_get_samples_window_fused: null pointer");
    if ((reinterpret_cast<uintptr_t>(header) & 15) || (reinterpret_cast<uintptr_t>(state) & 7))
        return fail("nsr_get_samples_window_fused: header must be 16-byte aligned (it is written with one 16-byte store), state 8-byte aligned");
    if (!indices && !indices_out) return fail("nsr_get_samples_window_fused: indices or indices_out is required");
    if (K == 0 || n == 0) return fail("nsr_get_samples_window_fused: empty window (the header and the zero span would stay unwritten)");
    return window_launch(indices, indices ? nullptr : indices_out, state, K, n, H0, H1, W0, W1, W_full, fx, fy, cx, cy, frames, rays_o, rays_d,
                         out_depth, out_color, bound_lo, bound_hi, keep, nullptr, stream, header, zero, zero_floats);
}

int nsr_get_samples_window_sharded(int64_t *indices_out, uint64_t *state, const uint64_t *peer_seeds, int32_t n_peers, int32_t K, int64_t n,
                                   int32_t H0, int32_t H1, int32_t W0, int32_t W1, int32_t W_full, float fx, float fy, float cx, float cy,
                                   const nsr_frame *frames, float *rays_o, float *rays_d, float *out_depth, float *out_color,
                                   const double *bound_lo, const double *bound_hi, uint8_t *keep, float *header,
                                   float *zero, int64_t zero_floats, void *stream) {
    if (!state || !header || !indices_out) return fail("nsr_get_samples_window_sharded: null pointer");
    if ((reinterpret_cast<uintptr_t>(header) & 15) || (reinterpret_cast<uintptr_t>(state) & 7)) return fail("nsr_get_samples_window_sharded: header must be 16-byte, state 8-byte aligned");
    if (K == 0 || n == 0) return fail("nsr_get_samples_window_sharded: empty window (the header and the zero span would stay unwritten)");
    return window_launch(nullptr, indices_out, state, K, n, H0, H1, W0, W1, W_full, fx, fy, cx, cy, frames, rays_o, rays_d,
                         out_depth, out_color, bound_lo, bound_hi, keep, nullptr, stream, header, zero, zero_floats, peer_seeds, n_peers);
}

int nsr_pose_grad(const int64_t *indices, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                  float fx, float fy, float cx, float cy, const float *d_rays_o, const float *d_rays_d,
                  float *d_c2w, int32_t out_stride, void *stream) {
    if (K < 0 || n < 0 || H1 <= H0 || W1 <= W0 || out_stride < 12) return fail("nsr_pose_grad: bad arguments");
    if (K == 0) return 0;
    if (!indices || !d_rays_o || !d_rays_d || !d_c2w) return fail("nsr_pose_grad: null pointer");
    nsr::PoseGradParams P;
    P.indices = reinterpret_cast<const long long *>(indices);
    P.n = n; P.H0 = H0; P.W0 = W0; P.crop_w = W1 - W0;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy;
    P.d_rays_o = d_rays_o; P.d_rays_d = d_rays_d; P.out = d_c2w; P.out_stride = out_stride;
    const int tb = 256;
    NSR_LAUNCH(nsr::pose_grad_kernel, dim3((unsigned)K), dim3(tb), 12 * tb * sizeof(float), stream, P);
    return finish("nsr_pose_grad");
}

int nsr_masked_adam_multi(const nsr_adam_grid *grids, int32_t n_grids, double beta1, double beta2, double eps,
                          int32_t zero_grad, float *scratch, void *stream) {
    if (n_grids < 0 || n_grids > 4) return fail("nsr_masked_adam_multi: 0..4 grids");
    if (n_grids == 0) return 0;
    if (!grids || !scratch) return fail("nsr_masked_adam_multi: null pointer");
    nsr::AdamMulti A;
    std::memset(&A, 0, sizeof(A));
    long long nmax = 0;
    for (int i = 0; i < n_grids; ++i) {
        const nsr_adam_grid &g = grids[i];
        if (!g.p || !g.g || !g.m || !g.v || !g.step || g.n_voxels < 0) return fail("nsr_masked_adam_multi: bad grid entry");
        A.p[i] = g.p; A.g[i] = g.g; A.m[i] = g.m; A.v[i] = g.v; A.mask[i] = g.voxel_mask; A.n_vox[i] = g.n_voxels;
        A.step[i] = g.step; A.lr[i] = g.lr;
        nmax = g.n_voxels > nmax ? g.n_voxels : nmax;
    }
    A.n = n_grids; A.b1 = (float)beta1; A.b2 = (float)beta2; A.eps = (float)eps; A.zero_grad = zero_grad; A.scal = scratch;
    A.b1d = beta1; A.b2d = beta2;
    NSR_LAUNCH(nsr::adam_tick_kernel, dim3(1), dim3(64), 0, stream, A);
    if (nmax > 0) {
        const int tb = 256;
        NSR_LAUNCH(nsr::masked_adam_multi_kernel, dim3((unsigned)((nmax * 8 + tb - 1) / tb), n_grids), dim3(tb), 0, stream, A);
    }
    return finish("nsr_masked_adam_multi");
}

int nsr_flat_adam(const nsr_adam_span *spans, int32_t n_spans, double beta1, double beta2, double eps, int32_t zero_grad, float *scratch,
                  void *stream) {
    if (n_spans < 0 || n_spans > 4) return fail("nsr_flat_adam: 0..4 spans");
    if (n_spans == 0) return 0;
    if (!spans || !scratch) return fail("nsr_flat_adam: null pointer");
    nsr::AdamMulti A;
    std::memset(&A, 0, sizeof(A));
    long long nmax = 0;
    for (int i = 0; i < n_spans; ++i) {
        const nsr_adam_span &g = spans[i];
        if (!g.p || !g.g || !g.m || !g.v || !g.step || g.n < 0) return fail("nsr_flat_adam: bad span entry");
        A.p[i] = g.p; A.g[i] = g.g; A.m[i] = g.m; A.v[i] = g.v; A.n_vox[i] = g.n; A.step[i] = g.step; A.lr[i] = g.lr;
        nmax = g.n > nmax ? g.n : nmax;
    }
    A.n = n_spans; A.b1 = (float)beta1; A.b2 = (float)beta2; A.eps = (float)eps; A.zero_grad = zero_grad; A.scal = scratch;
    A.omb1 = (float)(1.0 - beta1); A.omb2 = (float)(1.0 - beta2);
    A.b1d = beta1; A.b2d = beta2;
    NSR_LAUNCH(nsr::adam_tick_kernel, dim3(1), dim3(64), 0, stream, A);
    if (nmax > 0) {
        const int tb = 256;
        NSR_LAUNCH(nsr::flat_adam_kernel, dim3((unsigned)((nmax + tb - 1) / tb), n_spans), dim3(tb), 0, stream, A);
    }
    return finish("nsr_flat_adam");
}

int nsr_pack_rows(const nsr_rows *grids, int32_t n_grids, const nsr_span *spans, int32_t n_spans, float *packed,
                  int32_t unpack, void *stream) {
    if (n_grids < 0 || n_grids > 4 || n_spans < 0 || n_spans > 4) return fail("nsr_pack_rows: at most 4 grids and 4 spans");
    if ((n_grids && !grids) || (n_spans && !spans) || !packed) return fail("nsr_pack_rows: null pointer");
    nsr::PackParams P;
    std::memset(&P, 0, sizeof(P));
    long long total = 0;
    for (int i = 0; i < n_grids; ++i) {
        if (grids[i].n_rows < 0 || (grids[i].n_rows && (!grids[i].grid || !grids[i].rows))) return fail("nsr_pack_rows: bad grid entry");
        P.grid[i] = grids[i].grid; P.rows[i] = reinterpret_cast<const long long *>(grids[i].rows); P.n_rows[i] = grids[i].n_rows;
        total += grids[i].n_rows * nsr::kC;
    }
    for (int i = 0; i < n_spans; ++i) {
        if (spans[i].n < 0 || (spans[i].n && !spans[i].ptr)) return fail("nsr_pack_rows: bad span entry");
        P.span[i] = spans[i].ptr; P.span_n[i] = spans[i].n;
        total += spans[i].n;
    }
    P.n_grids = n_grids; P.n_spans = n_spans; P.unpack = unpack ? 1 : 0; P.packed = packed; P.total = total;
    if (total == 0) return 0;
    const int tb = 256;
    NSR_LAUNCH(nsr::pack_rows_kernel, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, stream, P);
    return finish("nsr_pack_rows");
}

int nsr_aabb_keep(const float *rays_o, const float *rays_d, const float *gt_depth, int64_t n,
                  const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream) {
    if (n < 0) return fail("nsr_aabb_keep: negative ray count");
    if (n == 0) return 0;
    if (!rays_o || !rays_d || !gt_depth || !bound_lo || !bound_hi || !keep) return fail("nsr_aabb_keep: null pointer");
    nsr::AabbParams P;
    P.rays_o = rays_o; P.rays_d = rays_d; P.gt_depth = gt_depth; P.n = n;
    for (int a = 0; a < 3; ++a) { P.lo[a] = bound_lo[a]; P.hi[a] = bound_hi[a]; }
    P.keep = keep; P.kept_max = kept_max;
    const int tb = 256;
    NSR_LAUNCH(nsr::aabb_keep_kernel, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, stream, P);
    return finish("nsr_aabb_keep");
}

int nsr_tracking_loss(int64_t n_rays, const float *gt_depth, const float *gt_color, const uint8_t *keep,
                      const double *depth, const double *var, const float *rgb,
                      int32_t handle_dynamic, int32_t use_color, float w_color,
                      double *loss, double *dl_depth, float *dl_rgb, void *stream) {
    if (n_rays < 0) return fail("nsr_tracking_loss: negative ray count");
    if (n_rays == 0) return 0;
    if (!gt_depth || !depth || !var || !loss || !dl_depth) return fail("nsr_tracking_loss: null pointer");
    if (use_color && (!gt_color || !rgb || !dl_rgb)) return fail("nsr_tracking_loss: the colour term needs gt_color, rgb and dl_rgb");
    nsr::TrackLossParams P;
    P.n = n_rays; P.gt_depth = gt_depth; P.gt_color = gt_color; P.rgb = rgb; P.keep = keep; P.depth = depth; P.var = var;
    P.handle_dynamic = handle_dynamic ? 1 : 0; P.use_color = use_color ? 1 : 0; P.w_color = w_color;
    P.loss = loss; P.dl_depth = dl_depth; P.dl_rgb = dl_rgb;
    const int tb = n_rays <= 256 ? 256 : 1024;                       // one block: the median is a property of the whole batch
    const int key_cap = (handle_dynamic && n_rays <= 4096) ? (int)n_rays : 0;      // tmp bit patterns cached in LDS (<= 32 KB), else recomputed
    // (key slots for every thread of the block: the one-ray-per-thread path of n <= tb pads the key list with skip marks)
    NSR_LAUNCH(nsr::tracking_loss_kernel, dim3(1), dim3(tb), 1024 + 32 + tb * 8 + (key_cap > 0 && key_cap < tb ? tb : key_cap) * 8, stream, P, key_cap);
    return finish("nsr_tracking_loss");
}

int nsr_camera_from_tensor(const float *cam, int64_t n, float *rt, const float *d_rt, float *d_cam, void *stream) {
    if (n < 0) return fail("nsr_camera_from_tensor: negative count");
    if (n == 0) return 0;
    if (!cam || (!d_rt && !rt) || (d_rt && !d_cam)) return fail("nsr_camera_from_tensor: null pointer");
    nsr::CamParams P;
    P.cam = cam; P.n = n; P.rt = rt; P.d_rt = d_rt; P.d_cam = d_cam;
    const int tb = 64;
    NSR_LAUNCH(nsr::camera_from_tensor_kernel, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, stream, P);
    return finish("nsr_camera_from_tensor");
}

}  // extern "C"
