#!/usr/bin/env python3
"""bench.py -- rendered rays/s (forward + backward) of the NICE-SLAM mapping render hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched under
torch.distributed.run, one rank per GPU.  One STEP = one mapping iteration's pass of the hot path over one
ray batch of BASELINE configs[1] (Replica room0 full config): get_samples for the 5-keyframe window ->
render_batch_ray -> mapping loss -> backward (grid, decoder and nothing else; no optimiser: Adam and the
masked write-back belong to the unchanged caller).  The stage of step i follows the reference
schedule of a 60-iteration frame batch (25 middle / 12 fine / 23 color, src/Mapper.py:403-410), so the
default K=60 reproduces the Replica mapping mix exactly.  Prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

RAYS_PER_GPU = 1000            # mapping.pixels (configs/Replica/replica.yaml)
WINDOW = 5                     # mapping_window_size -> 5 x 200 pixels
FP32_PEAK = 157.3e12           # MI355X dense fp32 MFMA peak (MI355X_MICROARCH.md)
# necessary backward FLOP per ray of the dominant kernel (color-stage backward): SURVEY §8(d)
#   (106 140 - 51 653) MAC/pt * 2 FLOP * 48 pts  (dX chain + stepped dW + d-embedding; the forward is a separate launch)
BWD_COLOR_FLOP_PER_RAY = (106140 - 51653) * 2 * 48
# what the colour backward kernel EXECUTES on the MFMA pipe: forward re-run + dX + dW of all three decoders, like the
# reference's autograd (SURVEY §8(d) "reference-equivalent" column: 154 959 MAC/pt)
BWD_COLOR_EXEC_FLOP_PER_RAY = 154959 * 2 * 48
FWD_FLOP_PER_RAY = {"middle": 15479 * 2 * 48, "fine": 36078 * 2 * 48, "color": 51653 * 2 * 48}


def _stage_cycle(n=60):
    """The 60 iterations of one optimize_map call (middle while it <= 0.4 n, fine while it <= 0.6 n, then colour:
    src/Mapper.py:402-410 -> 25 / 12 / 23), interleaved evenly so that ANY window of K timed steps carries the same mix."""
    counts = {"middle": 0, "fine": 0, "color": 0}
    for it in range(n):
        counts["middle" if it <= int(n * 0.4) else ("fine" if it <= int(n * 0.6) else "color")] += 1
    done = {k: 0 for k in counts}
    seq = []
    for i in range(1, n + 1):                       # largest deficit first (weighted round-robin)
        k = max(counts, key=lambda s: (counts[s] * i / n - done[s], counts[s]))
        done[k] += 1
        seq.append(k)
    return seq


_CYCLE = _stage_cycle()


def stage_of(it, n=60):
    return _CYCLE[it % n]


class HipEvents:
    """Raw hipEvent pairs (the kernels run on torch's current stream, recorded inside nsr_render_bwd)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so.7")      # already mapped by torch: same runtime instance
        self.pairs = {}

    def new(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def pair_for(self, stage):
        p = (self.new(), self.new())
        self.pairs.setdefault(stage, []).append(p)
        return p[0], p[1]

    def summary(self):
        out = {}
        for stage, pairs in self.pairs.items():
            ms = []
            for a, b in pairs:
                t = ctypes.c_float()
                if self.hip.hipEventElapsedTime(ctypes.byref(t), a, b) == 0:
                    ms.append(t.value)
            if ms:
                out[stage] = (sum(ms) / len(ms), len(ms))
        return out


# configs/Replica/room0.yaml + replica.yaml + nice_slam.yaml (BASELINE configs[1])
REPLICA_ROOM0 = {
    "scale": 1, "occupancy": True, "coarse": True,
    "mapping": {"bound": [[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]]},
    "grid_len": {"coarse": 2, "middle": 0.32, "fine": 0.16, "color": 0.16, "bound_divisible": 0.32},
    "model": {"c_dim": 32, "coarse_bound_enlarge": 2},
    "rendering": {"lindisp": False, "perturb": 0.0, "N_samples": 32, "N_surface": 16, "N_importance": 0},
    "cam": {"H": 680, "W": 1200, "fx": 600.0, "fy": 600.0, "cx": 599.5, "cy": 339.5},
}


def build_scene(dev, seed=0):
    """Synthetic workload of BASELINE configs[1], built with the product's own set-up code (no oracle involved): grids with
    the reference's init statistics (NICE_SLAM.py:223-247), random-init decoders (no pretrained weights exist here), one
    680x1200 RGB-D frame with depth U(1,4) m and 1 % zeros, a pose at the centre of the bound."""
    import math
    import types
    import nice_slam_amd as nsa
    from nice_slam_amd.common import set_decoder_bounds
    cfg = REPLICA_ROOM0
    torch.manual_seed(seed)
    bound = nsa.load_bound(cfg)
    cam = cfg["cam"]
    slam = types.SimpleNamespace(nice=True, bound=bound, H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    renderer = nsa.Renderer(cfg, None, slam)
    dec = nsa.NICE(coarse=True).to(dev)
    set_decoder_bounds(dec, bound, cfg["model"]["coarse_bound_enlarge"])
    grids = {k: v.to(dev) for k, v in nsa.grid_init(cfg, bound).items()}
    g = torch.Generator().manual_seed(seed + 1)
    depth = torch.rand((cam["H"], cam["W"]), generator=g) * 3.0 + 1.0
    depth[torch.rand((cam["H"], cam["W"]), generator=g) < 0.01] = 0.0
    color = torch.rand((cam["H"], cam["W"], 3), generator=g)
    ang = 0.15
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    c2w[:3, 3] = bound.mean(1).float()
    return {"cfg": cfg, "bound": bound, "renderer": renderer, "dec": dec, "grids": grids, "c2w": c2w,
            "depth_img": depth, "color_img": color, "intr": (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])}


def cpu_baseline(sc, n_rays, reps=1):
    """The CPU oracle (a torch restatement of the reference path, kind='port') on this box's host cores, fed with the SAME
    grids / decoder parameters / frame as the GPU run: one forward+backward per stage on an `n_rays` batch, stage-weighted
    like the GPU run.  The only place of this file that touches oracle/."""
    from oracle import nice_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))       # small-op torch CPU code scales poorly past ~16 threads
    H, W, fx, fy, cx, cy = sc["intr"]
    grids = {k: v.detach().cpu().contiguous() for k, v in sc["grids"].items()}                # NCDHW, standard strides
    params = {k: v.detach().cpu().clone() for k, v in sc["dec"].state_dict().items()}
    idx = torch.randint(H * W, (n_rays,), generator=torch.Generator().manual_seed(5))
    rays_o, rays_d, gt_depth, gt_color = orc.pixel_rays(idx, 0, H, 0, W, fx, fy, cx, cy, sc["c2w"], sc["depth_img"], sc["color_img"])

    def once(stage, n):
        G = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
        P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        depth, _, col = orc.render_batch_ray(G, P, rays_d[:n], rays_o[:n], stage, gt_depth[:n], sc["bound"])
        loss = (torch.abs(gt_depth[:n] - depth) * (gt_depth[:n] > 0)).sum()
        if stage == "color":
            loss = loss + 0.2 * torch.abs(gt_color[:n] - col).sum()
        loss.backward()

    t = {}
    for stage in ("middle", "fine", "color"):
        once(stage, 64)                                        # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            once(stage, n_rays)
        t[stage] = (time.perf_counter() - t0) / reps
    mix = (25 * t["middle"] + 12 * t["fine"] + 23 * t["color"]) / 60.0
    return {"value": n_rays / mix, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fwd+bwd, {n_rays} rays, 1 iter per stage (middle {t['middle']*1e3:.0f} ms, fine {t['fine']*1e3:.0f} ms, "
                      f"color {t['color']*1e3:.0f} ms), weighted 25/12/23"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage", default=None, help="pin every step to one stage (profiling)")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU)
    ap.add_argument("--eager", action="store_true", help="do not capture the iteration in a hipGraph")
    ap.add_argument("--stepped-grads-only", action="store_true",
                    help="parameter gradients only for the decoder the reference's optimiser steps (colour); default: all, like the reference autograd")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="multi-GPU: weak = --rays per GPU (default), strong = --rays in total, split over the ranks")
    ap.add_argument("--dense-exchange", action="store_true",
                    help="multi-GPU: all-reduce the whole feature-grid gradients instead of the frustum-selected voxel rows")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the result JSON: anything a library prints there (RCCL's version banner, ...) is
    # sent to stderr by pointing file descriptor 1 at it; the JSON goes to the saved original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import signal
        signal.alarm(int(os.environ.get("NSR_BENCH_DEADLINE_S", "900")))   # a wedged collective must not hang the node: die loudly
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NSR_SINGLE_DEVICE") == "1":                # CI on a 1-GPU box: every rank on device 0 (with gloo, see below)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("NSR_FORCE_SHARDED") == "1"        # exercise the RCCL path with a single rank (CI on a 1-GPU box)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("NSR_DIST_BACKEND", "nccl")       # "nccl" = RCCL; "gloo" only to exercise the multi-rank
        if backend == "nccl":                                      # code path where RCCL cannot run (ranks sharing one GPU)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import nice_slam_amd as nsa
    from nice_slam_amd.parallel import ShardedRenderer

    n_total = args.rays * world if args.scaling == "weak" else args.rays      # weak: fixed rays per GPU; strong: fixed batch
    sc = build_scene(dev)                                          # same seed -> identical scene on every rank
    renderer, dec, grids = sc["renderer"], sc["dec"], sc["grids"]
    grids = {k: v.requires_grad_(True) for k, v in grids.items()}
    for n_, p in dec.named_parameters():                          # reference: every decoder parameter has requires_grad=True
        p.requires_grad_(True)
    if args.stepped_grads_only:
        renderer.decoder_grads = ("color",)
    H, W, fx, fy, cx, cy = sc["intr"]
    depth_img, color_img, c2w = sc["depth_img"].to(dev), sc["color_img"].to(dev), sc["c2w"].to(dev)
    params = list(dec.parameters())
    ev = HipEvents()
    renderer.profile_events = ev.pair_for
    rend = ShardedRenderer(renderer) if (world > 1 or force_dist) else renderer
    exchange = "single GPU"
    if world > 1 or force_dist:
        exchange = f"ray-sharded x{world}, dense RCCL all-reduce of grid grads"
        if not args.dense_exchange:
            # The mapper optimises only the voxels inside the current frame's frustum mask (Mapper.py:315-333), identical
            # on every rank: exchange those voxel rows only (parallel.ShardedRenderer.set_voxel_masks).
            sel = nsa.FrustumSelector(sc["bound"], H, W, fx, fy, cx, cy)
            pose = torch.eye(4)
            pose[:3] = sc["c2w"][:3].detach().cpu().float()
            masks = {k: sel.voxel_mask(pose, k, v.shape[2:], depth_img) for k, v in grids.items() if k != "grid_coarse"}
            rend.set_voxel_masks(masks)
            frac = {k[5:]: round(float(m.float().mean()), 3) for k, m in masks.items()}
            exchange = (f"ray-sharded x{world}, one packed RCCL all-reduce per iteration over the frustum-selected voxel rows "
                        f"(selected fraction {frac}) + decoder grads")
    torch.manual_seed(1234)                                       # identical index draws on every rank
    per_frame = n_total // WINDOW

    def step(it, timed):
        stage = args.stage or stage_of(it)
        ro, rd, gd, gc = [], [], [], []
        for _ in range(WINDOW):
            o, d, dep, col = nsa.get_samples(0, H, 0, W, per_frame, H, W, fx, fy, cx, cy, c2w, depth_img, color_img, dev)
            ro.append(o); rd.append(d); gd.append(dep); gc.append(col)
        rays_o, rays_d, gt_depth, gt_color = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
        for g in grids.values():
            g.grad = None
        for p in params:
            p.grad = None
        if not timed:
            renderer.profile_events = None
        depth, unc, color = rend.render_batch_ray(grids, dec, rays_d, rays_o, dev, stage, gt_depth=gt_depth)
        # src/Mapper.py:487-489 sums |gt - depth| over gt > 0; written as a masked product so that no boolean-index
        # (nonzero -> host sync) sits inside the iteration: same value, the host keeps running ahead of the GPU
        loss = (torch.abs(gt_depth - depth) * (gt_depth > 0)).sum()
        if stage == "color":
            loss = loss + 0.2 * torch.abs(gt_color - color).sum()  # :490-493
        loss.backward()
        renderer.profile_events = ev.pair_for
        return stage

    # Warm-up: eager iterations.  First untimed ones (code load, allocator, LDS attribute), then 5 per stage with HIP
    # events around the backward kernel -- these feed `roofline` / `kernel_ms`.
    reps = tuple(_CYCLE.index(k) for k in ("middle", "fine", "color")) if args.stage is None else (0,)
    for i in range(max(args.warmup, 2 * len(reps))):
        step(reps[i % len(reps)], False)
    torch.cuda.synchronize()
    for st_i in reps:
        for _ in range(5):
            step(st_i, True)
    torch.cuda.synchronize()
    # RCCL collectives are capturable too; NSR_DIST_GRAPH=0 forces the eager path for multi-rank runs
    use_graph = not args.eager and ((world == 1 and not force_dist) or os.environ.get("NSR_DIST_GRAPH", "1") == "1") \
        and os.environ.get("NSR_DIST_BACKEND", "nccl") == "nccl"
    graphs = {}
    if use_graph:
        # The mapping iteration is launch-bound on the host (~25 small launches around three big kernels): capture one
        # hipGraph per stage (identical kernel sequence, fresh torch.randint draws on every replay) and replay it.
        renderer.profile_events = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for st_i in reps:
                for _ in range(2):
                    step(st_i, False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            for st_i in reps:
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph):
                    st_name = step(st_i, False)
                graphs[st_name] = gph
            torch.cuda.synchronize()
            for gph in graphs.values():                 # first replays pay the one-time upload of the executable graph:
                for _ in range(3):                      # part of the warm-up, not of the timed region
                    gph.replay()
            torch.cuda.synchronize()
        except Exception as e:                      # e.g. a collective that cannot be captured: run eagerly instead
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); falling back to eager", file=sys.stderr)
            graphs.clear()
            use_graph = False
            torch.cuda.synchronize()

    def timed_step(i):
        if not use_graph:
            return step(i, False)
        stage = args.stage or stage_of(i)
        graphs[stage].replay()
        return stage

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stages = [timed_step(i) for i in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ksum = ev.summary()
        res = {
            "metric": "rendered rays/sec (fwd+bwd) per mapping iter", "value": n_total * args.steps / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32 (f64 sample placement / depth)",
            "data": "synthetic",
            "config": {"workload": "Replica room0 full config (BASELINE configs[1]): grids 21x28x37 / 43x56x74 x2, 32 ch fp32, "
                                   "random-init decoders, 680x1200 synthetic RGB-D, 5x200 pixels/iter, S=32+16",
                       "rays_per_gpu": n_total // world, "rays_per_iteration": n_total, "stage_mix": {s: stages.count(s) for s in sorted(set(stages))},
                       "timed_region": "get_samples x5 + render_batch_ray + mapping loss (sync-free form) + backward (all grid + all decoder grads, like the reference autograd), no optimiser",
                       "decoder_grads": "colour decoder only (what Mapper's optimiser steps)" if args.stepped_grads_only else "all decoders (reference autograd semantics)",
                       "launch": "hipGraph replay (one captured graph per stage)" if use_graph else "eager",
                       "parallelism": exchange},
        }
        if "color" in ksum:
            ms, cnt = ksum["color"]
            rays_launch = n_total // world                      # rays per backward launch on this rank
            ach = rays_launch * BWD_COLOR_FLOP_PER_RAY / (ms * 1e-3)
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")      # from a separate rocprofv3 --pmc run (tools/pmc_summary.py)
            if os.path.exists(tpath) and rays_launch == RAYS_PER_GPU:
                tj = json.load(open(tpath)).get("nsr::render_bwd_kernel<3>", {})
                traffic, tsrc = tj.get("hbm_bytes_per_launch"), "profiles/r01_traffic.json: " + tj.get("note", "")
            res["roofline"] = {"bound": "mfma", "kernel": "render_bwd_kernel<color>", "achieved": ach / 1e12, "peak": FP32_PEAK / 1e12,
                               "unit": "TFLOP/s", "frac": ach / FP32_PEAK, "traffic": traffic, "traffic_source": tsrc,
                               "avg_kernel_ms": ms, "launches": cnt,
                               "measured": "HIP events recorded inside nsr_render_bwd on the launch stream, eager iterations of this process"
                                           + (" (the timed region replays the captured graph of the same kernels)" if use_graph else ""),
                               "algorithmic_flop_per_launch": rays_launch * BWD_COLOR_FLOP_PER_RAY,
                               "executed_frac": (rays_launch * BWD_COLOR_EXEC_FLOP_PER_RAY / (ms * 1e-3)) / FP32_PEAK
                               if not args.stepped_grads_only else None,
                               "executed_note": "MFMA work the kernel actually issues (forward re-run + dX + dW for every decoder, "
                                                "the reference autograd's semantics) over the same peak; `frac` counts only the necessary part"}
        if world > 1 or force_dist:
            res["config"]["grad_exchange_MB_last_iter"] = round(rend.last_exchange_floats * 4 / 1e6, 2)
        res["kernel_ms"] = {f"render_bwd<{s}>": round(v[0], 4) for s, v in ksum.items()}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(sc, args.rays)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
