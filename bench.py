#!/usr/bin/env python3
"""bench.py -- rendered rays/s (forward + backward) of the NICE-SLAM mapping render hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 either launched under torch.distributed.run
(one rank per GPU: WORLD_SIZE / RANK / LOCAL_RANK in the environment) or, as a plain ``python bench.py --gpus N``, this
script starts the N ranks itself (`spawn_ranks`); rank 0 prints the line, `--verify-shards` is on by default for N > 1.  One STEP = one mapping iteration's pass of the hot path over one ray batch:
window sampling (pixel draw inside the kernel, all keyframes, bounding-box mask) -> render forward -> mapping loss -> backward (every grid,
every decoder, like the reference's autograd; no optimiser: Adam and the masked write-back belong to the caller,
SURVEY §8(f)).  Default workload = BASELINE configs[1] (Replica room0 full config); ``--config {0,2,3,4,tracking}`` selects
the other BASELINE configurations.  For the staged configurations the stage of step i follows the reference schedule of a
60-iteration frame batch (25 middle / 12 fine / 23 color, src/Mapper.py:403-410), interleaved so that any window carries that
mix.  The K steps are timed ``--windows`` (default 3) times; the MEDIAN window is reported.  Prints ONE JSON line.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_PEAK = 157.3e12           # MI355X dense fp32 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12
# necessary / executed MAC per sample point, SURVEY §8(d)
FWD_MAC = {"coarse": 6176, "middle": 15479, "fine": 36078, "color": 51653}
NEC_MAC = {"coarse": 12352, "middle": 24727, "fine": 59694, "color": 106140}          # fwd + bwd, what the optimiser needs
# MACs per sample point of decoder forward + dX + dW of every decoder (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP, first measured on
# round 2's kernel that ran all three, profiles/r02_pmc_bench_kernels.txt; coarse: 3 x forward); the split backward issues this minus
# the forward (FWD_MAC) -- 7.6e6 MOPS per colour-stage dX launch in profiles/r04_pmc_bench_kernels.txt agree -- which is what
# `executed_frac` below uses
# of the backward's necessary MACs: the dW part = the weights of the decoder the mapper's optimiser steps in that stage (colour stage: the
# colour decoder, 32x96 + 3 x 32x32 + 32x128 + 5 x 32x32 + 3x32 = 15 456); the rest is dX (gradients to the stepped grids through all decoders)
DW_NEC_MAC = {"coarse": 0, "middle": 0, "fine": 0, "color": 15456}
EXEC_BWD_MAC = {"coarse": 3 * 6176, "middle": 46080, "fine": 102400, "color": 148480}

# ---- BASELINE.json configs ------------------------------------------------------------------------
GRID_LEN = {"coarse": 2, "middle": 0.32, "fine": 0.16, "color": 0.16, "bound_divisible": 0.32}
CONFIGS = {
    "0": dict(name="Replica room0, 32x32 pixel batch, coarse grid only (BASELINE configs[0])", bound=[[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]],
              cam=(680, 1200, 600.0, 600.0, 599.5, 339.5), rays=1024, window=1, stages=("coarse",), kind="mapping"),
    "1": dict(name="Replica room0 full config (BASELINE configs[1])", bound=[[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]],
              cam=(680, 1200, 600.0, 600.0, 599.5, 339.5), rays=1000, window=5, stages="mix", kind="mapping"),
    "2": dict(name="ScanNet scene0000_00 full config (BASELINE configs[2])", bound=[[-2.0, 11.0], [-2.0, 11.5], [-2.0, 5.5]],
              cam=(460, 620, 577.590698, 578.729797, 308.905426, 232.683609), rays=5000, window=10, stages="mix", kind="mapping"),
    "3": dict(name="Apartment large-scene config (BASELINE configs[3])", bound=[[-5.8, 11.3], [-4.0, 4.5], [-7.9, 4.9]],
              cam=(720, 1280, 607.4694213867188, 607.4534912109375, 636.9967041015625, 369.2689514160156), rays=5000, window=10,
              stages="mix", kind="mapping"),
    "4": dict(name="Synthetic 1024x1024 RGB-D, 64^3 fine grid, 100k rays/iter (BASELINE configs[4])",
              bound=[[-5.12, 5.11], [-5.12, 5.11], [-5.12, 5.11]], cam=(1024, 1024, 512.0, 512.0, 511.5, 511.5), rays=100000, window=5,
              stages=("color",), kind="mapping"),
    "tracking": dict(name="Replica room0 tracking (BASELINE configs[1], tracker side: 200 pixels, 100-pixel border, pose gradient only)",
                     bound=[[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]], cam=(680, 1200, 600.0, 600.0, 599.5, 339.5), rays=200, window=1,
                     stages=("color",), kind="tracking", crop=100),
}


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


class HipEvents:
    """Raw hipEvents (the kernels run on torch's current stream; the records happen inside nsr_render_fwd / nsr_render_bwd)."""

    def __init__(self, n=2):
        self.hip = ctypes.CDLL("libamdhip64.so.7")      # already mapped by torch: same runtime instance
        self.n = n                                      # events per record: 2 = (start, stop); 4 = (start, stop, behind dX, behind dW)
        self.recs = {}

    def new(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def pair_for(self, stage):
        p = tuple(self.new() for _ in range(self.n))
        self.recs.setdefault(stage, []).append(p)
        return p

    def _ms(self, a, b):
        t = ctypes.c_float()
        return t.value if self.hip.hipEventElapsedTime(ctypes.byref(t), a, b) == 0 else None

    def summary(self):
        """stage -> (mean ms start..stop, count)"""
        out = {}
        for stage, recs in self.recs.items():
            ms = [v for v in (self._ms(r[0], r[1]) for r in recs) if v is not None]
            if ms:
                out[stage] = (sum(ms) / len(ms), len(ms))
        return out

    def split(self):
        """stage -> {"dx": ms, "dw": ms, "finalize": ms} (means; only records whose four events were all reached)"""
        out = {}
        for stage, recs in self.recs.items():
            acc = []
            for r in recs:
                if len(r) < 4:
                    continue
                v = (self._ms(r[0], r[2]), self._ms(r[2], r[3]), self._ms(r[3], r[1]))
                if all(x is not None for x in v):
                    acc.append(v)
            if acc:
                out[stage] = {k: sum(a[i] for a in acc) / len(acc) for i, k in enumerate(("dx", "dw", "finalize"))}
        return out


def build_scene(cfg_id, dev, seed=0):
    """Synthetic workload of a BASELINE configuration, built with the product's own set-up code (no oracle involved): grids
    with the reference's init statistics (NICE_SLAM.py:223-247), random-init decoders (no pretrained weights exist here),
    `window` RGB-D frames with depth U(1,4) m and 1 % zeros, poses near the centre of the bound."""
    import nice_slam_amd as nsa
    from nice_slam_amd.common import set_decoder_bounds
    C = CONFIGS[cfg_id]
    H, W, fx, fy, cx, cy = C["cam"]
    cfg = {"scale": 1, "occupancy": True, "coarse": True, "mapping": {"bound": C["bound"]}, "grid_len": GRID_LEN,
           "model": {"c_dim": 32, "coarse_bound_enlarge": 2},
           "rendering": {"lindisp": False, "perturb": 0.0, "N_samples": 32, "N_surface": 16, "N_importance": 0}}
    torch.manual_seed(seed)
    bound = nsa.load_bound(cfg)
    slam = types.SimpleNamespace(nice=True, bound=bound, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy)
    renderer = nsa.Renderer(cfg, None, slam)
    dec = nsa.NICE(coarse=True).to(dev)
    set_decoder_bounds(dec, bound, cfg["model"]["coarse_bound_enlarge"])
    grids = {k: v.to(dev) for k, v in nsa.grid_init(cfg, bound).items()}
    g = torch.Generator().manual_seed(seed + 1)
    frames = []
    for k in range(C["window"]):
        depth = torch.rand((H, W), generator=g) * 3.0 + 1.0
        depth[torch.rand((H, W), generator=g) < 0.01] = 0.0
        color = torch.rand((H, W, 3), generator=g)
        ang = 0.15 + 0.05 * k
        c2w = torch.eye(4)
        c2w[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        c2w[:3, 3] = bound.mean(1).float() + 0.05 * k
        frames.append((c2w, depth, color))
    return {"cfg": cfg, "C": C, "bound": bound, "renderer": renderer, "dec": dec, "grids": grids, "frames": frames,
            "intr": (H, W, fx, fy, cx, cy)}


def cpu_baseline(sc, n_rays, stages, weights, track_crop=None):
    """The CPU oracle (a torch restatement of the reference path, kind='port') on this box's host cores, in the mode that
    calls ATen's grid_sampler_3d exactly like the reference (decoder.py:173; its backward is 59 % of the reference's CPU time),
    fed with the SAME grids / decoder parameters / frames as the GPU run: one forward+backward per stage on a bounded ray
    sample, stage-weighted like the GPU run.  The only place of this file that touches oracle/."""
    from oracle import nice_oracle as orc
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)                               # small-op torch CPU code scales poorly past ~16 threads
    orc.TRILINEAR_IMPL = "grid_sample"
    H, W, fx, fy, cx, cy = sc["intr"]
    grids = {k: v.detach().cpu().contiguous() for k, v in sc["grids"].items()}                # NCDHW, standard strides
    params = {k: v.detach().cpu().clone() for k, v in sc["dec"].state_dict().items()}
    c2w, depth_img, color_img = sc["frames"][0]
    n = min(n_rays, 1000)                                       # bounded sample: ~10-20 s of CPU work
    idx = torch.randint(H * W, (n,), generator=torch.Generator().manual_seed(5))
    rays_o, rays_d, gt_depth, gt_color = orc.pixel_rays(idx, 0, H, 0, W, fx, fy, cx, cy, c2w, depth_img, color_img)

    def once_track(m):                                          # Tracker.optimize_cam_in_batch (Tracker.py:86-125): gradient to the pose only
        pose = c2w[:3].clone().requires_grad_(True)
        e = track_crop
        ti = torch.randint((H - 2 * e) * (W - 2 * e), (m,), generator=torch.Generator().manual_seed(6))
        o, d, gd, gc = orc.pixel_rays(ti, e, H - e, e, W - e, fx, fy, cx, cy, pose, depth_img, color_img)
        with torch.no_grad():
            tt = (sc["bound"].unsqueeze(0) - o.detach().unsqueeze(-1)) / d.detach().unsqueeze(-1)
            inside = torch.min(torch.max(tt, dim=2)[0], dim=1)[0] >= gd
        o, d, gd, gc = o[inside], d[inside], gd[inside], gc[inside]
        kept_n.append(int(inside.sum()))
        depth, unc, col = orc.render_batch_ray(grids, params, d, o, "color", gd, sc["bound"])
        unc = unc.detach()
        tmp = torch.abs(gd - depth) / torch.sqrt(unc + 1e-10)
        mask = (tmp < 10 * tmp.median()) & (gd > 0)
        (tmp[mask].sum() + 0.5 * torch.abs(gc - col)[mask].sum()).backward()

    kept_n = []
    if track_crop is not None:
        once_track(64)
        kept_n.clear()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            once_track(n)
        dt = (time.perf_counter() - t0) / reps
        nk = sum(kept_n) / max(1, len(kept_n))
        return {"value": nk / dt, "unit": "rays/s", "sampled_rays_per_s": n / dt, "cores": cores, "kind": "port",
                "sample": "oracle (grid_sampler_3d mode) tracking iteration (sampling, pre-filter, colour-stage render, loss, backward to the "
                          "pose), %d sampled / %.0f rendered rays, mean of %d iterations (%.0f ms each); `value` counts rendered rays like the "
                          "GPU line" % (n, nk, reps, dt * 1e3)}

    def once(stage, m):
        G = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
        P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        o, d, gd, gc = rays_o[:m], rays_d[:m], gt_depth[:m], gt_color[:m]
        with torch.no_grad():                                   # the mapper's pre-filter: rays removed from the batch (Mapper.py:471-481)
            tt = (sc["bound"].unsqueeze(0) - o.unsqueeze(-1)) / d.unsqueeze(-1)
            inside = torch.min(torch.max(tt, dim=2)[0], dim=1)[0] >= gd
        o, d, gd, gc = o[inside], d[inside], gd[inside], gc[inside]
        kept_n.append(int(inside.sum()))
        depth, _, col = orc.render_batch_ray(G, P, d, o, stage, gd, sc["bound"])
        loss = (torch.abs(gd - depth) * (gd > 0)).sum()
        if stage == "color":
            loss = loss + 0.2 * torch.abs(gc - col).sum()
        loss.backward()

    t = {}
    reps = 3
    for stage in stages:
        once(stage, 64)                                        # warm-up
        kept_n.clear()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            once(stage, n)
            ts.append(time.perf_counter() - t0)
        t[stage] = sorted(ts)[reps // 2]
    mix = sum(weights[s] * t[s] for s in stages) / sum(weights.values())
    nk = kept_n[-1] if kept_n else n                            # the same pixels every time: the same rays survive the pre-filter
    return {"value": nk / mix, "unit": "rays/s", "sampled_rays_per_s": n / mix, "cores": cores, "kind": "port",
            "sample": "oracle (grid_sampler_3d mode) pre-filter + fwd+bwd, %d sampled / %d rendered rays, median of %d iterations per stage (%s), weighted "
                      "like the GPU run; `value` counts rendered rays like the GPU line"
                      % (n, nk, reps, ", ".join(f"{s} {t[s]*1e3:.0f} ms" for s in stages)),
            "port_vs_reference": "calibrated in the build container, where both run (tools/calibrate_cpu_baseline.py, "
                                 "profiles/README.md, 1000 rays at Replica shapes): this port takes 0.87 / 0.72 / 0.97 x the time of "
                                 "the reference's own Renderer + NICE modules (middle / fine / colour stage), i.e. the reference "
                                 "itself is up to 1.4x slower than this baseline"}


def verify_shards(nsa, shard, renderer, grids, dec, frames, per_frame, stage, H, W, world, rank, dev):
    """One iteration with pixel draws every rank knows (generator seeded identically, rank r takes columns [r, r+1) * per_frame
    of a [K, world * per_frame] draw): the sharded result (loss, grid / decoder / pose gradients after the all-reduce) against
    a single-GPU evaluation of the UNION batch computed redundantly on every rank.  -> {"max_rel_err", "ok", ...}; the worst
    rank's value is reported (MAX all-reduce)."""
    K = len(frames)
    idx = torch.randint(H * W, (K, world * per_frame), generator=torch.Generator().manual_seed(4242))

    def run(fn, indices, n):
        for g in grids.values():
            g.grad = None
        for p in dec.parameters():
            p.grad = None
        fr = [(f[0].detach().clone().requires_grad_(True), f[1], f[2]) for f in frames]
        loss = fn(fr, indices, n)
        loss.backward()
        shard.finish_exchange()                                 # (split exchange: the collective + the scatter; else a no-op)
        torch.cuda.synchronize()
        out = {"grid/" + k: g.grad.detach().clone() for k, g in grids.items() if g.grad is not None}
        out.update({"param/" + k: p.grad.detach().clone() for k, p in dec.named_parameters() if p.grad is not None})
        out.update({f"pose/{k}": f[0].grad.detach().clone() for k, f in enumerate(fr) if f[0].grad is not None})
        return loss, out

    _, got = run(lambda fr, i, n: shard.mapping_loss(grids, dec, fr, n, stage, indices=i),
                 idx[:, rank * per_frame:(rank + 1) * per_frame].reshape(-1).to(dev), per_frame)
    got_loss = float(shard.last_total_loss.item())
    masks = dict(shard._rows)                                   # frustum-selected rows: only those are exchanged (and compared)
    l_ref, ref = run(lambda fr, i, n: nsa.mapping_loss(renderer, grids, dec, fr, n, stage, indices=i),
                     idx.reshape(-1).to(dev), world * per_frame)
    worst, worst_key = abs(got_loss - float(l_ref)) / max(abs(float(l_ref)), 1e-30), "loss"
    for k, v in ref.items():
        a, b = got[k], v
        if k.startswith("grid/") and k[5:] in masks:            # voxel rows inside the mask (channels-last: [voxel][32])
            rows = masks[k[5:]].to(dev)
            a = a.permute(0, 2, 3, 4, 1).reshape(-1, a.shape[1])[rows]
            b = b.permute(0, 2, 3, 4, 1).reshape(-1, b.shape[1])[rows]
        den = float(b.abs().max())
        e = float((a - b).abs().max()) / den if den > 0 else float(a.abs().max())
        if e > worst:
            worst, worst_key = e, k
    t = torch.tensor([worst], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {"max_rel_err": float(t.item()), "ok": bool(float(t.item()) < 1e-4), "worst_tensor_on_rank0": worst_key, "stage": stage,
            "compared": "all-reduced loss + grid (frustum rows) / decoder / pose gradients of the sharded iteration vs one GPU "
                        "rendering the union batch, tensors: %d" % (len(ref) + 1)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one rank per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment, exactly what torch.distributed.run would set), wait for them, and return the
    worst exit code.  Rank 0 inherits stdout and prints the one JSON line.  Fewer than N devices is an error unless
    NSR_SINGLE_DEVICE=1 (CI on a 1-GPU box: every rank on device 0, NSR_DIST_BACKEND=gloo)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("NSR_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {n} but only {have} device(s) visible (set NSR_SINGLE_DEVICE=1 NSR_DIST_BACKEND=gloo to share one)", file=sys.stderr)
        return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + int(os.environ.get("NSR_BENCH_DEADLINE_S", "900")) + 60
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(timeout=max(1.0, deadline - time.time()))))
        except subprocess.TimeoutExpired:
            rc = max(rc, 124)
    for p in procs:                                               # a rank that died leaves the others waiting in a collective
        if p.poll() is None:
            p.kill()
    return rc


def measure(args, cfg_id, scaling, world, rank, dev, sharded, role="headline", rays_override=None, stage_override=None):
    """Time one BASELINE configuration and return its result record on rank 0 (None elsewhere).  role "headline": the full line
    (roofline block, per-kernel events, cpu_baseline); role "strong": the short record of the strong-scaling configuration that
    rides in the same line (no CPU baseline, no per-kernel events)."""
    import nice_slam_amd as nsa
    from nice_slam_amd.parallel import ShardedMapping

    C = CONFIGS[cfg_id]
    headline = role == "headline"
    rays_cfg = rays_override or C["rays"]
    n_total = rays_cfg * world if scaling == "weak" else rays_cfg      # weak: fixed rays per GPU; strong: fixed batch
    sc = build_scene(cfg_id, dev)                              # same seed -> identical scene on every rank
    renderer, dec = sc["renderer"], sc["dec"]
    grids = {k: v.requires_grad_(True) for k, v in sc["grids"].items()}
    tracking = C["kind"] == "tracking"
    for p in dec.parameters():                                      # reference: every decoder parameter has requires_grad=True
        p.requires_grad_(not tracking)                              # (the tracker works on a detached copy, Tracker.py:138)
    if tracking:
        grids = {k: v.detach() for k, v in grids.items()}
    consumed = args.consumed_grads_only or role == "consumed"
    if args.stepped_grads_only or consumed:
        renderer.decoder_grads = ("color",)
    if args.render_masked:
        renderer.skip_masked_rays = False
    H, W, fx, fy, cx, cy = sc["intr"]
    frames = [(c.to(dev), d.to(dev), col.to(dev)) for c, d, col in sc["frames"]]
    K = len(frames)
    params = list(dec.parameters())
    consumed_frac = None
    if consumed and not tracking:
        # frustum feature selection (Mapper.py:315-333): the optimiser holds val[mask] of the CURRENT frame's mask only; the backward
        # is told so (Renderer.grad_voxel_masks -> nsr_render_args.grad_voxel_mask) and skips the scatter into every other voxel
        sel = nsa.FrustumSelector(sc["bound"], H, W, fx, fy, cx, cy)
        pose = torch.eye(4)
        pose[:3] = sc["frames"][-1][0][:3].detach().cpu().float()
        renderer.grad_voxel_masks = {k: sel.voxel_mask(pose, k, v.shape[2:], frames[-1][1]) for k, v in grids.items() if k != "grid_coarse"}
        consumed_frac = {k[5:]: round(float(m.float().mean()), 3) for k, m in renderer.grad_voxel_masks.items()}
    ev = HipEvents(4)                       # backward, kernel by kernel: start, stop, behind dX, behind dW
    ev_tot = HipEvents(2)                   # backward as a whole: start, stop only (every event record between two kernels costs
                                            # 3-5 us of its own: with four events the sum read 196 us where rocprofv3 had 179)
    ev_fwd = HipEvents(2)                   # forward: around the decoder-pass kernel
    renderer.profile_events = None
    stages_cfg = ("middle", "fine", "color") if C["stages"] == "mix" else tuple(C["stages"])
    if stage_override:
        stages_cfg = (stage_override,)
    mix = {s: _CYCLE.count(s) for s in stages_cfg} if C["stages"] == "mix" and not stage_override else {s: 1 for s in stages_cfg}

    def stage_of(it):
        if len(stages_cfg) == 1:
            return stages_cfg[0]
        return _CYCLE[it % 60]

    exchange = "single GPU"
    shard = None
    if sharded:
        shard = ShardedMapping(renderer)
        exchange = f"rays sharded x{world}: each rank samples, renders and differentiates its share; " + \
                   ("the batch-global depth cap without a collective (every rank's window kernel re-draws the other ranks' pixels)"
                    if (shard.peer_seeds() and not tracking) else "one MAX all-reduce of 1 float (batch-global depth cap)") + \
                   " + ONE packed SUM all-reduce per iteration (dense grid gradients + decoder blob)"
        if not args.dense_exchange and not tracking:
            # The mapper optimises only the voxels inside the current frame's frustum mask (Mapper.py:315-333), identical
            # on every rank: exchange those voxel rows only
            sel = nsa.FrustumSelector(sc["bound"], H, W, fx, fy, cx, cy)
            pose = torch.eye(4)
            pose[:3] = sc["frames"][-1][0][:3].detach().cpu().float()
            masks = {k: sel.voxel_mask(pose, k, v.shape[2:], frames[-1][1]) for k, v in grids.items() if k != "grid_coarse"}
            shard.set_voxel_masks(masks)
            frac = {k[5:]: round(float(m.float().mean()), 3) for k, m in masks.items()}
            exchange = exchange.replace("dense grid gradients", f"frustum-selected voxel rows {frac}")
    torch.manual_seed(1234 + rank)                                 # every rank draws its own pixels
    per_frame = max(1, (n_total // world) // K)                    # this rank's pixels per frame
    rays_rank = per_frame * K
    crop = C.get("crop", 0)
    cam = None
    if tracking:
        cam = frames[0][0][:3].clone().requires_grad_(True)       # the pose under optimisation (gradient w.r.t. the 3x4 matrix)

    def step(it, timed, finish=True):
        stage = stage_of(it)
        for g in grids.values():
            g.grad = None
        for p in params:
            p.grad = None
        renderer.profile_fwd_events = ev_fwd.pair_for if timed == "split" else None
        renderer.profile_events = None if not timed else (ev.pair_for if timed == "split" else ev_tot.pair_for)
        if tracking and not args.unfused:                         # Tracker.optimize_cam_in_batch (Tracker.py:87-125) as one autograd node
            cam.grad = None
            loss = nsa.tracking_loss(renderer, grids, dec, cam, frames[0][1], frames[0][2], rays_rank, crop, crop, w_color=0.5)
            nsa.backward(loss)                                    # = loss.backward() without autograd's ones_like fill
        elif tracking:                                            # the same through the drop-in surface, sync-free form
            cam.grad = None
            o, d, gd, gc = nsa.get_samples(crop, H - crop, crop, W - crop, rays_rank, H, W, fx, fy, cx, cy, cam, frames[0][1], frames[0][2], dev)
            keep, kmax = nsa.aabb_keep(o, d, gd, sc["bound"])
            depth, unc, color = renderer.render_batch_ray(grids, dec, d, o, dev, "color", gt_depth=gd, gt_max=kmax)
            unc = unc.detach()
            tmp = torch.abs(gd - depth) / torch.sqrt(unc + 1e-10)
            med = torch.nanmedian(torch.where(keep, tmp.detach(), torch.full_like(tmp, float("nan"))))
            mask = (tmp < 10 * med) & (gd > 0) & keep
            loss = torch.where(mask, tmp, torch.zeros_like(tmp)).sum() + 0.5 * torch.where(mask[:, None], torch.abs(gc - color), torch.zeros_like(color)).sum()
            loss.backward()
        elif shard is not None:
            loss = shard.mapping_loss(grids, dec, frames, per_frame, stage)
            nsa.backward(loss)
            if finish:                                            # (split exchange: the collective + the scatter; otherwise a no-op)
                shard.finish_exchange()
        elif args.unfused:                                        # the reference's call sequence through the drop-in surface
            ro, rd, gd, gc = [], [], [], []
            for c2w, dimg, cimg in frames:
                o, d, dep, col = nsa.get_samples(0, H, 0, W, per_frame, H, W, fx, fy, cx, cy, c2w, dimg, cimg, dev)
                ro.append(o); rd.append(d); gd.append(dep); gc.append(col)
            rays_o, rays_d, gt_depth, gt_color = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
            depth, unc, color = renderer.render_batch_ray(grids, dec, rays_d, rays_o, dev, stage, gt_depth=None if stage == "coarse" else gt_depth)
            loss = (torch.abs(gt_depth - depth) * (gt_depth > 0)).sum()
            if stage == "color":
                loss = loss + 0.2 * torch.abs(gt_color - color).sum()
            loss.backward()
        else:                                                      # Mapper.py:437-503 as one autograd node (mapping.py)
            loss = nsa.mapping_loss(renderer, grids, dec, frames, per_frame, stage, w_color=0.2, coarse_mapper=(stage == "coarse"))
            nsa.backward(loss)
        renderer.profile_events = None
        renderer.profile_fwd_events = None
        return stage

    # Warm-up: eager iterations.  First untimed ones (code load, allocator, LDS attribute), then 5 per stage with HIP
    # events around the backward kernel -- these feed `roofline` / `kernel_ms`.
    reps = tuple(_CYCLE.index(k) if len(stages_cfg) > 1 else 0 for k in stages_cfg)
    for i in range(max(args.warmup, 2 * len(reps))):
        step(reps[i % len(reps)], False)
    torch.cuda.synchronize()
    for st_i in reps:
        for j in range(8 if headline else 4):         # (the short record: start / stop events only)
            # the GPU idles for ~1 ms first, so that the host has the whole iteration enqueued before the first kernel starts:
            # the events inside nsr_render_bwd then bracket the backward's kernels running back to back (as they do in the
            # replayed graphs of the timed region), not the host's launch pace.  Four iterations with (start, stop) around the
            # whole backward, four with an event behind every kernel (+ around the forward's pass kernel)
            torch.cuda._sleep(2_000_000)
            step(st_i, "total" if j < 4 else "split")
    torch.cuda.synchronize()
    # Multi-rank launch modes (NSR_DIST_GRAPH): "1" (default) everything captured, the RCCL collective included (one graph for the K
    # timed steps); "segments": the kernels before and behind the iteration's ONE collective as two captured segments per stage with
    # the collective eager between them (ShardedMapping.split_exchange) -- what a failed capture of a collective falls back to, and the
    # only graph mode of a backend that cannot be captured (gloo); "0": eager.  Fallback chain: captured -> segments -> eager.
    dist_env = os.environ.get("NSR_DIST_GRAPH", "1")
    backend_capturable = os.environ.get("NSR_DIST_BACKEND", "nccl") == "nccl"
    dist_mode = None
    if sharded:
        dist_mode = "eager" if (args.eager or dist_env == "0") else ("segments" if (dist_env == "segments" or not backend_capturable) else "captured")
        if tracking and dist_mode == "segments":
            dist_mode = "captured"                 # replicas only: the tracking iteration has no collective to keep out of a graph
    use_graph = not args.eager and (not sharded or dist_mode == "captured")
    segments = {}
    graphs = {}
    window_graph = None
    if use_graph:
        # The iteration is a handful of launches; one hipGraph per stage (identical kernel sequence, fresh torch.randint
        # draws on every replay) removes the host from the loop.
        renderer.profile_events = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for st_i in reps:
                for _ in range(2):
                    step(st_i, False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # (capture mode: "thread_local" while a process group is alive -- its watchdog thread polls events, nice_slam_amd/graphs.py)
        cap_mode = nsa.graphs.default_capture_error_mode()
        try:
            for st_i in reps:
                gph = torch.cuda.CUDAGraph()
                if shard is not None:                            # the ranks' own pixel generator takes part in the capture
                    gph.register_generator_state(shard.generator(dev))
                with torch.cuda.graph(gph, capture_error_mode=cap_mode):
                    st_name = step(st_i, False)
                graphs[st_name] = gph
            torch.cuda.synchronize()
            for gph in graphs.values():                 # first replays pay the one-time upload of the executable graph:
                for _ in range(3):                      # part of the warm-up, not of the timed region
                    gph.replay()
            torch.cuda.synchronize()
            # The K timed steps as ONE graph (what tools/slam_synthetic.py does with the iterations of a frame): a replay per step
            # leaves ~4 us between two graphs and a window of per-step replays pays its first launch and its last wait once per
            # step sequence anyway -- measured on one box, 20 steps: 3.9995 ms as twenty replays, 3.8648 ms as one (tools/
            # _window_probe of the round).  NSR_BENCH_WINDOW_GRAPH=0: one replay per step as before.
            if os.environ.get("NSR_BENCH_WINDOW_GRAPH", "1") == "1" and args.steps <= 512:
                try:
                    wg = torch.cuda.CUDAGraph()
                    if shard is not None:
                        wg.register_generator_state(shard.generator(dev))
                    stages_in_window = []
                    with torch.cuda.graph(wg, capture_error_mode=cap_mode):
                        for i in range(args.steps):
                            stages_in_window.append(step(i, False))
                    torch.cuda.synchronize()
                    for _ in range(2):
                        wg.replay()
                    torch.cuda.synchronize()
                    window_graph = (wg, stages_in_window)
                except Exception as e:
                    if rank == 0:
                        print(f"[bench] capture of the whole window failed ({type(e).__name__}: {e}); one replay per step", file=sys.stderr)
                    window_graph = None
                    torch.cuda.synchronize()
        except Exception as e:                      # e.g. a collective that cannot be captured: kernel segments instead (multi-rank), else eager
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); falling back to {'kernel segments' if shard is not None and not tracking else 'eager'}", file=sys.stderr)
            graphs.clear()
            use_graph = False
            if shard is not None and not tracking:
                dist_mode = "segments (captured mode fell back)"
            torch.cuda.synchronize()
    if shard is not None and dist_mode is not None and dist_mode.startswith("segments"):
        # two graphs per stage around the one eager collective: A = sampling + forward + backward + pack, B = scatter + publish
        shard.split_exchange = True
        try:
            for st_i in reps:
                def first(st_i=st_i):
                    step(st_i, False, finish=False)
                    rec = shard.deferred()
                    if rec is None:
                        raise RuntimeError("the iteration did not qualify for the split exchange")
                    return rec
                seg = nsa.graphs.SegmentedStep(first, shard.reduce_deferred, shard.scatter_deferred, warmup=2, device=dev,
                                               generators=(shard.generator(dev),))
                segments[stage_of(st_i)] = seg
            torch.cuda.synchronize()
            for seg in segments.values():
                for _ in range(3):
                    seg()
            torch.cuda.synchronize()
        except Exception as e:
            if rank == 0:
                print(f"[bench] segment capture failed ({type(e).__name__}: {e}); falling back to eager", file=sys.stderr)
            segments.clear()
            shard.split_exchange = False
            dist_mode = "eager (segment capture fell back)"
            torch.cuda.synchronize()

    def timed_step(i):
        stage = stage_of(i)
        if segments:
            segments[stage]()
            return stage
        if not use_graph:
            return step(i, False)
        graphs[stage].replay()
        return stage

    windows, stages = [], []
    for w in range(max(1, args.windows)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if use_graph and window_graph is not None:
            window_graph[0].replay()
            stages = list(window_graph[1])
        else:
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
        windows.append(dt)
    dt = sorted(windows)[len(windows) // 2]
    rays_iter = rays_rank * world
    # What share of the sampled rays passes the bounding-box pre-filter, i.e. is RENDERED (the others are removed from the batch
    # like the reference's compaction does, Mapper.py:471-481): the mean over 16 untimed draws of this rank (the keep bytes of
    # the window kernel; no_grad: no decoder work).  Every FLOP count of the roofline block is priced on these rays.
    kept_frac = None
    if not args.unfused:
        acc = torch.zeros((), device=dev)
        with torch.no_grad():
            for _ in range(16):
                info = {}
                if tracking:
                    nsa.tracking_loss(renderer, grids, dec, cam, frames[0][1], frames[0][2], rays_rank, crop, crop, w_color=0.5, out=info)
                elif shard is not None:
                    shard.mapping_loss(grids, dec, frames, per_frame, stages_cfg[-1], out=info)
                else:
                    nsa.mapping_loss(renderer, grids, dec, frames, per_frame, stages_cfg[-1], w_color=0.2, coarse_mapper=(stages_cfg[-1] == "coarse"), out=info)
                acc += info["keep"].float().mean()
        kept_frac = float(acc.item()) / 16
    removed = not (args.render_masked or args.unfused)                # are the rejected rays removed from the batch (default) or rendered?
    rendered_frac = kept_frac if (removed and kept_frac is not None) else 1.0
    shard_check = None
    if args.verify_shards and shard is not None:
        shard_check = verify_shards(nsa, shard, renderer, grids, dec, frames, per_frame, stages_cfg[-1], H, W, world, rank, dev)

    if rank == 0:
        ksum = ev_tot.summary()
        events_from = "eager iterations of this process, each enqueued behind a 1 ms GPU-side wait (kernels back to back, like in the replayed graphs)"
        dom = "color" if "color" in ksum else (list(ksum)[-1] if ksum else None)
        res = {
            "metric": "rendered rays/sec (fwd+bwd) per mapping iter", "value": rays_iter * rendered_frac * args.steps / dt, "unit": "rays/s",
            "value_counts": "rays RENDERED per second: the sampled rays that pass the callers' bounding-box pre-filter (what the reference hands "
                            "to render_batch_ray, Mapper.py:471-482); `sampled_rays_per_s` counts the pixels drawn per iteration instead "
                            "(the number rounds 1-4 reported as `value`)",
            "rendered_rays_per_s": rays_iter * rendered_frac * args.steps / dt, "sampled_rays_per_s": rays_iter * args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32 (f64 sample placement / depth)",
            "data": "synthetic",
            "config": {"workload": C["name"] + ": grids " + " / ".join("x".join(str(v) for v in grids[k].shape[2:]) for k in grids) +
                                   f", 32 ch fp32, random-init decoders, {H}x{W} synthetic RGB-D, {K}x{per_frame} pixels/iter/GPU, S=32+16",
                       "rays_per_gpu": rays_rank, "rays_per_iteration": rays_iter,
                       "rays_rendered_per_iteration": round(rays_iter * rendered_frac, 1),
                       "rays_kept_by_prefilter": kept_frac,
                       "prefilter": ("rays whose depth lies outside the bound (Mapper.py:471-481 / Tracker.py:95-104) are sampled (counted in "
                                     "`sampled_rays_per_s`), then " + ("rendered and masked out of the loss (--render-masked)" if args.render_masked or args.unfused else
                                                         "removed from the batch like the reference's boolean-mask compaction does: no decoder work, no gradient")),
                       "stage_mix": {s: stages.count(s) for s in sorted(set(stages))},
                       "timed_region": (("get_samples (crop) + bounding-box mask + render_batch_ray(color) + tracking loss (torch) + backward to the pose"
                                         if args.unfused else
                                         "window kernel (pixel draw: philox inside the kernel; crop, bounding-box mask) + render forward (colour stage) + tracking loss kernel "
                                         "(median mask) + render backward + pose gradient") if tracking else
                                        ("get_samples x window + cat + render_batch_ray + torch loss + backward" if args.unfused else
                                         "window sampling kernel (pixel draw: philox inside the kernel) + render forward (with the mapping loss) + render backward") +
                                        " (all grid + all decoder grads, like the reference autograd), no optimiser"),
                       "decoder_grads": "none (tracking)" if tracking else ("colour decoder only (what Mapper's optimiser steps)" if (args.stepped_grads_only or consumed) else "all decoders (reference autograd semantics)"),
                       "grid_grads": ("only the voxels of the current frame's frustum mask (what the optimiser holds with frustum_feature_selection, "
                                      "Mapper.py:315-333,394-401), selected share per grid %s" % consumed_frac) if consumed_frac else
                                     "dense (reference autograd semantics)",
                       "launch": ("hipGraph replay (the K timed steps captured as ONE graph, like the iterations of a frame in tools/slam_synthetic.py)" if window_graph is not None
                                  else "hipGraph replay (one captured graph per stage, one replay per step)") if use_graph else
                                 ("hipGraph replay of two kernel segments per iteration, the one all-reduce eager between them" if segments else "eager"),
                       "activations": "saved by the forward (832 B per point and decoder + 640 B of dY scratch) and consumed by the split "
                                      "backward (dX + dW kernels)",
                       "timed_windows_ms": [round(w_ * 1e3, 3) for w_ in windows], "reported": "median window",
                       "parallelism": exchange},
        }
        if dom is not None:
            ms, cnt = ksum[dom]
            pts = rays_rank * rendered_frac * (32 if dom == "coarse" else 48)     # sample points of the rays that are rendered
            nec = pts * (NEC_MAC[dom] - FWD_MAC[dom]) * 2
            ach = nec / (ms * 1e-3)
            traffic, tsrc = None, None
            tpath = next((p_ for p_ in (os.path.join(ROOT, "profiles", n_) for n_ in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json")) if os.path.exists(p_)), None)
            if tpath and cfg_id == "1" and rays_rank == 1000 and dom == "color":       # from a separate rocprofv3 --pmc run (tools/pmc_bench.sh)
                tall = json.load(open(tpath))                 # one entry per kernel: the backward = the sum over its kernels
                ks = [k for k in tall if any(n_ in k for n_ in ("render_bwd_dx_kernel<3", "render_bwd_dw_kernel<3", "bwd_finalize", "comp_bwd"))]
                if ks:
                    traffic = sum(tall[k]["hbm_bytes_per_launch"] for k in ks)
                    tsrc = "profiles/" + os.path.basename(tpath) + " (" + " + ".join(k.replace("nsr::", "") for k in ks) + "): " + tall[ks[0]].get("note", "")
            res["roofline"] = {"bound": "mfma",
                               "kernel": f"render backward, stage {dom}: comp_bwd + render_bwd_dx_kernel + render_bwd_dw_kernel + "
                                         "bwd_finalize (split backward over saved activations)",
                               "achieved": ach / 1e12, "peak": FP32_PEAK / 1e12,
                               "unit": "TFLOP/s", "frac": ach / FP32_PEAK, "traffic": traffic, "traffic_source": tsrc,
                               "traffic_measured_in_this_run": False,
                               "avg_kernel_ms": ms, "launches": cnt,
                               "measured": "HIP events recorded inside nsr_render_bwd on the launch stream around its kernels "
                                           "(compositor backward, dX, dW, finalize): " + events_from,
                               "algorithmic_flop_per_launch": nec,
                               "priced_on": "the %.1f rays per launch that are rendered (%d sampled x kept share %.3f) x %d sample points" %
                                            (rays_rank * rendered_frac, rays_rank, rendered_frac, 32 if dom == "coarse" else 48),
                               "executed_frac": (pts * (EXEC_BWD_MAC[dom] - FWD_MAC[dom]) * 2
                                                 / (ms * 1e-3)) / FP32_PEAK
                               if not (args.stepped_grads_only or consumed or tracking) else None,
                               "executed_note": "MFMA work the backward actually issues (dX + dW for every decoder -- the reference "
                                                "autograd's semantics) over the same peak; `frac` counts only the necessary part",
                               "peak_note": "`peak` is the dense fp32 MFMA rate.  On MI355X the fp32 matrix pipe and the vector ALU of a SIMD "
                                            "are one datapath -- an fp32-MFMA wave and a vector-instruction wave take the SUM of the two alone "
                                            "(tools/coexec_probe.hip, profiles/r06_coexec_probe.txt) -- so a kernel with vector work cannot reach it: "
                                            "against MFMA + vector time the dX / dW kernels run at ~0.75-0.85 (DESIGN.md sections 3 and 4)"}
        if dom is not None and "roofline" in res:
            # the kernels of the dominant stage one by one (eager iterations behind a 1 ms wait, like `avg_kernel_ms`), and the whole
            # iteration: necessary FLOP of forward + backward of the timed stage mix over the wall time of the timed region
            ker = {}
            f = ev_fwd.summary().get(dom)
            if f:
                ker["forward_pass"] = {"kernel": f"render_fwd_pass_kernel<{dom}>", "ms": round(f[0], 4), "frac": pts * FWD_MAC[dom] * 2 / (f[0] * 1e-3) / FP32_PEAK}
            sp = ev.split().get(dom)
            if sp and dom == "color" and not tracking:
                dw_nec = DW_NEC_MAC[dom]
                ker["dx"] = {"kernel": "render_bwd_dx_kernel<color>", "ms": round(sp["dx"], 4),
                             "frac": pts * (NEC_MAC[dom] - FWD_MAC[dom] - dw_nec) * 2 / (sp["dx"] * 1e-3) / FP32_PEAK,
                             "executed_frac": pts * 3 * 15360 * 2 / (sp["dx"] * 1e-3) / FP32_PEAK}
                ker["dw"] = {"kernel": "render_bwd_dw_kernel<color>", "ms": round(sp["dw"], 4),
                             "frac": pts * dw_nec * 2 / (sp["dw"] * 1e-3) / FP32_PEAK,
                             "executed_frac": pts * ((14336 if (args.stepped_grads_only or consumed) else 47104)) * 2 / (sp["dw"] * 1e-3) / FP32_PEAK}
                ker["finalize"] = {"kernel": "bwd_finalize_kernel", "ms": round(sp["finalize"], 4)}
                if cfg_id == "1" and rays_rank == 1000 and not (args.stepped_grads_only or consumed):
                    # this kernel alone lands in one of two modes from process to process on the same box and binary (twelve fresh
                    # processes: 93.5-98.0 us six times, 101.2-108.2 six times; profiles/r06_dx_modes_12_processes.txt -- not a function
                    # of any virtual address: the physical placement a process gets): say which one this run drew
                    ker["dx"]["mode"] = "fast" if sp["dx"] < 0.0995 else "slow"
                    ker["dx"]["mode_note"] = "dX<colour> of this process between events: < 99.5 us = the fast mode (93.5-98.0 in the 12-process study), else the slow one (101.2-108.2); +-5 % on `value`"
            elif sp:
                ker["dx"] = {"ms": round(sp["dx"], 4)}
                ker["dw"] = {"ms": round(sp["dw"], 4)}
                ker["finalize"] = {"ms": round(sp["finalize"], 4)}
            if ker:
                furthest = min((k for k in ker if "frac" in ker[k]), key=lambda k: ker[k]["frac"], default=None)
                res["roofline"]["kernels"] = ker
                res["roofline"]["kernels_note"] = ("separate eager iterations with a HIP event behind every kernel: each bracket carries the 3-5 us "
                                                   "of its own event records (rocprofv3 of the same command reads 5-10 % less per kernel); "
                                                   "`avg_kernel_ms` / `frac` come from iterations with (start, stop) only")
                res["roofline"]["furthest_from_peak"] = furthest
            # every stage of the timed mix, not only the dominant one (12 of the 20 steps of the driver's command are middle / fine)
            fsum, per_stage = ev_fwd.summary(), {}
            for st_, (ms_b, _) in ksum.items():
                pts_s = rays_rank * rendered_frac * (32 if st_ == "coarse" else 48)
                rec_ = {"backward_ms": round(ms_b, 4), "backward_frac": pts_s * (NEC_MAC[st_] - FWD_MAC[st_]) * 2 / (ms_b * 1e-3) / FP32_PEAK,
                        "backward_executed_frac": None if (args.stepped_grads_only or consumed or tracking) else
                        pts_s * (EXEC_BWD_MAC[st_] - FWD_MAC[st_]) * 2 / (ms_b * 1e-3) / FP32_PEAK,
                        "steps_in_timed_window": stages.count(st_)}
                if st_ in fsum:
                    rec_["forward_pass_ms"] = round(fsum[st_][0], 4)
                    rec_["forward_pass_frac"] = pts_s * FWD_MAC[st_] * 2 / (fsum[st_][0] * 1e-3) / FP32_PEAK
                per_stage[st_] = rec_
            res["roofline"]["stages"] = per_stage
            nst = {st_: stages.count(st_) for st_ in set(stages)}
            flop_iter = sum(cnt * rays_rank * rendered_frac * (32 if st_ == "coarse" else 48) * NEC_MAC[st_] * 2 for st_, cnt in nst.items()) / max(1, len(stages))
            if not tracking:
                res["roofline"]["iteration"] = {"necessary_flop_per_iteration": flop_iter, "ms": dt / args.steps * 1e3,
                                                "frac": flop_iter / (dt / args.steps) / FP32_PEAK,
                                                "note": "forward + backward necessary FLOP of the timed stage mix over the timed region's wall time per iteration"}
        if os.environ.get("NSR_DEBUG_PTRS") == "1":                 # measurement: where the iteration's buffers landed (mode study of the dX kernel)
            from nice_slam_amd import mapping as _mp
            res["buffer_ptrs"] = {st_: {k_: (hex(v_) if isinstance(v_, int) and k_ not in ("Z_bytes", "acts_bytes") else
                                              ({a_: hex(b_) for a_, b_ in v_.items()} if isinstance(v_, dict) else v_))
                                         for k_, v_ in d_.items()} for st_, d_ in (_mp.DEBUG_PTRS or {}).items()}
        if shard is not None:
            res["config"]["grad_exchange_MB_last_iter"] = round(shard.last_exchange_floats * 4 / 1e6, 2)
            res["rccl_ranks"] = world
            res["graph_capture"] = "ok" if use_graph else ("segments" if segments else ("off (--eager / NSR_DIST_GRAPH=0)" if args.eager or dist_env == "0" else "fell_back"))
            res["dist_mode"] = {"mode": dist_mode if (use_graph or segments or dist_mode.startswith("eager")) else "eager (fell back)",
                                "collectives_per_iteration": {"max": 0 if (shard.peer_seeds() and not tracking) else 1, "sum": 1},
                                "modes": "captured: kernels + the RCCL collective in one graph for the K steps | segments: two kernel graphs per "
                                         "iteration, the collective eager between them | eager: every launch from the host"}
        if shard_check is not None:
            res["shard_check"] = shard_check
        res["kernel_ms"] = {f"render_bwd<{s}>": round(v[0], 4) for s, v in ksum.items()}
        if headline and not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(sc, rays_rank, stages_cfg, mix, crop if tracking else None)
        if role != "headline":                                    # the short record that rides in the headline's line
            keep_keys = ("value", "unit", "rendered_rays_per_s", "sampled_rays_per_s", "n_gpus", "steps", "ms_per_step", "scaling", "shard_check",
                         "graph_capture", "dist_mode", "rccl_ranks", "kernel_ms")
            short = {k: res[k] for k in keep_keys if k in res}
            short["config"] = {k: res["config"][k] for k in ("workload", "rays_per_gpu", "rays_per_iteration", "rays_rendered_per_iteration",
                                                              "rays_kept_by_prefilter", "stage_mix", "launch", "timed_windows_ms", "parallelism",
                                                              "decoder_grads", "grid_grads")
                               if k in res["config"]}
            if "roofline" in res:
                short["roofline"] = {k: res["roofline"][k] for k in ("frac", "avg_kernel_ms", "kernel") if k in res["roofline"]}
                if "iteration" in res["roofline"]:
                    short["roofline"]["iteration_frac"] = res["roofline"]["iteration"]["frac"]
            res = short
        return res
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="1", choices=sorted(CONFIGS), help="BASELINE.json configuration (default 1 = the headline one)")
    ap.add_argument("--windows", type=int, default=3, help="the K timed steps are run this many times; the median window is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage", default=None, help="pin every step to one stage (profiling)")
    ap.add_argument("--rays", type=int, default=None, help="rays per iteration (default: the configuration's)")
    ap.add_argument("--verify-shards", action="store_true", default=None,
                    help="multi-GPU self check after the timed region: one iteration with shared fixed pixel draws, the all-reduced "
                         "loss / grid / decoder / pose gradients against a single-GPU evaluation of the union batch on every rank "
                         "(default: on whenever more than one rank runs)")
    ap.add_argument("--no-verify-shards", dest="verify_shards", action="store_false")
    ap.add_argument("--eager", action="store_true", help="do not capture the iteration in a hipGraph")
    ap.add_argument("--unfused", action="store_true", help="the drop-in call sequence (get_samples per frame, render_batch_ray, torch loss) instead of mapping_loss")
    ap.add_argument("--stepped-grads-only", action="store_true",
                    help="parameter gradients only for the decoder the reference's optimiser steps (colour); default: all, like the reference autograd")
    ap.add_argument("--consumed-grads-only", action="store_true",
                    help="--stepped-grads-only + grid gradients only for the voxels inside the current frame's frustum mask -- what the "
                         "mapper's optimiser consumes with frustum_feature_selection (Mapper.py:315-333,394-401); default: the reference's "
                         "dense grid gradient")
    ap.add_argument("--consumed-record", dest="consumed_record", action="store_true", default=None,
                    help="after the timed configuration also time it with --consumed-grads-only and report that as `consumed_grads_only` "
                         "in the same line (default: on for the default single-GPU command)")
    ap.add_argument("--no-consumed-record", dest="consumed_record", action="store_false")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="multi-GPU: weak = the configuration's rays per GPU, strong = in total (default: strong for config 3 / 4, else weak)")
    ap.add_argument("--render-masked", action="store_true",
                    help="render the rays the bounding-box pre-filter rejects too and only mask them out of the loss (default: they are "
                         "removed from the batch like the reference's compaction does, Mapper.py:471-481)")
    ap.add_argument("--strong-record", dest="strong_record", action="store_true", default=None,
                    help="after the timed configuration also time BASELINE configs[3] (Apartment, 5000 rays per iteration split over the "
                         "GPUs: strong scaling) and report it as `strong` in the same line (default: on for the default configuration)")
    ap.add_argument("--no-strong-record", dest="strong_record", action="store_false")
    ap.add_argument("--dense-exchange", action="store_true",
                    help="multi-GPU: all-reduce the whole feature-grid gradients instead of the frustum-selected voxel rows")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                          # `python bench.py --gpus N`: launch the N ranks ourselves
    if "WORLD_SIZE" in os.environ and args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")

    # stdout carries exactly ONE line, the result JSON: anything a library prints there (RCCL's version banner, ...) is
    # sent to stderr by pointing file descriptor 1 at it; the JSON goes to the saved original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.verify_shards is None:
        args.verify_shards = world > 1
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
    sharded = world > 1 or force_dist
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("NSR_DIST_BACKEND", "nccl")       # "nccl" = RCCL; "gloo" only to exercise the multi-rank
        if backend == "nccl":                                      # code path where RCCL cannot run (ranks sharing one GPU)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scaling = args.scaling or ("strong" if args.config in ("3", "4") else "weak")
    res = measure(args, args.config, scaling, world, rank, dev, sharded, "headline", rays_override=args.rays, stage_override=args.stage)
    # north_star's second target is written on STRONG scaling of the mapping ray batch (BASELINE configs[3]: Apartment, 5000 rays
    # per iteration split over the GPUs): that measurement rides in the same line as `strong`, so that the driver's N = 1, 2, 4, 8
    # runs of the default command give the weak-scaling curve of config 1 (`value`) AND the strong-scaling curve (`strong.value`).
    want_strong = args.strong_record if args.strong_record is not None else (args.config == "1" and not args.stage and not args.rays and not args.unfused)
    if want_strong:
        st = measure(args, "3", "strong", world, rank, dev, sharded, "strong")
        if res is not None:
            res["strong"] = st
    # The same configuration with the gradients restricted to what the mapper's optimiser consumes (colour decoder, frustum-selected
    # voxels): a SECOND record beside the headline (which stays the reference's dense autograd semantics), never instead of it.
    want_consumed = args.consumed_record if args.consumed_record is not None else \
        (args.config == "1" and world == 1 and not sharded and not args.stage and not args.rays and not args.unfused
         and not args.consumed_grads_only and not args.stepped_grads_only)
    if want_consumed:
        cr = measure(args, args.config, scaling, world, rank, dev, sharded, "consumed", rays_override=args.rays, stage_override=args.stage)
        if res is not None:
            res["consumed_grads_only"] = cr
    if res is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
