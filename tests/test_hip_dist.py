"""GPU: the N > 1 product path end to end -- two ranks (two processes sharing the one GPU of the test box, gloo transport
because RCCL refuses two ranks on one device) drive the HIP renderer through nice_slam_amd.parallel.ShardedRenderer; the
result must equal the single-process HIP result.  Covers what tests/test_dist_gloo.py (oracle stand-in, CPU) cannot: the
renderer's gradient hook with the temporary decoder-gradient blob, channels-last grid gradients in the packed
frustum-masked exchange, and the single output all-gather on device tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
N_RAYS = 61


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _masks(grids):
    g = torch.Generator().manual_seed(9)
    return {k: (torch.rand(tuple(v.shape[2:]), generator=g) < 0.4) for k, v in grids.items() if k != "grid_coarse"}


def _worker(rank, world, port, masked, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scene_util import make_scene, build_product, hip_render
    from nice_slam_amd.parallel import ShardedRenderer
    dev = "cuda:0"
    sc = make_scene(seed=5, n_rays=N_RAYS, small=True)
    renderer, dec, grids = build_product(sc, dev)
    sh = ShardedRenderer(renderer)
    if masked:
        sh.set_voxel_masks({k: m.to(dev) for k, m in _masks(grids).items()})
    out = {}
    for stage in ("color", "middle"):
        res = hip_render(sc, stage, dev, backward=True, product=(sh, dec, grids))
        out.update({f"{stage}/{k}": v.detach().cpu().numpy().copy() for k, v in res.items()})
    out["exchange_floats"] = np.array(sh.last_exchange_floats)
    # pre-existing .grad tensors (optimizer.zero_grad(set_to_none=False), the torch-1.10 default of the reference, or two
    # render calls before one step): the decoder gradients are ACCUMULATED, which must consume the reduced blob
    from scene_util import _loss
    c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids.items()}
    for p in dec.parameters():
        p.requires_grad_(True)
        p.grad = torch.zeros_like(p)
    o, d, gd = sc["rays_o"].to(dev), sc["rays_d"].to(dev), sc["gt_depth"].to(dev)
    depth, var, rgb = sh.render_batch_ray(c, dec, d, o, dev, "color", gt_depth=gd)
    _loss(depth, var, rgb, sc["w"]).backward()
    out.update({f"pre/dparam/{k}": p.grad.detach().cpu().numpy().copy() for k, p in dec.named_parameters()})
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("masked", [False, True])
def test_two_ranks_on_the_hip_renderer(masked):
    from scene_util import make_scene, build_product, hip_render, rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, masked, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = make_scene(seed=5, n_rays=N_RAYS, small=True)
    product = build_product(sc, "cuda:0")
    masks = _masks(product[2])
    for stage in ("color", "middle"):
        ref = hip_render(sc, stage, "cuda:0", backward=True, product=product)
        for rank in (0, 1):
            for k, v in ref.items():
                got = res[rank][f"{stage}/{k}"]
                v = v.detach().cpu().numpy()
                if masked and k.startswith("d_grid"):
                    m = masks[k[2:]].numpy()[None, None].repeat(32, 1)
                    got, v = got[m], v[m]                          # outside the mask: rank-local partial sums, by design
                # outputs, ray and grid gradients: same arithmetic per ray, sums of a few atomics -> 2e-6; decoder-parameter
                # gradients are sums over ALL samples whose order changes with the partition (the cancelling bias sums
                # carry ~1e-5 of their maximum in fp32 rounding, see scene_util.parity_failures)
                assert rel_err(got, v) < (2e-5 if k.startswith("dparam/") else 2e-6), (masked, stage, rank, k)
    ref = hip_render(sc, "color", "cuda:0", backward=True, product=product)
    for rank in (0, 1):                                           # accumulation into pre-existing .grad tensors sees reduced values
        for k, v in ref.items():
            if k.startswith("dparam/"):
                assert rel_err(res[rank]["pre/" + k], v.detach().cpu().numpy()) < 2e-5, (masked, rank, k)
    if masked:                                                    # last stage rendered: middle -> its masked rows + the decoder blob
        from nice_slam_amd.layout import param_count
        assert int(res[0]["exchange_floats"]) == int(masks["grid_middle"].sum()) * 32 + param_count("middle")


# ----------------------------------------------------------------------------------------------------------------------
# the fused mapping iteration sharded over two ranks (nice_slam_amd.parallel.ShardedMapping): each rank samples its own
# pixels; one MAX all-reduce (depth cap) + one packed SUM all-reduce (voxel rows, decoder blobs, pose gradients, loss)
# ----------------------------------------------------------------------------------------------------------------------
K_FR, M_PIX = 3, 40


def _map_frames(sc, dev, grad):
    H, W = sc["intr"][:2]
    g = torch.Generator().manual_seed(33)
    out = []
    for k in range(K_FR):
        c2w = sc["c2w"].clone()
        c2w[:3, 3] += 0.02 * k
        out.append((c2w.to(dev).requires_grad_(grad), (sc["depth_img"] * (1.0 + 0.05 * k)).to(dev), torch.rand((H, W, 3), generator=g).to(dev)))
    return out


def _map_indices(sc):
    H, W = sc["intr"][:2]
    return torch.randint(H * W, (K_FR, 2 * M_PIX), generator=torch.Generator().manual_seed(34))


def _map_worker(rank, world, port, masked, q, own_draws=False, split=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scene_util import make_scene, build_product
    from nice_slam_amd.parallel import ShardedMapping
    dev = "cuda:0"
    sc = make_scene(seed=5, n_rays=8, small=True)
    renderer, dec, grids = build_product(sc, dev)
    sh = ShardedMapping(renderer, split_exchange=split)
    if masked:
        sh.set_voxel_masks({k: m.to(dev) for k, m in _masks(grids).items()})
    idx = None if own_draws else _map_indices(sc)[:, rank * M_PIX:(rank + 1) * M_PIX].reshape(-1)
    split_used = 0
    torch.manual_seed(1234)                      # every process seeds torch identically, like the reference's setup_seed
    out = {}
    for stage in ("color", "fine"):
        frames = _map_frames(sc, dev, grad=not split)       # (local BA keeps the blocking exchange: split iterations optimise no pose)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids.items()}
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        info = {}
        loss = sh.mapping_loss(c, dec, frames, M_PIX, stage, indices=idx, out=info)
        loss.backward()
        if split:                                            # backward() only packed: nothing is summed yet
            split_used += sh.deferred() is not None and sh.sum_collectives == split_used
            sh.finish_exchange()
        out[f"{stage}/indices"] = info["indices"].cpu().numpy().copy()
        out[f"{stage}/loss_total"] = sh.last_total_loss.cpu().numpy().copy()
        out.update({f"{stage}/d_{k}": v.grad.cpu().numpy().copy() for k, v in c.items() if v.grad is not None})
        out.update({f"{stage}/dparam/{k}": p.grad.cpu().numpy().copy() for k, p in dec.named_parameters() if p.grad is not None})
        out.update({f"{stage}/dpose/{i}": f[0].grad.cpu().numpy().copy() for i, f in enumerate(frames) if f[0].grad is not None})
    out["exchange_floats"] = np.array(sh.last_exchange_floats)
    out["collectives"] = np.array([sh.max_collectives, sh.sum_collectives, split_used])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("masked,own_draws,split", [(False, False, False), (True, False, False), (True, True, False), (True, True, True), (False, False, True)])
def test_two_ranks_sharded_fused_mapping(masked, own_draws, split):
    """own_draws: every rank draws its pixels itself (ShardedMapping's per-rank state of the in-kernel draw; all processes seed
    torch identically) -- the draws must differ between the ranks, and the all-reduced loss / gradients must be those of ONE GPU
    rendering the union of the two draws.  With own draws the batch-global depth cap needs NO collective (round 6: every rank's
    window kernel re-draws the other rank's pixels, nsr_get_samples_window_sharded).  split: `split_exchange` -- backward() only
    packs, the one SUM all-reduce and the scatter into the `.grad` tensors run behind it (finish_exchange)."""
    import nice_slam_amd as nsa
    from scene_util import make_scene, build_product, rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_map_worker, args=(r, 2, port, masked, q, own_draws, split)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in (0, 1):
        n_max, n_sum, n_split = (int(v) for v in res[r]["collectives"])
        assert n_sum == 2 and n_max == (0 if own_draws else 2), (r, n_max, n_sum)    # two iterations: one SUM each; MAX only without the peer re-draw
        assert n_split == (2 if split else 0), (r, n_split)
    sc = make_scene(seed=5, n_rays=8, small=True)
    renderer, dec, grids = build_product(sc, "cuda:0")
    masks = _masks(grids)
    for stage in ("color", "fine"):
        # both ranks' pixels, frame-major: [frame][rank 0's draw | rank 1's draw]
        i0, i1 = (torch.from_numpy(res[r][f"{stage}/indices"]).reshape(K_FR, M_PIX) for r in (0, 1))
        assert not torch.equal(i0, i1), "the ranks drew the same pixels"
        idx = torch.cat([i0, i1], 1).reshape(-1)
        if not own_draws:
            assert torch.equal(idx, _map_indices(sc).reshape(-1))
        frames = _map_frames(sc, "cuda:0", grad=not split)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids.items()}
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        loss = nsa.mapping_loss(renderer, c, dec, frames, 2 * M_PIX, stage, indices=idx)
        loss.backward()
        for rank in (0, 1):
            r = res[rank]
            assert abs(float(r[f"{stage}/loss_total"][0]) - float(loss)) < 1e-5 * abs(float(loss)), (stage, rank)
            for k, v in c.items():
                if v.grad is None:
                    continue
                got, ref = r[f"{stage}/d_{k}"], v.grad.cpu().numpy()
                if masked:
                    m = masks[k].numpy()[None, None].repeat(32, 1)
                    got, ref = got[m], ref[m]
                assert rel_err(got, ref) < 2e-6, (masked, stage, rank, k)
            for k, p in dec.named_parameters():
                if p.grad is not None:
                    assert rel_err(r[f"{stage}/dparam/{k}"], p.grad.cpu().numpy()) < 2e-5, (masked, stage, rank, k)
            for i, f in enumerate(frames):
                if not split:
                    assert rel_err(r[f"{stage}/dpose/{i}"], f[0].grad.cpu().numpy()) < 1e-4, (masked, stage, rank, i)
    if masked:
        from nice_slam_amd.layout import param_count
        want = sum(int(masks[k].sum()) for k in ("grid_middle", "grid_fine")) * 32 + param_count("middle") + param_count("fine") + (0 if split else K_FR * 16) + 1
        assert int(res[0]["exchange_floats"]) == want


def test_bench_gpus_2_spawns_two_ranks_and_checks_its_shards():
    """The driver's command line, `python bench.py --gpus 2`, on a 1-GPU box: both ranks on device 0 over gloo (RCCL refuses two
    ranks on one device).  ONE JSON line, n_gpus / rccl_ranks = 2, and the run's own shard check (the all-reduced loss and
    gradients against one GPU on the union batch) green."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NSR_SINGLE_DEVICE="1", NSR_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--windows", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak", d
    assert d["shard_check"]["ok"], d["shard_check"]
    # gloo cannot be captured: the kernels run as two graph segments per iteration around the eager all-reduce; with two ranks the
    # batch-global depth cap comes from the peer re-draw inside the window kernel, i.e. ONE collective per iteration
    assert d["graph_capture"] == "segments" and d["dist_mode"]["mode"] == "segments", (d["graph_capture"], d["dist_mode"])
    assert d["dist_mode"]["collectives_per_iteration"] == {"max": 0, "sum": 1}
    assert d["config"]["rays_per_iteration"] == 2 * d["config"]["rays_per_gpu"]
    # `value` counts the rays that are rendered (the pre-filter's kept share of the sampled ones); the sampled rate rides beside it
    assert 0.5 < d["config"]["rays_kept_by_prefilter"] <= 1.0
    assert abs(d["value"] - d["sampled_rays_per_s"] * d["config"]["rays_kept_by_prefilter"]) <= 1e-6 * d["value"]
    # the strong-scaling record of the same line (north_star: BASELINE configs[3], the Apartment batch of 5000 rays SPLIT over the GPUs)
    st = d["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["config"]["rays_per_iteration"] == 5000 and st["config"]["rays_per_gpu"] == 2500, st
    assert st["shard_check"]["ok"], st["shard_check"]
    assert st["value"] > 0 and st["ms_per_step"] > 0


@pytest.mark.parametrize("mode", ["1", "segments", "0"])
def test_bench_launch_modes_of_a_sharded_run_over_rccl(mode):
    """bench.py's three launch modes of a multi-rank run (NSR_DIST_GRAPH): everything captured, the RCCL collective included /
    kernel segments captured with the one SUM all-reduce eager between them (ShardedMapping.split_exchange + graphs.SegmentedStep)
    / eager -- each over RCCL with the one rank a 1-GPU box allows (NSR_FORCE_SHARDED=1: process group, sharder, pack / all-reduce /
    scatter all run), each with the run's own shard check green, and the line says which mode ran.  (No N > 1 RCCL execution
    exists on this pool: DESIGN §5.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NSR_FORCE_SHARDED="1", NSR_DIST_GRAPH=mode, MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--windows", "1", "--no-cpu-baseline",
                        "--no-strong-record", "--no-consumed-record", "--verify-shards"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["rccl_ranks"] == 1 and d["shard_check"]["ok"], d.get("shard_check")
    want = {"1": ("ok", "captured"), "segments": ("segments", "segments"), "0": ("off (--eager / NSR_DIST_GRAPH=0)", "eager")}[mode]
    assert d["graph_capture"] == want[0] and d["dist_mode"]["mode"] == want[1], (d["graph_capture"], d["dist_mode"])
    assert d["dist_mode"]["collectives_per_iteration"] == {"max": 1, "sum": 1}      # one rank: no peer to re-draw, the MAX stays
