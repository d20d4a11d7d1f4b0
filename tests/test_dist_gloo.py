"""CPU, world_size 2, gloo: the ray-sharding logic of nice_slam_amd.parallel.

The inner renderer is an oracle-backed stand-in with the same ``render_batch_ray`` / ``_gt_max`` protocol
(the HIP renderer needs a GPU); what is under test is the distributed plumbing: contiguous sharding, the
batch-global max(gt_depth) taken before slicing, all-gather of outputs, all-gather of ray gradients and the
SUM all-reduce of the replicated feature-grid gradients.  Result must equal the single-process result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleRenderer:
    """Renderer-protocol stand-in built on the CPU oracle."""

    def __init__(self, sc):
        self.sc, self._gt_max, self._reduce_hook = sc, None, None

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None):
        from oracle import nice_oracle as orc
        if gt_depth is not None and self._gt_max is not None and gt_depth.numel() > 0:
            # emulate the kernel's `gt_max` argument: plant the batch-global maximum as an extra ray, drop it after
            extra_o, extra_d = rays_o[:1].detach(), rays_d[:1].detach()
            ro, rd = torch.cat([rays_o, extra_o]), torch.cat([rays_d, extra_d])
            gd = torch.cat([gt_depth, self._gt_max.reshape(1)])
            d, v, col = orc.render_batch_ray(c, decoders, rd, ro, stage, gd, self.sc["bound"])
            return d[:-1], v[:-1], col[:-1]
        return orc.render_batch_ray(c, decoders, rays_d, rays_o, stage, gt_depth, self.sc["bound"])


def _voxel_masks(sc, frac=0.4):
    g = torch.Generator().manual_seed(77)
    return {k: torch.rand(tuple(v.shape[2:]), generator=g) < frac for k, v in sc["grids"].items()}


def _worker(rank, world, port, stage, q, masked=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from scene_util import make_scene
    from nice_slam_amd.parallel import ShardedRenderer, shard_range
    sc = make_scene(seed=5, n_rays=23, small=True)           # 23 rays: uneven shards (12 + 11)
    grids = {k: v.clone().requires_grad_(True) for k, v in sc["grids"].items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sc["params"].items()}
    o = sc["rays_o"].clone().requires_grad_(True)
    d = sc["rays_d"].clone().requires_grad_(True)
    rend = ShardedRenderer(OracleRenderer(sc))
    if masked:
        rend.set_voxel_masks(_voxel_masks(sc))
    depth, var, rgb = rend.render_batch_ray(grids, params, d, o, "cpu", stage, gt_depth=sc["gt_depth"])
    w = sc["w"]
    ((depth * w["depth"]).sum() + (var * w["var"]).sum() + (rgb * w["rgb"]).sum()).backward()
    # decoder parameters are plain autograd leaves here (the HIP path reduces its flat blob in _reduce_flat)
    for p in params.values():
        if p.grad is not None:
            dist.all_reduce(p.grad)
    out = {"depth": depth.detach(), "var": var.detach(), "rgb": rgb.detach(), "d_rays_o": o.grad, "d_rays_d": d.grad}
    out.update({"d_" + k: v.grad for k, v in grids.items() if v.grad is not None})
    out.update({"dparam/" + k: v.grad for k, v in params.items() if v.grad is not None})
    out["shard"] = torch.tensor(shard_range(23, world, rank))
    out["exchange_floats"] = torch.tensor(rend.last_exchange_floats)
    q.put((rank, {k: v.detach().numpy().copy() for k, v in out.items()}))   # by value: the worker may exit first
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stage", ["color", "coarse"])
def test_two_rank_sharding_matches_single_process(stage):
    from scene_util import make_scene, oracle_render, rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, stage, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = oracle_render(make_scene(seed=5, n_rays=23, small=True), stage, backward=True)
    assert res[0]["shard"].tolist() == [0, 12] and res[1]["shard"].tolist() == [12, 23]
    for rank in (0, 1):
        for k, v in ref.items():
            assert rel_err(res[rank][k], v) < 2e-5, (rank, k)


def test_two_rank_masked_gradient_exchange():
    """With the frame's frustum masks set, only the selected voxel rows travel (one packed all-reduce); inside the masks
    the gradients equal the single-process ones, outside they are left rank-local (nothing reads them)."""
    from scene_util import make_scene, oracle_render, rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "color", q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = make_scene(seed=5, n_rays=23, small=True)
    ref = oracle_render(sc, "color", backward=True)
    masks = _voxel_masks(sc)
    used = [k for k in masks if "d_" + k in ref]
    assert sorted(used) == ["grid_color", "grid_fine", "grid_middle"]
    expect = sum(int(masks[k].sum()) * 32 for k in used)
    dense = sum(sc["grids"][k].numel() for k in used)
    for rank in (0, 1):
        assert int(res[rank]["exchange_floats"]) == expect < 0.5 * dense
        for k, v in ref.items():
            if k.startswith("d_grid"):
                m = masks[k[2:]].numpy()[None, None].repeat(32, 1)
                assert rel_err(res[rank][k][m], v.numpy()[m] if hasattr(v, "numpy") else v[m]) < 2e-5, (rank, k)
            else:
                assert rel_err(res[rank][k], v) < 2e-5, (rank, k)
    k = "d_grid_fine"                     # outside the mask: rank-local partial sums, which add up to the full gradient
    m = ~masks["grid_fine"].numpy()[None, None].repeat(32, 1)
    full = ref[k].numpy() if hasattr(ref[k], "numpy") else ref[k]
    assert rel_err((res[0][k] + res[1][k])[m], full[m]) < 2e-5
    assert rel_err(res[0][k][m], full[m]) > 1e-2


def _tiny_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from scene_util import make_scene
    from nice_slam_amd.parallel import ShardedRenderer
    sc = make_scene(seed=5, n_rays=1, small=True)
    grids = {k: v.clone().requires_grad_(True) for k, v in sc["grids"].items()}
    rend = ShardedRenderer(OracleRenderer(sc))
    depth, var, rgb = rend.render_batch_ray(grids, sc["params"], sc["rays_d"], sc["rays_o"], "cpu", "middle", gt_depth=sc["gt_depth"])
    depth.sum().backward()
    q.put((rank, float(depth[0]), float(grids["grid_middle"].grad.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_fewer_rays_than_ranks_does_not_shard():
    """1 ray on 2 ranks: no empty shard (whose missing backward would leave the other rank alone in a collective)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tiny_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:] == res[1][1:] and res[0][2] > 0


def test_shard_range_partition():
    from nice_slam_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1000, 100003):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, w, r) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


# ----------------------------------------------------------------------------------------------------------------------
# the renderer <-> ShardedRenderer hook protocol (nice_slam_amd/renderer.py:_RenderFn.backward): decoder gradients come
# as ONE flat blob handed to `_reduce_hook(d_grids, gflat, publish)`; with voxel masks set the blob is parked and reduced
# later together with the grid rows, and only THEN published as Parameter.grad
# ----------------------------------------------------------------------------------------------------------------------
class _ToyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, rays_o, grid):
        ctx.owner, ctx.hook = owner, owner._reduce_hook          # the hook is captured at forward time, like _RenderFn's meta
        ctx.save_for_backward(rays_o, grid)
        s = rays_o.sum(1).double()
        return s * float(grid.sum()) + float(owner.theta.sum()), s * 0.0, rays_o * 0.0

    @staticmethod
    def backward(ctx, g_depth, g_var, g_rgb):
        owner = ctx.owner
        rays_o, grid = ctx.saved_tensors
        w = (g_depth * rays_o.sum(1).double()).sum().float()
        d_grid = torch.ones_like(grid) * w
        gflat = torch.ones_like(owner.theta) * g_depth.sum().float()

        def publish():
            owner.theta.grad = gflat.clone() if owner.theta.grad is None else owner.theta.grad + gflat
            owner.published_after_reduce = owner.reduced

        deferred = ctx.hook is not None and bool(ctx.hook([d_grid], gflat, publish))
        if not deferred:
            publish()
        return None, None, d_grid


class _ToyRenderer:
    def __init__(self):
        self.theta = torch.arange(5, dtype=torch.float32)
        self._gt_max, self._reduce_hook, self.reduced, self.published_after_reduce = None, None, False, None

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None):
        return _ToyFn.apply(self, rays_o, c["grid_middle"])


def _hook_worker(rank, world, port, q, pre_existing):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nice_slam_amd.parallel import ShardedRenderer
    toy = _ToyRenderer()
    sh = ShardedRenderer(toy)
    _all_reduce = sh._all_reduce

    def traced(t):
        _all_reduce(t)
        toy.reduced = True

    sh._all_reduce = traced
    g = torch.Generator().manual_seed(3)
    grid = torch.rand((1, 32, 2, 3, 4), generator=g).requires_grad_(True)
    rays = torch.rand((9, 3), generator=g)
    w = torch.rand((9,), generator=g, dtype=torch.float64)
    sh.set_voxel_masks({"grid_middle": torch.rand((2, 3, 4), generator=g) < 0.5})
    if pre_existing:
        toy.theta.grad = torch.full_like(toy.theta, 10.0)
    depth, _, _ = sh.render_batch_ray({"grid_middle": grid}, None, rays, rays, "cpu", "middle")
    (depth * w).sum().backward()
    q.put((rank, toy.theta.grad.numpy().copy(), bool(toy.published_after_reduce), float(w.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pre_existing", [False, True])
def test_decoder_grads_are_published_after_the_masked_exchange(pre_existing):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hook_worker, args=(r, 2, port, q, pre_existing)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grad, after, wsum in res:
        assert after, "Parameter.grad was published before the gradient exchange"
        want = wsum + (10.0 if pre_existing else 0.0)          # full-batch value on BOTH ranks, accumulated once
        assert abs(grad - want).max() < 1e-5 * want, (rank, grad, want)
