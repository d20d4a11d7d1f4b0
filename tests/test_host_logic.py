"""CPU: host-side logic of the drop-in modules (no kernel is launched)."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

import nice_slam_amd as nsa
from nice_slam_amd.layout import param_count, param_spec


def test_state_dict_surface_matches_reference(golden):
    dec = nsa.NICE(coarse=True)
    ref_keys = [k[len("param/"):] for k in golden if k.startswith("param/")]
    assert list(dec.state_dict().keys()) == ref_keys                      # same names, same order as decoder.py
    dec.load_state_dict({k: torch.from_numpy(golden["param/" + k]) for k in ref_keys})
    for slot in ("coarse", "middle", "fine", "color"):
        sub = dec.sub(slot)
        flat = sub.flat_params()
        assert flat.numel() == param_count(slot)
        off = 0
        for name, shape in param_spec(slot):
            n = int(np.prod(shape))
            assert np.array_equal(flat[off:off + n].view(shape).numpy(), golden[f"param/{slot}_decoder.{name}"])
            off += n
    assert [p.numel() for p in dec.middle_decoder.parameters()] == [int(np.prod(s)) for _, s in param_spec("middle")]


def test_parameters_are_views_and_survive_module_ops():
    dec = nsa.NICE(coarse=True)
    sub = dec.color_decoder
    f = sub.flat_params()
    w = sub.pts_linears[3].weight
    with torch.no_grad():
        w.add_(1.0)
    off = dict(zip([n for n, _ in param_spec("color")], sub._offsets))["pts_linears.3.weight"]
    assert torch.equal(f[off:off + w.numel()].view_as(w), w)             # in-place updates land in the blob
    d2 = copy.deepcopy(dec)                                               # src/Tracker.py:138
    assert d2.color_decoder.flat_params().data_ptr() != f.data_ptr()
    assert torch.equal(d2.color_decoder.flat_params(), f)
    d3 = dec.to(torch.float32)                                            # Module._apply path
    assert torch.equal(d3.color_decoder.flat_params(), f)
    dec.share_memory()
    assert torch.equal(dec.color_decoder.flat_params(), f)


def test_pack_cache_key_sees_optimizer_steps():
    """regression: ``p.data = view`` gives the Parameter its own version counter, so the key must look at every
    Parameter, not only at the flat buffer (stale packed weights after Adam otherwise)"""
    dec = nsa.NICE(coarse=False)
    sub = dec.color_decoder
    for p in sub._views:                      # emulate the re-flatten path (.to(device))
        p.data = p.data.clone()
    k0 = sub._pack_key()
    assert sub._pack_key() == k0
    opt = torch.optim.Adam(sub.parameters(), lr=1e-2)
    for p in sub.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert sub._pack_key() != k0
    k1 = sub._pack_key()
    sub.load_state_dict(sub.state_dict())
    assert sub._pack_key() != k1


def test_publish_grads_are_views_of_one_blob():
    dec = nsa.NICE(coarse=False)
    sub = dec.middle_decoder
    g = torch.arange(param_count("middle"), dtype=torch.float32)
    sub.publish_grads(g)
    off = 0
    for p in sub.parameters():
        assert torch.equal(p.grad.reshape(-1), g[off:off + p.numel()])
        assert p.grad.data_ptr() == g.data_ptr() + 4 * off
        off += p.numel()
    sub.publish_grads(g)                      # accumulation when .grad already exists
    assert torch.equal(next(sub.parameters()).grad.reshape(-1), 2 * g[:next(sub.parameters()).numel()])


def test_grid_init_matches_reference_shapes():
    import json, os
    from conftest import GOLDEN
    shapes = json.load(open(os.path.join(GOLDEN, "scene_shapes.json")))
    for path in ("configs/Replica/room0.yaml", "configs/ScanNet/scene0000.yaml", "configs/Apartment/apartment.yaml"):
        rec = shapes[path]
        cfg = {"scale": rec["scale"], "mapping": {"bound": rec["bound_cfg"]}, "coarse": True,
               "grid_len": dict(rec["grid_len"], bound_divisible=rec["bound_divisible"]),
               "model": {"c_dim": 32, "coarse_bound_enlarge": rec["coarse_bound_enlarge"]}}
        bound = nsa.load_bound(cfg)
        assert np.array_equal(bound.numpy(), np.array(rec["bound"]))
        c = nsa.grid_init(cfg, bound)
        assert {k: list(v.shape[2:]) for k, v in c.items()} == rec["shapes"]
        for v in c.values():
            assert v.is_contiguous(memory_format=torch.channels_last_3d) and v.shape[1] == 32


def test_renderer_rejects_imap_configuration():
    import types, pytest
    cfg = {"rendering": {"lindisp": False, "perturb": 0.0, "N_samples": 32, "N_surface": 16, "N_importance": 12},
           "scale": 1, "occupancy": True}
    slam = types.SimpleNamespace(nice=True, bound=torch.zeros(3, 2), H=4, W=4, fx=1., fy=1., cx=1., cy=1.)
    with pytest.raises(NotImplementedError):
        nsa.Renderer(cfg, None, slam)
    cfg["rendering"]["N_importance"] = 0
    r = nsa.Renderer(cfg, None, slam)
    import pickle
    r2 = pickle.loads(pickle.dumps(r))        # constructed before the processes are spawned (NICE_SLAM.py:91,296-301)
    assert r2.N_samples == 32 and r2._ws == {}
    with pytest.raises(NotImplementedError):
        r.regulation(None, None, None, None, None, "cpu")


def test_grad_target_state_machine():
    """persistent gradient blob: overwrite when every .grad is None, accumulate when every .grad is the cached view,
    fallback (None) for foreign .grad tensors; the views survive across calls, are rebuilt after .to()/deepcopy."""
    import copy
    dec = nsa.NICE(coarse=False)
    sub = dec.color_decoder
    flat, mode = sub.grad_target()
    assert mode == "overwrite" and flat.numel() == param_count("color")
    flat.copy_(torch.arange(flat.numel(), dtype=torch.float32))
    sub.grad_done(mode)
    off = 0
    for p in sub.parameters():
        assert p.grad.data_ptr() == flat.data_ptr() + 4 * off and torch.equal(p.grad.reshape(-1), flat[off:off + p.numel()])
        off += p.numel()
    views = [p.grad for p in sub.parameters()]
    flat2, mode2 = sub.grad_target()
    assert mode2 == "accumulate" and flat2 is flat
    sub.grad_done(mode2)
    assert all(a is b for a, b in zip(views, (p.grad for p in sub.parameters())))
    for p in sub.parameters():                        # optimizer.zero_grad(set_to_none=True)
        p.grad = None
    assert sub.grad_target()[1] == "overwrite"
    next(sub.parameters()).grad = torch.zeros_like(next(sub.parameters()))      # a foreign gradient tensor
    assert sub.grad_target() == (None, None)
    for p in sub.parameters():
        p.grad = None
    sub.output_linear.weight.requires_grad_(False)    # frozen parameters are ignored by the state test and get no .grad
    flat3, mode3 = sub.grad_target()
    sub.grad_done(mode3)
    assert sub.output_linear.weight.grad is None and sub.output_linear.bias.grad is not None
    cp = copy.deepcopy(sub)
    f4, m4 = cp.grad_target()
    assert m4 == "overwrite" and f4.data_ptr() != flat.data_ptr()


def test_bench_gpus_n_fails_loudly_without_n_devices():
    """`python bench.py --gpus 2` starts its ranks itself (bench.spawn_ranks); with fewer than two devices and no
    NSR_SINGLE_DEVICE it must say so and exit non-zero instead of silently measuring one rank."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NSR_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "--gpus 2" in r.stderr and "device" in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])


def test_tile_liveness_reciprocal_is_an_exact_division():
    """nsr_kernels.h::tile_live divides a sample-point index by the samples per ray S with one multiplication by ceil(2^32 / S)
    (RenderParams.s_magic, nsr_api.cpp::build_params): exact for every point index a call can hold (< 2^25 points, S <= 64)."""
    import numpy as np
    rng = np.random.default_rng(0)
    for S in range(1, 65):
        magic = ((1 << 32) + S - 1) // S
        assert magic < (1 << 32) or S == 1          # (S = 1: 2^32 does not fit the 32-bit field; build_params passes 0 and ray_of_point returns the index)
        x = np.concatenate([np.arange(0, 4096), (1 << 25) - 1 - np.arange(0, 4096), rng.integers(0, 1 << 25, 200_000),
                            np.arange(1, 1 << 25, 48 * 1021)[:50_000], (np.arange(1, 20_000) * S) - 1, np.arange(0, 20_000) * S]).astype(np.uint64)
        x = x[x < (1 << 25)]
        q = (x * np.uint64(magic & 0xFFFFFFFF if S > 1 else 0)) >> np.uint64(32)
        if S > 1:
            assert np.array_equal(q, x // np.uint64(S)), S
