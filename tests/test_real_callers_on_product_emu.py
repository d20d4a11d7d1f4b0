"""The UNMODIFIED reference callers on the product's Python surface (build container only).

``src/Mapper.py`` (``Mapper.optimize_map``, :230-540) and ``src/Tracker.py`` (``Tracker.optimize_cam_in_batch``, :71-128) are
imported from /root/reference exactly as tests/golden/make_golden_callers.py imports them (same arithmetic-neutral stubs) and
run with ``nice_slam_amd.Renderer`` / ``NICE`` / ``get_samples`` / channels-last grids in place of the reference's own -- the
drop-in claim of BASELINE.json's north_star, exercised by the real code: ``val[mask] = val_grad`` on channels-last grids
(Mapper.py:394-401,511-519), the per-stage ``optimizer.param_groups`` surgery (:412-419), ``deepcopy`` / ``state_dict`` of the
decoders, ``loss.backward(retain_graph=False)`` into ``torch.optim.Adam``.

No GPU exists here, so -- in THIS TEST ONLY -- the C ABI is served by tests/emu/libnsr_emu.so (the same kernel sources run
lane by lane on the CPU) and the product's "tensor must be on the GPU" checks are patched out; the product itself has no such
path.  The recorded pixel draws of tests/golden/caller_steps.npz are replayed through ``torch.randint``; the first iteration
(identical state) must reproduce the reference's loss and every optimiser gradient at 1e-4, later iterations -- which follow
the product's own Adam trajectory -- its losses to 1 %.  Skipped where /root/reference does not exist (the GPU box)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="needs the reference tree (build container only)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a).max())


@pytest.fixture(scope="module")
def env():
    """reference modules (with the golden script's stubs), the product on the emulator, the fixture"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    mk = importlib.import_module("make_golden_callers")
    import nice_slam_amd as nsa
    from nice_slam_amd import _capi, common, decoders, frustum, mapping, optim, renderer
    from emu_harness import emu_lib
    saved = {"get_lib": _capi.get_lib, "randint": torch.randint, "adam": torch.optim.Adam, "backward": torch.Tensor.backward}
    patched = []
    lib = emu_lib()
    _capi.get_lib = lambda: lib
    for mod in (common, decoders, frustum, mapping, optim, renderer):
        for name, fn in (("_require_cuda", lambda t, what: None), ("_stream", lambda device: None)):
            if hasattr(mod, name):
                patched.append((mod, name, getattr(mod, name)))
                setattr(mod, name, fn)
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "caller_steps.npz")))
    yield mk, nsa, gold
    _capi.get_lib = saved["get_lib"]
    torch.randint, torch.optim.Adam, torch.Tensor.backward = saved["randint"], saved["adam"], saved["backward"]
    for mod, name, fn in patched:
        setattr(mod, name, fn)


def _product_objects(mk, nsa, gold):
    cfg, bound, _, _, _ = mk.build()
    dec = nsa.NICE(dim=3, c_dim=32, coarse=True, coarse_grid_len=0.8, middle_grid_len=0.4, fine_grid_len=0.25, color_grid_len=0.25,
                   hidden_size=32, pos_embedding_method="fourier")
    dec.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in gold.items() if k.startswith("param/")})
    dec.bound = bound
    dec.middle_decoder.bound = dec.fine_decoder.bound = dec.color_decoder.bound = bound
    dec.coarse_decoder.bound = bound * cfg["model"]["coarse_bound_enlarge"]
    grids = {k[5:]: nsa.to_channels_last(torch.from_numpy(v.copy())) for k, v in gold.items() if k.startswith("grid/")}
    slam = mk.make_slam(cfg, bound, grids, dec)
    slam.renderer = nsa.Renderer(cfg, None, slam)            # same constructor as src/utils/Renderer.py:5-21
    return cfg, bound, grids, dec, slam


def _replay_draws(gold, pre):
    draws = [torch.from_numpy(gold[f"{pre}draw/{i}"]) for i in range(int(gold[pre + "n_draws"]))]
    it = iter(draws)

    def randint(*a, **k):
        return next(it).clone()
    return randint


@pytest.mark.parametrize("pre,ba,coarse,n_kf,iters", [("map/", False, False, 2, 5), ("ba/", True, False, 3, 5), ("coarse/", False, True, 2, 3)])
def test_unmodified_optimize_map_runs_on_the_product(env, pre, ba, coarse, n_kf, iters):
    mk, nsa, gold = env
    import src.Mapper as ref_mapper_mod
    cfg, bound, grids, dec, slam = _product_objects(mk, nsa, gold)
    poses = [torch.from_numpy(gold[f"frame/{i}/c2w"]) for i in range(4)]
    frames = [(torch.from_numpy(gold[f"frame/{i}/depth"]), torch.from_numpy(gold[f"frame/{i}/color"])) for i in range(4)]
    m = ref_mapper_mod.Mapper(cfg, None, slam, coarse_mapper=coarse)
    m.BA = ba
    kfd = []
    for i in range(n_kf):
        est = poses[i + 1].clone()
        est[:3, 3] += 0.01 * (i + 1)
        kfd.append({"gt_c2w": poses[i + 1].clone(), "idx": 10 * (i + 1), "color": frames[i + 1][1].clone(), "depth": frames[i + 1][0].clone(), "est_c2w": est})
    kfl = [d["idx"] for d in kfd]
    m.keyframe_dict, m.keyframe_list = kfd, kfl
    cur = torch.from_numpy(gold[pre + "cur_c2w"])
    torch.manual_seed(11); np.random.seed(11)                  # (keyframe_selection_overlap permutes with numpy's generator, Mapper.py:222)
    mk.reset_rec()
    torch.randint = _replay_draws(gold, pre)
    torch.optim.Adam = mk.RecAdam
    torch.Tensor.backward = mk._rec_backward
    _gs = ref_mapper_mod.get_samples
    ref_mapper_mod.get_samples = nsa.get_samples             # `from src.common import get_samples` -> the product's
    try:
        m.optimize_map(iters, 1.0, 40, frames[0][1], frames[0][0], poses[0], kfd, kfl, cur_c2w=cur)
    finally:
        ref_mapper_mod.get_samples = _gs
        torch.randint, torch.optim.Adam, torch.Tensor.backward = env_saved(env)
    losses = np.array(mk.REC["losses"])
    ref_losses = gold[pre + "losses"]
    assert len(losses) == iters == len(ref_losses)
    assert abs(losses[0] - ref_losses[0]) < 1e-5 * abs(ref_losses[0]), (losses, ref_losses)
    assert np.all(np.abs(losses - ref_losses) < 1e-2 * np.abs(ref_losses)), (losses, ref_losses)
    # first iteration: every gradient the reference's Adam saw (group order of Mapper.py:368-379)
    dec_names = [f"color_decoder.{n}" for n, _ in dec.color_decoder.named_parameters()]
    names = [dec_names, ["grid_coarse"], ["grid_middle"], ["grid_fine"], ["grid_color"]]
    st = mk.REC["steps"][0]
    if ba:
        names.append([f"cam{i}" for i in range(len(st["grads"][5]))])
    n_checked = 0
    for gi, group in enumerate(names):
        for pi, nm in enumerate(group):
            key = f"{pre}it0/grad/{nm}"
            if key not in gold:
                continue
            g = st["grads"][gi][pi]
            assert g is not None, nm
            assert rel_err(g.numpy(), gold[key]) < 1e-4, (nm, rel_err(g.numpy(), gold[key]))
            n_checked += 1
    assert n_checked >= 1
    # the write-back of Mapper.py:511-519 landed in the shared grids, which are still channels-last
    for k, v in m.c.items():
        assert v.is_contiguous(memory_format=torch.channels_last_3d), k
        assert rel_err(v.detach().numpy(), gold[f"{pre}final/{k}"]) < 2e-2, k


def env_saved(env):
    import torch as _t
    mk = env[0]
    return mk._randint, _ADAM, _BACKWARD


_ADAM, _BACKWARD = torch.optim.Adam, torch.Tensor.backward


def test_unmodified_optimize_cam_in_batch_runs_on_the_product(env):
    mk, nsa, gold = env
    import src.Tracker as ref_tracker_mod
    cfg, bound, grids, dec, slam = _product_objects(mk, nsa, gold)
    for p in dec.parameters():                                # the tracker works on a detached copy (src/Tracker.py:138)
        p.requires_grad_(False)
    t = ref_tracker_mod.Tracker(cfg, None, slam)
    t.c, t.decoders = grids, dec
    cam = torch.autograd.Variable(torch.from_numpy(gold["track/init/cam"]).clone(), requires_grad=True)
    mk.reset_rec()
    torch.randint = _replay_draws(gold, "track/")
    torch.optim.Adam = mk.RecAdam
    torch.Tensor.backward = mk._rec_backward
    _gs = ref_tracker_mod.get_samples
    ref_tracker_mod.get_samples = nsa.get_samples
    try:
        opt = torch.optim.Adam([cam], lr=cfg["tracking"]["lr"])
        losses = [t.optimize_cam_in_batch(cam, torch.from_numpy(gold["frame/0/color"]), torch.from_numpy(gold["frame/0/depth"]),
                                          t.tracking_pixels, opt) for _ in range(2)]
    finally:
        ref_tracker_mod.get_samples = _gs
        torch.randint, torch.optim.Adam, torch.Tensor.backward = env_saved(env)
    ref = gold["track/ret_losses"]
    assert abs(losses[0] - ref[0]) < 1e-5 * abs(ref[0]), (losses, ref)
    assert abs(losses[1] - ref[1]) < 1e-2 * abs(ref[1]), (losses, ref)
    g0 = mk.REC["steps"][0]["grads"][0][0]
    assert rel_err(g0.numpy(), gold["track/it0/grad/cam"]) < 1e-4
