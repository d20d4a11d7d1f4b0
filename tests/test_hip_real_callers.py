"""GPU: the product against the REAL callers.  The fixtures hold what the unmodified reference ``Mapper.optimize_map``
(src/Mapper.py:230-540: mapper, mapper with local BA, coarse mapper) and ``Tracker.optimize_cam_in_batch``
(src/Tracker.py:71-128) computed on the CPU for recorded states and pixel draws (tests/golden/make_golden_callers.py);
tests/caller_replay.py drives ``nice_slam_amd`` through a loop of the same shape -- masked leaves written into
channels-last grids with ``val[mask] = val_grad``, camera tensors -> get_samples, boolean-mask ray compaction,
render_batch_ray, the callers' losses, autograd -- teacher-forced on the reference's state.  Per iteration the loss and the
gradient of every tensor the reference's optimiser held must agree in max-norm (1e-4 of the tensor's maximum)."""
import pytest
import torch

import caller_replay as cr

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gold():
    return cr.load()


@pytest.mark.parametrize("pre", ["map/", "ba/", "coarse/"])
def test_product_reproduces_real_optimize_map(gold, pre):
    got = cr.replay_mapper(gold, pre, cr.ProductOps(gold))
    truth = cr.replay_mapper(gold, pre, cr.OracleOps(gold, lo=torch.float64))      # only for the cancelling-sum bias gradients
    bad, n = cr.compare(gold, pre, got, TOL, truth=truth)
    assert not bad, bad
    assert n >= {"map/": 30, "ba/": 45, "coarse/": 3}[pre]


def test_product_reproduces_real_optimize_cam_in_batch(gold):
    got = cr.replay_tracker(gold, cr.ProductOps(gold))
    bad, n = cr.compare(gold, "track/", got, TOL)
    assert not bad and n == 3, bad


@pytest.mark.parametrize("pre", ["map/", "ba/", "coarse/"])
def test_fused_mapping_loss_reproduces_real_optimize_map(gold, pre):
    """The entry point bench.py times -- ``mapping_loss``: window kernel, bounding-box MASK instead of compaction, loss and its
    derivative in the forward epilogue, split backward -- against the same fixtures, fed with the recorded draws."""
    ops = cr.ProductOps(gold)
    for p in ops.dec.parameters():
        p.requires_grad_(True)
    got = cr.replay_mapper_fused(gold, pre, ops)
    truth = cr.replay_mapper(gold, pre, cr.OracleOps(gold, lo=torch.float64))
    bad, n = cr.compare(gold, pre, got, TOL, truth=truth)
    assert not bad, bad
    assert n >= {"map/": 30, "ba/": 45, "coarse/": 3}[pre]


def test_fused_tracking_loss_reproduces_real_optimize_cam_in_batch(gold):
    ops = cr.ProductOps(gold)
    for p in ops.dec.parameters():                  # the tracker works on detached decoders (src/Tracker.py:138)
        p.requires_grad_(False)
    got = cr.replay_tracker_fused(gold, ops)
    bad, n = cr.compare(gold, "track/", got, TOL)
    assert not bad and n == 3, bad
