"""CPU: the oracle and the reference-shaped replay loops (tests/caller_replay.py) pinned against fixtures minted by running
the UNMODIFIED reference callers -- ``Mapper.optimize_map`` (mapper, mapper with local BA, coarse mapper) and
``Tracker.optimize_cam_in_batch`` -- on the CPU (tests/golden/make_golden_callers.py).  Every loss and every gradient the
reference's optimiser saw (masked grid leaves, colour-decoder parameters, BA / tracking camera tensors) must be reproduced
by the oracle in the restated loop; the GPU twin of this test (tests/test_hip_real_callers.py) then holds the product to
the same fixtures."""
import pytest

import caller_replay as cr

TOL = 2e-5          # fp32 CPU vs fp32 CPU, different operation order


@pytest.fixture(scope="module")
def gold():
    return cr.load()


@pytest.mark.parametrize("pre", ["map/", "ba/", "coarse/"])
def test_oracle_reproduces_real_optimize_map(gold, pre):
    got = cr.replay_mapper(gold, pre, cr.OracleOps(gold))
    bad, n = cr.compare(gold, pre, got, TOL)
    assert not bad, bad
    assert n >= {"map/": 30, "ba/": 45, "coarse/": 3}[pre]
    want = {"map/": ["middle"] * 3 + ["fine", "color"], "ba/": ["middle"] * 3 + ["fine", "color"], "coarse/": ["coarse"] * 3}[pre]
    assert [r["stage"] for r in got] == want


def test_oracle_reproduces_real_optimize_cam_in_batch(gold):
    got = cr.replay_tracker(gold, cr.OracleOps(gold))
    bad, n = cr.compare(gold, "track/", got, TOL)
    assert not bad and n == 3, bad
    assert abs(got[0]["loss"] - float(gold["track/ret_losses"][0])) < 1e-5 * abs(got[0]["loss"])
