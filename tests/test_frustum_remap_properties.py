"""CPU: hand-derived exact cases for the one leg of the frustum mask that cannot be pinned against
the reference here -- ``cv2.remap(depth, u, v, INTER_LINEAR)`` (src/Mapper.py:134; OpenCV is absent from the image, so
oracle/frustum_oracle.remap_bilinear restates OpenCV's published fixed-point algorithm: INTER_BITS = 5, cvRound,
BORDER_CONSTANT 0).  Every expectation below follows from that algorithm by hand, none from running the restatement;
the GPU kernel is held bit-equal to the restatement elsewhere (tests/test_hip_frustum.py, tests/test_emu_parity.py).
The row stays "parity unpinned" until a real cv2 is available."""
import numpy as np

from oracle.frustum_oracle import remap_bilinear

RNG = np.random.RandomState(0)
IMG = (RNG.rand(9, 13) * 4 + 0.5).astype(np.float32)
H, W = IMG.shape


def rm(u, v):
    return remap_bilinear(IMG, np.asarray(u, dtype=np.float32), np.asarray(v, dtype=np.float32))


def test_pixel_centres_return_the_pixel():
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    assert np.array_equal(rm(xs.ravel(), ys.ravel()), IMG.ravel())


def test_half_pixel_is_the_exact_mean_of_two_taps():
    # weights (1-0.5, 0.5) are exact in fp32; the four products are summed left to right
    x, y = 4, 3
    want = (IMG[y, x] * np.float32(0.5) + IMG[y, x + 1] * np.float32(0.5)).astype(np.float32)
    assert rm([x + 0.5], [y])[0] == want
    want = (IMG[y, x] * np.float32(0.5) + IMG[y + 1, x] * np.float32(0.5)).astype(np.float32)
    assert rm([x], [y + 0.5])[0] == want
    q = np.float32(0.25)
    want = ((IMG[y, x] * q + IMG[y, x + 1] * q) + IMG[y + 1, x] * q) + IMG[y + 1, x + 1] * q
    assert rm([x + 0.5], [y + 0.5])[0] == np.float32(want)


def test_coordinates_snap_to_the_1_32_grid():
    x, y = 5.0, 2.0
    base = rm([x + 7 / 32], [y + 3 / 32])[0]
    for dx in (-0.4 / 32, 0.4 / 32):                       # closer than half a step: same fixed-point coordinate
        assert rm([x + 7 / 32 + dx], [y + 3 / 32])[0] == base
        assert rm([x + 7 / 32], [y + 3 / 32 + dx])[0] == base
    assert rm([x + 8 / 32], [y + 3 / 32])[0] != base


def test_ties_round_half_to_even():
    # x*32 = 160 + k + 0.5: cvRound -> the even neighbour.  (5 + 0.5/32)*32 = 160.5 -> 160; (5 + 1.5/32)*32 = 161.5 -> 162
    y = 4.0
    assert rm([5 + 0.5 / 32], [y])[0] == rm([5.0], [y])[0]
    assert rm([5 + 1.5 / 32], [y])[0] == rm([5 + 2 / 32], [y])[0]
    assert rm([5 + 2.5 / 32], [y])[0] == rm([5 + 2 / 32], [y])[0]


def test_constant_zero_border():
    a = np.float32(8 / 32)
    # u = -1 + 8/32: the left tap is outside (0), the right tap is column 0 with weight 8/32
    assert rm([-1 + 8 / 32], [3.0])[0] == IMG[3, 0] * a
    # one step beyond the last column / row: only the in-image tap contributes, weight (1 - 8/32)
    assert rm([W - 1 + 8 / 32], [3.0])[0] == IMG[3, W - 1] * (np.float32(1) - a)
    assert rm([4.0], [H - 1 + 8 / 32])[0] == IMG[H - 1, 4] * (np.float32(1) - a)
    # fully outside, far outside, NaN: zero
    out = rm([-1.0, -5.0, W + 0.0, 1e9, -1e9, np.nan, 3.0], [2.0, 2.0, 2.0, 2.0, 2.0, 2.0, -1.0])
    assert np.array_equal(out, np.zeros(7, dtype=np.float32))


def test_linear_ramp_is_reproduced_exactly_on_the_1_32_grid():
    # depth = x/4 + y/2 (exact in fp32 together with every weight k/32): bilinear interpolation of a ramp is the ramp
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ramp = (xs / 4 + ys / 2).astype(np.float32)
    u = np.array([0.0, 1 + 5 / 32, 7 + 31 / 32, 11.0, 3 + 16 / 32], dtype=np.float32)
    v = np.array([0.0, 2 + 9 / 32, 6 + 1 / 32, 7 + 30 / 32, 4.0], dtype=np.float32)
    got = remap_bilinear(ramp, u, v)
    assert np.array_equal(got, (u / 4 + v / 2).astype(np.float32))
