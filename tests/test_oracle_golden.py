"""Pin the oracle (oracle/moving.py) to golden vectors produced by the reference's own code
(tests/golden/make_golden.py ran SonicSim-SonicSet/SonicSim_moving.py unmodified)."""
import numpy as np

from oracle import moving
from util import golden, golden_inputs, rel_rms


def test_fixed_cfg1_bitwise():
    g = golden("g1_fixed_cfg1.npz")
    y = moving.convolve_fixed_receiver(g["x"], g["h"])
    assert str(y.dtype) == str(g["y_dtype"])
    assert np.array_equal(y.astype(np.float32), g["y"])


def test_fixed_torch_inputs():
    g = golden("g2_fixed_torch.npz")
    assert np.array_equal(moving.convolve_fixed_receiver(g["x"], g["h"]).astype(np.float32), g["y"])


def test_setup_dynamic_interp_rng_coupling():
    g = golden("g3_interp.npz")
    for i in range(int(g["n"])):
        np.random.seed(int(g[f"seed{i}"]))
        idx, w = moving.setup_dynamic_interp(g[f"pos{i}"], int(g[f"T{i}"]))
        assert np.array_equal(idx, np.repeat(np.arange(len(g[f"seg_len{i}"])), g[f"seg_len{i}"]))
        assert w.dtype == np.float32 and np.array_equal(w, g[f"w{i}"])
        np.random.seed(int(g[f"seed{i}"]))
        n = moving.segment_lengths(g[f"pos{i}"], int(g[f"T{i}"]))
        assert np.array_equal(n, g[f"seg_len{i}"])
        i2, w2 = moving.expand_segments(n)
        assert np.array_equal(i2, idx) and np.array_equal(w2, w)


def test_moving_small_bitwise_and_chunked():
    g = golden("g4_moving_small.npz")
    y = moving.convolve_moving_receiver(g["x"], g["bank"], g["idx"], g["w"])
    assert np.array_equal(y, g["y"])
    for pc in (1, 2, 3):
        assert np.array_equal(moving.convolve_moving_receiver(g["x"], g["bank"], g["idx"], g["w"], p_chunk=pc), g["y"])


def test_moving_medium_from_seed():
    g = golden("g5_moving_medium.npz")
    x, bank, pos = golden_inputs(int(g["seed"]), int(g["T"]), int(g["P"]), int(g["C"]), int(g["L"]))
    np.random.seed(int(g["np_seed"]))
    idx, w = moving.setup_dynamic_interp(pos, int(g["T"]))
    assert np.array_equal(np.bincount(idx, minlength=int(g["P"]) - 1), g["seg_len"])
    y = moving.convolve_moving_receiver(x, bank, idx, w, p_chunk=5)
    assert np.array_equal(y, g["y"])


def test_edges_and_arbitrary_index():
    g = golden("g6_edges.npz")
    assert np.array_equal(moving.convolve_moving_receiver(g["x"], g["bank"], g["idx"], g["w"]), g["y"])
    assert np.array_equal(moving.convolve_moving_receiver(g["x1"], g["bank1"], g["idx1"], g["w1"]), g["y1"])
    g = golden("g8_arbitrary_idx.npz")
    assert np.array_equal(moving.convolve_moving_receiver(g["x"], g["bank"], g["idx"], g["w"]), g["y"])
    assert np.array_equal(moving.convolve_moving_receiver(g["x"], g["bank"], g["idx"], g["w"], p_chunk=2), g["y"])


def test_interpolate_moving_audio():
    g = golden("g7_interpolate.npz")
    x, bank, pos = golden_inputs(int(g["seed"]), int(g["T"]), int(g["P"]), int(g["C"]), int(g["L"]))
    np.random.seed(int(g["np_seed"]))
    y = moving.interpolate_moving_audio(x[None, :], bank[:, None], list(pos))
    assert np.array_equal(y, g["y"])


def test_independent_checkers_agree_with_reference():
    """The float64 closed form and the segment-wise reformulation (the algorithm the GPU implements)
    reproduce the reference output to its own float32 noise floor."""
    g = golden("g4_moving_small.npz")
    pts = np.array([0, 1, 17, 2999, 3000, 5555, 11999])
    d = moving.direct_form_f64(g["x"], g["bank"], g["idx"], g["w"], pts)
    assert np.abs(d - g["y"][:, pts]).max() < 2e-4 * np.abs(g["y"]).max()
    seg = np.bincount(g["idx"], minlength=g["bank"].shape[0] - 1)
    ys = moving.segmentwise_f64(g["x"], g["bank"], seg)
    assert rel_rms(ys, g["y"]) < 2e-6
