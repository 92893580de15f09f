"""Accuracy of the deterministic exp/tanh/softmax restatements (oracle/detmath.c) against
numpy/scipy -- the routines the reference actually calls (mct.py:41, tree.py:157,180)."""
import numpy as np
from scipy import special

from oracle import mcts_ref


def _ulp_diff64(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def test_tanh64_within_3ulp_and_odd(oracle_lib):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32).astype(np.float64) * s
                        for s in (1e-3, 0.1, 1.0, 5.0)] + [np.array([0.0, 100.0, -100.0, 19.9, 20.0, 0.5, -0.5,
                                                                     2.0 ** -28, 2.0 ** -29, 1e-300])])
    t = mcts_ref.det_tanh(x)
    ref = np.tanh(x)
    assert _ulp_diff64(t, ref).max() <= 3
    assert np.array_equal(mcts_ref.det_tanh(-x), -t)
    assert mcts_ref.det_tanh(np.array([100.0]))[0] == 1.0 and mcts_ref.det_tanh(np.array([0.0]))[0] == 0.0


def test_softmax32_close_to_scipy(oracle_lib):
    rng = np.random.default_rng(2)
    lowest = np.finfo(np.float32).min
    for n in (26, 82, 170, 362):
        for _ in range(20):
            x = (rng.standard_normal(n) * rng.choice([0.5, 3.0, 10.0])).astype(np.float32)
            mask = rng.random(n) < 0.3
            x[mask] = lowest                  # invalid moves after data.batch_invalid_values
            x[-1] = rng.standard_normal()
            got, want = mcts_ref.det_softmax32(x), special.softmax(x)
            assert got.dtype == np.float32
            assert np.all(got[mask[:-1].nonzero()[0]] == 0)
            assert np.allclose(got, want, rtol=2e-6, atol=1e-30)
            assert abs(float(got.sum()) - 1.0) < 1e-5


def test_softmax64_masked_close_to_scipy(oracle_lib):
    rng = np.random.default_rng(3)
    for n in (26, 82, 362):
        x = rng.standard_normal(n) * 3
        valid = (rng.random(n) < 0.7).astype(np.uint8)
        valid[-1] = 1
        got = mcts_ref.det_softmax64_masked(x, valid)
        want = np.zeros(n)
        want[valid.nonzero()[0]] = special.softmax(x[valid.nonzero()[0]])
        assert np.allclose(got, want, rtol=1e-14, atol=0)
        assert np.all(got[valid == 0] == 0)
