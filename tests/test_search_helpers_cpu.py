"""The policy helpers of go_ai/search/__init__.py (host fp64) against vectors recorded from the reference's own
functions (/root/reference/go_ai/search/__init__.py:9-97; tests/golden/search_helpers.npz, made by
tests/golden/make_golden.py): invert_vals, greedy_pi, batch_greedy_pi, temp_softmax, temp_norm and
batch_temperate_pi at temperatures 0, 0.5, 1 and 2.  Greedy and T = 1 results are exact; the others differ
from sklearn / scipy only in how the normalising sum is rounded (bound: 4 ulp)."""
import os

import numpy as np

from golden_util import GOLDEN_DIR

ULP4 = 4 * np.finfo(np.float64).eps


def test_helpers_match_reference_vectors():
    from go_ai import search
    z = np.load(os.path.join(GOLDEN_DIR, "search_helpers.npz"))
    q, valid, counts, temps = z["q"], z["valid"], z["counts"], z["temps"]
    n = len(q)
    assert np.array_equal(search.invert_vals(q), z["invert"])
    assert np.array_equal(np.stack([search.greedy_pi(q[i], valid[i]) for i in range(n)]), z["greedy_pi"])
    assert np.array_equal(np.stack([search.greedy_pi(counts[i], valid[i]) for i in range(n)]), z["greedy_pi_counts"])
    assert np.array_equal(search.batch_greedy_pi(q.copy(), valid), z["batch_greedy_pi"])
    for ti, t in enumerate(temps):
        exact = t <= 0
        for name, fn in (("temp_softmax", lambda i: search.temp_softmax(q[i], t, valid[i])),
                         ("temp_norm_exp", lambda i: search.temp_norm(np.exp(q[i]), t, valid[i])),
                         ("temp_norm_counts", lambda i: search.temp_norm(counts[i], t, valid[i]))):
            got, want = np.stack([fn(i) for i in range(n)]), z[f"{name}_{ti}"]
            assert got.shape == want.shape and np.array_equal(got == 0, want == 0), (name, t)
            if exact:
                assert np.array_equal(got, want), (name, t)
            else:
                assert np.allclose(got, want, rtol=ULP4, atol=0), (name, t, np.abs(got / np.where(want == 0, 1, want) - 1).max())
        got, want = search.batch_temperate_pi(q.copy(), t, valid), z[f"batch_temperate_pi_{ti}"]
        assert np.array_equal(got, want) if exact else np.allclose(got, want, rtol=ULP4, atol=0), t
    # T = 1 on integer counts is exact (what Value.__call__ does with visit counts)
    one = int(np.flatnonzero(temps == 1.0)[0])
    assert np.array_equal(np.stack([search.temp_norm(counts[i], 1.0, valid[i]) for i in range(n)]), z[f"temp_norm_counts_{one}"])
