"""Loading / checking of the training fixtures (tests/golden/train_*.npz, made by make_golden.gen_train from the
reference's own train_step on its real nn.Module)."""
import os

import numpy as np

from golden_util import GOLDEN_DIR

TRAIN_FIXTURES = ("train_ac5", "train_ac9", "train_val9")


def load_train(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def check_against_golden(g, s, metrics, grads, params, stats, rtol, atol):
    """metrics: 4 values (None / nan where the net has none); grads / params / stats: dict key -> numpy array."""
    want = g[f"s{s}_metrics"]
    for got, w in zip(metrics, want):
        if np.isnan(w):
            assert got is None or np.isnan(got)
        else:
            assert abs(got - w) <= 1e-4 * max(1.0, abs(w)), (s, metrics, want)
    keys = [k[len("idx_"):] for k in g if k.startswith("idx_")]
    # the largest gradient rms of the step: a convolution bias in front of a BatchNorm has a gradient that is zero in
    # exact arithmetic, so its recorded values are rounding noise, to be judged against the step's real gradients
    top = {what: max(g[f"s{s}_{what}_{k}"][-1] / np.sqrt(g["idx_" + k].size and np.asarray(table[k]).size)
                     for k in keys) for what, table in (("grad", grads), ("param", params))}
    for key in keys:
        idx = g["idx_" + key]
        for what, table in (("grad", grads), ("param", params)):
            ref = g[f"s{s}_{what}_{key}"]
            a = np.asarray(table[key], dtype=np.float64).reshape(-1)
            scale = max(ref[-1] / np.sqrt(a.size), 1e-3 * top[what])   # the tensor's rms: errors are judged against it
            # Adam normalises every gradient by its own magnitude: where the gradient is rounding noise (plus weight
            # decay), the update is up to lr in either direction whatever the implementation
            noise_grad = g[f"s{s}_grad_{key}"][-1] / np.sqrt(a.size) < 1e-3 * top["grad"]
            if what == "param" and noise_grad:
                assert np.allclose(a[idx], ref[:-1], rtol=0, atol=1.1e-3 * (s + 1)), (s, what, key)
                continue
            assert np.allclose(a[idx], ref[:-1], rtol=rtol, atol=atol * scale), \
                (s, what, key, np.abs(a[idx] - ref[:-1]).max(), scale)
            assert abs(np.linalg.norm(a) - ref[-1]) <= rtol * max(ref[-1], 1e-12) + atol * scale * np.sqrt(a.size), (s, what, key)
    for k in [k for k in g if k.startswith(f"s{s}_stat_")]:
        key = k[len(f"s{s}_stat_"):]
        assert np.allclose(np.asarray(stats[key]), g[k], rtol=rtol, atol=atol), (s, key)
