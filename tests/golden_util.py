"""Loading of the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def mcts_fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "mcts_*.npz")))


def mctsq_fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "mctsq_*.npz")))


def load_mcts(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cases = []
    for i in range(int(z["n_cases"])):
        pre = f"c{i}_"
        cases.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    return int(z["size"]), int(z["sims"]), float(z["temp"]), cases


def load_nn(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


class ReplayNet:
    """Feeds recorded NN outputs back, one leaf per call, checking the leaf state."""

    def __init__(self, case):
        self.case, self.i = case, 0

    def __call__(self, states):
        c, i = self.case, self.i
        assert np.array_equal(np.asarray(states[0]).astype(np.uint8), c["leaf_states"][i]), f"leaf {i} differs"
        self.i += 1
        return c["pi_logits"][i][None].copy(), np.asarray([[c["val_logits"][i]]], dtype=np.float32)


class ReplayCritic:
    """Feeds a recorded critic-mode search back: calls alternate between the selected leaf
    (batch of one) and, for non-terminal leaves, its valid children (tests/golden/mctsq_*.npz)."""

    def __init__(self, case):
        self.case, self.step, self.expect_children = case, 0, False

    def __call__(self, states):
        c = self.case
        states = np.asarray(states)
        if not self.expect_children:
            i = self.step
            assert len(states) == 1 and np.array_equal(states[0].astype(np.uint8), c["leaf_states"][i]), f"leaf {i} differs"
            if c["terminal"][i]:
                self.step += 1
            else:
                self.expect_children = True
            return np.asarray([[c["leaf_vals"][i]]], dtype=np.float32)
        i = self.step
        vals = c["child_vals"][i]
        vals = vals[~np.isnan(vals)]
        assert len(states) == len(vals), f"step {i}: {len(states)} children, expected {len(vals)}"
        self.expect_children = False
        self.step += 1
        return vals.astype(np.float32)[:, None]
