"""The MCTS/policy oracle (oracle/mcts_ref.py) against golden searches recorded from the
reference's own tree.py / mct.py / actorcritic.py (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from golden_util import ReplayCritic, ReplayNet, load_mcts, mcts_fixture_names, mctsq_fixture_names
from oracle import mcts_ref
from oracle.gogame import GoEnv, GoGame


def _env(root):
    env = GoEnv(root.shape[-1])
    env.state = root.astype(np.float64)   # canonical root: canonical_state() is the identity on it
    return env


@pytest.mark.parametrize("name", mcts_fixture_names())
def test_oracle_reproduces_reference_search(oracle_lib, name):
    size, sims, temp, cases = load_mcts(name)
    for c in cases:
        net = ReplayNet(c)
        pi, counts, root = mcts_ref.actor_critic_policy(_env(c["root"]), net, sims, temp, math="numpy")
        assert net.i == sims + 1
        assert np.array_equal(counts.astype(np.int32), c["visits"])
        assert root.visits == c["root_visits"]
        assert np.array_equal(root.prior_pi, c["root_prior"])
        assert np.array_equal(pi, c["pi"])            # fp64, bit for bit


@pytest.mark.parametrize("name", mcts_fixture_names())
def test_deterministic_math_gives_same_visits(oracle_lib, name):
    """math='det' (what the GPU reproduces) never flips a PUCT argmax on the golden searches."""
    size, sims, temp, cases = load_mcts(name)
    for c in cases:
        net = ReplayNet(c)
        pi, counts, root = mcts_ref.actor_critic_policy(_env(c["root"]), net, sims, temp, math="det")
        assert np.array_equal(counts.astype(np.int32), c["visits"])
        assert np.allclose(root.prior_pi, c["root_prior"], rtol=1e-6, atol=1e-9)
        assert np.allclose(pi, c["pi"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", mctsq_fixture_names())
@pytest.mark.parametrize("math", ["numpy", "det"])
def test_oracle_reproduces_reference_q_search(oracle_lib, name, math):
    """Critic-driven search + Value policy against the reference's own (mct.py:27-47, value.py:19-45)."""
    size, sims, temp, cases = load_mcts(name)
    for c in cases:
        net = ReplayCritic(c)
        pi, root = mcts_ref.value_policy(_env(c["root"]), net, sims, temp, math=math)
        assert net.step == sims + 1
        assert np.array_equal(root.get_visit_counts().astype(np.int32), c["visits"])
        assert root.visits == c["root_visits"]
        assert np.array_equal(root.inverted_children_values(), c["inv_child_vals"])
        if math == "numpy":
            assert np.array_equal(root.prior_pi, c["root_prior"])
            assert np.array_equal(pi, c["pi"])
        else:
            assert np.allclose(root.prior_pi, c["root_prior"], rtol=1e-12, atol=1e-300)
            assert np.allclose(pi, c["pi"], rtol=1e-12, atol=0)


def test_reference_kat_greedy_obvious_move(oracle_lib):
    # /root/reference/tests/test_policy.py:34-43 through the oracle's Value policy
    z = np.load(__import__("os").path.join(__import__("golden_util").GOLDEN_DIR, "policy_kat_greedy5.npz"))

    def greedy_val_func(states):   # /root/reference/go_ai/policies/baselines.py:11-30
        vals = []
        for st in states:
            b, w = GoGame.areas(st)
            if GoGame.game_ended(st):
                vals.append(100 if b > w else -100 if b < w else 0)
            else:
                vals.append((b - w) / 25)
        return np.array(vals, dtype=float)[:, None]

    env = GoEnv(5)
    env.state = z["state"].astype(np.float64)
    pi, _ = mcts_ref.value_policy(env, greedy_val_func, 0, 1)
    assert pi[-1] == 1 and np.allclose(pi[:-1], 0)
    assert np.array_equal(pi, z["pi"])


def test_pairwise_sum_matches_numpy(oracle_lib):
    rng = np.random.default_rng(0)
    for n in list(range(1, 140)) + [255, 256, 257, 699]:
        a = rng.standard_normal(n)
        assert mcts_ref.pairwise_sum(a) == np.add.reduce(a)
        assert mcts_ref.pairwise_sum(a) / n == np.mean(a)
