"""GPU parity of the Q-value (critic-driven) tree search through the C ABI
(GOAI_SEARCH_Q: goai_mcts_select / goai_mcts_expand / goai_mcts_backup_q / goai_mcts_run).

1. Replay of golden searches recorded from the reference's own Value policy
   (tests/golden/mctsq_*.npz; /root/reference/go_ai/search/mct.py:27-47,
   /root/reference/go_ai/policies/value.py:19-45): fed the reference's critic outputs, the GPU
   must select the same leaf states, build the same children and end with bit-identical visit
   counts and child values; priors / pi agree to the last few ulp (the reference's prior is a
   scipy fp64 softmax, the GPU's the deterministic restatement).
2. Full GPU search with its own value network vs the sequential oracle (math='det') fed the
   GPU's network outputs: visit counts, prior and pi bit for bit.
3. goai_mcts_run (compacted leaf + children batch, fused overrides) == the step-wise path.
"""
import numpy as np
import pytest

from golden_util import load_mcts, mctsq_fixture_names
from helpers import random_positions
from oracle import mcts_ref, nn_ref
from oracle.gogame import GoEnv, GoGame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", mctsq_fixture_names())
def test_replay_reference_q_search(name):
    from go_ai.engine import Context
    size, sims, temp, cases = load_mcts(name)
    n = len(cases)
    c = Context(size)
    c.mcts_init(n, sims, mode=Context.SEARCH_Q)
    c.mcts_reset(np.stack([cs["root"] for cs in cases]))
    for s in range(sims + 1):
        leaves = c.mcts_select().cpu().numpy()
        kids, valid = [t.cpu().numpy() for t in c.mcts_expand()]
        for i, cs in enumerate(cases):
            assert np.array_equal(leaves[i], cs["leaf_states"][s]), f"{name}: case {i} step {s}: different leaf"
            if cs["terminal"][s]:
                assert not valid[i].any() and not kids[i].any()
            else:
                st = cs["leaf_states"][s].astype(np.float64)
                assert np.array_equal(valid[i], GoGame.valid_moves(st).astype(np.uint8))
                assert np.array_equal(kids[i], np.asarray(GoGame.children(st, canonical=True, padded=True)).astype(np.uint8))
        leaf_val = np.array([cs["leaf_vals"][s] for cs in cases], dtype=np.float32)
        child_vals = np.nan_to_num(np.stack([cs["child_vals"][s] for cs in cases]), nan=0.0).astype(np.float32)
        c.mcts_backup_q(leaf_val, child_vals, vals_are_raw=False)
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    inv_vals = c.mcts_child_values().cpu().numpy()
    pi = c.mcts_policy_value(temp, from_values=(sims == 0)).cpu().numpy()
    assert c.mcts_counters()["errors"] == 0
    c.close()
    for i, cs in enumerate(cases):
        assert np.array_equal(visits[i], cs["visits"])
        assert root_visits[i] == cs["root_visits"]
        assert np.array_equal(inv_vals[i], cs["inv_child_vals"])
        assert np.allclose(prior[i], cs["root_prior"], rtol=1e-12, atol=1e-300)
        if temp <= 0:
            assert np.array_equal(pi[i], cs["pi"])        # search.greedy_pi: uniform over the best moves, exact
        elif sims > 0 and temp == 1.0:
            assert np.array_equal(pi[i], cs["pi"])        # integer counts: exact
        elif temp == 1.0:
            assert np.allclose(pi[i], cs["pi"], rtol=1e-12, atol=0)     # exp(-child values): the deterministic exp
        else:
            # temp_norm's ** (1 / T) (search/__init__.py:85): the device pow(), and for mcts = 0 the deterministic
            # exp under it; bound 1e-12 relative on every entry
            assert np.array_equal(pi[i] == 0, cs["pi"] == 0)
            assert np.allclose(pi[i], cs["pi"], rtol=1e-12, atol=0)


def _stepwise(c, sims, log=None):
    """select -> expand -> value forward of the leaf and of its valid children -> backup_q."""
    n_games, A = c.n_games, c.A
    for s in range(sims + 1):
        leaves = c.mcts_select()
        kids, valid = c.mcts_expand()
        _, lv = c.nn_forward(leaves, mode=c.NN_NUMPY_SHIM, want_pi=False)
        _, kv = c.nn_forward(kids.reshape(n_games * A, 6, c.size, c.size), mode=c.NN_NUMPY_SHIM, want_pi=False)
        c.mcts_backup_q(lv, kv.reshape(n_games, A), vals_are_raw=False)
        if log is not None:
            log.append((leaves.cpu().numpy(), valid.cpu().numpy().astype(bool), lv.cpu().numpy().reshape(-1),
                        kv.cpu().numpy().reshape(n_games, A)))


@pytest.mark.parametrize("size,n_games,sims,kind", [(5, 16, 10, "val"), (9, 12, 8, "val"), (9, 4, 5, "ac"), (13, 3, 4, "val"),
                                                   (19, 2, 3, "val")])     # 362 actions: six ballot words per node
def test_full_q_search_matches_oracle(size, n_games, sims, kind):
    from go_ai.engine import Context
    sd = nn_ref.make_state_dict(kind, size, 70 + size)
    roots = random_positions(size, n_games, seed=500 + size, max_moves=size * size // 2, canonical=True, pass_prob=0.1)
    c = Context(size)
    c.nn_init(Context.NET_VAL if kind == "val" else Context.NET_AC, 3, 256)
    c.load_weights(sd)
    c.mcts_init(n_games, sims, mode=Context.SEARCH_Q)
    c.mcts_reset(roots)
    log = []
    _stepwise(c, sims, log)
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    pis = c.mcts_policy_value(1.0).cpu().numpy()
    inv_vals = c.mcts_child_values().cpu().numpy()
    cnt = c.mcts_counters()
    assert cnt["errors"] == 0 and cnt["leaf_evals"] == n_games * (sims + 1)

    # the internal path (one compacted batch per step, overrides fused into the backup)
    c.mcts_reset(roots)
    c.mcts_run(sims)
    visits2, prior2, _ = [t.cpu().numpy() for t in c.mcts_readout()]
    assert np.array_equal(visits, visits2) and np.array_equal(prior, prior2)
    assert np.array_equal(pis, c.mcts_policy_value(1.0).cpu().numpy())
    assert np.array_equal(inv_vals, c.mcts_child_values().cpu().numpy())
    assert c.mcts_counters()["errors"] == 0
    c.close()

    for g in range(n_games):
        class Feed:
            step, kids = 0, False

            def __call__(self, states):
                leaves, valid, lv, kv = log[self.step]
                states = np.asarray(states)
                if not self.kids:
                    assert len(states) == 1 and np.array_equal(states[0].astype(np.uint8), leaves[g]), \
                        f"game {g} step {self.step}: GPU and oracle selected different leaves"
                    if GoGame.game_ended(states[0]):
                        self.step += 1
                    else:
                        self.kids = True
                    return np.asarray([[lv[g]]], dtype=np.float32)
                assert len(states) == valid[g].sum()
                self.kids = False
                self.step += 1
                return kv[g][valid[g]].astype(np.float32)[:, None]

        env = GoEnv(size)
        env.state = roots[g].astype(np.float64)
        feed = Feed()
        pi, root = mcts_ref.value_policy(env, feed, sims, 1.0, math="det")
        assert feed.step == sims + 1
        assert np.array_equal(root.get_visit_counts().astype(np.int32), visits[g])
        assert np.array_equal(root.prior_pi, prior[g])
        assert np.array_equal(root.inverted_children_values(), inv_vals[g])
        assert np.array_equal(pi, pis[g])


def test_q_policy_from_values_matches_oracle():
    """Value policy with mcts == 0 (value.py:34-37): exp(-child values) -> temp_norm."""
    from go_ai.engine import Context
    size, n_games = 9, 24
    sd = nn_ref.make_state_dict("val", size, 3)
    roots = random_positions(size, n_games, seed=77, max_moves=60, canonical=True, pass_prob=0.1)
    c = Context(size)
    c.nn_init(Context.NET_VAL, 3, 512)
    c.load_weights(sd)
    c.mcts_init(n_games, 0, mode=Context.SEARCH_Q)
    c.mcts_reset(roots)
    log = []
    _stepwise(c, 0, log)
    pis = c.mcts_policy_value(1.0, from_values=True).cpu().numpy()
    leaves, valid, lv, kv = log[0]
    for g in range(n_games):
        calls = []

        def feed(states, g=g):
            calls.append(len(states))
            return np.asarray([[lv[g]]], np.float32) if len(calls) == 1 else kv[g][valid[g]].astype(np.float32)[:, None]

        env = GoEnv(size)
        env.state = roots[g].astype(np.float64)
        pi, root = mcts_ref.value_policy(env, feed, 0, 1.0, math="det")
        assert np.array_equal(pi, pis[g])
    c.close()
