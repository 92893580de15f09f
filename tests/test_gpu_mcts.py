"""GPU parity of the tree kernels (select / expand / backup / policy) through the C ABI.

1. Replay of golden searches recorded from the reference's own tree.py / mct.py
   (tests/golden/mcts_*.npz): the GPU is fed the reference's NN outputs step by step (val
   logits and the fp32 prior scipy produced) and must select the same leaf STATES at every
   step and end with bit-identical visit counts, root prior and policy pi.
2. Same replay but feeding the recorded pi_logits: the prior comes from the GPU's own
   deterministic fp32 softmax; visit counts must still match the reference on every fixture.
3. Full GPU search (own CNN) against the CPU oracle running math='det' with the GPU's NN
   outputs fed back: leaf states, visits and pi bit-exact at the sizes of BASELINE.json.
"""
import numpy as np
import pytest
import torch

from golden_util import load_mcts, mcts_fixture_names
from helpers import random_positions
from oracle import mcts_ref, nn_ref
from oracle.gogame import GoEnv

pytestmark = pytest.mark.gpu


def _replay(name, feed):
    from go_ai.engine import Context
    size, sims, temp, cases = load_mcts(name)
    n = len(cases)
    c = Context(size)
    c.mcts_init(n, sims)
    c.mcts_reset(np.stack([cs["root"] for cs in cases]))
    for s in range(sims + 1):
        leaves = c.mcts_select().cpu().numpy()
        for i, cs in enumerate(cases):
            assert np.array_equal(leaves[i], cs["leaf_states"][s]), f"{name}: case {i} step {s}: different leaf"
        val = np.array([cs["val_logits"][s] for cs in cases], dtype=np.float32)
        if feed == "prior":
            c.mcts_backup(val, prior=np.stack([cs["priors"][s] for cs in cases]))
        else:
            c.mcts_backup(val, pi_logits=np.stack([cs["pi_logits"][s] for cs in cases]), logits_are_raw=False)
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    pi = c.mcts_policy(temp).cpu().numpy()
    assert c.mcts_counters()["errors"] == 0
    c.close()
    return cases, visits, prior, root_visits, pi


@pytest.mark.parametrize("name", mcts_fixture_names())
def test_replay_reference_search_fed_priors(name):
    cases, visits, prior, root_visits, pi = _replay(name, "prior")
    temp = load_mcts(name)[2]
    for i, cs in enumerate(cases):
        assert np.array_equal(visits[i], cs["visits"])
        assert root_visits[i] == cs["root_visits"]
        assert np.array_equal(prior[i], cs["root_prior"])
        if temp == 1.0:
            assert np.array_equal(pi[i], cs["pi"]), np.abs(pi[i] - cs["pi"]).max()
        else:
            # counts ** (1 / T) (actorcritic.py:34): the device pow() against numpy's, bound 4 ulp on every entry
            assert np.array_equal(pi[i] == 0, cs["pi"] == 0)
            assert np.allclose(pi[i], cs["pi"], rtol=4 * np.finfo(np.float64).eps, atol=0), np.abs(pi[i] / np.where(cs["pi"] == 0, 1, cs["pi"]) - 1).max()


@pytest.mark.parametrize("name", mcts_fixture_names())
def test_replay_reference_search_fed_logits(name):
    cases, visits, prior, root_visits, pi = _replay(name, "logits")
    for i, cs in enumerate(cases):
        assert np.array_equal(visits[i], cs["visits"])
        assert np.allclose(prior[i], cs["root_prior"], rtol=2e-6, atol=1e-30)
        assert np.array_equal(prior[i], mcts_ref.det_softmax32(cs["pi_logits"][0]).astype(np.float64))


@pytest.mark.parametrize("size,n_games,sims,layers", [(5, 24, 8, 3), (9, 16, 81, 3), (13, 4, 40, 3),
                                                      (19, 2, 30, 10),      # BASELINE.json configs[4]: 362 actions, deeper CNN
                                                      (19, 3, 81, 3),
                                                      # more than 128 simulations: nodes with more than 128 values, whose
                                                      # mean follows numpy's recursion over halves (Tree::vlog)
                                                      (5, 6, 300, 3), (9, 3, 200, 3)])
def test_full_search_matches_oracle(size, n_games, sims, layers):
    """GPU search with its own CNN vs the sequential oracle fed the same NN outputs."""
    from go_ai.engine import Context
    c = Context(size)
    c.nn_init(Context.NET_AC, layers, n_games)
    sd = nn_ref.make_state_dict("ac", size, 40 + size, layers)
    c.load_weights(sd)
    c.mcts_init(n_games, sims)
    roots = random_positions(size, n_games, seed=300 + size, max_moves=size * size // 2, canonical=True,
                             pass_prob=0.1)
    c.mcts_reset(roots)
    leaf_log, pi_log, val_log = [], [], []
    for s in range(sims + 1):
        leaves = c.mcts_select()
        pi, val = c.nn_forward(leaves, mode=c.NN_NUMPY_SHIM)
        raw_pi, raw_val = c.nn_forward(leaves, mode=c.NN_RAW)
        c.mcts_backup(raw_val, pi_logits=raw_pi, logits_are_raw=True)     # fused shim inside the backup
        leaf_log.append(leaves.cpu().numpy())
        pi_log.append(pi.cpu().numpy())
        val_log.append(val.cpu().numpy())
    visits = c.mcts_readout()[0].cpu().numpy()
    pis = c.mcts_policy(1.0).cpu().numpy()
    assert c.mcts_counters() == dict(leaf_evals=n_games * (sims + 1), transitions=c.mcts_counters()["transitions"],
                                     errors=0, forwards=0)
    for g in range(n_games):
        case = dict(leaf_states=[l[g] for l in leaf_log], pi_logits=[p[g] for p in pi_log],
                    val_logits=[v[g, 0] for v in val_log])

        class Feed:
            i = 0

            def __call__(self, states):
                assert np.array_equal(np.asarray(states[0]).astype(np.uint8), case["leaf_states"][self.i]), \
                    f"game {g} step {self.i}: GPU and oracle selected different leaves"
                out = case["pi_logits"][self.i][None].copy(), np.asarray([[case["val_logits"][self.i]]], np.float32)
                self.i += 1
                return out

        env = GoEnv(size)
        env.state = roots[g].astype(np.float64)
        pi, counts, root = mcts_ref.actor_critic_policy(env, Feed(), sims, 1.0, math="det")
        assert np.array_equal(counts.astype(np.int32), visits[g])
        assert np.array_equal(pi, pis[g])
    c.close()


def test_run_equals_stepwise_and_advance():
    """goai_mcts_run == explicit select/forward/backup loop; advance() plays the move."""
    from go_ai.engine import Context
    from oracle.gogame import GoGame
    size, n_games, sims = 9, 32, 20
    sd = nn_ref.make_state_dict("ac", size, 5)
    roots = random_positions(size, n_games, seed=9, max_moves=30, canonical=True)
    res = []
    for mode in ("run", "steps"):
        c = Context(size)
        c.nn_init(Context.NET_AC, 3, n_games)
        c.load_weights(sd)
        c.mcts_init(n_games, sims)
        c.mcts_reset(roots)
        if mode == "run":
            c.mcts_run(sims)
        else:
            for s in range(sims + 1):
                leaves = c.mcts_select()
                pi, val = c.nn_forward(leaves, mode=c.NN_RAW)
                c.mcts_backup(val, pi_logits=pi, logits_are_raw=True)
        res.append(c.mcts_readout()[0].cpu().numpy())
        if mode == "run":
            pi = c.mcts_policy(1.0)
            acts = c.sample_actions(pi, seed=1234, step=0)
            acts2 = c.sample_actions(pi, seed=1234, step=0)
            assert torch.equal(acts, acts2)
            a = acts.cpu().numpy()
            assert (pi.cpu().numpy()[np.arange(n_games), a] > 0).all()
            c.mcts_advance(acts)
            states, turn, moves, result = [t.cpu().numpy() for t in c.mcts_roots()]
            want, status = GoGame.next_states_u8(roots, a, canonical=True)
            assert (status == 0).all() and np.array_equal(states, want)
            assert (turn == 1).all() and (moves == 1).all()
            assert c.mcts_counters()["errors"] == 0
        c.close()
    assert np.array_equal(res[0], res[1])
    assert (res[0].sum(axis=1) == sims).all()


def test_sampled_actions_follow_pi():
    """goai_sample_actions = GoGame.random_weighted_action(pi) (game.py:91): 102,400 Philox draws (4096 games x 25
    steps) from one skewed distribution with zero entries: never a zero-probability action, and a chi-square test of
    the counts against pi (the reference draws from numpy's unseeded generator, so only the distribution is comparable)."""
    from scipy import stats
    from go_ai.engine import Context
    size, n_games, steps = 9, 4096, 25
    A = size * size + 1
    rng = np.random.default_rng(5)
    p = rng.random(A) ** 3
    p[rng.random(A) < 0.3] = 0.0
    p[-1] = 0.02
    p /= p.sum()
    c = Context(size)
    c.mcts_init(n_games, 1)
    pi = torch.from_numpy(np.tile(p, (n_games, 1))).to(c.device)
    draws = torch.stack([c.sample_actions(pi, seed=99, step=s) for s in range(steps)]).cpu().numpy()
    counts = np.bincount(draws.reshape(-1), minlength=A).astype(np.float64)
    assert counts[p == 0].sum() == 0
    support = p > 0
    chi2 = float((((counts - p * draws.size) ** 2)[support] / (p * draws.size)[support]).sum())
    pval = float(stats.chi2.sf(chi2, int(support.sum()) - 1))
    print(f"chi2 = {chi2:.1f} over {int(support.sum()) - 1} dof, p = {pval:.3f}")
    assert pval > 1e-3
    # streams differ per game, per step and per seed, and do not depend on the sharding offset arithmetic
    assert len({tuple(d) for d in draws.T[:64]}) == 64
    again = c.sample_actions(pi, seed=99, step=3).cpu().numpy()
    assert np.array_equal(again, draws[3])
    other = c.sample_actions(pi, seed=100, step=3).cpu().numpy()
    assert (other != draws[3]).mean() > 0.5
    shifted = c.sample_actions(pi, seed=99, step=3, game_id_offset=7).cpu().numpy()
    assert np.array_equal(shifted[:-7], draws[3][7:])
    c.close()


@pytest.mark.parametrize("size,games,sims,layers", [(5, 7, 12, 3), (9, 33, 20, 3), (13, 5, 6, 2), (19, 3, 5, 10)])
def test_fused_step_equals_separate_launches(size, games, sims, layers):
    """goai_mcts_run with everything between two trunk launches in one kernel (k_post_forward: heads, backup, the next
    step's select, first convolution) against the same search as separate launches (GOAI_NO_STEP_FUSION=1): visit
    counts, priors, pi and the search counters bit for bit, odd game counts and every board size included."""
    import os
    from go_ai.engine import Context
    roots = random_positions(size, games, seed=3 + size, max_moves=size * size // 2, canonical=True).astype(np.uint8)
    sd = nn_ref.make_state_dict("ac", size, 5, layers=layers)
    outs = []
    for no_fusion in ("0", "1"):
        os.environ["GOAI_NO_STEP_FUSION"] = no_fusion
        try:
            c = Context(size, 0)
            c.nn_init(Context.NET_AC, layers, games)
            c.load_weights(sd)
            c.mcts_init(games, sims)
            c.mcts_reset(roots)
            c.mcts_run(sims)
            visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
            outs.append((visits, prior, root_visits, c.mcts_policy(1.0).cpu().numpy(), c.mcts_counters()))
            c.close()
        finally:
            os.environ.pop("GOAI_NO_STEP_FUSION", None)
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert np.array_equal(a, b)
    assert outs[0][4] == outs[1][4] and outs[0][4]["errors"] == 0
    assert int(outs[0][2].min()) == sims + 1


@pytest.mark.parametrize("games,sims,layers,dtype", [(64, 20, 3, "f16x3"), (8, 81, 3, "f16x3"), (1024, 12, 3, "f16x3"), (36, 9, 2, "f16"),
                                                     (30, 9, 3, "f16x3")])
def test_steps_in_one_launch_equal_separate_launches(games, sims, layers, dtype):
    """goai_mcts_run's lock-step loop as ONE launch (k_search_steps: a workgroup owns four games for all steps, trunk passes
    and the search's work between them in one kernel) against the same search with one launch per trunk pass
    (the default; the one-launch form is opt-in, GOAI_STEPS_KERNEL=1: it measured slower): visit counts, priors, pi and the
    search counters bit for bit; a game count that is not a multiple of four, and the reduced-precision tier (F(2x2,3x3) on
    9x9), keep the separate launches."""
    import os
    from go_ai.engine import Context
    roots = random_positions(9, games, seed=11 + games, max_moves=40, canonical=True).astype(np.uint8)
    sd = nn_ref.make_state_dict("ac", 9, 7, layers=layers)
    outs, how = [], []
    for on in ("1", "0"):
        os.environ["GOAI_STEPS_KERNEL"] = on
        try:
            c = Context(9, 0)
            c.nn_init(Context.NET_AC, layers, games, dtype=Context.DTYPE_F16X3 if dtype == "f16x3" else Context.DTYPE_F16)
            c.load_weights(sd)
            c.mcts_init(games, sims)
            c.mcts_reset(roots)
            c.mcts_run(sims)
            how.append(c.mcts_steps_in_one_launch())
            visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
            outs.append((visits, prior, root_visits, c.mcts_policy(1.0).cpu().numpy(), c.mcts_counters()))
            c.close()
        finally:
            os.environ.pop("GOAI_STEPS_KERNEL", None)
    assert how == ([sims + 1, 0] if (dtype == "f16x3" and games % 4 == 0) else [0, 0]), how
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert np.array_equal(a, b)
    assert outs[0][4] == outs[1][4] and outs[0][4]["errors"] == 0
    assert int(outs[0][2].min()) == sims + 1
