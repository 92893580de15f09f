"""BASELINE.json's full bench configuration (configs[1]: 9x9 actor-critic, mcts = 81, 1024 concurrent
games on one GPU) through size-independent properties: the oracle cannot replay 84 K leaf
evaluations in test time, so this checks what must hold whatever the network says --
visit-count conservation (mct.py:50-62), priors and policies supported on legal moves only and
normalised (actorcritic.py:30-44), legality of every sampled action on the CPU oracle's rules,
run-to-run determinism, independence from how the games are sharded over contexts, and a sample
of the games replayed leaf by leaf against the sequential oracle fed with the GPU's own network."""
import numpy as np
import pytest
import torch

from oracle import mcts_ref
from oracle.gogame import GoEnv, GoGame

pytestmark = pytest.mark.gpu

SIZE, GAMES, SIMS = 9, 1024, 81
A = SIZE * SIZE + 1


def _search(n_games, offset, roots, max_batch=GAMES, sd=None):
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    c = Context(SIZE)
    c.nn_init(Context.NET_AC, 3, max_batch)
    c.load_weights(sd if sd is not None else random_state_dict("ac", SIZE, 0, 3))
    c.mcts_init(n_games, SIMS)
    c.mcts_reset(roots)
    c.mcts_run(SIMS)
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    pi = c.mcts_policy(1.0)
    acts = c.sample_actions(pi, 1234, 0, offset).cpu().numpy()
    cnt = c.mcts_counters()
    algo = c.nn_conv_algo()
    c.close()
    return visits, prior, root_visits, pi.cpu().numpy(), acts, cnt, algo


@pytest.fixture(scope="module")
def full():
    from go_ai.game import BatchedSelfPlay
    sp = BatchedSelfPlay(SIZE, GAMES, SIMS, seed=1234)
    sp.randomize_starts(40)                       # the bench's start positions
    roots = sp.ctx.mcts_roots()[0].clone()
    for c, _, _ in sp.groups:
        c.close()
    return roots, _search(GAMES, 0, roots)


def test_full_config_search_invariants(full):
    from go_ai.engine import Context
    roots, (visits, prior, root_visits, pi, acts, cnt, algo) = full
    assert algo == Context.CONV_WINOGRAD_F3_PAIRED    # 1024 boards of 9x9 = 4 per CU: the bench's kernel
    assert cnt["errors"] == 0 and cnt["leaf_evals"] == GAMES * (SIMS + 1)
    st = roots.cpu().numpy()
    valid = np.concatenate([1 - st[:, 3].reshape(GAMES, -1), np.ones((GAMES, 1), np.uint8)], axis=1).astype(bool)
    assert (root_visits == SIMS + 1).all()                     # the first step "doesn't count" (mct.py:55-56)
    assert (visits.sum(axis=1) == SIMS).all() and (visits >= 0).all()
    assert (visits[~valid] == 0).all() and (prior[~valid] == 0).all() and (pi[~valid] == 0).all()
    assert np.allclose(prior.sum(axis=1), 1, atol=1e-6) and np.allclose(pi.sum(axis=1), 1, atol=1e-12)
    # ActorCritic.__call__ at temp 1: 0.98 * visits / sum + 0.02 * uniform over valid (actorcritic.py:30-40)
    want = 0.98 * (visits / visits.sum(axis=1, keepdims=True)) + 0.02 * valid / valid.sum(axis=1, keepdims=True)
    assert np.allclose(pi, want, rtol=1e-12, atol=1e-15)
    assert valid[np.arange(GAMES), acts].all()
    for g in range(0, GAMES, 64):                              # the oracle agrees that the move is legal
        assert GoGame.valid_moves(st[g].astype(np.float64))[acts[g]] == 1
    assert len({tuple(v) for v in visits}) > GAMES // 2        # the trees really differ


def test_full_config_is_deterministic_and_shard_invariant(full):
    roots, (visits, prior, _, pi, acts, _, _) = full
    again = _search(GAMES, 0, roots)
    assert np.array_equal(again[0], visits) and np.array_equal(again[3], pi) and np.array_equal(again[4], acts)
    # two shards of 512 games (what two GPUs would own), same network batch capacity -> same kernel:
    # every game's search and sampled action is bit-identical to the 1024-game run
    for lo in (0, 512):
        v, p, _, q, a, _, _ = _search(512, lo, roots[lo:lo + 512])
        assert np.array_equal(v, visits[lo:lo + 512]) and np.array_equal(p, prior[lo:lo + 512])
        assert np.array_equal(q, pi[lo:lo + 512]) and np.array_equal(a, acts[lo:lo + 512])


def test_full_config_sampled_games_match_sequential_oracle(full):
    """Eight of the 1024 games, each replayed by the sequential oracle search (math='det') that is
    fed the GPU network's own logits for the states it asks about: visit counts bit for bit."""
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    roots, (visits, prior, _, pi, _, _, _) = full
    c = Context(SIZE)
    c.nn_init(Context.NET_AC, 3, GAMES)           # same kernel as the search used
    c.load_weights(random_state_dict("ac", SIZE, 0, 3))
    st = roots.cpu().numpy()

    def net(states):
        states = np.asarray(states)
        p, v = c.nn_forward(states.astype(np.uint8), mode=c.NN_NUMPY_SHIM)
        return p.cpu().numpy(), v.cpu().numpy()

    for g in range(5, GAMES, 128):
        env = GoEnv(SIZE)
        env.state = st[g].astype(np.float64)
        opi, counts, _ = mcts_ref.actor_critic_policy(env, net, SIMS, 1.0, math="det")
        assert np.array_equal(counts.astype(np.int32), visits[g]), f"game {g}"
        assert np.array_equal(opi, pi[g])
    c.close()
