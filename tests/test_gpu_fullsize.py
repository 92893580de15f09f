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


# =================================================================================================
# BASELINE.json configs[2] (C3): 9x9 Q-value model, 4096 games, mcts = 8 -- every leaf expansion makes
# all valid children (k_q_expand) and sends leaf + children through the value net in one compacted
# batch -- and configs[3] (C4): 13x13 actor-critic, 8192 games per GPU, mcts = 81.  Same
# size-independent properties as C2 above, at the configurations' own sizes.
# =================================================================================================
C3 = dict(size=9, games=4096, sims=8, nn_batch=32768, random_moves=40)
C4 = dict(size=13, games=8192, sims=81, random_moves=80)


def _roots_of(size, games, sims, random_moves, search):
    from go_ai.game import BatchedSelfPlay
    sp = BatchedSelfPlay(size, games, sims, seed=1234, search=search)
    sp.randomize_starts(random_moves)
    roots = sp.ctx.mcts_roots()[0].clone()
    for c, _, _ in sp.groups:
        c.close()
    return roots


def _valid_of(roots_np):
    n = roots_np.shape[0]
    return np.concatenate([1 - roots_np[:, 3].reshape(n, -1), np.ones((n, 1), np.uint8)], axis=1).astype(bool)


def _search_q(n_games, offset, roots):
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    c = Context(C3["size"])
    c.nn_init(Context.NET_VAL, 3, C3["nn_batch"])
    c.load_weights(random_state_dict("val", C3["size"], 0, 3))
    c.mcts_init(n_games, C3["sims"], Context.SEARCH_Q)
    c.mcts_reset(roots)
    c.mcts_run(C3["sims"])
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    inv_vals = c.mcts_child_values().cpu().numpy()
    pi = c.mcts_policy_value(1.0)
    acts = c.sample_actions(pi, 1234, 0, offset).cpu().numpy()
    cnt = c.mcts_counters()
    algo = c.nn_conv_algo()
    c.close()
    return visits, prior, root_visits, pi.cpu().numpy(), acts, cnt, algo, inv_vals


@pytest.fixture(scope="module")
def full_c3():
    roots = _roots_of(C3["size"], C3["games"], C3["sims"], C3["random_moves"], "q")
    return roots, _search_q(C3["games"], 0, roots)


def test_c3_q_search_invariants(full_c3):
    from go_ai.engine import Context
    roots, (visits, prior, root_visits, pi, acts, cnt, algo, inv_vals) = full_c3
    G, S = C3["games"], C3["sims"]
    assert algo == Context.CONV_WINOGRAD_F3            # 32768-board slices: more than 4 boards per CU
    assert cnt["errors"] == 0 and cnt["leaf_evals"] == G * (S + 1)
    st = roots.cpu().numpy()
    valid = _valid_of(st)
    assert (root_visits == S + 1).all()
    assert (visits.sum(axis=1) == S).all() and (visits >= 0).all()
    assert (visits[~valid] == 0).all() and (prior[~valid] == 0).all() and (pi[~valid] == 0).all()
    assert (inv_vals[~valid] == 0).all()
    assert np.allclose(prior.sum(axis=1), 1, atol=1e-9) and np.allclose(pi.sum(axis=1), 1, atol=1e-12)
    # Value.__call__ at temp 1 (value.py:30-37): temp_norm of the visit counts = counts / sum
    assert np.allclose(pi, visits / visits.sum(axis=1, keepdims=True), rtol=1e-12, atol=1e-15)
    # the root's prior is softmax(-child values) over the valid moves (tree.py:152-157)
    e = np.where(valid, np.exp(inv_vals - inv_vals.max(axis=1, keepdims=True, where=valid, initial=-np.inf)), 0.0)
    assert np.allclose(prior, e / e.sum(axis=1, keepdims=True), rtol=1e-9, atol=1e-300)
    # every expansion sent 1 + #valid states through the value net (mct.py:30,44-46): the root's share is known
    assert cnt["forwards"] >= G * (S + 1) + valid.sum() and cnt["transitions"] >= valid.sum()
    assert valid[np.arange(G), acts].all()
    for g in range(0, G, 256):
        assert GoGame.valid_moves(st[g].astype(np.float64))[acts[g]] == 1
    assert len({tuple(v) for v in visits}) > G // 4


def test_c3_is_deterministic_and_shard_invariant(full_c3):
    roots, (visits, prior, _, pi, acts, cnt, _, inv_vals) = full_c3
    again = _search_q(C3["games"], 0, roots)
    assert np.array_equal(again[0], visits) and np.array_equal(again[3], pi) and np.array_equal(again[4], acts)
    assert again[5] == cnt
    half = C3["games"] // 2
    for lo in (0, half):                     # what each of two GPUs would own
        v, p, _, q, a, _, _, iv = _search_q(half, lo, roots[lo:lo + half])
        assert np.array_equal(v, visits[lo:lo + half]) and np.array_equal(p, prior[lo:lo + half])
        assert np.array_equal(q, pi[lo:lo + half]) and np.array_equal(a, acts[lo:lo + half])
        assert np.array_equal(iv, inv_vals[lo:lo + half])


def test_c3_sampled_games_match_sequential_oracle(full_c3):
    """Four of the 4096 games replayed by the oracle's sequential Q-value search (math='det') whose
    critic is the GPU value net itself (same context shape, hence the same kernel)."""
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    roots, (visits, prior, _, pi, _, _, _, inv_vals) = full_c3
    c = Context(C3["size"])
    c.nn_init(Context.NET_VAL, 3, C3["nn_batch"])
    c.load_weights(random_state_dict("val", C3["size"], 0, 3))
    st = roots.cpu().numpy()

    def critic(states):
        return c.nn_forward(np.asarray(states).astype(np.uint8), mode=c.NN_NUMPY_SHIM, want_pi=False)[1].cpu().numpy()

    for g in range(7, C3["games"], 1024):
        env = GoEnv(C3["size"])
        env.state = st[g].astype(np.float64)
        opi, root = mcts_ref.value_policy(env, critic, C3["sims"], 1.0, math="det")
        assert np.array_equal(root.get_visit_counts().astype(np.int32), visits[g]), f"game {g}"
        assert np.array_equal(root.prior_pi, prior[g]) and np.array_equal(root.inverted_children_values(), inv_vals[g])
        assert np.array_equal(opi, pi[g])
    c.close()


def _search_c4(n_games, offset, roots):
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    c = Context(C4["size"])
    c.nn_init(Context.NET_AC, 3, C4["games"])
    c.load_weights(random_state_dict("ac", C4["size"], 0, 3))
    c.mcts_init(n_games, C4["sims"])
    c.mcts_reset(roots)
    c.mcts_run(C4["sims"])
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    pi = c.mcts_policy(1.0)
    acts = c.sample_actions(pi, 1234, 0, offset).cpu().numpy()
    cnt = c.mcts_counters()
    algo = c.nn_conv_algo()
    c.close()
    return visits, prior, root_visits, pi.cpu().numpy(), acts, cnt, algo


@pytest.fixture(scope="module")
def full_c4():
    roots = _roots_of(C4["size"], C4["games"], C4["sims"], C4["random_moves"], "ac")
    return roots, _search_c4(C4["games"], 0, roots)


def test_c4_search_invariants(full_c4):
    from go_ai.engine import Context
    roots, (visits, prior, root_visits, pi, acts, cnt, algo) = full_c4
    G, S = C4["games"], C4["sims"]
    assert algo == Context.CONV_WINOGRAD               # F(2x2,3x3) on 13x13
    assert cnt["errors"] == 0 and cnt["leaf_evals"] == G * (S + 1)
    st = roots.cpu().numpy()
    valid = _valid_of(st)
    assert (root_visits == S + 1).all()
    assert (visits.sum(axis=1) == S).all() and (visits >= 0).all()
    assert (visits[~valid] == 0).all() and (prior[~valid] == 0).all() and (pi[~valid] == 0).all()
    assert np.allclose(prior.sum(axis=1), 1, atol=1e-6) and np.allclose(pi.sum(axis=1), 1, atol=1e-12)
    want = 0.98 * (visits / visits.sum(axis=1, keepdims=True)) + 0.02 * valid / valid.sum(axis=1, keepdims=True)
    assert np.allclose(pi, want, rtol=1e-12, atol=1e-15)
    assert valid[np.arange(G), acts].all()
    for g in range(0, G, 512):
        assert GoGame.valid_moves(st[g].astype(np.float64))[acts[g]] == 1
    assert len({tuple(v) for v in visits}) > G // 2


def test_c4_is_deterministic_and_shard_invariant(full_c4):
    roots, (visits, prior, _, pi, acts, _, _) = full_c4
    half = C4["games"] // 2
    for lo in (0, half):
        v, p, _, q, a, _, _ = _search_c4(half, lo, roots[lo:lo + half])
        assert np.array_equal(v, visits[lo:lo + half]) and np.array_equal(p, prior[lo:lo + half])
        assert np.array_equal(q, pi[lo:lo + half]) and np.array_equal(a, acts[lo:lo + half])


def test_c4_sampled_games_match_sequential_oracle(full_c4):
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    roots, (visits, prior, _, pi, _, _, _) = full_c4
    c = Context(C4["size"])
    c.nn_init(Context.NET_AC, 3, C4["games"])
    c.load_weights(random_state_dict("ac", C4["size"], 0, 3))
    st = roots.cpu().numpy()

    def net(states):
        p, v = c.nn_forward(np.asarray(states).astype(np.uint8), mode=c.NN_NUMPY_SHIM)
        return p.cpu().numpy(), v.cpu().numpy()

    for g in range(11, C4["games"], 2048):
        env = GoEnv(C4["size"])
        env.state = st[g].astype(np.float64)
        opi, counts, _ = mcts_ref.actor_critic_policy(env, net, C4["sims"], 1.0, math="det")
        assert np.array_equal(counts.astype(np.int32), visits[g]), f"game {g}"
        assert np.array_equal(opi, pi[g])
    c.close()


# =================================================================================================
# The arithmetic bench.py runs by default: GOAI_DTYPE_F16X3 (Winograd products on the fp16 matrix cores, two-way
# split operands scaled per board), and round 2's earlier GOAI_DTYPE_BF16X6 (bf16, three-way split).  Same properties
# at C2's and C4's full sizes, and sampled games replayed by the sequential oracle fed the network's own logits: the
# search is bit-exact whatever produced the logits; halves of the batch searched alone give bit-identical results
# (a board's activation scale depends on that board alone).
# =================================================================================================
def _search_x6(size, games, sims, roots, offset=0, max_batch=None, dtype="f16x3", layers=3):
    from go_ai.engine import Context
    from go_ai.models import random_state_dict
    c = Context(size)
    c.nn_init(Context.NET_AC, layers, max_batch or games, dtype=Context.DTYPE_F16X3 if dtype == "f16x3" else Context.DTYPE_BF16X6)
    c.load_weights(random_state_dict("ac", size, 0, layers))
    c.mcts_init(games, sims)
    c.mcts_reset(roots)
    c.mcts_run(sims)
    visits, prior, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    pi = c.mcts_policy(1.0)
    acts = c.sample_actions(pi, 1234, 0, offset).cpu().numpy()
    cnt = c.mcts_counters()
    return c, visits, prior, root_visits, pi.cpu().numpy(), acts, cnt


@pytest.mark.parametrize("size,games,sims,random_moves,step,dtype,layers", [
    (9, 1024, 81, 40, 128, "f16x3", 3), (9, 1024, 81, 40, 256, "bf16x6", 3), (13, 8192, 81, 80, 2048, "f16x3", 3),
    # BASELINE.json configs[4] at its full size (what `bench.py --config c5` runs): 19x19, 362 actions, the deeper CNN
    # (10 blocks), 4096 concurrent games = a 4096-leaf batch per step, mid-game positions with large groups
    (19, 4096, 81, 250, 1500, "f16x3", 10)])
def test_split_dtype_full_config_search(size, games, sims, random_moves, step, dtype, layers):
    from go_ai.engine import Context
    roots = _roots_of(size, games, sims, random_moves, "ac")
    c, visits, prior, root_visits, pi, acts, cnt = _search_x6(size, games, sims, roots, dtype=dtype, layers=layers)
    assert c.nn_conv_algo() == {("f16x3", True): Context.CONV_WINOGRAD_F3_F16X3, ("f16x3", False): Context.CONV_WINOGRAD_F16X3,
                                ("bf16x6", True): Context.CONV_WINOGRAD_F3_BF16X6,
                                ("bf16x6", False): Context.CONV_WINOGRAD_BF16X6}[(dtype, size == 9)]
    assert cnt["errors"] == 0 and cnt["leaf_evals"] == games * (sims + 1)
    st = roots.cpu().numpy()
    valid = _valid_of(st)
    assert (root_visits == sims + 1).all() and (visits.sum(axis=1) == sims).all() and (visits >= 0).all()
    assert (visits[~valid] == 0).all() and (prior[~valid] == 0).all() and (pi[~valid] == 0).all()
    assert np.allclose(prior.sum(axis=1), 1, atol=1e-6) and np.allclose(pi.sum(axis=1), 1, atol=1e-12)
    want = 0.98 * (visits / visits.sum(axis=1, keepdims=True)) + 0.02 * valid / valid.sum(axis=1, keepdims=True)
    assert np.allclose(pi, want, rtol=1e-12, atol=1e-15)
    assert valid[np.arange(games), acts].all()

    def net(states):
        p, v = c.nn_forward(np.asarray(states).astype(np.uint8), mode=c.NN_NUMPY_SHIM)
        return p.cpu().numpy(), v.cpu().numpy()

    for g in range(3, games, step):            # a sample of the games, replayed sequentially on the oracle
        env = GoEnv(size)
        env.state = st[g].astype(np.float64)
        assert GoGame.valid_moves(st[g].astype(np.float64))[acts[g]] == 1        # legal on the oracle's rules too
        opi, counts, _ = mcts_ref.actor_critic_policy(env, net, sims, 1.0, math="det")
        assert np.array_equal(counts.astype(np.int32), visits[g]), f"game {g}"
        assert np.array_equal(opi, pi[g])
    c.close()
    # two shards of half the games (what two GPUs would own): bit-identical searches and draws
    half = games // 2
    for lo in (0, half):
        c2, v, p, _, q, a, _ = _search_x6(size, half, sims, roots[lo:lo + half], offset=lo, max_batch=games, dtype=dtype,
                                          layers=layers)
        c2.close()
        assert np.array_equal(v, visits[lo:lo + half]) and np.array_equal(q, pi[lo:lo + half]) and np.array_equal(a, acts[lo:lo + half])
