"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own code
(/root/reference/go_ai/{search,models,policies}) in the build container.

Run from the repo root:   python tests/golden/make_golden.py

The reference cannot be imported as-is (gym, gym_go and mpi4py are not installed, and it uses
the removed np.float / np.int aliases): this script injects two stub modules -- `gym`, whose
make() hands out the build's CPU Go-rules restatement (oracle/gogame.py) in place of GymGo,
and `mpi4py.MPI` -- and restores the two numpy aliases.  Everything else that runs is the
reference's code: its nn.Modules, RLNet._numpy, tree.Node, mct.mct_search, ActorCritic /
Value policies.  One semantic shim, recorded in SURVEY.md section 8c: priors are widened to
float64 in Node.set_prior_pi, because under NumPy >= 2 `1.5 * np.float32` stays float32 whereas
the reference's NumPy 1.x evaluates tree.py:182 in float64.

The fixtures hold data only (inputs, recorded per-step NN outputs, expected outputs); the
network weights are re-created from a seed by oracle/nn_ref.make_state_dict and guarded by a
sha256 digest stored in each fixture.  Nothing from /root/reference is read at test time.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)

np.float = float   # noqa: removed aliases the reference still uses
np.int = int

from oracle import gogame as og  # noqa: E402
from oracle import nn_ref  # noqa: E402

# ---- stubs for the two absent third-party modules -------------------------------------
gym = types.ModuleType("gym")
gym.make = lambda name, size=0, **kw: og.GoEnv(size, **kw)
sys.modules["gym"] = gym
mpi4py = types.ModuleType("mpi4py")
MPI = types.ModuleType("mpi4py.MPI")
MPI.Intracomm = object
MPI.SUM = "sum"
mpi4py.MPI = MPI
sys.modules["mpi4py"] = mpi4py
sys.modules["mpi4py.MPI"] = MPI

import torch  # noqa: E402
from go_ai import data as ref_data  # noqa: E402
from go_ai.models import ac_net, val_net  # noqa: E402
from go_ai.policies import actorcritic as ref_ac, value as ref_value, baselines as ref_baselines  # noqa: E402
from go_ai.search import mct as ref_mct, tree as ref_tree  # noqa: E402

assert ref_tree.__file__.startswith(REFERENCE), ref_tree.__file__

from helpers import random_positions  # noqa: E402

# ---- the NumPy-1.x promotion shim -------------------------------------------------------
_orig_set_prior = ref_tree.Node.set_prior_pi
_prior_log = []


def _set_prior_pi(self, prior_pi):
    if prior_pi is not None:
        _prior_log.append(np.array(prior_pi, copy=True))
        _orig_set_prior(self, np.asarray(prior_pi, dtype=np.float64))
    else:
        _orig_set_prior(self, None)
        _prior_log.append(np.array(self.prior_pi, copy=True))


ref_tree.Node.set_prior_pi = _set_prior_pi


def ref_net(kind, size, seed):
    sd = nn_ref.make_state_dict(kind, size, seed)
    net = ac_net.ActorCriticNet(size) if kind == "ac" else val_net.ValueNet(size)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("game_head" in k or "num_batches_tracked" in k for k in missing), missing
    net.eval()
    return net, sd


def env_from_state(state):
    env = og.GoEnv(state.shape[-1])
    env.state = np.array(state, dtype=np.float64)
    env.done = bool(og.GoGame.game_ended(env.state))
    return env


def nn_states(size, n, seed):
    """Mid-game positions plus the special cases the shim treats: pass-flag set, game over."""
    sts = list(random_positions(size, n - 4, seed, canonical=True, pass_prob=0.05))
    extra = random_positions(size, 2, seed + 1, min_moves=size, canonical=True, pass_prob=0.0)
    for st in extra:
        p1 = og.GoGame.next_state(st, size * size, canonical=True)
        p2 = og.GoGame.next_state(p1, size * size, canonical=True)
        sts += [p1.astype(np.uint8), p2.astype(np.uint8)]
    return np.asarray(sts, dtype=np.uint8)


def gen_nn(kind, size, seed, n):
    net, sd = ref_net(kind, size, seed)
    states = nn_states(size, n, 1000 + seed)
    out = dict(states=states, seed=seed, size=size, kind=kind, digest=nn_ref.state_dict_digest(sd))
    with torch.no_grad():
        x = torch.tensor(states).type(net.dtype())
        if kind == "ac":
            pi_raw, val_raw = net(x)
            out["pi_raw"] = pi_raw.numpy()
            out["val_raw"] = val_raw.numpy()
            pi_np, val_np = net.create_numpy("actor_critic")(states.astype(np.float64))
            out["pi_np"], out["val_np"] = pi_np, val_np
        else:
            out["val_raw"] = net(x).numpy()
            out["val_np"] = net.create_numpy("critic")(states.astype(np.float64))
    # our torch restatement must reproduce the reference module (same torch, same machine)
    pi_o, val_o = nn_ref.forward(sd, torch.tensor(states), kind)
    assert np.allclose(val_o.numpy(), out["val_raw"], rtol=1e-5, atol=1e-6)
    if kind == "ac":
        assert np.allclose(pi_o.numpy(), out["pi_raw"], rtol=1e-5, atol=1e-6)
    path = os.path.join(HERE, f"nn_{kind}{size}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", v) for k, v in out.items()})


class Recorder:
    """Wraps an actor_critic callable and records every (state, pi_logits, val_logits)."""

    def __init__(self, fn):
        self.fn = fn
        self.states, self.pis, self.vals = [], [], []

    def __call__(self, states):
        pi, val = self.fn(states)
        assert len(states) == 1
        self.states.append(np.asarray(states[0]).astype(np.uint8))
        self.pis.append(np.array(pi[0], dtype=np.float32))
        self.vals.append(np.float32(val.item()))
        return pi, val


def fake_ac(style, rng):
    """Table-free synthetic networks that stress ties, saturation and deep narrow trees."""
    def fn(states):
        b = len(states)
        A = states.shape[-1] ** 2 + 1
        if style == "uniform":          # all-equal priors and values: pure tie-breaking
            pi = np.zeros((b, A), dtype=np.float32)
            val = np.full((b, 1), 0.25, dtype=np.float32)
        elif style == "peaky":          # one dominant move: deep, narrow tree
            pi = (rng.standard_normal((b, A)) * 6).astype(np.float32)
            val = (rng.standard_normal((b, 1)) * 0.3).astype(np.float32)
        elif style == "saturated":      # |val| large: tanh saturates, +-1 sums
            pi = rng.standard_normal((b, A)).astype(np.float32)
            val = (rng.choice([-30.0, -3.0, 0.0, 2.5, 40.0], size=(b, 1))).astype(np.float32)
        else:
            raise ValueError(style)
        pi = pi + ref_data.batch_invalid_values(states)
        pi = pi.astype(np.float32)
        for i, st in enumerate(states):   # same terminal override as RLNet._numpy
            if ref_data.GoGame.game_ended(st):
                val[i] = 100 * ref_data.GoGame.winning(st)
        return pi, val
    return fn


def gen_mcts_ac(tag, size, sims, roots, fn_factory, temp=1.0):
    """Runs the reference's ActorCritic policy (mct_search inside) from each root."""
    cases = []
    for ri, root in enumerate(roots):
        rec = Recorder(fn_factory(ri))
        pol = ref_ac.ActorCritic.__new__(ref_ac.ActorCritic)
        pol.name, pol.temp, pol.pt_model, pol.mcts = "golden", temp, None, sims
        pol.ac_func = rec
        env = env_from_state(root)
        del _prior_log[:]
        pi, qs, rootnode = pol(env, debug=True)
        priors = [p for p in _prior_log if p is not None]
        nsteps = len(rec.states)
        assert nsteps == sims + 1
        # the reference only sets a prior on non-terminal leaves
        term = np.array([og.GoGame.game_ended(s) for s in rec.states], dtype=np.uint8)
        assert len(priors) == int((1 - term).sum()), (len(priors), term.sum())
        prior_arr = np.zeros((nsteps, size * size + 1), dtype=np.float32)
        it = iter(priors)
        for s in range(nsteps):
            if not term[s]:
                p = next(it)
                assert p.dtype == np.float32
                prior_arr[s] = p
        cases.append(dict(root=np.asarray(og.GoGame.canonical_form(root)).astype(np.uint8),
                          leaf_states=np.asarray(rec.states), pi_logits=np.asarray(rec.pis),
                          val_logits=np.asarray(rec.vals, dtype=np.float32), priors=prior_arr, terminal=term,
                          visits=np.asarray(rootnode.get_visit_counts(), dtype=np.int32),
                          root_prior=np.asarray(rootnode.prior_pi, dtype=np.float64),
                          root_visits=np.int32(rootnode.visits), pi=np.asarray(pi, dtype=np.float64)))
    out = {"n_cases": len(cases), "size": size, "sims": sims, "temp": temp}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{i}_{k}"] = v
    path = os.path.join(HERE, f"mcts_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "cases", len(cases), "steps", sims + 1)


class CriticRecorder:
    """Wraps a critic callable and records every call (states, values)."""

    def __init__(self, fn):
        self.fn = fn
        self.calls = []

    def __call__(self, states):
        vals = self.fn(states)
        self.calls.append((np.asarray(states).astype(np.uint8), np.array(vals, dtype=np.float32).reshape(-1)))
        return vals


def fake_critic(style, rng):
    def fn(states):
        b = len(states)
        if style == "flat":             # all-equal values: uniform priors, pure tie-breaking
            val = np.full((b, 1), -0.125, dtype=np.float32)
        elif style == "wide":           # spread-out values: peaked softmax(-val) priors
            val = (rng.standard_normal((b, 1)) * 2.5).astype(np.float32)
        else:
            raise ValueError(style)
        for i, st in enumerate(states):   # same terminal override as RLNet._numpy
            if ref_data.GoGame.game_ended(st):
                val[i] = 100 * ref_data.GoGame.winning(st)
        return val
    return fn


def gen_mcts_q(tag, size, sims, roots, fn_factory, temp=1.0):
    """Runs the reference's Value policy (critic-driven mct_search inside) from each root."""
    A = size * size + 1
    cases = []
    for ri, root in enumerate(roots):
        rec = CriticRecorder(fn_factory(ri))
        args = types.SimpleNamespace(temp=temp, mcts=sims)
        pol = ref_value.Value("golden", rec, args)
        env = env_from_state(root)
        pi, qs, rootnode = pol(env, debug=True)
        leaf_states, leaf_vals, child_vals, term = [], [], [], []
        i = 0
        while i < len(rec.calls):
            st, v = rec.calls[i]
            assert len(st) == 1
            i += 1
            leaf_states.append(st[0])
            leaf_vals.append(v[0])
            ended = bool(og.GoGame.game_ended(st[0]))
            term.append(ended)
            cv = np.full(A, np.nan, dtype=np.float32)
            if not ended:
                kst, kv = rec.calls[i]
                i += 1
                valid = np.flatnonzero(og.GoGame.valid_moves(st[0].astype(np.float64)))
                assert len(kst) == len(valid), (len(kst), len(valid))
                kids = og.GoGame.children(st[0].astype(np.float64), canonical=True, padded=True)
                assert np.array_equal(np.asarray(kids)[valid].astype(np.uint8), kst)
                cv[valid] = kv
            child_vals.append(cv)
        assert len(leaf_states) == sims + 1
        cases.append(dict(root=np.asarray(og.GoGame.canonical_form(root)).astype(np.uint8),
                          leaf_states=np.asarray(leaf_states), leaf_vals=np.asarray(leaf_vals, dtype=np.float32),
                          child_vals=np.asarray(child_vals, dtype=np.float32), terminal=np.asarray(term, dtype=np.uint8),
                          visits=np.asarray(rootnode.get_visit_counts(), dtype=np.int32),
                          root_prior=np.asarray(rootnode.prior_pi, dtype=np.float64),
                          root_visits=np.int32(rootnode.visits),
                          inv_child_vals=np.asarray(rootnode.inverted_children_values(), dtype=np.float64),
                          pi=np.asarray(pi, dtype=np.float64)))
    out = {"n_cases": len(cases), "size": size, "sims": sims, "temp": temp}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{i}_{k}"] = v
    path = os.path.join(HERE, f"mctsq_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "cases", len(cases), "steps", sims + 1)


def gen_policy_kat():
    """/root/reference/tests/test_policy.py:34-43 run for real: greedy Value policy, mcts=0."""
    env = og.GoEnv(5)
    env.reset()
    for a in (0, 1, 2, None):
        env.step(a)
    args = types.SimpleNamespace(temp=1, mcts=0)
    policy = ref_value.Value("Greedy", ref_baselines.greedy_val_func, args)
    pi = policy(env)
    assert pi[-1] == 1 and np.allclose(pi[:-1], 0), pi
    print("reference KAT test_greedy_obvious_move passes on the oracle rules: pi[-1] =", pi[-1])
    np.savez_compressed(os.path.join(HERE, "policy_kat_greedy5.npz"), state=env.state.astype(np.uint8),
                        canonical=env.canonical_state().astype(np.uint8), pi=pi)


def gen_search_helpers():
    """Direct input / output vectors of the reference's policy helpers
    (/root/reference/go_ai/search/__init__.py:9-97): invert_vals, greedy_pi, batch_greedy_pi, temp_softmax,
    temp_norm, batch_temperate_pi at temperatures 0, 0.5, 1 and 2, on random q values with ties, random
    valid masks (the pass always valid) and visit-count-like integer inputs."""
    from go_ai import search as ref_search
    rng = np.random.default_rng(77)
    n, A = 24, 26
    valid = (rng.random((n, A)) < 0.7).astype(np.float64)
    valid[:, -1] = 1
    q = rng.standard_normal((n, A)) * 2
    q[0:4] = np.round(q[0:4])                       # ties
    q[4] = 0.0                                      # all equal
    counts = rng.integers(0, 12, (n, A)).astype(np.float64) * valid
    counts[:, -1] += 1                              # never all zero
    temps = np.array([0.0, 0.5, 1.0, 2.0])
    out = dict(q=q, valid=valid, counts=counts, temps=temps, invert=ref_search.invert_vals(q))
    out["greedy_pi"] = np.stack([ref_search.greedy_pi(q[i], valid[i]) for i in range(n)])
    out["greedy_pi_counts"] = np.stack([ref_search.greedy_pi(counts[i], valid[i]) for i in range(n)])
    out["batch_greedy_pi"] = ref_search.batch_greedy_pi(q.copy(), valid)
    for ti, t in enumerate(temps):
        out[f"temp_softmax_{ti}"] = np.stack([ref_search.temp_softmax(q[i], t, valid[i]) for i in range(n)])
        out[f"temp_norm_exp_{ti}"] = np.stack([ref_search.temp_norm(np.exp(q[i]), t, valid[i]) for i in range(n)])
        out[f"temp_norm_counts_{ti}"] = np.stack([ref_search.temp_norm(counts[i], t, valid[i]) for i in range(n)])
        out[f"batch_temperate_pi_{ti}"] = ref_search.batch_temperate_pi(q.copy(), t, valid)
    path = os.path.join(HERE, "search_helpers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def gen_temps():
    """The reference's policies at temperatures other than 1 (actorcritic.py:33-40, value.py:31-37,
    search/__init__.py:78-87): counts ** (1 / T) for the actor-critic policy; temp_norm -- and, for T <= 0,
    greedy_pi -- for the Value policy, with and without search."""
    net5, _ = ref_net("ac", 5, 11)
    val5, _ = ref_net("val", 5, 21)
    roots5 = [s.astype(np.float64) for s in random_positions(5, 6, 7, max_moves=18, canonical=False)]
    roots5[0] = og.GoGame.get_init_board(5)
    roots5[1] = og.GoGame.next_state(roots5[1], 25, canonical=False)
    for tag, temp in (("t05", 0.5), ("t2", 2.0)):
        gen_mcts_ac(f"ac5_n8_{tag}", 5, 8, roots5, lambda i: net5.create_numpy("actor_critic"), temp=temp)
        gen_mcts_q(f"val5_n12_{tag}", 5, 12, roots5, lambda i: val5.create_numpy("critic"), temp=temp)
        gen_mcts_q(f"val5_n0_{tag}", 5, 0, roots5, lambda i: val5.create_numpy("critic"), temp=temp)
    gen_mcts_q("val5_n12_t0", 5, 12, roots5, lambda i: val5.create_numpy("critic"), temp=0.0)
    gen_mcts_q("val5_n0_t0", 5, 0, roots5, lambda i: val5.create_numpy("critic"), temp=0.0)
    gen_mcts_q("fake_flat5_n30_t0", 5, 30, roots5[:3], lambda i: fake_critic("flat", np.random.default_rng(300 + i)), temp=0.0)


def train_batch(size, B, seed):
    """A synthetic replay batch in the layout train_step receives (data.py:92-105): states, actions, children,
    wins, pi -- positions from seeded random play, one valid action each, a random target policy over the valid moves."""
    rng = np.random.default_rng(seed)
    states = random_positions(size, B, seed, canonical=True, pass_prob=0.05).astype(np.float64)
    A = size * size + 1
    children = np.stack([np.asarray(og.GoGame.children(st, canonical=True, padded=True)) for st in states])
    valid = np.stack([og.GoGame.valid_moves(st) for st in states])
    actions = np.array([int(rng.choice(np.flatnonzero(v))) for v in valid])
    wins = rng.choice([-1, 0, 1], size=B, p=[0.45, 0.1, 0.45])
    pi = rng.random((B, A)) ** 4 * valid
    pi = pi / pi.sum(axis=1, keepdims=True)
    return states, actions, children, wins, pi


def gen_train(kind, size, seed, B, steps=2):
    """The reference's own train_step (ac_net.py:67-86 / val_net.py:29-43) with torch.optim.Adam(lr=1e-3,
    weight_decay=1e-4) (train.py:52) on its real nn.Module, `steps` consecutive batches.  One shim, so that the
    inputs of the step are data and not a draw of numpy's global generator: data.batch_random_symmetries is the
    identity here (the HIP entry point receives next_states after that augmentation anyway).
    Stored: the batches, the metrics, and after every step 24 sampled entries + the L2 norm of every parameter,
    of the gradient the optimiser consumed, and every BatchNorm running statistic in full."""
    net, sd0 = ref_net(kind, size, seed)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    saved_sym = ref_data.batch_random_symmetries
    ref_data.batch_random_symmetries = lambda states: np.asarray(states)
    names = {id(p): n for n, p in net.named_parameters()}
    keys = [k for k in sd0 if not k.endswith(("running_mean", "running_var"))]
    stat_keys = [k for k in sd0 if k.endswith(("running_mean", "running_var"))]
    rng = np.random.default_rng(4242)
    sample_idx = {k: rng.integers(0, sd0[k].numel(), 24) for k in keys}
    out = dict(kind=kind, size=size, seed=seed, steps=steps, digest=nn_ref.state_dict_digest(sd0))
    for k in keys:
        out["idx_" + k] = sample_idx[k]
    try:
        for s in range(steps):
            states, actions, children, wins, pi = train_batch(size, B, 9000 + 10 * seed + s)
            grads = {}
            orig_step = opt.step

            def recording_step(*a, **kw):
                for group in opt.param_groups:
                    for p in group["params"]:
                        if p.grad is not None:
                            grads[names[id(p)]] = p.grad.detach().clone()
                return orig_step(*a, **kw)

            opt.step = recording_step
            metrics = net.train_step(opt, states, actions, None, children, None, wins, pi)
            opt.step = orig_step
            out[f"s{s}_states"] = states.astype(np.uint8)
            out[f"s{s}_next_states"] = children[np.arange(B), actions].astype(np.uint8)
            out[f"s{s}_wins"] = wins.astype(np.int64)
            out[f"s{s}_pi"] = pi.astype(np.float32)
            out[f"s{s}_metrics"] = np.array([np.nan if m is None else m for m in metrics], dtype=np.float64)
            now = net.state_dict()
            for k in keys:
                assert k in grads, k
                g = grads[k].numpy().reshape(-1)
                w = now[k].detach().numpy().reshape(-1)
                out[f"s{s}_grad_{k}"] = np.concatenate([g[sample_idx[k]], [np.linalg.norm(g.astype(np.float64))]])
                out[f"s{s}_param_{k}"] = np.concatenate([w[sample_idx[k]], [np.linalg.norm(w.astype(np.float64))]])
            for k in stat_keys:
                out[f"s{s}_stat_{k}"] = now[k].detach().numpy().copy()
            assert not any(k.startswith("game_head") for k in grads), "game_head takes no part in train_step"
    finally:
        ref_data.batch_random_symmetries = saved_sym
    path = os.path.join(HERE, f"train_{kind}{size}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "metrics", [out[f"s{s}_metrics"] for s in range(steps)])


def gen_long():
    """Searches of more than 128 simulations: nodes on the principal line collect more than 128 values, where
    np.mean's pairwise sum (tree.py:180) recurses over halves split at n/2 - (n/2) % 8."""
    roots5 = [s.astype(np.float64) for s in random_positions(5, 6, 7, max_moves=18, canonical=False)]
    roots5[0] = og.GoGame.get_init_board(5)
    gen_mcts_ac("fake_peaky5_n300", 5, 300, roots5[:3], lambda i: fake_ac("peaky", np.random.default_rng(500 + i)))
    gen_mcts_ac("fake_saturated5_n200", 5, 200, roots5[:2], lambda i: fake_ac("saturated", np.random.default_rng(520 + i)))
    gen_mcts_q("fake_wide5_n150", 5, 150, roots5[:2], lambda i: fake_critic("wide", np.random.default_rng(540 + i)))


def main():
    torch.manual_seed(0)
    if "--only-long" in sys.argv:        # add the > 128-simulation fixtures without rewriting the others
        gen_long()
        return
    if "--only-train" in sys.argv:       # add the training fixtures without rewriting the others
        gen_train("ac", 5, 31, 16)
        gen_train("ac", 9, 32, 32)
        gen_train("val", 9, 33, 32)
        return
    if "--only-temps" in sys.argv:       # add the temperature / helper fixtures without rewriting the others
        gen_search_helpers()
        gen_temps()
        return
    only_q = "--only-q" in sys.argv      # add the Q-search fixtures without rewriting the others
    if only_q:
        global gen_policy_kat, gen_nn, gen_mcts_ac
        gen_policy_kat = lambda *a, **k: None
        gen_nn = lambda *a, **k: None
        gen_mcts_ac = lambda *a, **k: None
    gen_policy_kat()
    gen_nn("ac", 5, 11, 20)
    gen_nn("ac", 9, 12, 24)
    gen_nn("val", 9, 13, 16)

    # MCTS driven by the reference's real network shim (RLNet._numpy, assist on)
    net5, _ = ref_net("ac", 5, 11)
    net9, _ = ref_net("ac", 9, 12)
    roots5 = [s.astype(np.float64) for s in random_positions(5, 6, 7, max_moves=18, canonical=False)]
    roots9 = [s.astype(np.float64) for s in random_positions(9, 5, 8, max_moves=50, canonical=False)]
    roots5[0] = og.GoGame.get_init_board(5)
    roots9[0] = og.GoGame.get_init_board(9)
    # a root right after a pass: the pass child ends the game (assist path, terminal revisits)
    roots5[1] = og.GoGame.next_state(roots5[1], 25, canonical=False)
    gen_mcts_ac("ac5_n8", 5, 8, roots5, lambda i: net5.create_numpy("actor_critic"))
    gen_mcts_ac("ac5_n40", 5, 40, roots5, lambda i: net5.create_numpy("actor_critic"))
    gen_mcts_ac("ac9_n81", 9, 81, roots9, lambda i: net9.create_numpy("actor_critic"))
    for style in ("uniform", "peaky", "saturated"):
        gen_mcts_ac(f"fake_{style}5_n60", 5, 60, roots5[:4],
                    lambda i, style=style: fake_ac(style, np.random.default_rng(100 + i)))
    gen_mcts_ac("fake_peaky9_n120", 9, 120, roots9[:3], lambda i: fake_ac("peaky", np.random.default_rng(200 + i)))
    # late-game 5x5 roots: many terminal leaves
    late = [s.astype(np.float64) for s in random_positions(5, 4, 9, min_moves=22, max_moves=40, canonical=False,
                                                           pass_prob=0.2)]
    gen_mcts_ac("ac5_late_n30", 5, 30, late, lambda i: net5.create_numpy("actor_critic"))

    # Q-value searches (Value policy, critic-driven mct_search): real value nets and fake critics
    val5, _ = ref_net("val", 5, 21)
    val9, _ = ref_net("val", 9, 13)
    gen_mcts_q("val5_n0", 5, 0, roots5, lambda i: val5.create_numpy("critic"))
    gen_mcts_q("val5_n12", 5, 12, roots5, lambda i: val5.create_numpy("critic"))
    gen_mcts_q("val9_n8", 9, 8, roots9, lambda i: val9.create_numpy("critic"))
    gen_mcts_q("accrit5_n6", 5, 6, roots5[:3], lambda i: net5.create_numpy("critic"))
    gen_mcts_q("val5_late_n20", 5, 20, late, lambda i: val5.create_numpy("critic"))
    for style in ("flat", "wide"):
        gen_mcts_q(f"fake_{style}5_n30", 5, 30, roots5[:3],
                   lambda i, style=style: fake_critic(style, np.random.default_rng(300 + i)))
    gen_mcts_q("fake_wide9_n20", 9, 20, roots9[:2], lambda i: fake_critic("wide", np.random.default_rng(400 + i)))
    gen_search_helpers()
    gen_temps()
    gen_long()
    gen_train("ac", 5, 31, 16)
    gen_train("ac", 9, 32, 32)
    gen_train("val", 9, 33, 32)


if __name__ == "__main__":
    main()
