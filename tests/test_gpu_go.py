"""GPU parity of the Go-rule kernels against the CPU oracle, through the C ABI (bit-exact)."""
import numpy as np
import pytest

from helpers import from_ascii, random_positions
from oracle.gogame import GoGame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs():
    from go_ai.engine import Context
    cache = {}

    def get(size):
        if size not in cache:
            cache[size] = Context(size)
        return cache[size]
    yield get
    for c in cache.values():
        c.close()


# The stateless entry points have three implementations (csrc/go_kernels.hip): one board per lane with the byte
# records staged through LDS (the default for 16-byte aligned arrays), one board per lane with wave-wide plane
# accesses (any alignment), one board per wavefront (GOAI_GO_KERNELS=wave; also what the search kernels use).
IMPLS = ("staged", "lanes", "wave")


class _impl:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        import os
        self.old = os.environ.get("GOAI_GO_KERNELS")
        if self.name == "wave":
            os.environ["GOAI_GO_KERNELS"] = "wave"
        else:
            os.environ.pop("GOAI_GO_KERNELS", None)
        return self

    def __exit__(self, *exc):
        import os
        os.environ.pop("GOAI_GO_KERNELS", None)
        if self.old is not None:
            os.environ["GOAI_GO_KERNELS"] = self.old

    def states(self, states):
        """The states as a device tensor; for "lanes" at an address that is not 16-byte aligned."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(states) != 0).astype(np.uint8)).cuda()
        if self.name != "lanes":
            return t
        pad = torch.zeros((1,) + tuple(t.shape[1:]), dtype=torch.uint8, device="cuda")
        shifted = torch.cat([pad, t])[1:]
        assert shifted.data_ptr() % 16 != 0 and shifted.is_contiguous()
        return shifted


def _random_actions(states, rng, illegal_frac=0.1):
    acts = []
    for st in states:
        valid = GoGame.valid_moves(st)
        if rng.random() < illegal_frac and valid.min() == 0:
            acts.append(int(rng.choice(np.flatnonzero(valid == 0))))
        else:
            acts.append(int(rng.choice(np.flatnonzero(valid))))
    return np.asarray(acts, dtype=np.int32)


@pytest.mark.parametrize("size,n", [(5, 600), (7, 300), (9, 1500), (13, 400), (19, 200)])
def test_transition_matches_oracle(ctxs, size, n):
    rng = np.random.default_rng(size)
    states = random_positions(size, n, seed=size, max_moves=int(1.3 * size * size), canonical=False)
    acts = _random_actions(states, rng)
    for canonical in (True, False):
        want, wstatus = GoGame.next_states_u8(states, acts, canonical=canonical)
        for name in IMPLS:
            with _impl(name) as impl:
                got, gstatus = ctxs(size).transition(impl.states(states), acts, canonical=canonical)
                got, gstatus = got.cpu().numpy(), gstatus.cpu().numpy()
            assert np.array_equal(gstatus != 0, wstatus != 0), name
            assert np.array_equal(got, want), name


@pytest.mark.parametrize("size,n", [(9, 700), (19, 130)])
def test_any_nonzero_byte_is_a_stone(ctxs, size, n):
    """The boundary's uint8 planes: nonzero = set (np.asarray(state) != 0 is what the rules see).  The one-board-per-lane
    kernels gather 0 / 1 bytes with dot products and redo a record that holds other values the general way: same
    results whether a stone is written as 1, 255 or anything between, in a batch where only some records are odd."""
    import torch
    rng = np.random.default_rng(size)
    states = random_positions(size, n, seed=50 + size, max_moves=int(1.2 * size * size), canonical=False)
    acts = _random_actions(states, rng)
    odd = states.copy()
    scale = rng.integers(1, 256, size=odd.shape).astype(np.uint8)
    scale[::3] = 1                                              # every third record stays 0 / 1
    odd = (odd.astype(np.uint16) * scale).astype(np.uint8)
    assert np.array_equal(odd != 0, states != 0) and odd.max() > 1
    c = ctxs(size)
    for canonical in (True, False):
        want, wstatus = c.transition(states, acts, canonical=canonical)
        got, gstatus = c.transition(torch.from_numpy(odd).cuda(), acts, canonical=canonical)
        assert torch.equal(got, want) and torch.equal(gstatus, wstatus)
    for a, b in zip(c.score(states), c.score(torch.from_numpy(odd).cuda())):
        assert torch.equal(a, b)


@pytest.mark.parametrize("size,n", [(5, 64), (9, 96), (13, 16), (19, 6)])
def test_children_valid_win_match_oracle(ctxs, size, n):
    states = random_positions(size, n, seed=10 + size, max_moves=int(1.2 * size * size), canonical=True,
                              pass_prob=0.15)
    kids, valid, win = [t.cpu().numpy() for t in ctxs(size).children(states, canonical=True)]
    with _impl("wave") as impl:                                # the one-board-per-wavefront kernel gives the same bytes
        for a, b in zip((kids, valid, win), ctxs(size).children(impl.states(states), canonical=True)):
            assert np.array_equal(a, b.cpu().numpy())
    for i, st in enumerate(states):
        wk, wv = GoGame.children_u8(st, canonical=True)
        assert np.array_equal(valid[i], wv)
        assert np.array_equal(valid[i].astype(np.float64), GoGame.valid_moves(st))
        assert np.array_equal(kids[i], wk)
        wwin = [GoGame.winning(c) if GoGame.game_ended(c) else 0 for c in wk]
        assert np.array_equal(win[i], np.asarray(wwin, dtype=np.int8))


@pytest.mark.parametrize("size,n", [(5, 300), (9, 600), (19, 100)])
def test_score_matches_oracle(ctxs, size, n):
    states = random_positions(size, n, seed=20 + size, max_moves=2 * size * size, canonical=False, pass_prob=0.05)
    b, w, ended, winner = [t.cpu().numpy() for t in ctxs(size).score(states)]
    for i, st in enumerate(states):
        assert (b[i], w[i]) == GoGame.areas(st)
        assert ended[i] == GoGame.game_ended(st)
        assert winner[i] == GoGame.winning(st)
    for name in ("lanes", "wave"):
        with _impl(name) as impl:
            for a, t in zip((b, w, ended, winner), ctxs(size).score(impl.states(states))):
                assert np.array_equal(a, t.cpu().numpy()), name
            lb, lw = [t.cpu().numpy() for t in ctxs(size).liberties(impl.states(states))]
        lb0, lw0 = [t.cpu().numpy() for t in ctxs(size).liberties(states)]
        assert np.array_equal(lb, lb0) and np.array_equal(lw, lw0), name


def test_edge_cases(ctxs):
    c = ctxs(5)
    # n = 0 is a no-op
    out, status = c.transition(np.zeros((0, 6, 5, 5), np.uint8), np.zeros(0, np.int32))
    assert out.shape[0] == 0
    # ko, capture and pass-pass known answers (same positions as tests/test_oracle_go.py)
    st = from_ascii([".XO..", "XO.O.", ".XO..", ".....", "....."], to_move=0).astype(np.uint8)
    out, status = c.transition(st[None], np.asarray([7], np.int32), canonical=False)
    out = out.cpu().numpy()[0]
    assert status.item() == 0 and out[3, 1, 1] == 1 and out[1, 1, 1] == 0 and out[2].min() == 1
    p1, _ = c.transition(st[None], np.asarray([25], np.int32), canonical=True)
    p2, _ = c.transition(p1, np.asarray([25], np.int32), canonical=True)
    assert p1.cpu().numpy()[0, 4].min() == 1 and p1.cpu().numpy()[0, 5].max() == 0
    assert p2.cpu().numpy()[0, 5].min() == 1
    # out-of-range and occupied actions are rejected with a zero state
    out, status = c.transition(st[None].repeat(2, 0), np.asarray([26, 1], np.int32))
    assert (status.cpu().numpy() != 0).all() and out.cpu().numpy().max() == 0
