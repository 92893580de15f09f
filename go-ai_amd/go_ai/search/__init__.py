"""go_ai.search helpers (/root/reference/go_ai/search/__init__.py): value sign flip and the
temperature / greedy normalisations used by the policies.  Host-side fp64, as the reference."""
import numpy as np


def invert_vals(vals):
    return -vals


def _l1_normalize(x):
    # sklearn.preprocessing.normalize(x[None], norm='l1')[0]: divide by sum |x|, zero rows untouched
    norm = np.abs(x).sum()
    return x / norm if norm != 0 else x


def greedy_pi(qvals, valid_moves):
    expq = np.exp(qvals - np.max(qvals))
    expq = expq * valid_moves
    pi = (expq == np.max(expq)).astype(float)
    return _l1_normalize(pi)


def temp_norm(qs, temp, valid_moves):
    if temp <= 0:
        return greedy_pi(qs, valid_moves)
    pi = np.zeros(valid_moves.shape)
    where_valid = np.where(valid_moves)
    pi[where_valid] = _l1_normalize(qs[where_valid])
    return _l1_normalize(pi ** (1 / temp))


def temp_softmax(qvals, temp, valid_moves):
    if temp <= 0:
        return greedy_pi(qvals, valid_moves)
    pi = np.zeros(valid_moves.shape)
    valid_indcs = np.where(valid_moves)
    if qvals.shape == valid_moves.shape:
        qvals = qvals[valid_indcs]
    z = qvals * (1 / temp)
    e = np.exp(z - np.max(z))
    pi[valid_indcs] = e / e.sum()
    return pi


# ---- Q values of all children from a value function (search/__init__.py:13-43) --------------
def vals_to_qs(canonical_childvals, valid_moves):
    qvals = np.zeros(valid_moves.shape)
    qvals[np.where(valid_moves)] = invert_vals(np.asarray(canonical_childvals).flatten())
    return qvals


def qs_from_valfunc(state, val_func, group_map=None):
    """Q(s, a) = -V(child_a) for every valid a, 0 elsewhere, and the valid canonical children in
    ascending action order.  The children come from the all-actions transition kernel."""
    from ..env import GoGame
    canonical_children = GoGame.children(state, group_map, canonical=True)
    child_vals = val_func(np.array(canonical_children))
    valid_moves = GoGame.valid_moves(state)
    return vals_to_qs(child_vals, valid_moves), canonical_children


def batchqs_from_valfunc(states, val_func, group_maps=None):
    """Reference semantics (one val_func call per state), but the children of the whole batch
    are built by ONE kernel launch."""
    from ..env import context
    states = np.asarray(states)
    if len(states) == 0:
        return np.array([]), []
    kids, valid, _ = context(states.shape[-1]).children(states, canonical=True)
    kids, valid = kids.cpu().numpy().astype(np.float64), valid.cpu().numpy().astype(bool)
    batch_qvals, batch_canon_children = [], []
    for k, v in zip(kids, valid):
        children = list(k[v])
        batch_qvals.append(vals_to_qs(val_func(np.array(children)), v.astype(np.float64)))
        batch_canon_children.append(children)
    return np.array(batch_qvals), batch_canon_children


def _l1_normalize_rows(x):
    norm = np.abs(x).sum(axis=1, keepdims=True)
    return x / np.where(norm == 0, 1, norm)


def batch_greedy_pi(batch_qvals, batch_valid_moves):
    expq = np.exp(batch_qvals - np.max(batch_qvals, axis=1, keepdims=True))
    expq = expq * batch_valid_moves
    pi = (expq == np.max(expq, axis=1, keepdims=True)).astype(float)
    return _l1_normalize_rows(pi)


def batch_temperate_pi(batch_qvals, temp, batch_valid_moves):
    if temp <= 0:
        return batch_greedy_pi(batch_qvals, batch_valid_moves)
    z = np.asarray(batch_qvals) * (1 / temp)
    e = np.exp(z - np.max(z, axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)
