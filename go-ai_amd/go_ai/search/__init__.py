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
