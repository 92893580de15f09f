"""The drop-in seam of /root/reference/go_ai/policies/baselines.py: the greedy value function
(areas by the HIP scoring kernel), the Random policy, `create_policy` and `load_weights`."""
import numpy as np
import torch

from ..env import context
from . import Policy
from .actorcritic import ActorCritic
from .value import Value


def greedy_val_func(states):
    # baselines.py:11-30
    if len(states) <= 0:
        return np.array([])
    states = np.asarray(states)
    board_area = states.shape[-1] ** 2
    b, w, ended, _ = [t.cpu().numpy().astype(np.float64) for t in context(states.shape[-1]).score(states)]
    vals = np.where(ended > 0, 100.0 * np.sign(b - w), (b - w) / board_area)
    return vals[:, np.newaxis]


def smart_greedy_val_func(states):
    # baselines.py:33-55: (6 * area difference + liberty difference) / 7, +-1 / 0 once the game is over
    if len(states) <= 0:
        return np.array([])
    states = np.asarray(states)
    board_area = states.shape[-1] ** 2
    c = context(states.shape[-1])
    b, w, ended, _ = [t.cpu().numpy().astype(np.float64) for t in c.score(states)]
    bl, wl = [t.cpu().numpy().astype(np.float64) for t in c.liberties(states)]
    area_val = (b - w) / board_area
    libs_val = (bl - wl) / board_area
    vals = np.where(ended > 0, np.sign(b - w), (6 * area_val + libs_val) / 7)
    return vals[:, np.newaxis]


class Random(Policy):
    def __init__(self):
        super().__init__('Random')

    def __call__(self, go_env, **kwargs):
        valid_moves = go_env.valid_moves()
        return valid_moves / np.sum(valid_moves)


def create_policy(args, name=''):
    # baselines.py:129-157 ('attn' and 'human' are outside this build's scope)
    from ..models import ac_net, val_net
    model, size = args.model, args.size
    if model == 'val':
        net = val_net.ValueNet(size)
        pi = Value(name, net, args)
    elif model == 'ac':
        net = ac_net.ActorCriticNet(size)
        pi = ActorCritic(name, net, args)
    elif model == 'rand':
        return Random(), None
    elif model == 'greedy':
        return Value('Greedy', greedy_val_func, args), greedy_val_func
    elif model == 'smartgreedy':          # tests/test_pi_v_pi.py:46 asks for it; baselines.py:125 builds it
        return Value('Smart Greedy', smart_greedy_val_func, args), smart_greedy_val_func
    else:
        raise Exception("Unknown model argument", model)
    load_weights(args, net)
    return pi, net


def load_weights(args, net):
    # baselines.py:160-172; plain state_dict files, tensors only
    path = None
    if getattr(args, 'baseline', False):
        path = args.basepath
    elif getattr(args, 'latest_checkpoint', False):
        path = args.checkpath
    elif getattr(args, 'customdir', '') != '':
        path = args.custompath
    if path is not None:
        net.load_state_dict(torch.load(path, map_location='cpu', weights_only=True))
