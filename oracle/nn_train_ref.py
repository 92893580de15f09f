"""oracle/nn_train_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain PyTorch fp32 (CPU, autograd) restatement of one optimisation step of the reference's networks:

  ActorCriticNet.train_step   /root/reference/go_ai/models/ac_net.py:67-86
      cl, ca = critic_step(next_states, -wins);  al, aa = actor_step(states, pi);  (2 * cl + al).backward();  optimizer.step()
  ValueNet.train_step         /root/reference/go_ai/models/val_net.py:29-43      (critic only)
  RLNet.critic_step           /root/reference/go_ai/models/__init__.py:115-134    mse(tanh(val_logits), wins)
  RLNet.actor_step            /root/reference/go_ai/models/__init__.py:167-186    cross_entropy(pi_logits, argmax(pi))
  optimizer                   /root/reference/train.py:52   torch.optim.Adam(lr=1e-3, weight_decay=1e-4)
  average_model               /root/reference/go_ai/models/__init__.py:279-282    parameters only (not BN statistics)

The networks run in train() mode (models/__init__.py:212): BatchNorm normalises with the batch's own statistics and
moves its running statistics (momentum 0.1, unbiased variance) -- twice per actor-critic step, once per trunk pass.
The host-side preparation of a step (next_states = children[actions] under a random board symmetry,
ac_net.py:69-71) is the caller's: this restatement, like the HIP entry point it checks, takes next_states as given.

State is a `state_dict` in the reference's key layout (oracle/nn_ref.make_state_dict).  Pinned against the
reference's real nn.Modules + torch.optim.Adam by tests/golden/train_*.npz (tests/golden/make_golden.py).
Only tests/ may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nn_ref

BN_MOMENTUM = 0.1
ADAM_BETAS = (0.9, 0.999)
ADAM_EPS = 1e-8


def param_keys(sd):
    """The keys nn.Module.parameters() would yield (weights and biases; not the BN statistics), in order."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def _bn_train(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=True, momentum=BN_MOMENTUM, eps=nn_ref.BN_EPS)


def trunk_train(sd, x):
    x = F.relu(_bn_train(F.conv2d(x, sd["main.0.weight"], sd["main.0.bias"], padding=1), sd, "main.1"))
    for i in range(nn_ref.count_layers(sd)):
        b = f"main.{3 + i}.main"
        y = F.relu(_bn_train(F.conv2d(x, sd[b + ".0.weight"], sd[b + ".0.bias"], padding=1), sd, b + ".1"))
        y = _bn_train(F.conv2d(y, sd[b + ".3.weight"], sd[b + ".3.bias"], padding=1), sd, b + ".4")
        x = F.relu(y + x)
    return x


def _critic_logits(sd, kind, states):
    head = "crit_head" if kind == "ac" else "convs"
    h = trunk_train(sd, states - 0.5 if kind == "ac" else states)
    v = F.relu(_bn_train(F.conv2d(h, sd[head + ".0.weight"], sd[head + ".0.bias"]), sd, head + ".1")).flatten(1)
    v = F.relu(F.linear(v, sd[head + ".4.weight"], sd[head + ".4.bias"]))
    return F.linear(v, sd[head + ".6.weight"], sd[head + ".6.bias"])


def _actor_logits(sd, states):
    h = trunk_train(sd, states - 0.5)
    a = F.relu(_bn_train(F.conv2d(h, sd["act_head.0.weight"], sd["act_head.0.bias"]), sd, "act_head.1")).flatten(1)
    return F.linear(a, sd["act_head.4.weight"], sd["act_head.4.bias"])


class AdamState:
    """torch.optim.Adam(lr, weight_decay) on a state_dict's parameters, written out (L2 decay added to the gradient)."""

    def __init__(self, sd):
        self.t = 0
        self.m = {k: torch.zeros_like(sd[k]) for k in param_keys(sd)}
        self.v = {k: torch.zeros_like(sd[k]) for k in param_keys(sd)}

    def step(self, sd, grads, lr, weight_decay):
        self.t += 1
        b1, b2 = ADAM_BETAS
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:
                    continue
                g = g + weight_decay * sd[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = self.v[k].sqrt() / np.sqrt(bc2) + ADAM_EPS
                sd[k].addcdiv_(self.m[k], denom, value=-lr / bc1)


def train_step(sd, kind, adam, states, next_states, wins, pi, lr=1e-3, weight_decay=1e-4):
    """One train_step, in place on `sd` (parameters, BN running statistics) and `adam`.
    states / next_states: [B,6,s,s]; wins [B] (the caller passes the reference's `wins`; the critic target is -wins,
    ac_net.py:76); pi [B,A] target policy (actor-critic only).  Returns (metrics, grads): metrics =
    (critic_loss, critic_acc, actor_loss, actor_acc) (None for the value net's actor entries), grads = the gradients
    the optimiser consumed, by key."""
    keys = param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    nxt = torch.as_tensor(np.asarray(next_states)).to(torch.float32)
    target = torch.as_tensor(-np.asarray(wins, dtype=np.float64)[:, None]).to(torch.float32)
    vals = torch.tanh(_critic_logits(sd, kind, nxt))
    cl = F.mse_loss(vals, target)
    ca = torch.mean((torch.sign(vals) == target).to(torch.float32)).item()
    al = aa = None
    loss = cl
    if kind == "ac":
        st = torch.as_tensor(np.asarray(states)).to(torch.float32)
        greedy = torch.argmax(torch.as_tensor(np.asarray(pi)).to(torch.float32), dim=1)
        logits = _actor_logits(sd, st)
        al = F.cross_entropy(logits, greedy)
        aa = torch.mean((torch.argmax(logits, dim=1) == greedy).to(torch.float32)).item()
        loss = 2 * cl + al
    loss.backward()
    grads = {k: (sd[k].grad.detach().clone() if sd[k].grad is not None else None) for k in keys}
    for k in keys:
        sd[k].requires_grad_(False)
        sd[k].grad = None
    adam.step(sd, grads, lr, weight_decay)
    return (cl.item(), ca, al.item() if al is not None else None, aa), grads


def average_state_dicts(sds):
    """average_model over `sds` (one per rank): parameters are averaged, BN statistics are left alone."""
    keys = param_keys(sds[0])
    with torch.no_grad():
        for k in keys:
            mean = sum(sd[k] for sd in sds) / len(sds)
            for sd in sds:
                sd[k].copy_(mean)
