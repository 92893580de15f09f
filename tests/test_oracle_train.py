"""The training oracle (oracle/nn_train_ref.py: torch fp32 autograd + written-out Adam) against fixtures recorded from
the reference's own train_step on its real nn.Modules with torch.optim.Adam (tests/golden/train_*.npz): losses and
accuracies, the gradients the optimiser consumed, the updated parameters and every BatchNorm running statistic,
over two consecutive steps."""
import numpy as np
import pytest

from golden_train_util import TRAIN_FIXTURES, check_against_golden, load_train
from oracle import nn_ref, nn_train_ref


@pytest.mark.parametrize("name", TRAIN_FIXTURES)
def test_oracle_train_step_matches_reference(name):
    g = load_train(name)
    kind, size, seed = str(g["kind"]), int(g["size"]), int(g["seed"])
    sd = nn_ref.make_state_dict(kind, size, seed)
    assert nn_ref.state_dict_digest(sd) == str(g["digest"])
    adam = nn_train_ref.AdamState(sd)
    for s in range(int(g["steps"])):
        metrics, grads = nn_train_ref.train_step(sd, kind, adam, g[f"s{s}_states"], g[f"s{s}_next_states"], g[f"s{s}_wins"],
                                                 g[f"s{s}_pi"])
        params = {k: sd[k].numpy() for k in nn_train_ref.param_keys(sd)}
        stats = {k: sd[k].numpy() for k in sd if k.endswith(("running_mean", "running_var"))}
        check_against_golden(g, s, metrics, {k: v.numpy() for k, v in grads.items()}, params, stats, rtol=2e-4, atol=2e-3)


def test_average_state_dicts():
    sds = [nn_ref.make_state_dict("ac", 5, s) for s in (1, 2, 3)]
    want = sum(sd["main.0.weight"] for sd in sds) / 3
    rm = [sd["main.1.running_mean"].clone() for sd in sds]
    nn_train_ref.average_state_dicts(sds)
    assert all(np.allclose(sd["main.0.weight"], want) for sd in sds)
    assert all(np.array_equal(sd["main.1.running_mean"], r) for sd, r in zip(sds, rm))    # statistics are not averaged
