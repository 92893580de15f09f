"""The torch fp32 NN restatement (oracle/nn_ref.py) against outputs of the reference's real
nn.Modules and numpy shim (tests/golden/nn_*.npz)."""
import numpy as np
import pytest
import torch

from golden_util import load_nn
from oracle import nn_ref

RTOL, ATOL = 1e-5, 1e-5   # same torch build on both sides; differences are summation order only


@pytest.mark.parametrize("name", ["nn_ac5", "nn_ac9", "nn_val9"])
def test_nn_ref_matches_reference_modules(oracle_lib, name):
    g = load_nn(name)
    kind, size, seed = str(g["kind"]), int(g["size"]), int(g["seed"])
    sd = nn_ref.make_state_dict(kind, size, seed)
    assert nn_ref.state_dict_digest(sd) == str(g["digest"]), "weight recipe drifted from the fixture"
    pi, val = nn_ref.forward(sd, torch.tensor(g["states"]), kind)
    assert np.allclose(val.numpy(), g["val_raw"], rtol=RTOL, atol=ATOL)
    if kind == "ac":
        assert np.allclose(pi.numpy(), g["pi_raw"], rtol=RTOL, atol=ATOL)
        pi_np, val_np = nn_ref.numpy_shim(sd, kind, "actor_critic", g["states"])
        assert pi_np.dtype == np.float32 and val_np.dtype == np.float32
        assert np.array_equal(pi_np == np.finfo(np.float32).min, g["pi_np"] == np.finfo(np.float32).min)
        assert np.allclose(pi_np, g["pi_np"], rtol=RTOL, atol=ATOL)
        assert np.allclose(val_np, g["val_np"], rtol=RTOL, atol=ATOL)
        # the fixture exercises the assist paths: a +-100 pass logit and a +-100/0 terminal value
        assert (np.abs(g["pi_np"][:, -1]) > 50).any()
        assert np.isin(g["val_np"], (-100.0, 0.0, 100.0)).any()
    else:
        val_np = nn_ref.numpy_shim(sd, kind, "critic", g["states"])
        assert np.allclose(val_np, g["val_np"], rtol=RTOL, atol=ATOL)
