"""Two steps of a training fixture on the GPU and in the oracle: per-tensor differences (diagnostic; GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "go-ai_amd"), os.path.join(ROOT, "tests")]
from oracle import nn_ref, nn_train_ref
from go_ai.engine import Context
from golden_train_util import load_train

g = load_train(sys.argv[1])
kind, size, seed = str(g["kind"]), int(g["size"]), int(g["seed"])
sd = nn_ref.make_state_dict(kind, size, seed)
pk = nn_train_ref.param_keys(sd)
pshape = {k: tuple(sd[k].shape) for k in pk}
c = Context(size, 0)
c.train_init(0 if kind == "ac" else 1, 3, 64)
c.train_load(sd)
adam = nn_train_ref.AdamState(sd)
for s in range(2):
    if s and "--sync" in sys.argv:
        c.train_load(sd)          # the oracle's parameters and statistics after step 0
    args = (g[f"s{s}_states"], g[f"s{s}_next_states"], g[f"s{s}_wins"], g[f"s{s}_pi"])
    got = c.train_step(*args)
    want, wg = nn_train_ref.train_step(sd, kind, adam, *args)
    print(s, got, want)
    grads = c.train_export(pshape, 1)
    params = c.train_export(pshape)
    for k in pk:
        w = wg[k].numpy().astype(np.float64); a = grads[k].astype(np.float64)
        rms = np.sqrt(np.mean(w ** 2))
        pd = np.abs(params[k] - sd[k].numpy())
        if s == 0 and "-v" not in sys.argv:
            continue
        print(f"  {k:24s} grad rms {rms:.2e} rel {np.sqrt(np.mean((a - w) ** 2)) / max(rms, 1e-30):.2e} | param maxdiff {pd.max():.2e} frac>2e-5 {np.mean(pd > 2e-5):.4f}")
