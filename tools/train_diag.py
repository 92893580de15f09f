"""Worst per-tensor gradient error of the HIP train step against the torch oracle over configs x seeds (diagnostic; GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "go-ai_amd"), os.path.join(ROOT, "tests")]
from oracle import nn_ref, nn_train_ref
from go_ai.engine import Context


def run(kind, size, n, layers, seed):
    rng = np.random.default_rng(seed)
    sd = nn_ref.make_state_dict(kind, size, 11, layers=layers)
    pk = nn_train_ref.param_keys(sd)
    pshape = {k: tuple(sd[k].shape) for k in pk}
    c = Context(size, 0)
    c.train_init(0 if kind == "ac" else 1, layers, n)
    c.train_load(sd)
    adam = nn_train_ref.AdamState(sd)
    states = (rng.random((n, 6, size, size)) < 0.3).astype(np.uint8)
    nxt = (rng.random((n, 6, size, size)) < 0.3).astype(np.uint8)
    wins = rng.integers(-1, 2, n).astype(np.float32)
    pi = rng.random((n, size * size + 1)).astype(np.float32)
    got = c.train_step(states, nxt, wins, pi)
    want, wg = nn_train_ref.train_step(sd, kind, adam, states.astype(np.float32), nxt.astype(np.float32), wins, pi)
    grads = c.train_export(pshape, 1)
    top = max(float(np.sqrt(np.mean(v.numpy().astype(np.float64) ** 2))) for v in wg.values())
    worst, wk = 0.0, None
    for k in pk:
        w = wg[k].numpy().astype(np.float64); a = grads[k].astype(np.float64)
        rms = np.sqrt(np.mean(w ** 2))
        if rms < 1e-3 * top:
            continue
        rel = np.sqrt(np.mean((a - w) ** 2)) / rms
        if rel > worst:
            worst, wk = rel, k
    md = max(abs(a - b) / max(1, abs(b)) for a, b in zip(got, want) if b is not None)
    print(f"{kind} s{size} n{n} L{layers} seed{seed}: worst grad rel-rms {worst:.2e} ({wk}) metrics diff {md:.1e}", flush=True)
    c.close()


cfgs = [("ac", 19, 4, 3), ("ac", 9, 32, 3), ("ac", 13, 5, 1), ("ac", 7, 33, 2)]
if len(sys.argv) > 4:
    cfgs = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))]
for cfg in cfgs:
    for seed in range(6):
        run(*cfg, seed)
