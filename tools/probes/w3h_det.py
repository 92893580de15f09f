import os, sys, numpy as np, torch
sys.path.insert(0, "go-ai_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
from go_ai.engine import Context
from oracle import nn_ref
from helpers import random_positions
for n, mb in ((1024, 1024), (4, 8), (16, 16)):
    c = Context(9); c.nn_init(Context.NET_AC, 3, mb, dtype=Context.DTYPE_F16X3)
    sd = nn_ref.make_state_dict("ac", 9, 103, 3); c.load_weights(sd)
    st = random_positions(9, n, seed=7 + n, canonical=True, pass_prob=0.05)
    outs = [c.nn_forward(st, mode=c.NN_RAW) for _ in range(3)]
    want_pi, want_val = nn_ref.forward(sd, torch.tensor(st), "ac")
    for i, (pi, val) in enumerate(outs):
        v = val.cpu().numpy(); d = np.abs(v - want_val.numpy())
        print(n, "run", i, "max err vs torch", d.max(), "boards off by > 1e-3:", int((d > 1e-3).sum()), "first bad", np.nonzero(d > 1e-3)[0][:8])
    c.close()
