import os, sys, numpy as np, torch, time
sys.path.insert(0, "/root/repo/go-ai_amd"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from go_ai.engine import Context
from oracle import nn_ref
from helpers import random_positions
def run(old, n, layers, max_batch, kind="ac", nofuse=False):
    if old: os.environ["GOAI_WINO3S_OLD"] = "1"
    else: os.environ.pop("GOAI_WINO3S_OLD", None)
    if nofuse: os.environ["GOAI_NO_TRUNK_FUSION"] = "1"
    else: os.environ.pop("GOAI_NO_TRUNK_FUSION", None)
    c = Context(9); c.nn_init(Context.NET_AC if kind == "ac" else Context.NET_VAL, layers, max_batch, dtype=Context.DTYPE_F16X3)
    sd = nn_ref.make_state_dict(kind, 9, 100 + layers, layers); c.load_weights(sd)
    st = random_positions(9, n, seed=7 + n, canonical=True, pass_prob=0.05)
    pi, val = c.nn_forward(st, mode=c.NN_RAW)
    pi2, val2 = c.nn_forward(st, mode=c.NN_RAW)
    assert torch.equal(val, val2)
    out = (pi.cpu().numpy() if kind == "ac" else None, val.cpu().numpy())
    t = None
    if n >= 1024:
        stt = torch.from_numpy(st).cuda()
        for _ in range(3): c.nn_forward(stt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): c.nn_forward(stt)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20 * 1e6
    c.close()
    return out, sd, st, t
for (n, layers, mb, kind, nofuse) in [(1024, 3, 1024, "ac", False), (333, 3, 128, "ac", False), (77, 3, 256, "val", False), (1, 3, 8, "ac", False),
                                      (64, 10, 64, "ac", False), (1500, 3, 2048, "ac", False), (77, 3, 64, "ac", True)]:
    new, sd, st, tn = run(False, n, layers, mb, kind, nofuse)
    old, _, _, to = run(True, n, layers, mb, kind, nofuse)
    want_pi, want_val = nn_ref.forward(sd, torch.tensor(st), kind)
    ev = np.abs(new[1] - want_val.numpy()).max()
    same = np.array_equal(new[1], old[1]) and (kind != "ac" or np.array_equal(new[0], old[0]))
    print(f"n={n} L={layers} mb={mb} {kind} nofuse={nofuse}: bit-equal to old kernel: {same}; max |val err| vs torch {ev:.2e}; forward us new {tn} old {to}", flush=True)
    assert np.isfinite(new[1]).all()
