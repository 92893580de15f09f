import os, sys, time, numpy as np, torch
sys.path.insert(0, "go-ai_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
from go_ai.engine import Context
from oracle import nn_ref
def run(n=32, size=9):
    c = Context(size); c.train_init(Context.NET_AC, 3, n); sd = nn_ref.make_state_dict("ac", size, 1); c.train_load(sd)
    rng = np.random.default_rng(0); P = size * size
    st = torch.from_numpy((rng.random((n, 6, size, size)) < 0.3).astype(np.uint8)).cuda(); nx = st.clone()
    w = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32)).cuda()
    pi = torch.softmax(torch.randn(n, P + 1), 1).cuda()
    for _ in range(3): c.train_step(st, nx, w, pi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.train_step(st, nx, w, pi)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    c.close(); return dt
for env in (None, "1"):
    if env: os.environ["GOAI_NO_TRAIN_GRAPH"] = env
    else: os.environ.pop("GOAI_NO_TRAIN_GRAPH", None)
    print("graph" if not env else "plain launches", f"9x9 batch 32: {run():.3f} ms per step; 9x9 batch 512: {run(512):.3f}; 19x19 batch 32: {run(32, 19):.3f}", flush=True)
