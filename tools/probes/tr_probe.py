import sys, time, torch
sys.path.insert(0, "go-ai_amd"); sys.path.insert(0, ".")
from go_ai.game import BatchedSelfPlay
sp = BatchedSelfPlay(9, 1024, 81, temp=1.0, device=0, n_blocks=3, seed=1234, game_id_offset=0, groups=1, search="ac", dtype=3, refill=True)
import bench
cfg = bench.CONFIGS["c2"] if hasattr(bench, "CONFIGS") else None
print("cfg random_moves", cfg and cfg.get("random_moves"))
sp.randomize_starts(cfg["random_moves"] if cfg else 0)
sp.play(2)
c = sp.groups[0][0]
st = c.mcts_roots()[0]
print("stones avg", float((st[:, 0] | st[:, 1]).sum()) / st.shape[0])
for n in (262144, 1048576, 262144, 1048576):
    r = sp.profile_transition(n)
    print(n, r["avg_launch_us"], r["frac"])
time.sleep(3)
for n in (262144, 1048576):
    r = sp.profile_transition(n)
    print("after sleep", n, r["avg_launch_us"], r["frac"])

sp.play(6)
ms, n = sp.profile_conv()
print("after 6 more moves + profile_conv: conv launch us", ms / n * 1e3)
st = c.mcts_roots()[0]
print("stones avg", float((st[:, 0] | st[:, 1]).sum()) / st.shape[0])
for n in (262144, 1048576, 1048576):
    r = sp.profile_transition(n)
    print("right after", n, r["avg_launch_us"], r["frac"])
time.sleep(3)
for n in (262144, 1048576):
    r = sp.profile_transition(n)
    print("after sleep", n, r["avg_launch_us"], r["frac"])

t0 = time.time(); sp.play(40); torch.cuda.synchronize(); print("40 moves", time.time() - t0, "s")
for n in (1048576, 1048576):
    r = sp.profile_transition(n)
    print("right after 40 moves", n, r["avg_launch_us"], r["frac"])
st = c.mcts_roots()[0]
print("stones avg", float((st[:, 0] | st[:, 1]).sum()) / st.shape[0])
