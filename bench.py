#!/usr/bin/env python3
"""bench.py -- MCTS leaf evals/sec (9x9, transition + NN forward), BASELINE.json's metric.

Workload (config.workload): BASELINE.json configs[1] -- 9x9 actor-critic net, mcts=81 per
move (82 leaf evaluations per move and game), 1024 concurrent self-play games per GPU from
seeded random mid-game positions (k ~ U{0..40} random legal moves), synthetic random-init
weights (the reference ships none).  One "step" = one move of every game = 82 lock-step
passes of  select -> Go transition -> CNN forward over the 1024-leaf batch -> expand/backup,
plus the temperature policy, action sampling and the move itself.  Everything inside the
timed region runs on the GPU from HBM-resident state.

    python bench.py [--gpus N --steps K --warmup W]
        N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
        (WORLD_SIZE set: this process is one rank) or run plainly, and then bench.py starts the N ranks
        itself as a child torch.distributed.run.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how each field is obtained.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
sys.path.insert(0, ROOT)

SIZE, SIMS, GAMES_PER_GPU, LAYERS = 9, 81, 1024, 3      # the default workload: BASELINE.json configs[1]
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, matrix fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0        # same guide, dense bf16; the split path issues 3 bf16 MFMA flops per algorithmic flop

# The other BASELINE.json configs, runnable with --config for DESIGN.md's tables (the driver's
# bench line is always c2).  random_moves = upper bound of the random opening length.
CONFIGS = {
    "c2": dict(size=9, sims=81, games=1024, layers=3, search="ac", random_moves=40,
               name="BASELINE.json configs[1]: 9x9 actor-critic (3 blocks x 128 ch), mcts=81/move"),
    "c3": dict(size=9, sims=8, games=4096, layers=3, search="q", random_moves=40,
               name="BASELINE.json configs[2]: 9x9 Q-value model, next_states over all 82 actions per leaf + value "
                    "forward, mcts=8/move"),
    "c4": dict(size=13, sims=81, games=8192, layers=3, search="ac", random_moves=80,
               name="BASELINE.json configs[3]: 13x13 actor-critic self-play, mcts=81/move"),
    "c5": dict(size=19, sims=81, games=4096, layers=10, search="ac", random_moves=250,
               name="BASELINE.json configs[4]: 19x19 actor-critic, deeper CNN (10 blocks x 128 ch), mcts=81/move"),
}


# --------------------------------------------------------------------------------------
# CPU baseline: the oracle's sequential search (reference execution shape: Python tree, one
# leaf at a time, batch-1 torch CPU forward, children of every leaf for the "assist" path),
# on `cores` independent single-threaded worker processes like the reference's
# multi_proc_play (/root/reference/go_ai/utils.py:211-232).  Runs BEFORE the GPU is touched.
# --------------------------------------------------------------------------------------
def _cpu_worker(args):
    wid, seconds, threads = args
    import numpy as np
    import torch
    if threads:
        torch.set_num_threads(threads)
    from oracle import mcts_ref, nn_ref
    from oracle.gogame import GoEnv, GoGame
    sd = nn_ref.make_state_dict("ac", SIZE, 0, LAYERS)
    rng = np.random.default_rng(1234 + wid)
    env = GoEnv(SIZE)
    env.reset()
    for _ in range(int(rng.integers(0, 41))):
        valid = env.valid_moves()
        valid[-1] = 0
        if valid.sum() == 0:
            break
        env.step(int(rng.choice(len(valid), p=valid / valid.sum())))
    evals = 0

    def ac(states):
        nonlocal evals
        evals += len(states)
        return nn_ref.numpy_shim(sd, "ac", "actor_critic", states)

    t0 = time.perf_counter()
    moves = 0
    while time.perf_counter() - t0 < seconds:
        if env.done:
            env.reset()
        pi, _, _ = mcts_ref.actor_critic_policy(env, ac, SIMS, 1.0, math="numpy")
        env.step(GoGame.random_weighted_action(pi))
        moves += 1
    return evals, time.perf_counter() - t0, moves, torch.get_num_threads()


def host_cpu_share():
    """How many host cores this process may really use: the smallest of the scheduler affinity, the
    cgroup CPU quota and the harness's worker-pool hint (OMP_NUM_THREADS / MAX_JOBS: the GPU box exports
    its per-GPU CPU share there while nproc shows the whole machine).  Returns (cores, explanation)."""
    on_box = os.cpu_count() or 1
    limits = {"os.cpu_count": on_box}
    try:
        limits["sched_getaffinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            limits["cgroup cpu.max"] = max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                limits["cgroup cfs_quota"] = max(1, q // per)
        except (OSError, ValueError):
            pass
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            limits[var] = int(v)
    if os.environ.get("GOAI_CPU_WORKERS", "").isdigit():
        limits = {"GOAI_CPU_WORKERS": int(os.environ["GOAI_CPU_WORKERS"]), "os.cpu_count": on_box}
        return max(1, limits["GOAI_CPU_WORKERS"]), limits
    return max(1, min(limits.values())), limits


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds):
    """All-cores figure (the one the >= 1000x target is judged on): one
    single-threaded worker process per usable core, independent games, like the reference's
    multi_proc_play; then the reference's per-process figure: ONE process with torch's default
    thread count."""
    import multiprocessing as mp
    from oracle import gogame
    gogame.build()
    cores, limits = host_cpu_share()
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(w, seconds, 1) for w in range(cores)])
    with ctx.Pool(1) as pool:
        one = pool.map(_cpu_worker, [(0, max(4.0, seconds / 2), 0)])[0]
    value = sum(e / t for e, t, _, _ in res)
    return {"value": round(value, 2), "unit": "leaf evals/s", "cores": cores, "kind": "port",
            "per_core": round(value / cores, 2),
            "cores_on_box": limits["os.cpu_count"], "core_limits": limits, "cpu_model": cpu_model(),
            "one_process": {"value": round(one[0] / one[1], 2), "torch_num_threads": one[3],
                            "sample": f"1 process x {one[1]:.0f} s, {one[2]} moves, {one[0]} leaf evals"},
            "sample": f"{cores} single-threaded worker processes x {seconds:.0f} s of whole moves "
                      f"(9x9 AC, mcts={SIMS}, oracle search + batch-1 torch CPU forward), "
                      f"{sum(m for _, _, m, _ in res)} moves, {sum(e for e, _, _, _ in res)} leaf evals"}


# --------------------------------------------------------------------------------------
# --gpus N without a launcher: this process (which never touches the GPU) times the CPU baseline,
# then starts the N ranks itself as a child `torch.distributed.run` -- as the reference's
# multi_proc_play starts its own workers (/root/reference/go_ai/utils.py:211-232) -- relays rank 0's
# JSON line and attaches the baseline to it.
# --------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, cpu):
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", args.config,
           "--dtype", args.dtype, "--groups", str(args.groups), "--no-cpu-baseline"]
    if args.games:
        cmd += ["--games", str(args.games)]
    if args.backend != "nccl":
        cmd += ["--backend", args.backend]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout)
        raise SystemExit(f"bench.py: the {args.gpus}-rank child run failed (exit code {proc.returncode})")
    out = json.loads(line)
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1)
    print(json.dumps(out), flush=True)


def gloo_rehearsal(args, rank, world):
    """--backend gloo: the N-rank path without a GPU -- rendezvous, game sharding by rank, barriers, the
    single gather, max-over-ranks time, rank 0's JSON line -- with a stub in place of the search."""
    import torch.distributed as dist
    from go_ai.sharding import gather_stats, reduce_stats, shard_games
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus
    n_games = args.games or CONFIGS[args.config]["games"]
    lo, hi = shard_games(n_games * world, world, rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    evals = (hi - lo) * (SIMS + 1) * args.steps          # what the search would have counted
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tot = reduce_stats(gather_stats(dict(leaf_evals=evals, seconds=elapsed, games=hi - lo)))
    if rank == 0:
        print(json.dumps({"metric": "rehearsal (gloo, no GPU work)", "value": round(tot["leaf_evals"] / tot["seconds"], 1),
                          "unit": "leaf evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "games_total": int(tot["games"]), "leaf_evals": int(tot["leaf_evals"]), "data": "none"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def executed_mfma_flop(algo, conv_rows, conv_flop, cpl, n_cu):
    """Flops the matrix pipe really executes per launch of the trunk kernel `algo` (goai_nn_conv_algo) over `conv_rows`
    activation rows; conv_flop = the convolutions' own (algorithmic) flops of the launch."""
    nb = int(round(conv_rows / (SIZE * SIZE)))
    if algo == 2:
        return 3.0 * conv_flop       # hi*hi + lo*hi + hi*lo on the bf16 pipe
    if algo == 1:
        # Winograd F(2x2,3x3): 16 products per 2x2 output tile and (cin, cout) instead of 36, on
        # workgroups of 112 tile rows holding NBW whole boards (conv_wino_kernels.hip)
        t2 = ((SIZE + 1) // 2) ** 2
        wgs = -(-nb // (112 // t2))
        rows = 100 if SIZE in (9, 13, 19) else 112       # six 16-row tiles + one round of 4x4 blocks, or seven tiles
        return 2.0 * wgs * rows * 16 * 128 * 128
    if algo in (9, 10):        # one fp16 product per transform-domain multiply-add
        t2 = ((SIZE + 1) // 2) ** 2
        wgs = -(-nb // 4) if algo == 9 else -(-nb // (112 // t2))
        return 2.0 * wgs * (48 * 25 * cpl if algo == 9 else 112 * 16) * 128 * 128
    if algo in (6, 8):
        # Winograd F(2x2,3x3) on the bf16 / fp16 pipe: workgroups of seven 16-column tiles (112 wtile rows), 16 positions,
        # six bf16 (three fp16) products per transform-domain multiply-add
        t2 = ((SIZE + 1) // 2) ** 2
        wgs = -(-nb // (112 // t2))
        return (6 if algo == 6 else 3) * 2.0 * wgs * 112 * 16 * 128 * 128
    if algo in (5, 7):
        # Winograd F(3x3,3x3) on the bf16 / fp16 pipe: 4 boards = 36 wtile rows run as three 16-column tiles
        wgs = -(-nb // 4)
        return (6 if algo == 5 else 3) * 2.0 * wgs * 48 * 25 * 128 * 128 * cpl
    if algo in (3, 4):
        # Winograd F(3x3,3x3): 25 products per 3x3 output tile; workgroups of 48 tile rows = 4 boards of 9x9
        # (5 boards once 4 per workgroup would need more than one round of CUs)
        wgs = -(-nb // 4) if -(-nb // 4) <= n_cu else -(-nb // 5)
        # (the two-chunk kernel runs 32 rows as 16x16x4 tiles and the other 4 as 4x4x1 blocks: no padded rows)
        return 2.0 * wgs * (36 if algo == 4 else 48) * 25 * 128 * 128 * cpl
    return conv_flop


KERNEL_NAMES = {0: "k_conv3x3", 1: "k_conv3x3_wino", 2: "k_conv3x3_bf16", 3: "k_conv3x3_wino3", 4: "k_conv3x3_wino3p",
                5: "k_conv3x3_wino3s", 6: "k_conv3x3_wino2s", 7: "k_conv3x3_wino3h", 8: "k_conv3x3_wino2s",
                9: "k_conv3x3_wino3h", 10: "k_conv3x3_wino2s"}


def exact_fp32_record(cfg, args, n_games, rank, local_rank, n_cu):
    """The same workload with GOAI_DTYPE_F32 (exact fp32 MFMA in every convolution), a few steps after the timed
    region: the figure to hold next to `value` when the split-operand arithmetic is in question."""
    import torch
    from go_ai.game import BatchedSelfPlay
    sp = BatchedSelfPlay(SIZE, n_games, SIMS, temp=1.0, device=local_rank, n_blocks=LAYERS, seed=1234,
                         game_id_offset=rank * n_games, groups=args.groups, search=cfg["search"], dtype=0,
                         refill=not args.no_refill)
    sp.randomize_starts(cfg["random_moves"])
    steps = max(2, min(args.steps, {"c2": 12, "c3": 2, "c4": 2, "c5": 1}[args.config]))
    sp.play(2 if args.config == "c2" else 1)
    e0 = sp.counters()["leaf_evals"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.play(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evals = sp.stats()["leaf_evals"] - e0
    conv_ms, conv_n = sp.profile_conv()
    cpl, algo = sp.ctx.nn_convs_per_launch(), sp.ctx.nn_conv_algo()
    rec = {"dtype": "f32", "value": round(evals / dt, 1), "unit": "leaf evals/s", "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 3), "kernel": KERNEL_NAMES[algo] + f"<{SIZE}>"}
    if cfg["search"] == "ac" and conv_n:
        rows = n_games // len(sp.groups) * SIZE * SIZE
        flop = 2.0 * rows * 128 * (9 * 128) * cpl
        launch_s = conv_ms / conv_n * 1e-3
        ex = executed_mfma_flop(algo, rows, flop, cpl, n_cu)
        rec.update({"avg_launch_us": round(launch_s * 1e6, 2), "convs_per_launch": cpl, "peak": PEAK_F32_MFMA_TFLOPS,
                    "achieved": round(flop / launch_s / 1e12, 2), "frac": round(flop / launch_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "achieved_executed": round(ex / launch_s / 1e12, 2),
                    "frac_executed": round(ex / launch_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "note": "frac = algorithmic flops over the fp32 MFMA peak; above 1 because Winograd executes 3.24x fewer "
                            "products than the convolution has multiply-adds (frac_executed is the pipe's own fraction)"})
    sp.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: about 5 s of GPU time for the config)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps before them (default: 5 for c2, else 1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--games", type=int, default=0, help="games per GPU (default: the config's)")
    ap.add_argument("--dtype", choices=["auto", "f32", "bf16x3", "bf16x6", "f16x3", "f16"], default="auto",
                    help="arithmetic of the trunk convolutions.  auto = f16x3 (fp16 matrix cores on operands split into two "
                         "fp16 parts and scaled by exact powers of two per board: fp32-grade results, error at the exact-fp32 "
                         "Winograd kernels' own level); bf16x6 = the same kernels with three bf16 parts and six products; "
                         "f32 = exact fp32 MFMA everywhere; bf16x3 = two-part bf16 split (1e-5 class, opt-in); f16 = ONE fp16 part per "
                         "operand: REDUCED PRECISION (north_star's 1e-2 tier), its own line in profiles/, never the headline")
    ap.add_argument("--groups", type=int, default=1, help="independent game groups on separate HIP streams")
    ap.add_argument("--no-refill", action="store_true",
                    help="leave the slot of a finished game idle instead of starting the next game in it")
    ap.add_argument("--c-gather", action="store_true",
                    help="also sum the per-rank results with the C-ABI's own RCCL all-reduce (goai_gather_stats) and "
                         "check it against the torch.distributed gather (always on when more than one rank runs)")
    ap.add_argument("--no-c-gather", action="store_true", help="N > 1: skip the C-ABI all-reduce cross-check")
    ap.add_argument("--c-gather-timeout", type=float, default=120.0, help="seconds the C-ABI all-reduce leg may take before the line goes out without it")
    ap.add_argument("--no-exact-fp32", action="store_true",
                    help="skip the exact_fp32 sub-record (a few untimed-for-value steps of the same workload with --dtype f32)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: CPU rehearsal of the N-rank launch / rendezvous / gather path with a stub body (no GPU)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    # defaults sized so that the timed region is seconds, not a blink (a step of c2 is 42 ms, of c5 2.8 s)
    if args.steps <= 0:
        args.steps = {"c2": 120, "c3": 6, "c4": 6, "c5": 3}.get(args.config, 8)
    if args.warmup < 0:
        args.warmup = 5 if args.config == "c2" else 1
    if args.dtype == "auto":
        args.dtype = "f16x3"
    global SIZE, SIMS, LAYERS
    SIZE, SIMS, LAYERS = cfg["size"], cfg["sims"], cfg["layers"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ
    if launched and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "c2" and args.backend == "nccl":
        cpu = cpu_baseline(args.cpu_seconds)          # before any GPU initialisation
    if not launched and args.gpus > 1:
        return launch_ranks(args, cpu)                # child ranks; this process never initialises the GPU

    import torch
    import torch.distributed as dist

    if args.backend == "gloo":
        return gloo_rehearsal(args, rank, world)
    from go_ai.game import BatchedSelfPlay

    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_games = args.games or cfg["games"]
    sp = BatchedSelfPlay(SIZE, n_games, SIMS, temp=1.0, device=local_rank, n_blocks=LAYERS, seed=1234,
                         game_id_offset=rank * n_games, groups=args.groups, search=cfg["search"],
                         dtype={"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3, "f16": 4}[args.dtype], refill=not args.no_refill)
    sp.randomize_starts(cfg["random_moves"])
    sp.play(args.warmup)
    cnt0 = sp.counters()
    evals0 = cnt0["leaf_evals"]

    barrier()
    t0 = time.perf_counter()
    sp.play(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0

    st = sp.stats()
    my_evals = st["leaf_evals"] - evals0
    assert st["errors"] == 0, st

    # dominant kernel (the trunk's 3x3 convolutions): per-launch duration by HIP events on the
    # kernel's own stream over one more move of the same workload
    my_forwards = st["forwards"] - cnt0["forwards"]
    fwd0 = st["forwards"]
    conv_ms, conv_n = sp.profile_conv()
    cpl = sp.ctx.nn_convs_per_launch()       # F(3x3,3x3) runs the whole trunk (2 * blocks convolutions) in one launch
    if cfg["search"] == "ac":
        conv_rows = n_games // len(sp.groups) * SIZE * SIZE
    else:   # data-dependent batch sizes: rows = states sent through the net during the profiled move
        conv_rows = (sp.counters()["forwards"] - fwd0) * SIZE * SIZE / max(1, conv_n * cpl // (2 * LAYERS))
    conv_flop = 2.0 * conv_rows * 128 * (9 * 128) * cpl              # algorithmic: the convolutions' own multiply-adds
    launch_s = conv_ms / conv_n * 1e-3
    peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    algo = sp.ctx.nn_conv_algo()
    kname = KERNEL_NAMES[algo] + f"<{SIZE}>"
    n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
    executed_flop = executed_mfma_flop(algo, conv_rows, conv_flop, cpl, n_cu)
    wgs = -(-int(round(conv_rows / (SIZE * SIZE))) // 4)
    lds_info = {}
    if algo == 7:
        # What this kernel really waits for besides instruction issue: LDS.  Per convolution and workgroup (DESIGN.md section 5,
        # profiles/r02_wino3s_ablation.txt): V fragment reads 50 stages x 8 waves x 6 tile steps x 2 parts x 1 KB = 4.8 MB, the
        # input transform's patch + table reads 1152 items x 1536 B = 1.77 MB, its V writes 1152 x 25 x 2 x 8 B = 0.46 MB,
        # image hand-over 0.17 MB; peak 128 B per clock and CU at 2.4 GHz.
        lds_bytes = wgs * cpl * (4.8e6 + 1.77e6 + 0.46e6 + 0.17e6)
        lds_peak = n_cu * 128 * 2.4e9
        lds_info = {"lds": {"bytes_per_launch": lds_bytes, "achieved_TBps": round(lds_bytes / launch_s / 1e12, 2),
                            "peak_TBps": round(lds_peak / 1e12, 1), "frac": round(lds_bytes / launch_s / lds_peak, 4),
                            "note": "analytic LDS bytes of the launch at 128 B per clock and CU; SQ_LDS_IDX_ACTIVE measures 0.31 of the "
                                    "launch's CU cycles (profiles/r02_wino3s_ablation.txt).  Neither the matrix pipe nor LDS is what "
                                    "this kernel waits for: its SIMDs are bound by instruction issue.  Per SIMD a stage = 2 waves x 18 MFMAs "
                                    "(16 cycles each = 576) and takes ~1,100 cycles with the fragment reads, weight loads and the output "
                                    "transform's plain v_fmac_f32 in the MFMA gaps (two per gap issue for free, packed fp32 never overlaps: "
                                    "tools/ubench/mfma_f16_valu2.hip / valu3.hip, profiles/r03_wino3h_notes.txt sections 1, 5, 6); the input "
                                    "transforms between the stages (14.6 of a layer's 57 us) and the epilogue (11.5 us) run no MFMAs at all"}}
    executed = executed_flop / launch_s / 1e12           # TFLOP/s the matrix pipe sustains
    achieved = conv_flop / launch_s / 1e12               # ALGORITHMIC: the convolutions' own flops per second (Winograd executes fewer)

    # the transition side (north_star subsystem 1): k_transition over 1,048,576 boards per launch taken from the games'
    # current positions, HIP events on the launch stream around 20 launches; algorithmic bytes = 6P in + 6P out per board.
    # (Round 3: 262,144 boards per launch -- one Q-search make_children step of 4096 games -- take the kernel ~60 us at
    # 9x9, less than the Python call that launches it: the events then timed the interpreter, 100 us per launch, where
    # rocprofv3 shows 60 us of kernel (profiles/r03_c2_kernel_stats.csv).  Four times the boards keep the queue ahead.)
    tr = sp.profile_transition(1048576)

    # the one collective of the run: gather per-rank results (utils.py:201-204's allreduces)
    from go_ai.sharding import gather_stats, reduce_stats
    per_rank = gather_stats(dict(leaf_evals=my_evals, seconds=elapsed, games=st["games"], finished=st["finished"],
                                 black_wins=st["black_wins"], white_wins=st["white_wins"],
                                 total_steps=st["total_steps"]), device="cuda")
    tot = reduce_stats(per_rank)
    total_evals, t_max = tot["leaf_evals"], tot["seconds"]

    c_gather, c_gather_stuck = None, False
    if (args.c_gather or (world > 1 and not args.no_c_gather)) and len(sp.groups) == 1:
        # the same sums through goai_gather_stats: rank 0's RCCL id goes round by torch.distributed, the all-reduce is the library's.
        # The leg runs beside the main thread with a time limit: the line's value and roofline are measured by now, and a communicator
        # that will not form on some node (this path has run with one rank only so far) must not take them with it.
        import threading
        ident = [sp.ctx.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        box = {}

        def leg():
            try:
                torch.cuda.set_device(local_rank)
                comm = sp.ctx.comm_init(ident[0], world, rank)
                sums, extra = sp.ctx.gather_stats(comm, extra=[float(my_evals), float(st["games"])])
                sp.ctx.comm_destroy(comm)
                box["ok"] = (sums, extra)
            except Exception as e:      # noqa: BLE001 -- reported in the line
                box["error"] = f"{type(e).__name__}: {e}"

        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(args.c_gather_timeout)
        if "ok" in box:
            sums, extra = box["ok"]
            # (reported, not asserted: a rank that raised here would leave the others waiting in the next collective)
            same = int(extra[0]) == int(total_evals) and int(extra[1]) == int(tot["games"])
            c_gather = dict(sums, ranks=world, leaf_evals_timed=int(extra[0]), games=int(extra[1]), agrees_with_torch_gather=same)
        else:
            c_gather_stuck = th.is_alive()
            c_gather = {"ranks": world, "error": box.get("error", f"no answer within {args.c_gather_timeout:.0f} s (given up; the rest of the line does not depend on it)")}

    exact = None
    if rank == 0 and world == 1 and args.dtype != "f32" and not args.no_exact_fp32:
        exact = exact_fp32_record(cfg, args, n_games, rank, local_rank, n_cu)

    if rank == 0:
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "conv3x3_traffic.json")
        if os.path.exists(tpath):          # recorded PMC passes of THIS workload (default dtype, default games), per kernel
            rec = json.load(open(tpath))
            if args.config == "c2":
                traffic = rec.get("per_kernel", {}).get(kname.split("<")[0], {}).get("hbm_bytes_per_launch")
            elif rec.get("per_config", {}).get(args.config, {}).get("kernel") == kname.split("<")[0] and args.dtype == "f16x3":
                traffic = rec["per_config"][args.config]["hbm_bytes_per_launch"]
            if traffic is not None:
                traffic_src = f"recorded rocprofv3 --pmc passes, profiles/conv3x3_traffic.json ({rec.get('recorded', 'round 1')})"
        out = {
            "metric": f"MCTS leaf evals/sec ({SIZE}x{SIZE}, transition + NN forward)",
            "value": round(total_evals / t_max, 1),
            "unit": "leaf evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            **({"c_gather": c_gather} if c_gather else {}),
            "precision": {"f32": "fp32 storage and arithmetic (fp32 MFMA)",
                          "bf16x6": "fp32 storage; trunk products as 6 bf16 MFMAs on operands split exactly into three bf16 parts, "
                                    "fp32 accumulate and transforms: 6e-6 max rel. error vs the torch fp32 oracle on the 3-block net "
                                    "(the exact-fp32 F(3x3,3x3) kernel: 7e-6), tolerance 1e-3 (tests/test_gpu_nn.py)",
                          "f16x3": "fp32 storage; trunk products as 3 fp16 MFMAs on operands split into two fp16 parts (22 bits + sign), "
                                   "scaled by exact powers of two per board, fp32 accumulate and transforms; error vs the torch fp32 "
                                   "oracle at the exact-fp32 kernels' own level (tests/test_gpu_nn.py), tolerance 1e-3",
                          "bf16x3": "fp32 storage; trunk products as 3 bf16 MFMAs on two-part splits: 2e-5..8e-5 vs torch fp32",
                          "f16": "REDUCED PRECISION (opt-in, not the headline): fp32 storage; trunk products as ONE fp16 MFMA on operands "
                                 "rounded to fp16 (11 bits) and scaled by powers of two per board; fp32 accumulate and transforms; "
                                 "1e-3..4e-3 vs the torch fp32 oracle, tolerance 1e-2 (BASELINE.json north_star's second tier)"}[args.dtype],
            "config": {"workload": f"{cfg['name']}, {n_games} concurrent games per GPU, step = one move of every game "
                                   f"({SIMS + 1} leaf evals per game)",
                       "board": SIZE, "mcts": SIMS, "games_per_gpu": n_games, "parallelism": f"games sharded x{world}",
                       "stream_groups_per_gpu": len(sp.groups),
                       "finished_games": "slot starts the next game from the empty board" if not args.no_refill else "slot idles",
                       "weights": "random init, seed 0", "leaf_evals_timed": int(total_evals)},
            "roofline": {"bound": "mfma", "kernel": kname, **lds_info,
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "frac_algorithmic": round(achieved / peak, 4),
                         "achieved_executed": round(executed, 2), "frac_executed": round(executed / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_us": round(launch_s * 1e6, 2), "launches_timed": conv_n,
                         "executed_mfma_flop_per_launch": executed_flop, "convs_per_launch": cpl,
                         "algorithm": {0: "direct implicit GEMM, fp32 MFMA", 1: "Winograd F(2x2,3x3), fp32 MFMA",
                                       2: "direct implicit GEMM, bf16 MFMA on hi+lo split operands (3 products per multiply-add)",
                                       3: "Winograd F(3x3,3x3), fp32 MFMA",
                                       4: "Winograd F(3x3,3x3), fp32 MFMA (16x16x4 + 4x4x1), two 32-channel chunks per output transform",
                                       6: "Winograd F(2x2,3x3), bf16 MFMA 16x16x32 on operands split exactly into three bf16 parts "
                                          "(6 products per multiply-add, fp32 accumulate, fp32 transforms): fp32-grade results",
                                       5: "Winograd F(3x3,3x3), bf16 MFMA 16x16x32 on operands split exactly into three bf16 parts "
                                          "(6 products per multiply-add, fp32 accumulate, fp32 transforms): fp32-grade results",
                                       7: "Winograd F(3x3,3x3), fp16 MFMA 16x16x32 on operands split into two fp16 parts and scaled by "
                                          "powers of two per board (3 products per multiply-add, fp32 accumulate, fp32 transforms): fp32-grade results",
                                       8: "Winograd F(2x2,3x3), fp16 MFMA 16x16x32 on operands split into two fp16 parts and scaled by "
                                          "powers of two per board (3 products per multiply-add, fp32 accumulate, fp32 transforms): fp32-grade results",
                                       9: "Winograd F(3x3,3x3), ONE fp16 MFMA 16x16x32 per multiply-add on operands rounded to fp16 (reduced precision)",
                                       10: "Winograd F(2x2,3x3), ONE fp16 MFMA 16x16x32 per multiply-add on operands rounded to fp16 (reduced precision)"}[algo],
                         "algorithmic_flop_per_launch": conv_flop,
                         "achieved_algorithmic": round(achieved, 2),
                         "note": "achieved / frac (= frac_algorithmic) count the convolutions' own 2*M*128*1152 flops per launch "
                                 "(SURVEY.md 8d) over the HIP-event launch time; achieved_executed / frac_executed count the flops "
                                 "the matrix pipe really executes (Winograd needs fewer products, the operand split three per product)"},
            "roofline_transition": tr,
            **({"exact_fp32": exact} if exact else {}),
            "games": {"total": int(tot["games"]), "finished": int(tot["finished"]),
                      "black_wins": int(tot["black_wins"]), "white_wins": int(tot["white_wins"]),
                      "moves_played": int(tot["total_steps"])},
        }
        if cfg["search"] == "q":
            out["value_forwards_per_s"] = round(my_forwards / elapsed, 1)
            out["forwards_per_leaf"] = round(my_forwards / max(1, my_evals), 2)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if c_gather_stuck:       # a thread is still inside RCCL: leave without the destructors that would wait for it
        globals()["_real_stdout"].flush()
        sys.stderr.flush()
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    # ONE JSON line on stdout, whatever the libraries print: RCCL writes a version banner to stdout when a communicator is made
    # (torch's process group or goai_comm_init).  File descriptor 1 points at stderr for the run; the line goes to the real stdout.
    sys.stdout.flush()
    _real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    _print = print

    def print(*a, **kw):      # noqa: A001 -- the JSON lines are the only print() calls without file=
        if kw.get("file") is None:
            kw["file"] = _real_stdout
        _print(*a, **kw)

    main()
    _real_stdout.flush()
