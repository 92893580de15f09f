"""World-size-2 gloo test (CPU) of the multi-GPU plumbing: game sharding and the single
result gather.  The search itself never communicates."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from go_ai import sharding
    lo, hi = sharding.shard_games(2049, world, rank)
    rec = dict(leaf_evals=1000 * (rank + 1), seconds=1.0 + rank, games=hi - lo, finished=rank, black_wins=3,
               white_wins=1, total_steps=10 * (hi - lo))
    per_rank = sharding.gather_stats(rec)
    total = sharding.reduce_stats(per_rank)
    torch.save(dict(per_rank=per_rank, total=total, shard=(lo, hi)), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_sharding_world2(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert res[0]["shard"] == (0, 1025) and res[1]["shard"] == (1025, 2049)
    for r in res:
        assert [p["leaf_evals"] for p in r["per_rank"]] == [1000.0, 2000.0]
        assert r["total"]["games"] == 2049 and r["total"]["seconds"] == 2.0
        assert r["total"]["leaf_evals_per_s"] == 1500.0
        assert r["total"]["total_steps"] == 20490


def test_shards_cover_all_games():
    sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
    from go_ai import sharding
    for total in (1, 7, 8, 1024, 8192 * 8 + 3):
        for world in (1, 2, 4, 8):
            spans = [sharding.shard_games(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts the two ranks itself as a child
    torch.distributed.run (as the reference's multi_proc_play starts its own workers,
    /root/reference/go_ai/utils.py:211-232) and relays rank 0's line; here with the gloo stub body
    (rendezvous, game sharding, barriers, gather) so that it runs without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--backend", "gloo", "--games", "100"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=env, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["games_total"] == 200
    assert out["leaf_evals"] == 200 * 82 * 3


def test_bench_rejects_a_world_size_that_differs_from_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert proc.returncode != 0 and "started 2 ranks" in proc.stderr


def _avg_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from go_ai import data
    from go_ai.models import average_model

    class Model:                      # what average_model needs of a net: its flat parameter buffer
        def __init__(self):
            self.flat = torch.arange(10, dtype=torch.float32) * (rank + 1)

        def flat_parameters(self):
            return self.flat

    comm, m = data.TorchComm(), Model()
    average_model(comm, m)
    metrics = comm.allreduce(np.array([1.0 + rank, np.nan if rank else 2.0])) / comm.Get_size()
    scalar = comm.allreduce(3.0 * (rank + 1))
    torch.save(dict(flat=m.flat, metrics=metrics, scalar=scalar), os.path.join(out_dir, f"a{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_average_model_world2(tmp_path):
    """average_model (/root/reference/go_ai/models/__init__.py:279-282) and the metric all-reduce of optimize()
    (models/__init__.py:226) over two gloo ranks."""
    world, port = 2, 31500 + os.getpid() % 2000
    mp.spawn(_avg_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"a{r}.pt"), weights_only=False) for r in range(world)]
    for r in res:
        assert torch.equal(r["flat"], torch.arange(10, dtype=torch.float32) * 1.5)
        assert r["metrics"][0] == 1.5 and r["metrics"][1] != r["metrics"][1]      # nan + 2 stays nan, as with MPI.SUM
        assert r["scalar"] == 9.0


def test_solo_comm_and_metrics_text():
    sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
    from go_ai import data
    from go_ai.models import Adam, ModelMetrics, average_model
    assert data.SoloComm().allreduce(4.5) == 4.5
    average_model(data.SoloComm(), None)                  # one rank: nothing to do, the model is not touched
    assert str(ModelMetrics(0.5, 0.25, float("nan"), None)) == "C[25.0% 0.500L] "
    assert str(ModelMetrics(0.5, 0.25, 1.25, 0.5)) == "C[25.0% 0.500L] A[50.0% 1.250L] "
    opt = Adam(lr=2e-3)
    assert (opt.lr, opt.weight_decay) == (2e-3, 1e-4)


class _FakeNet:
    """Stands in for RLNet in average_model: a flat parameter tensor (CPU here, a GPU view in the product)."""

    def __init__(self, values):
        self.flat = torch.tensor(values, dtype=torch.float32)

    def flat_parameters(self):
        return self.flat


class _ReturningComm:
    """mpi4py semantics (what the reference passes, /root/reference/go_ai/models/__init__.py:279-282): lowercase
    allreduce returns a NEW object with the sum and leaves its argument alone."""

    def __init__(self, world, others):
        self.world, self.others = world, others

    def Get_size(self):
        return self.world

    def allreduce(self, x, op=None):
        return (x.clone() + self.others).numpy()        # a new object, not even a tensor


class _InPlaceComm(_ReturningComm):
    """data.TorchComm semantics: sums in place, returns its argument."""

    def allreduce(self, x, op=None):
        x.add_(self.others)
        return x


class _InPlaceNoneComm(_ReturningComm):
    def allreduce(self, x, op=None):
        x.add_(self.others)


@pytest.mark.parametrize("comm_cls", [_ReturningComm, _InPlaceComm, _InPlaceNoneComm])
def test_average_model_uses_the_sum_the_communicator_returns(comm_cls):
    sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
    from go_ai.models import average_model
    net = _FakeNet([1.0, 2.0, 3.0])
    average_model(comm_cls(2, torch.tensor([3.0, 2.0, 1.0])), net)
    assert torch.equal(net.flat, torch.tensor([2.0, 2.0, 2.0]))
    one = _FakeNet([5.0])
    average_model(comm_cls(1, torch.tensor([100.0])), one)          # a single rank: untouched
    assert torch.equal(one.flat, torch.tensor([5.0]))
