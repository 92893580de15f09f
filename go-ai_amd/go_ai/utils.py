"""The two helpers of /root/reference/go_ai/utils.py that sit on the accelerated path: `mpi_play`
(utils.py:179-208: every rank plays its share of the episodes, the results are summed over the ranks) on the batched
GPU game loop, and `count_parameters` (utils.py:83-84).  Argument parsing, logging and checkpoint shuffling stay the
reference's own (out of scope)."""
import logging
import math
import time

from . import data, game


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def mpi_play(comm, go_env, pi1, pi2, requested_episodes, slots=256, seed=1234, record=True, device=None):
    """Plays `requested_episodes` games of pi1 against pi2, shared out over the ranks of `comm` (anything with
    Get_rank / Get_size / allreduce: data.SoloComm(), data.TorchComm(), an mpi4py communicator), each rank's share
    as ONE batched run on its GPU (game.BatchedPlay: `slots` games in lock step, colours alternating by episode).
    Returns (pi1's win rate, black's win rate, this rank's replay) like the reference.  The action streams are keyed
    by the global episode number (rank * episodes per rank + local number)."""
    comm = comm if comm is not None else data.SoloComm()
    world_size, rank = comm.Get_size(), comm.Get_rank()
    worker_episodes = int(math.ceil(requested_episodes / world_size))
    episodes = worker_episodes * world_size
    if device is None:
        import torch
        device = rank % max(torch.cuda.device_count(), 1)
    bp = game.BatchedPlay(go_env.size, pi1, pi2, slots=min(slots, worker_episodes), device=device, seed=seed,
                          episode_offset=rank * worker_episodes)
    bp.record(record)
    timestart = time.time()
    p1wr, black_wr, replay, steps = bp.play(worker_episodes)
    duration = time.time() - timestart
    avg_time = comm.allreduce(duration / worker_episodes) / world_size
    p1wr = comm.allreduce(p1wr) / world_size
    black_wr = comm.allreduce(black_wr) / world_size
    avg_steps = comm.allreduce(float(sum(steps))) / episodes
    if rank == 0:
        logging.debug(f'{pi1} V {pi2} | {episodes} GAMES, {avg_time:.3f} SEC/GAME, {avg_steps:.0f} STEPS/GAME, '
                      f'{100 * p1wr:.1f}% WIN({100 * black_wr:.1f}% BLACK_WIN)')
    return p1wr, black_wr, replay
