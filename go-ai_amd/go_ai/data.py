"""The mask / children half of /root/reference/go_ai/data.py:15-60 on the HIP Go-rule kernels.

Same names, argument meaning and return shapes as the reference.  Of the replay half
(data.py:63-105) the in-memory conversions are here (symmetries, state+action planes, replay ->
events -> numpy batches) and so is the disk replay (data.py:108-182): one pickle of a Trajectory
collection, rank-serialised appends, balanced win / non-win sampling.  The reference passes an
mpi4py communicator; here `comm` is anything with Get_rank / Get_size / Barrier -- `SoloComm()`
(one process) or `TorchComm()` over an initialised torch.distributed group (gloo on CPU, RCCL on GPUs).
"""
import collections
import os
import pickle
import random

import numpy as np

from .env import GoGame, GoVars, context  # noqa: F401  (GoGame / GoVars are re-exported like the reference's)


def batch_invalid_moves(states):
    """1 where a move is invalid, 0 where it is valid; the pass is always valid (data.py:15-24)."""
    states = np.asarray(states)
    assert len(states.shape) == 4
    batchsize, board_size = states.shape[0], states.shape[2]
    invalid_moves = states[:, GoVars.INVD_CHNL].reshape((batchsize, -1))
    return np.insert(invalid_moves, board_size ** 2, 0, axis=1)


def batch_valid_moves(states):
    return 1 - batch_invalid_moves(states)


def batch_invalid_values(states):
    """np.finfo(float32).min at invalid moves, (-)0 at valid ones (data.py:31-38)."""
    return np.finfo(np.float32).min * batch_invalid_moves(states)


def batch_padded_children(states):
    """GoGame.children(state, canonical=True, padded=True) for every state (data.py:54-60): one
    launch of the all-actions transition kernel instead of B x A Python calls."""
    states = np.asarray(states)
    kids, _, _ = context(states.shape[-1]).children(states, canonical=True)
    return list(kids.cpu().numpy().astype(np.float64))


def batch_win_children(batch_children):
    """winning(child) where the child ended the game, else 0 (data.py:41-51), by the scoring
    kernel; zero-padded (invalid) children give 0."""
    kids = np.asarray(batch_children)
    b, a = kids.shape[:2]
    _, _, ended, winner = context(kids.shape[-1]).score(kids.reshape((b * a,) + kids.shape[2:]))
    win = np.where(ended.cpu().numpy() > 0, winner.cpu().numpy(), 0)
    return win.reshape(b, a)


def batch_random_symmetries(states):
    """A random one of the 8 board symmetries per state (data.py:63-68)."""
    states = np.asarray(states)
    assert len(states.shape) == 4
    return np.array([GoGame.random_symmetry(state) for state in states])


def batch_combine_state_actions(states, actions):
    """[B, 7, s, s]: the state planes plus a one-hot plane of the action (all zero for the pass),
    data.py:71-82."""
    states = np.asarray(states)
    new_shape = np.array(states.shape)
    new_shape[1] = GoVars.NUM_CHNLS + 1
    size = new_shape[-1]
    state_actions = np.zeros(new_shape)
    state_actions[:, :-1] = states
    for i, a in enumerate(actions):
        if a < size ** 2:
            state_actions[i, -1, a // size, a % size] = 1
    return state_actions


def replay_to_events(replay):
    """Flatten trajectories (go_ai.game.Trajectory, e.g. BatchedSelfPlay.trajectories()) into the
    event tuples (state, action, reward, children, terminal, won, pi) of data.py:85-89."""
    events = []
    for traj in replay:
        events.extend(traj.get_events())
    return events


def events_to_numpy(events):
    """Column arrays with the reference's dtypes (data.py:92-105)."""
    if len(events) == 0:
        return [], [], [], [], [], []
    cols = list(zip(*events))
    states = np.array(list(cols[0]), dtype=np.float32)
    actions = np.array(list(cols[1]), dtype=int)
    rewards = np.array(list(cols[2]), dtype=np.float32).reshape((-1,))
    next_states = np.array(list(cols[3]), dtype=np.float32)
    terminals = np.array(list(cols[4]), dtype=np.uint8)
    wins = np.array(list(cols[5]), dtype=int)
    pis = np.array(list(cols[6]), dtype=np.float32)
    return states, actions, rewards, next_states, terminals, wins, pis


# ---- disk replay (/root/reference/go_ai/data.py:108-182) ---------------------------------------
class SoloComm:
    """Single-process stand-in for the reference's MPI communicator."""

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def Barrier(self):
        pass

    def allreduce(self, x):
        """comm.allreduce(x, op=MPI.SUM) of one process: x itself."""
        return x


class TorchComm:
    """The communicator calls the replay code needs, over the default torch.distributed group
    (one process per GPU, as bench.py runs)."""

    def Get_rank(self):
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def Get_size(self):
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def Barrier(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    def allreduce(self, x):
        """comm.allreduce(x, op=MPI.SUM) (models/__init__.py:226,282).  A torch tensor is summed IN PLACE over the
        ranks and returned (a GPU tensor goes over RCCL when the group's backend is nccl, through the host
        otherwise); numbers and numpy arrays come back as new float64 arrays."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return x
        if torch.is_tensor(x):
            if x.is_cuda and dist.get_backend() != "nccl":
                host = x.detach().cpu()
                dist.all_reduce(host)
                x.copy_(host)
            else:
                dist.all_reduce(x)
            return x
        t = torch.as_tensor(np.asarray(x, dtype=np.float64))
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t)
        out = t.cpu().numpy()
        return out if np.ndim(x) else float(out)


def load_replay(replay_path):
    """The Trajectory collection this package wrote to `replay_path` (data.py:108-117)."""
    with open(replay_path, 'rb') as f:
        return pickle.load(f)


def mpi_sample_eventdata(comm, replay_path, batches, batchsize):
    """Every rank, one at a time, loads the replay and draws min(batchsize * batches / 2, n) events
    from the games black won and as many from the others (n = the smaller pool), shuffles them and
    splits the columns into batches (data.py:120-159).  Returns (batches, number of games on disk)."""
    comm = comm if comm is not None else SoloComm()
    rank, sample_data, replay_len = comm.Get_rank(), None, None
    for worker in range(comm.Get_size()):
        if rank == worker:
            replay = load_replay(replay_path)
            replay_len = len(replay)
            black_wins = replay_to_events([t for t in replay if t.get_winner() == 1])
            black_nonwins = replay_to_events([t for t in replay if t.get_winner() != 1])
            sample_size = min(batchsize * batches // 2, len(black_wins), len(black_nonwins))
            sample_data = random.sample(black_wins, sample_size) + random.sample(black_nonwins, sample_size)
            del replay
        comm.Barrier()
    random.shuffle(sample_data)
    columns = events_to_numpy(sample_data)
    sample_size = len(columns[0])
    assert all(len(col) == sample_size for col in columns)
    splits = max(sample_size // batchsize, 1)
    batched = list(zip(*[np.array_split(col, splits) for col in columns]))
    return batched, replay_len


def mpi_disk_append_replay(comm, args, replays):
    """Ranks take turns extending the collection at args.replay_path (data.py:162-174); a deque
    written by reset_replay keeps its maxlen, so old games fall out."""
    comm = comm if comm is not None else SoloComm()
    rank = comm.Get_rank()
    for worker in range(comm.Get_size()):
        if rank == worker:
            if os.path.exists(args.replay_path):
                all_replays = load_replay(args.replay_path)
                all_replays.extend(replays)
            else:
                all_replays = replays
            with open(args.replay_path, 'wb') as f:
                pickle.dump(all_replays, f)
        comm.Barrier()


def reset_replay(args):
    """An empty deque(maxlen=args.replaysize) at args.replay_path (data.py:177-182)."""
    if os.path.exists(args.replay_path):
        os.remove(args.replay_path)
    with open(args.replay_path, 'wb') as f:
        pickle.dump(collections.deque(maxlen=args.replaysize), f)
