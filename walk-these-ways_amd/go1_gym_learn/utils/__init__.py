"""Trajectory helpers of the reference's go1_gym_learn/utils (utils.py:5-43).  They only serve recurrent policies
(`RolloutStorage.reccurent_mini_batch_generator`); the ppo_cse policy is feed-forward (actor_critic.py:20
`is_recurrent = False`) and the runners never call them — kept for the import surface, pinned by
tests/golden/traj_utils.npz."""
import torch


def split_and_pad_trajectories(tensor, dones):
    """Cut the (T, N, ...) rollout of every environment at its `dones` (the last step always ends a piece), and stack the
    pieces — environment by environment, in time order — as columns of a zero-padded (longest piece, pieces, ...) tensor.
    Also returns the (T, pieces) validity mask: row t of column k is True while t < length of piece k."""
    T, N = tensor.shape[0], tensor.shape[1]
    ends_flag = dones.reshape(T, N).clone().bool()
    ends_flag[-1] = True
    ends = ends_flag.t().reshape(-1).nonzero(as_tuple=False)[:, 0]            # positions in env-major order
    starts = torch.cat((ends.new_zeros(1), ends[:-1] + 1))
    lengths = ends - starts + 1
    rows = tensor.transpose(0, 1).flatten(0, 1)                               # (N * T, ...), env-major like `ends`
    steps = torch.arange(int(lengths.max()), device=tensor.device).unsqueeze(1)
    inside = steps < lengths.unsqueeze(0)                                     # (longest, pieces)
    padded = rows.new_zeros((steps.shape[0], lengths.numel()) + tuple(rows.shape[1:]))
    padded[inside] = rows[(starts.unsqueeze(0) + steps)[inside]]
    masks = lengths > torch.arange(0, T, device=tensor.device).unsqueeze(1)
    return padded, masks


def unpad_trajectories(trajectories, masks):
    """Inverse of split_and_pad_trajectories for a (T, pieces, D) tensor: the valid entries, piece after piece, folded back into
    rows of T steps -> (T, valid // T, D)."""
    valid = trajectories.transpose(0, 1)[masks.transpose(0, 1)]
    return valid.reshape(-1, trajectories.shape[0], trajectories.shape[-1]).transpose(0, 1)
