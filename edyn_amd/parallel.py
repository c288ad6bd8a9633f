"""Multi-GPU sharding of independent islands.

Islands never exchange data inside a step, so the path shards with NO data-path collective: every rank
owns a contiguous block of islands (here: sites of a mini-pile grid, or whole replicas of a pile) plus
its own copy of the static bodies, and steps them independently. The only collective is the gather of
the integrated state (13 floats per body: pos3, orn4, linvel3, angvel3) that the host registry
write-back needs - one all_gather per step over RCCL ("nccl" backend on ROCm; "gloo" in CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total, rank, world_size):
    """Contiguous, balanced [first, first+count) of `total` items for `rank`."""
    base, rem = divmod(total, world_size)
    first = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return first, count


def gather_state(local_state: torch.Tensor, counts):
    """All-gather per-rank state blocks [n_r, 13] into one [sum n_r, 13] tensor on every rank.
    `counts` = bodies per rank (known on all ranks from shard_range); ragged blocks are padded to the max."""
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    if world_size == 1:
        return local_state
    nmax = max(counts)
    pad = local_state
    if local_state.shape[0] < nmax:
        pad = torch.zeros((nmax, local_state.shape[1]), dtype=local_state.dtype, device=local_state.device)
        pad[: local_state.shape[0]] = local_state
    out = torch.empty((world_size * nmax, local_state.shape[1]), dtype=local_state.dtype, device=local_state.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    return torch.cat([out[r * nmax: r * nmax + counts[r]] for r in range(world_size)], 0)


class StateGather:
    """The per-step state gather with persistent buffers (no allocation inside the timed loop): every rank contributes
    its [n_r, 13] block, ragged blocks padded to the largest; `gather()` returns the [world, nmax, 13] buffer, rows beyond
    counts[r] of block r are padding. On ROCm the "nccl" backend is RCCL (device buffers); "gloo" stages through the host."""

    def __init__(self, counts, device, backend):
        self.counts = [int(c) for c in counts]
        self.world = len(self.counts)
        self.nmax = max(self.counts)
        self.backend = backend
        self.local = torch.zeros((self.nmax, 13), dtype=torch.float32, device=device)
        self.out = torch.empty((self.world, self.nmax, 13), dtype=torch.float32, device=device if backend == "nccl" else "cpu")

    def gather(self):
        """`self.local` holds this rank's packed state (rows beyond its count stay zero)."""
        if self.world == 1 or not dist.is_initialized():
            self.out[0].copy_(self.local)
            return self.out
        src = self.local if self.backend == "nccl" else self.local.cpu()
        dist.all_gather_into_tensor(self.out.view(-1), src.view(-1))
        return self.out

    def split(self):
        """Per-rank [n_r, 13] views of the last gather."""
        return [self.out[r, : self.counts[r]] for r in range(self.world)]


def pack_state(pos, orn, linvel, angvel):
    return np.concatenate([pos, orn, linvel, angvel], axis=1).astype(np.float32)
