"""Agent-batch sharding across the GPUs of one node (one process per GPU).

Agents are independent `ErgodicControl` instances, so the batch shards with no data-path
collective; the only exchange is the all-gather of the per-agent trajectory coefficients c_k
(K^2 reals per agent) that decentralised ergodic control shares between agents (reference
README ref. [2]).  `backend="nccl"` is RCCL over xGMI on ROCm; `gloo` is used by the CPU tests.
"""


def shard_range(n_agents, rank, world):
    """Contiguous block of agents owned by `rank`: [first, last)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    first = (n_agents * rank) // world
    last = (n_agents * (rank + 1)) // world
    return first, last


def shard_sizes(n_agents, world):
    return [shard_range(n_agents, r, world)[1] - shard_range(n_agents, r, world)[0] for r in range(world)]


def gather_ck(ck_local, out=None, group=None, async_op=False):
    """All-gather of per-agent c_k: ck_local [B_local, K2] -> [world * B_local, K2] in rank
    order (equal shard sizes, one in-place collective).  Returns (tensor, work)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * ck_local.shape[0],) + tuple(ck_local.shape[1:]),
                          dtype=ck_local.dtype, device=ck_local.device)
    work = dist.all_gather_into_tensor(out, ck_local.contiguous(), group=group, async_op=async_op)
    return out, work


def gather_ck_ragged(ck_local, n_agents, group=None):
    """All-gather for unequal shards (n_agents not divisible by the world size): shards are
    padded to the largest one so a single equal-size collective still does the exchange."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_agents, world)
    smax = max(sizes)
    padded = torch.zeros((smax,) + tuple(ck_local.shape[1:]), dtype=ck_local.dtype, device=ck_local.device)
    padded[:ck_local.shape[0]] = ck_local
    out = torch.empty((world * smax,) + tuple(ck_local.shape[1:]), dtype=ck_local.dtype,
                      device=ck_local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * smax:r * smax + sizes[r]] for r in range(world)], 0)


def consensus_ck(ck_all):
    """Mean of the agents' c_k: the shared statistic of decentralised ergodic control."""
    return ck_all.mean(dim=0)


def consensus_ck_allreduce(ck_local, group=None):
    """The same consensus without gathering anything: each rank sums its own agents' c_k, ONE all-reduce of
    K^2 + 1 reals (the agent count rides along) gives the global sum and count.  This is what
    eea_comm_consensus_ck does on the device over RCCL; the result feeds eea_batch_io::d_ck_shared."""
    import torch
    import torch.distributed as dist
    buf = torch.cat([ck_local.sum(dim=0), torch.tensor([float(ck_local.shape[0])], dtype=ck_local.dtype,
                                                        device=ck_local.device)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:-1] / buf[-1]


# ---- grid-tiled phi_k (BASELINE config 5): the target grid is sharded by rows ----------------
def grid_row_tile(ny, rank, world):
    """Rows [row0, row0 + nrows) of the target grid owned by `rank`."""
    row0, row1 = shard_range(ny, rank, world)
    return row0, row1 - row0


def reduce_phik(phik_partial, group=None, total_mass=None):
    """Sum of the per-rank K^2 partials (one all-reduce over RCCL/xGMI; 7.2 KB at K = 30).  If
    `total_mass` (this rank's sum of un-normalised target values) is given it rides in the same
    collective and phi_k is divided by the global mass (Target::fill's normalisation,
    reference target.cpp:87)."""
    import torch
    import torch.distributed as dist
    if total_mass is None:
        buf = phik_partial.clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return buf
    buf = torch.cat([phik_partial.reshape(-1), total_mass.reshape(1).to(phik_partial.dtype)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:-1] / buf[-1]


def reduce_occupancy_sums(sums_partial, group=None):
    """phi_k of an occupancy target tiled by rows (Engine.spatial_coeff_occupancy_rows): one
    all-reduce of the K^2 un-normalised sums; element 0 (mode (0,0): cos 0 = 1) of the total is the
    sum of the cell entropies, i.e. the normaliser of the reference's target.cpp:87."""
    import torch.distributed as dist
    buf = sums_partial.clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf / buf.reshape(-1)[0]
