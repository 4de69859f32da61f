"""Multi-GPU: envs shard by global id, one process per GPU, no data-path collective.

The reference's only parallelism is env-level data parallelism (one SubprocVecEnv worker per
env, src/baselines/common/vec_env/subproc_vec_env.py:31-50; per-env seed = seed + rank,
src/utils.py:39).  Here rank r of W owns the contiguous global env ids shard_range(N, r, W); the
Philox subsequence is the GLOBAL id, so trajectories are identical for every W.  The single
collective is an all-gather of the fixed 4 x int64 episode-statistics record per logging interval
(the epinfobuf aggregation of src/ppo_multi_agent.py:288,331,366-390), RCCL over xGMI on GPUs
(backend "nccl"), gloo on CPU in the tests.
"""


def shard_range(total_envs, rank, world):
    """(first global env id, count) of `rank`'s contiguous shard; counts differ by at most 1."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(total_envs), world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


STAT_KEYS = ("episodes", "ep_len_sum", "ep_return_sum", "env_steps")


def gather_stats(local, device=None, group=None):
    """all-gather the per-rank record; returns (list of per-rank dicts, totals dict).

    `local` is MultiSnakeVecEnv.stats() (or any dict with STAT_KEYS).  Without an initialised
    process group this is the single-rank identity."""
    import torch
    import torch.distributed as dist

    live = dist.is_available() and dist.is_initialized()
    backend = str(dist.get_backend(group)) if live else "none"
    if device is None:
        # RCCL ("nccl") moves device tensors only; gloo moves host tensors
        on_gpu = live and "nccl" in backend and torch.cuda.is_available()
        device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    rec = torch.tensor([int(local[k]) for k in STAT_KEYS], dtype=torch.int64, device=device)
    if live:
        world = dist.get_world_size(group)
        out = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(out, rec, group=group)
    else:
        out = [rec]
    per_rank = [dict(zip(STAT_KEYS, (int(x) for x in o.tolist()))) for o in out]
    total = {k: sum(r[k] for r in per_rank) for k in STAT_KEYS}
    total["mean_ep_return"] = total["ep_return_sum"] / max(1, total["episodes"])
    total["mean_ep_len"] = total["ep_len_sum"] / max(1, total["episodes"])
    # where the gathered records lived and what moved them (logging; "cuda:N" + "nccl" = RCCL)
    total["record_device"], total["backend"] = str(out[0].device), backend
    return per_rank, total


def make_sharded(total_envs, rank=None, world=None, device=None, **kw):
    """MultiSnakeVecEnv for this rank's shard of `total_envs` global envs (env ids stay global).

    rank / world default to torchrun's RANK / WORLD_SIZE; the device defaults to cuda:LOCAL_RANK
    (one process per GPU), so a rank that never called torch.cuda.set_device still lands on its own
    card.  The handle and every buffer of the env live on that device."""
    import os
    from .vec_env import MultiSnakeVecEnv

    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if device is None:
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    start, count = shard_range(total_envs, rank, world)
    return MultiSnakeVecEnv(count, env_id_base=start, device=device, **kw)
