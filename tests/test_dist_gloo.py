"""N>1 path on CPU: two gloo ranks each own a shard of the global env ids (stepped here by the CPU
oracle standing in for the GPU engine), trajectories must equal the unsharded run, and the only
collective -- the all-gather of the episode-statistics record -- must sum correctly."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import msnake

TOTAL, STEPS, SEED = 96, 60, 11


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _actions():
    return np.random.default_rng(5).integers(0, 5, (STEPS, TOTAL, 3)).astype(np.int32)


def _run_shard(start, count):
    from oracle.snake_oracle import Oracle
    ora = Oracle(count, dim=19, n_snakes=3, rules="snake_env", seed=SEED, env_id_base=start)
    obs0 = ora.reset().copy()
    acts = _actions()[:, start:start + count]
    crc, eps, lens, rets = [], 0, 0, 0
    for t in range(STEPS):
        obs, rew, done, ns, er, el = ora.step(acts[t])
        crc.append(np.stack([obs.reshape(count, -1).astype(np.int64).sum(1), rew.astype(np.int64), done.astype(np.int64)], 1))
        eps += int(done.sum()); lens += int(el.sum()); rets += int(er.sum())
    return obs0, np.stack(crc), {"episodes": eps, "ep_len_sum": lens, "ep_return_sum": rets, "env_steps": STEPS * count}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = msnake.shard_range(TOTAL, rank, world)
    obs0, crc, st = _run_shard(start, count)
    per_rank, total = msnake.gather_stats(st)
    q.put((rank, start, count, obs0, crc, st, per_rank, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_stats_allgather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    obs0_ref, crc_ref, st_ref = _run_shard(0, TOTAL)
    assert np.array_equal(np.concatenate([r[3] for r in res]), obs0_ref)
    assert np.array_equal(np.concatenate([r[4] for r in res], axis=1), crc_ref)
    for r in res:
        per_rank, total = r[6], r[7]
        assert [p["env_steps"] for p in per_rank] == [res[0][5]["env_steps"], res[1][5]["env_steps"]]
        for k in ("episodes", "ep_len_sum", "ep_return_sum", "env_steps"):
            assert total[k] == st_ref[k] == res[0][5][k] + res[1][5][k]


def test_gather_stats_single_process_identity():
    per_rank, total = msnake.gather_stats({"episodes": 4, "ep_len_sum": 80, "ep_return_sum": -2, "env_steps": 100})
    assert len(per_rank) == 1 and total["mean_ep_len"] == 20 and total["mean_ep_return"] == -0.5
