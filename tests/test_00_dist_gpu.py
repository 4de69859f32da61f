"""The N>1 path with the PRODUCT on a real GPU: two fresh processes (one gloo rank each, both on
cuda:0 of the one-GPU box) each own a shard built by msnake.make_sharded, step the HIP env on the
same global action tape, and all-gather their episode statistics with msnake.gather_stats.  The
shards' trajectories must equal the unsharded single-handle run (global env ids are the Philox
subsequence, so sharding must be invisible), and the gathered totals must equal that run's stats.

Reference counterpart: one worker process per env with seed + rank, src/baselines/common/vec_env/
subproc_vec_env.py:31-50 and src/utils.py:39; the epinfo aggregation of src/ppo_multi_agent.py:288,331.

This file sorts first on purpose: the children are spawned BEFORE this process initialises the GPU
(a process that has must not start other programs on the GPU box)."""
import os
import socket
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOTAL, STEPS, SEED, WORLD = 16384, 48, 23, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tape():
    return np.random.default_rng(77).integers(0, 5, (STEPS, TOTAL, 3)).astype(np.int32)


def _digest(obs):
    """per-env CRC32 of the observation bytes"""
    return np.array([zlib.crc32(o.tobytes()) for o in obs], np.uint32)


def _play(env, acts):
    """reset + STEPS steps; returns (obs0 crc, per-step [crc, rew, done, num_snakes, ep_r, ep_l], last obs)"""
    crc0 = _digest(env.reset())
    rows = []
    obs = None
    for t in range(STEPS):
        obs, rew, done, infos = env.step(acts[t])
        rows.append(np.stack([_digest(obs).astype(np.int64), rew.astype(np.int64), done.astype(np.int64),
                              infos._ns.astype(np.int64), infos._r.astype(np.int64), infos._l.astype(np.int64)], 1))
    return crc0, np.stack(rows), obs


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import msnake
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = msnake.make_sharded(TOTAL, dim=19, n_snakes=3, rules="snake_env", seed=SEED)  # RANK / WORLD_SIZE / LOCAL_RANK from the env
    start, count = msnake.shard_range(TOTAL, rank, world)
    assert env.num_envs == count and env.cfg.env_id_base == start and str(env.device) == "cuda:0"
    crc0, rows, last = _play(env, _tape()[:, start:start + count])
    st = env.stats()
    per_rank, total = msnake.gather_stats(st)
    q.put((rank, start, count, crc0, rows, last[:8].copy(), st, per_rank, total))
    dist.barrier()
    env.close()
    dist.destroy_process_group()


def _rccl_worker(port, q):
    """ONE rank on RCCL: the "nccl" branches of msnake.dist / bench.py run for real (a one-GPU box cannot
    give two RCCL ranks a card each, and RCCL refuses two ranks on one device)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import msnake
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    env = msnake.make_sharded(4096, dim=19, n_snakes=3, rules="snake_env", seed=SEED)
    assert env.num_envs == 4096 and env.cfg.env_id_base == 0 and str(env.device) == "cuda:0"
    env.reset_device()
    tape = torch.from_numpy(_tape()[:32, :4096].copy()).to(env.device)
    for t in range(32):
        env.step_device(tape[t])
    st = env.stats()
    per_rank, total = msnake.gather_stats(st)
    # the same collective by hand on a device tensor, and a barrier: RCCL itself moved these
    rec = torch.tensor([st["episodes"], st["env_steps"]], dtype=torch.int64, device="cuda:0")
    out = [torch.zeros_like(rec)]
    dist.all_gather(out, rec)
    dist.barrier()
    torch.cuda.synchronize()
    q.put({"backend": str(dist.get_backend()), "world": dist.get_world_size(), "per_rank": per_rank, "total": total, "st": st,
           "by_hand": out[0].tolist(), "by_hand_device": str(out[0].device), "nccl_version": list(torch.cuda.nccl.version())})
    env.close()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_nccl_branches():
    """First in this file: its child is spawned before this process has touched the GPU."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; run tests/test_00_dist_gpu.py first (it sorts first)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    r = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert "nccl" in r["backend"] and r["world"] == 1
    assert r["total"]["record_device"] == "cuda:0" and "nccl" in r["total"]["backend"]  # the record was a device tensor
    assert r["by_hand_device"] == "cuda:0" and r["by_hand"] == [r["st"]["episodes"], r["st"]["env_steps"]]
    assert len(r["per_rank"]) == 1 and r["per_rank"][0] == {k: r["st"][k] for k in ("episodes", "ep_len_sum", "ep_return_sum", "env_steps")}
    assert r["st"]["env_steps"] == 32 * 4096 and r["st"]["episodes"] > 1000 and r["st"]["errors"] == 0
    assert not torch.cuda.is_initialized()  # (the two-process test below still finds this process untouched)
    print("RCCL version", r["nccl_version"])


def test_two_processes_shard_the_hip_env_and_gather_stats():
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; run tests/test_00_dist_gpu.py first (it sorts first)")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(WORLD)], key=lambda x: x[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # only now does this process touch the GPU: the unsharded run
    import msnake
    whole = msnake.MultiSnakeVecEnv(TOTAL, dim=19, n_snakes=3, rules="snake_env", seed=SEED, device="cuda:0")
    crc0, rows, last = _play(whole, _tape())
    st = whole.stats()
    whole.close()
    assert [r[1] for r in res] == [0, TOTAL // 2] and [r[2] for r in res] == [TOTAL // 2, TOTAL // 2]
    assert np.array_equal(np.concatenate([r[3] for r in res]), crc0)
    assert np.array_equal(np.concatenate([r[4] for r in res], axis=1), rows)
    assert np.array_equal(res[1][5], last[TOTAL // 2:TOTAL // 2 + 8])
    assert st["errors"] == 0 and st["episodes"] > 1000
    for r in res:
        per_rank, total = r[7], r[8]
        assert per_rank == [{k: res[i][6][k] for k in msnake.dist.STAT_KEYS} for i in range(WORLD)]
        for k in ("episodes", "ep_len_sum", "ep_return_sum", "env_steps"):
            assert total[k] == st[k] == res[0][6][k] + res[1][6][k], k
