#!/usr/bin/env python3
"""bench.py -- env-steps/s of the HIP batched multi-snake step (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the N ranks itself (a
child `python -m torch.distributed.run ... bench.py ...` started before anything touches the GPU; the
parent only waits and exits with the child's code).

A "step" is ONE lockstep step of every env of the batch = one msnake_step kernel launch through
the C-ABI (include/msnake.h), actions already resident in HBM, observations / rewards / dones /
info written to HBM.  Workload at every N: 4 096 envs per GPU, 19x19 grid, 3 snakes, snake_env
rules, auto-reset on (BASELINE.json configs[2]; configs[3] = the same per-GPU shard on 8 GPUs), so
scaling is weak.  Envs never communicate: ranks shard the global env ids (msnake.make_sharded) with
no data-path collective; the one collective is msnake.gather_stats after the timed regions (RCCL).

Timing ("warm-up, time K steps, repeats, median"): after W warm-up steps the K-step
region is timed R = max(5, ceil(2048 / K)) times.  Every repeat is bracketed by barrier +
torch.cuda.synchronize() on both sides and timed with HIP events recorded on the launch stream
right around its K launches (the first event is queued behind one untimed msnake_render launch, so
the region does not start on an idle queue); per repeat the MAX over ranks is taken, and `value`, `ms_per_step` and
`roofline` all come from the MEDIAN repeat -- one clock for all three, independent of K.  The host
wall clock around the same region (it adds the launch latency of the first step and the wake-up
after the final synchronize, a fixed 30-40 us per region) is reported as `wall_ms_per_step`.

Prints ONE JSON line (rank 0).  Extra keys beside the driver's contract:
  roofline     : dominant kernel vs the HBM roofline (algorithmic bytes per launch from
                 msnake_algorithmic_bytes_per_env_step() x envs, duration = the median above;
                 `traffic` = PMC bytes per launch of the newest committed profiles/hbm_traffic_*.json,
                 named in `traffic_source` -- a recorded counter pass of this command, not a live one)
  rollout_tape : the same steps through the persistent msnake_rollout_tape launch, every step's
                 observations going to their own slice of a T x 16.3 MB buffer (never `value`)
  per_step_strided : the per-step launches again, every step's observations going to their own 16.3 MB slice
                 of a 64-step buffer, as the caller keeps them (ppo_multi_agent.py:178 mb_obs.append); the
                 headline leg re-writes ONE buffer (never `value`)
  selfplay_rollout : BASELINE configs[4] -- 4 096 envs x 2 snakes driving a 16-step self-play rollout
                 (msnake.selfplay.Runner: CnnPolicy learner + opponent on the same GPU, ppo_multi_agent.py:
                 170-216): frames/s of the rollout and the env's share of its time (never `value`)
  cpu_baseline : the CPU oracle (oracle/snake_oracle.c, the parity-pinned port of the reference)
                 timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
DIM, N_SNAKES = 19, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TAPE_STEPS = 256


def cpu_quota():
    """CPUs this process may actually burn (cgroup v2 cpu.max), e.g. 16 on a one-GPU box whose
    affinity mask shows all 256 hardware threads: more threads than that only oversubscribe."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return max(1, int(int(quota) / int(period))) if quota != "max" else 1 << 30
    except Exception:  # noqa: BLE001
        return 1 << 30


def cpu_baseline(actions_host, seconds=12.0):
    """The oracle on all host cores, same action tape, bounded to ~`seconds` of wall time.  One
    persistent thread team per 256-step tape (every thread owns an env range for all its steps,
    no per-step fork/join): the strongest fair figure for this port, not a strawman."""
    from oracle import snake_oracle
    n = actions_host.shape[1]
    cores = max(1, min(snake_oracle.lib().orc_max_threads(), len(os.sched_getaffinity(0)), cpu_quota()))
    ora = snake_oracle.Oracle(n, dim=DIM, n_snakes=N_SNAKES, rules="snake_env", seed=0)
    ora.reset()
    T = len(actions_host)
    ora.rollout(actions_host[:8], cores)  # warm
    t0 = time.perf_counter()
    k = 0
    while True:
        ora.rollout(actions_host, cores)
        k += T
        el = time.perf_counter() - t0
        if el >= seconds or k >= 4000000:
            break
    # the same port on ONE core (a short sample), so the per-core rate is on record too
    t1 = time.perf_counter()
    k1 = 0
    while time.perf_counter() - t1 < 2.0:
        ora.step(actions_host[k1 % T], threads=1)
        k1 += 1
    one = n * k1 / (time.perf_counter() - t1)
    return {"value": round(n * k / el, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{k} lockstep steps of {n} envs (19x19, 3 snakes, obs render included) in {el:.1f}s, OpenMP, "
                      f"one thread team per {T}-step tape, each thread owning an env range",
            "one_core_value": round(one, 1)}


def selfplay_rollout_leg(dev, n=4096, nsteps=16):
    """BASELINE configs[4]: n envs x 2 snakes x 19x19, self-play rollouts with the PyTorch policy in the loop
    (observations, actions, rewards and dones never leave the GPU).  Reports the rollout's frames/s and how much of
    a rollout is the env: the same `nsteps` msnake_step launches timed alone with HIP events.  Bounded: one warm-up
    rollout (MIOpen picks its convolution solvers there) and two timed ones."""
    import torch
    import msnake
    from msnake import selfplay
    env = msnake.MultiSnakeVecEnv(n, dim=DIM, n_snakes=2, rules="snake_env", seed=0, device=dev)
    H, W, _ = env.obs_shape
    torch.manual_seed(0)
    model, opp = selfplay.CnnPolicy((H, W, 3)).to(dev), selfplay.CnnPolicy((H, W, 3)).to(dev)
    runner = selfplay.Runner(env, model, [opp], nsteps, 0.99, 0.95)
    runner.run()
    torch.cuda.synchronize()
    wall = []
    for _ in range(2):
        t0 = time.perf_counter()
        runner.run()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
    full = runner.multi_step()[3]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.step_device(full)
    e0.record()
    for _ in range(nsteps):
        env.step_device(full)
    e1.record()
    torch.cuda.synchronize()
    env_ms = e0.elapsed_time(e1)
    st = env.stats()
    env.close()
    t = statistics.median(wall)
    return {"what": f"{n} envs x 2 snakes x {DIM}x{DIM} (BASELINE configs[4]), {nsteps}-step self-play rollout: "
                    "selfplay.Runner.run = learner + opponent CnnPolicy inference (fp32, MIOpen) + msnake_step + GAE, all on the GPU",
            "frames_per_s": round(n * nsteps / t, 1), "rollout_ms": round(t * 1e3, 2),
            "env_us_per_step": round(env_ms * 1e3 / nsteps, 2), "env_share_of_rollout": round(env_ms * 1e-3 / t, 5),
            "env_errors": int(st["errors"])}


def load_pmc_traffic(bytes_per_launch):
    """(HBM bytes per launch, source) from the newest committed PMC pass of THIS workload
    (profiles/hbm_traffic_*.json: separate FETCH_SIZE / WRITE_SIZE passes of this same command,
    FETCH doubled for gfx950; matched by its algorithmic bytes per launch), or (None, None)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "hbm_traffic_*.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except Exception:  # noqa: BLE001
            continue
        if d.get("algorithmic_bytes_per_launch") == bytes_per_launch and (best is None or d.get("recorded", "") >= best[1].get("recorded", "")):
            best = (f, d)
    if best is None:
        return None, None
    f, d = best
    return d.get("hbm_bytes_per_launch"), (f"profiles/{os.path.basename(f)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                            f"command recorded {d.get('recorded')}; not measured by this run)")


def load_gap():
    """(inter-kernel gap in us, source) of back-to-back msnake_step launches, from the newest committed span / gap record
    (profiles/r*_span_gap_4096.json: wave stamps on the GPU's 100 MHz clock) -- a recorded figure, not a live one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_span_gap_4096.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            d = json.load(fh)
        return float(d["reading"]["gap_us"]), f"profiles/{os.path.basename(files[-1])} (tools/span_gap.py; not measured by this run)"
    except Exception:  # noqa: BLE001
        return None, None


def self_launch(args):
    """--gpus N > 1 without a torchrun environment: become the launcher.  Nothing in this process has
    touched the GPU (torch is not even imported yet), the ranks are ordinary child processes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of the K-step region (default max(5, ceil(2048/K)))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--python-loop", action="store_true", help="issue launches from Python instead of msnake_step_tape")
    ap.add_argument("--no-rollout", action="store_true",
                    help="skip the secondary legs: msnake_rollout_tape, per_step_strided, selfplay_rollout (profiling)")
    ap.add_argument("--preroll", type=int, default=1, help="untimed msnake_render launches queued ahead of each timed region's first event")
    ap.add_argument("--legs", default="tape,strided,selfplay", help="which secondary legs to run (comma list; --no-rollout = none)")
    ap.add_argument("--obs-scale", type=int, default=1, choices=[1, 4], help="4 = the fused 84x84x9 WarpFrame series (profiling; not the headline workload)")
    ap.add_argument("--epb", type=int, default=0, help="msnake_config.envs_per_block (0 = auto)")
    ap.add_argument("--record-policy", default="auto", choices=["auto", "full", "short"])
    ap.add_argument("--store-policy", default="auto", choices=["auto", "plain", "stream"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + MSNAKE_BENCH_ONE_DEVICE=1 rehearses the N>1 path with every rank on cuda:0")
    args = ap.parse_args()
    legs = set() if args.no_rollout or args.python_loop else {x for x in args.legs.split(",") if x}

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")  # (selfplay_rollout leg: no exhaustive convolution search on first use)
    import torch
    import msnake

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torchrun -- also with ONE rank: the collective path (RCCL) is then the real one
    if all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT")):
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MSNAKE_BENCH_ONE_DEVICE") == "1":
            local_rank = 0  # rehearsal on a one-GPU box: all ranks share the card (RCCL refuses that, use gloo)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        dist = None
        torch.cuda.set_device(0)
    if args.gpus != world:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())
    n = args.envs_per_gpu
    K, Wm = args.steps, args.warmup
    R = args.repeats if args.repeats > 0 else max(5, math.ceil(2048 / max(1, K)))

    # rank r owns the global env ids [r*n, (r+1)*n): the shipped sharding API, not a hand-rolled one
    env = msnake.make_sharded(n * world, rank, world, device=dev, dim=DIM, n_snakes=N_SNAKES, rules="snake_env", seed=0,
                              envs_per_block=args.epb, record_policy=args.record_policy, obs_store_policy=args.store_policy,
                              obs_scale=args.obs_scale)
    assert env.num_envs == n and env.cfg.env_id_base == rank * n
    # synthetic input: uniform random actions in [0,5), one tape shared by the GPU and CPU runs
    # (large batches wrap a shorter tape, kept below 32 MiB: what cycles through the 256 MB Infinity Cache beside the launch
    #  decides whether the env state (640 B per env) and, in the plain-store range, the observation buffer stay resident --
    #  at 262 144 envs the same launch takes 191 us with an 8- or 32-step tape, 210 with 64 steps (201 MB), 222 with 128;
    #  a caller with a policy in the loop has no tape at all: its actions are one 12-byte-per-env tensor per step)
    T = TAPE_STEPS
    while T > 8 and T * n * N_SNAKES * 4 > (32 << 20):
        T //= 2
    tape_h = np.random.default_rng(1234 + rank).integers(0, 5, (T, n, N_SNAKES)).astype(np.int32)
    tape = torch.from_numpy(tape_h).to(dev)
    env.reset_device()
    H, Wd, C = env.obs_shape
    L, h = env._L, env._h
    obs, rew, done, info = env._obs, env._rew, env._done, env._info

    def run(nsteps, start):
        """nsteps launches; the tape wraps every T steps."""
        k = 0
        while k < nsteps:
            off = (start + k) % T
            m = min(nsteps - k, T - off)
            if args.python_loop:
                for j in range(m):
                    env.step_device(tape[off + j])
            else:
                msnake._capi.check(L.msnake_step_tape(h, tape[off].data_ptr(), N_SNAKES, m, obs.data_ptr(), 0,
                                                      rew.data_ptr(), done.data_ptr(), info.data_ptr(), 0,
                                                      env._stream()), "msnake_step_tape")
            k += m

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(Wm, 0)
    torch.cuda.synchronize()
    env.stats(reset=True)
    dev_ms, wall_ms = [], []
    for r in range(R):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        t0 = time.perf_counter()
        # one msnake_render launch (no env state or statistic changes) goes ahead of the first event, so
        # that the event fires behind a running kernel and the K timed launches follow it back to back
        # as in the steady state; without it every region starts on an idle queue and pays one launch
        # latency (~10 us, i.e. 0.5 us per step at K = 20) inside the timed interval
        for _ in range(args.preroll):
            env.render_device()
        ev0.record()
        run(K, Wm + r * K)
        ev1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dev_ms.append(ev0.elapsed_time(ev1))
        wall_ms.append((t1 - t0) * 1e3)

    st = env.stats()
    assert st["errors"] == 0 and st["env_steps"] == R * K * n, st

    # secondary figure: the same K steps through msnake_rollout_tape (ONE persistent launch per tape
    # chunk, env state kept in registers across steps).  Only usable when the actions of several
    # steps exist up front, so it is reported beside the headline, not as it.  Every step's
    # observations go to their own slice of a T-step buffer (T x 16.3 MB), so all of them reach HBM.
    rollout = None
    if rank == 0 and "tape" in legs:
        Tr = min(T, max(1, K))
        S = H * Wd * C
        obs_t = torch.empty((Tr, n, H, Wd, C), dtype=torch.uint8, device=dev)
        rew_t = torch.empty((Tr, n), dtype=torch.float32, device=dev)
        done_t = torch.empty((Tr, n), dtype=torch.uint8, device=dev)
        info_t = torch.empty((Tr, n, 4), dtype=torch.int32, device=dev)

        def run_rollout(nsteps):
            k = 0
            while k < nsteps:
                m = min(nsteps - k, Tr)
                msnake._capi.check(L.msnake_rollout_tape(h, tape.data_ptr(), N_SNAKES, m, obs_t.data_ptr(), n * S,
                                                         rew_t.data_ptr(), done_t.data_ptr(), info_t.data_ptr(), n,
                                                         env._stream()), "msnake_rollout_tape")
                k += m
        run_rollout(min(Wm, Tr))
        torch.cuda.synchronize()
        ro = []
        for _ in range(5):
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record()
            run_rollout(max(K, Tr))
            r1.record()
            torch.cuda.synchronize()
            ro.append(r0.elapsed_time(r1) * 1e3 / max(K, Tr))  # us per step
        rollout = statistics.median(ro)
        env.stats(reset=True)
        del obs_t

    # secondary figure: per-step launches whose observations go to a per-step slice, as Runner.run keeps them
    strided = None
    if rank == 0 and "strided" in legs:
        S = H * Wd * C
        Ts = max(2, min(64, (8 << 30) // (n * S)))  # 64 steps of 16.3 MB at the bench batch; at most 8 GB
        obs_s = torch.empty((Ts, n, H, Wd, C), dtype=torch.uint8, device=dev)

        def run_strided(nsteps):
            k = 0
            while k < nsteps:
                off = k % T
                m = min(nsteps - k, Ts, T - off)
                msnake._capi.check(L.msnake_step_tape(h, tape[off].data_ptr(), N_SNAKES, m, obs_s.data_ptr(), n * S,
                                                      rew.data_ptr(), done.data_ptr(), info.data_ptr(), 0,
                                                      env._stream()), "msnake_step_tape")
                k += m
        Ks = max(K, Ts)
        run_strided(Ts)
        torch.cuda.synchronize()
        ss = []
        for _ in range(5):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            env.render_device()
            s0.record()
            run_strided(Ks)
            s1.record()
            torch.cuda.synchronize()
            ss.append(s0.elapsed_time(s1) * 1e3 / Ks)
        strided = (statistics.median(ss), Ts, Ks)
        env.stats(reset=True)
        del obs_s

    # the only collective of the path: all-gather of the per-rank episode statistics (RCCL on GPUs)
    per_rank, total = msnake.gather_stats(st)
    cdev = dev if (dist and args.backend == "nccl") else torch.device("cpu")  # collectives run where the backend lives
    tmax = torch.tensor([dev_ms, wall_ms], dtype=torch.float64, device=cdev)
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dev_ms, wall_ms = tmax[0].tolist(), tmax[1].tolist()

    if rank == 0:
        assert len(per_rank) == world and total["env_steps"] == R * K * n * world, total
        med_ms = statistics.median(dev_ms)
        steps_per_region = K * n * world
        value = steps_per_region / (med_ms * 1e-3)
        bytes_per_env_step = env.algorithmic_bytes_per_env_step()
        bytes_per_launch = bytes_per_env_step * n
        launch_us = med_ms * 1e3 / K
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        traffic, traffic_src = load_pmc_traffic(bytes_per_launch)
        out = {
            "metric": "env-steps/sec (agent·step) at 4 096×19×19×3-snake, 1/2/4/8 GPU + CPU ref",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(med_ms / K, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{n} envs/GPU x {DIM}x{DIM} x {N_SNAKES} snakes, snake_env rules, auto-reset, "
                                   f"{'native' if args.obs_scale == 1 else 'fused x%d' % args.obs_scale} {H}x{Wd}x{C} uint8 obs, uniform random actions (BASELINE configs[2]"
                                   + ("/[3]" if world > 1 else "") + ")",
                       "envs_total": n * world, "agent_steps_per_s": round(value * N_SNAKES, 1),
                       "launch": "python per-step" if args.python_loop else "msnake_step_tape (C loop, 1 launch/step)",
                       "mean_episode_len": round(total["mean_ep_len"], 2),
                       "mean_episode_return": round(total["mean_ep_return"], 3),
                       "env_steps_per_rank": [p["env_steps"] for p in per_rank]},
            "timing": {"clock": "HIP events on the launch stream around each K-step region (the first event queued behind one "
                                "untimed msnake_render launch), max over ranks, median of repeats",
                       "repeats": R, "repeats_us_per_step": [round(x * 1e3 / K, 3) for x in dev_ms],
                       "wall_ms_per_step": round(statistics.median(wall_ms) / K, 6)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": env.kernel_name(), "launch_us": round(launch_us, 3),
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        gap_us, gap_src = load_gap()
        if gap_us is not None and n == ENVS_PER_GPU and launch_us > gap_us:
            # `launch_us` is the cadence of back-to-back launches = kernel span + the launch boundary (a fixed ~2 us between
            # the last acknowledged store of one kernel and the first wave of the next); `achieved` / `frac` above are by the
            # cadence.  The kernel's own span, and the fraction of peak it moves its bytes at, for the record:
            out["roofline"]["launch_boundary_gap_us"] = gap_us
            out["roofline"]["launch_boundary_gap_source"] = gap_src
            out["roofline"]["kernel_span_us_est"] = round(launch_us - gap_us, 3)
            out["roofline"]["frac_of_kernel_span_est"] = round(bytes_per_launch / (launch_us - gap_us) / 1e3 / HBM_PEAK_GBS, 4)
        if rollout is not None:
            out["rollout_tape"] = {
                "what": f"same steps via msnake_rollout_tape: one persistent launch per {Tr}-step tape chunk, "
                        f"observations of every step stored to their own slice of a {Tr} x {n * S / 1e6:.1f} MB buffer",
                "us_per_step": round(rollout, 3), "env_steps_per_s": round(n / rollout * 1e6, 1),
                "algorithmic_GBs": round(bytes_per_launch / rollout / 1e3, 1),
                "algorithmic_frac_of_hbm_peak": round(bytes_per_launch / rollout / 1e3 / HBM_PEAK_GBS, 4)}
        if strided is not None:
            us, Ts, Ks = strided
            out["per_step_strided"] = {
                "what": f"msnake_step launches whose observations go to per-step slices of a {Ts} x {n * H * Wd * C / 1e6:.1f} MB buffer "
                        f"(obs_step_stride = one batch), {Ks} steps, median of 5",
                "us_per_step": round(us, 3), "env_steps_per_s": round(n / us * 1e6, 1),
                "algorithmic_frac_of_hbm_peak": round(bytes_per_launch / us / 1e3 / HBM_PEAK_GBS, 4)}
        out["collective"] = {"backend": total.get("backend"), "record_device": total.get("record_device"), "ranks": len(per_rank)}
        if world == 1 and "selfplay" in legs:
            out["selfplay_rollout"] = selfplay_rollout_leg(dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(tape_h)
        print(json.dumps(out), flush=True)
    env.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
