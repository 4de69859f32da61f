"""Parity tests proper: the HIP path, called through the C-ABI, against the golden fixtures
recorded from the reference and against the CPU oracle on the same seeded inputs.  Bit-exact:
this is integer / byte work, there is no tolerance anywhere."""
import numpy as np
import pytest

from golden_util import blob_to_obs, crc_rows, edge_cases, load_tape, state_view, tape_names, unpack_state

pytestmark = pytest.mark.gpu

SUPPORTED_RULES = (0, 1, 2)


def _mk(**kw):
    import msnake
    return msnake.MultiSnakeVecEnv(**kw)


def _state(env, e):
    from oracle.snake_oracle import flat_to_state
    return flat_to_state(env.get_state_words(e))


def _set_state(env, e, st):
    from oracle.snake_oracle import state_to_flat
    env.set_state_words(e, state_to_flat(st, env.n_snakes))


@pytest.mark.parametrize("case", [c for c in edge_cases() if c["rules"] in SUPPORTED_RULES], ids=lambda c: c["name"])
def test_edge_case(case):
    rules = case["rules"]
    env = _mk(num_envs=1, dim=case["dim"], n_snakes=case["n_snakes"], n_fruits=case["n_fruits"], rules=rules,
              seed=case["seed"], env_id_base=case["env_id"], auto_reset=False)
    env.reset()
    _set_state(env, 0, case["state0"])
    assert state_view(_state(env, 0), rules) == state_view(case["state0"], rules)
    assert np.array_equal(env.render()[0], blob_to_obs(case["obs0"]))
    for i, step in enumerate(case["steps"]):
        if step["actions"] == "reset":
            obs = env.reset()
        else:
            obs, rew, done, infos = env.step(np.array([step["actions"]], np.int32))
            assert float(rew[0]) == step["reward"], (case["name"], i)
            assert bool(done[0]) == step["done"], (case["name"], i)
            assert infos[0]["num_snakes"] == step["num_snakes"], (case["name"], i)
        assert state_view(_state(env, 0), rules) == state_view(step["state"], rules), (case["name"], i)
        assert np.array_equal(obs[0], blob_to_obs(step["obs"])), (case["name"], i)
    assert env.stats()["errors"] == 0
    env.close()


@pytest.mark.parametrize("name", [n for n in tape_names() if load_tape(n)[0]["rules"] in SUPPORTED_RULES])
def test_golden_tape(name):
    meta, z = load_tape(name)
    rules, E, T = meta["rules"], meta["num_envs"], meta["steps"]
    env = _mk(num_envs=E, dim=meta["dim"], n_snakes=meta["n_snakes"], n_fruits=meta["n_fruits"], rules=rules,
              seed=meta["seed"], env_id_base=meta["env_id_base"], max_steps=meta["max_steps"],
              auto_reset=meta["auto_reset"])
    assert np.array_equal(env.reset(), z["obs0"])
    for e in range(E):
        assert state_view(_state(env, e), rules) == unpack_state(z, "s0_", 0, e, rules)
    full_t = {int(t): i for i, t in enumerate(z["full_obs_t"])}
    actions = z["actions"].astype(np.int32)
    for t in range(T):
        obs, rew, done, infos = env.step(actions[t])
        assert np.array_equal(rew, z["reward"][t]), (name, t)
        assert np.array_equal(done, z["done"][t].astype(bool)), (name, t)
        assert np.array_equal(infos._ns, z["num_snakes"][t].astype(np.int32)), (name, t)
        assert np.array_equal(infos._r, z["ep_return"][t]), (name, t)
        assert np.array_equal(infos._l, z["ep_len"][t]), (name, t)
        assert np.array_equal(crc_rows(obs), z["obs_crc"][t]), (name, t)
        if t in full_t:
            assert np.array_equal(obs, z["full_obs"][full_t[t]]), (name, t)
        if t % 16 == 0 or t == T - 1:
            for e in range(0, E, 3):
                assert state_view(_state(env, e), rules) == unpack_state(z, "st_", t, e, rules), (name, t, e)
    assert env.stats()["errors"] == 0
    env.close()


def _run_vs_oracle(num_envs, dim, n_snakes, n_fruits, rules, steps, seed, env_id_base=0, greedy=0.0, **tuning):
    """Same seeded action stream into the HIP env and the oracle; everything must match.
    `tuning`: msnake_config's launch-tuning fields (envs_per_block, record_policy, obs_store_policy)."""
    from oracle.snake_oracle import Oracle
    env = _mk(num_envs=num_envs, dim=dim, n_snakes=n_snakes, n_fruits=n_fruits, rules=rules, seed=seed,
              env_id_base=env_id_base, **tuning)
    ora = Oracle(num_envs, dim=dim, n_snakes=n_snakes, n_fruits=n_fruits, rules=rules, seed=seed,
                 env_id_base=env_id_base)
    assert np.array_equal(env.reset(), ora.reset())
    rs = np.random.default_rng(seed + 17)
    n_done = 0
    for t in range(steps):
        act = rs.integers(0, 5, (num_envs, n_snakes)).astype(np.int32)
        if greedy:  # bias towards "keep going" so snakes live longer and grow
            keep = rs.random((num_envs, n_snakes)) < greedy
            act = np.where(keep, 0, act).astype(np.int32)
        obs, rew, done, infos = env.step(act)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(act, threads=8)
        assert np.array_equal(rew, o_rew), t
        assert np.array_equal(done, o_done.astype(bool)), t
        assert np.array_equal(infos._ns, o_ns), t
        assert np.array_equal(infos._r, o_er), t
        assert np.array_equal(infos._l, o_el), t
        assert np.array_equal(obs, o_obs), t
        n_done += int(done.sum())
    st = env.stats()
    assert st["errors"] == 0
    assert st["episodes"] == n_done and st["env_steps"] == steps * num_envs
    for e in range(0, num_envs, max(1, num_envs // 64)):
        got, want = _state(env, e), ora.get_state(e)
        assert got == want, e
    env.close()
    return n_done


def test_config2_4096_envs_10x10_1snake():
    assert _run_vs_oracle(4096, 10, 1, 1, "snake_env", 200, seed=0) > 1000


def test_config3_4096_envs_19x19_3snakes():
    assert _run_vs_oracle(4096, 19, 3, 3, "snake_env", 200, seed=0) > 1000


@pytest.mark.parametrize("rules,dim,ns,nf", [("snake_env", 62, 3, 3), ("snake_env", 33, 2, 2), ("snake_env", 47, 3, 3),
                                             ("adversarial", 62, 3, 3), ("new_world", 62, 3, 20), ("new_world", 41, 4, 32)])
def test_large_boards_respawn_cell_split(rules, dim, ns, nf):
    """Boards up to MSNAKE_MAX_DIM: a respawned fruit's cell comes from x / dim, x % dim of a free-cell index
    x < dim^2, which the kernel computes with a multiplier from the launch glue instead of a division.  On a big
    board random play rarely eats, so the snakes are steered less (more straight runs over more fruits)."""
    n = 192 if dim > 50 else 256
    assert _run_vs_oracle(n, dim, ns, nf, rules, 160, seed=dim, greedy=0.7) >= 0


def test_config3_with_plain_observation_stores():
    """4 096 envs stream their observation stores by default; batches of 64-130 MiB use plain ones."""
    assert _run_vs_oracle(4096, 19, 3, 3, "snake_env", 60, seed=5, obs_store_policy="plain") > 300


def test_config5_shape_19x19_2snakes_long_lived():
    _run_vs_oracle(2048, 19, 2, 2, "snake_env", 300, seed=4, greedy=0.7)


def test_adversarial_vs_oracle():
    _run_vs_oracle(4096, 10, 2, 2, "adversarial", 200, seed=6)
    _run_vs_oracle(1024, 10, 3, 3, "adversarial", 300, seed=7, greedy=0.6)
    _run_vs_oracle(512, 19, 3, 3, "adversarial", 200, seed=8)


def test_new_world_4096_envs():
    _run_vs_oracle(4096, 19, 3, 3, "new_world", 100, seed=2)
    _run_vs_oracle(1024, 10, 2, 4, "new_world", 100, seed=3)
    _run_vs_oracle(256, 10, 4, 4, "new_world", 100, seed=3)


@pytest.mark.parametrize("stores", ["plain", "streaming"])
@pytest.mark.parametrize("num_envs", [1, 2, 5, 17, 63, 130])
def test_ragged_batch_sizes_and_unaligned_images(num_envs, stores):
    """3969-byte images are not 16-byte multiples: every misalignment 0..15 of the per-env image
    and the byte-store edges are exercised; guard bytes around the tensor must stay untouched.
    Both copy-out variants (plain and nt/streaming stores; the library picks one by batch size,
    msnake_config.obs_store_policy forces it) must produce the same bytes."""
    import torch
    from oracle.snake_oracle import Oracle
    env = _mk(num_envs=num_envs, dim=19, n_snakes=3, rules="snake_env", seed=9,
              obs_store_policy="stream" if stores == "streaming" else "plain")
    ora = Oracle(num_envs, dim=19, n_snakes=3, rules="snake_env", seed=9)
    H, W, C = env.obs_shape
    nbytes = num_envs * H * W * C
    for lead in (0, 3, 16):
        buf = torch.full((nbytes + 64,), 0xAB, dtype=torch.uint8, device=env.device)
        out = buf[lead:lead + nbytes].view(num_envs, H, W, C)
        got = env.reset_device(out=out).cpu().numpy()
        assert np.array_equal(got, ora.reset())
        act = np.random.default_rng(lead).integers(0, 5, (num_envs, 3)).astype(np.int32)
        obs, _, _, _ = env.step_device(torch.from_numpy(act).to(env.device), out=out)
        assert np.array_equal(obs.cpu().numpy(), ora.step(act)[0])
        host = buf.cpu().numpy()
        assert (host[:lead] == 0xAB).all() and (host[lead + nbytes:] == 0xAB).all()
    env.close()


@pytest.mark.parametrize("dim,n_snakes,rules,scale", [(19, 3, "snake_env", 4), (10, 1, "snake_env", 7),
                                                      (10, 2, "new_world", 7), (10, 2, "adversarial", 7),
                                                      (19, 2, "snake_env", 4), (19, 4, "new_world", 4)])
def test_fused_warpframe_is_pixel_replication(dim, n_snakes, rules, scale):
    """obs_scale = the reference's WarpFrame (src/utils.py:15-31): 84x84 frames.  cv2 is not
    installed, so parity with cv2.resize(INTER_AREA) is UNPINNED; the contract tested here is exact
    integer pixel replication (np.repeat on both axes) of the oracle's native frame."""
    from oracle.snake_oracle import Oracle
    n = 300
    env = _mk(num_envs=n, dim=dim, n_snakes=n_snakes, rules=rules, seed=12, obs_scale=scale)
    ora = Oracle(n, dim=dim, n_snakes=n_snakes, rules=rules, seed=12)
    assert env.obs_shape[:2] == (84, 84)
    up = lambda o: np.repeat(np.repeat(o, scale, axis=1), scale, axis=2)
    assert np.array_equal(env.reset(), up(ora.reset()))
    rs = np.random.default_rng(3)
    for t in range(40):
        act = rs.integers(0, 5, (n, n_snakes)).astype(np.int32)
        obs, rew, done, _ = env.step(act)
        o_obs, o_rew, o_done, _, _, _ = ora.step(act)
        assert np.array_equal(rew, o_rew) and np.array_equal(done, o_done.astype(bool)), t
        assert np.array_equal(obs, up(o_obs)), t
    env.close()


def test_sharding_is_invisible():
    """Two handles owning global envs [0,1024) and [1024,2048) == one handle owning [0,2048)."""
    whole = _mk(num_envs=2048, dim=19, n_snakes=3, rules="snake_env", seed=5)
    lo = _mk(num_envs=1024, dim=19, n_snakes=3, rules="snake_env", seed=5, env_id_base=0)
    hi = _mk(num_envs=1024, dim=19, n_snakes=3, rules="snake_env", seed=5, env_id_base=1024)
    assert np.array_equal(whole.reset(), np.concatenate([lo.reset(), hi.reset()]))
    rs = np.random.default_rng(1)
    for _ in range(60):
        act = rs.integers(0, 5, (2048, 3)).astype(np.int32)
        a = whole.step(act)
        b, c = lo.step(act[:1024]), hi.step(act[1024:])
        for k in range(3):
            assert np.array_equal(a[k], np.concatenate([b[k], c[k]]))
    for env in (whole, lo, hi):
        env.close()


def test_step_tape_equals_single_steps():
    import ctypes
    import torch
    import msnake
    n, T = 512, 16
    a = _mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=8)
    b = _mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=8)
    a.reset(); b.reset()
    tape = torch.randint(0, 5, (T, n, 3), dtype=torch.int32, device=a.device)
    H, W, C = a.obs_shape
    obs = torch.empty((T, n, H, W, C), dtype=torch.uint8, device=a.device)
    rew = torch.empty((T, n), dtype=torch.float32, device=a.device)
    done = torch.empty((T, n), dtype=torch.uint8, device=a.device)
    info = torch.empty((T, n, 4), dtype=torch.int32, device=a.device)
    msnake._capi.check(a._L.msnake_step_tape(a._h, tape.data_ptr(), 3, T, obs.data_ptr(), n * H * W * C,
                                             rew.data_ptr(), done.data_ptr(), info.data_ptr(), n, a._stream()))
    for t in range(T):
        o, r, d, i = b.step_device(tape[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t]) and torch.equal(i, info[t])
    a.close(); b.close()


@pytest.mark.parametrize("rules,dim,ns,nf,scale", [("snake_env", 19, 3, 3, 1), ("new_world", 10, 2, 4, 1),
                                                   ("adversarial", 10, 3, 3, 1), ("snake_env", 19, 2, 2, 4)])
def test_rollout_tape_persistent_launch_equals_single_steps(rules, dim, ns, nf, scale):
    """msnake_rollout_tape (one persistent launch for T steps) == T msnake_step launches."""
    import torch
    import msnake
    n, T = (777, 48) if scale == 1 else (150, 12)
    a = _mk(num_envs=n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=21, obs_scale=scale)
    b = _mk(num_envs=n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=21, obs_scale=scale)
    a.reset(); b.reset()
    tape = torch.randint(0, 5, (T, n, ns), dtype=torch.int32, device=a.device)
    H, W, C = a.obs_shape
    obs = torch.empty((T, n, H, W, C), dtype=torch.uint8, device=a.device)
    rew = torch.empty((T, n), dtype=torch.float32, device=a.device)
    done = torch.empty((T, n), dtype=torch.uint8, device=a.device)
    info = torch.empty((T, n, 4), dtype=torch.int32, device=a.device)
    msnake._capi.check(a._L.msnake_rollout_tape(a._h, tape.data_ptr(), ns, T, obs.data_ptr(), n * H * W * C,
                                                rew.data_ptr(), done.data_ptr(), info.data_ptr(), n, a._stream()))
    for t in range(T):
        o, r, d, i = b.step_device(tape[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t]) and torch.equal(i, info[t]), t
    for e in range(0, n, 37):
        assert _state(a, e) == _state(b, e)
    assert a.stats() == b.stats()
    # in-place outputs (stride 0): the last step's outputs remain
    msnake._capi.check(a._L.msnake_rollout_tape(a._h, tape.data_ptr(), ns, 5, obs.data_ptr(), 0, rew.data_ptr(),
                                                done.data_ptr(), info.data_ptr(), 0, a._stream()))
    for t in range(5):
        o, r, d, i = b.step_device(tape[t])
    assert torch.equal(o, obs[0]) and torch.equal(r, rew[0]) and torch.equal(d, done[0])
    a.close(); b.close()


def test_rollout_tape_with_bodies_longer_than_one_chunk():
    """Bodies over 64 cells re-read ring cells the same wave stored earlier in the same launch."""
    import torch
    import msnake
    from oracle.snake_oracle import Oracle
    case = [c for c in edge_cases() if c["name"] == "S_full_board_10"][0]
    env = _mk(num_envs=3, dim=10, n_snakes=1, rules="snake_env", seed=case["seed"], env_id_base=case["env_id"], auto_reset=False)
    ora = Oracle(3, dim=10, n_snakes=1, rules="snake_env", seed=case["seed"], env_id_base=case["env_id"], auto_reset=False)
    env.reset(); ora.reset()
    st = dict(case["state0"]); st["grow_to"] = [90]; st["snakes"] = [st["snakes"][0][:80]]  # 80 cells, shrinking to 90? no: stays
    # walk the boustrophedon backwards is impossible; instead let it advance along the free last row
    for e in range(3):
        _set_state(env, e, st); ora.set_state(e, st)
    T = 6
    acts = np.zeros((T, 3, 1), np.int32)  # keep heading (-1,0) along row 9... then it leaves the grid
    tape = torch.from_numpy(acts).to(env.device)
    H, W, C = env.obs_shape
    obs = torch.empty((T, 3, H, W, C), dtype=torch.uint8, device=env.device)
    rew = torch.empty((T, 3), dtype=torch.float32, device=env.device)
    done = torch.empty((T, 3), dtype=torch.uint8, device=env.device)
    msnake._capi.check(env._L.msnake_rollout_tape(env._h, tape.data_ptr(), 1, T, obs.data_ptr(), 3 * H * W * C,
                                                  rew.data_ptr(), done.data_ptr(), None, 3, env._stream()))
    for t in range(T):
        o_obs, o_rew, o_done, _, _, _ = ora.step(acts[t])
        assert np.array_equal(obs[t].cpu().numpy(), o_obs), t
        assert np.array_equal(rew[t].cpu().numpy(), o_rew) and np.array_equal(done[t].cpu().numpy(), o_done), t
    env.close()


def test_step_without_observation_buffer():
    """obs_dev = NULL skips the render; everything else is unchanged."""
    import torch
    import msnake
    a = _mk(num_envs=300, dim=19, n_snakes=3, rules="snake_env", seed=2)
    b = _mk(num_envs=300, dim=19, n_snakes=3, rules="snake_env", seed=2)
    a.reset(); b.reset()
    acts = torch.randint(0, 5, (30, 300, 3), dtype=torch.int32, device=a.device)
    for t in range(30):
        msnake._capi.check(a._L.msnake_step(a._h, acts[t].data_ptr(), 3, None, a._rew.data_ptr(), a._done.data_ptr(),
                                            a._info.data_ptr(), a._stream()))
        _, r, d, i = b.step_device(acts[t])
        assert torch.equal(a._rew, r) and torch.equal(a._done, d) and torch.equal(a._info, i)
    assert np.array_equal(a.render(), b.render())
    a.close(); b.close()


def test_errors_are_exceptions():
    import torch
    with pytest.raises(RuntimeError, match="n_snakes"):
        _mk(num_envs=4, dim=19, n_snakes=5, rules="snake_env")
    with pytest.raises(RuntimeError, match="dim"):
        _mk(num_envs=4, dim=200, n_snakes=3, rules="snake_env")
    env = _mk(num_envs=4, dim=10, n_snakes=2, rules="snake_env")
    env.reset()
    with pytest.raises(ValueError):
        env.step(np.zeros((4, 1), np.int32))
    with pytest.raises(ValueError):
        env.step(np.zeros((3, 2), np.int32))
    with pytest.raises(RuntimeError, match="snakes"):
        env.set_state_words(0, np.zeros(8, np.int32))
    acts = torch.zeros((4, 2), dtype=torch.int32, device="cuda")
    for bad in (torch.empty((4, 12, 12, 3), dtype=torch.uint8, device="cuda"),      # too small: the kernel would overrun it
                torch.empty((4, 12, 12, 9), dtype=torch.float32, device="cuda"),
                torch.empty((4, 12, 12, 18), dtype=torch.uint8, device="cuda")[..., ::2],  # not contiguous
                torch.empty((4, 12, 12, 9), dtype=torch.uint8)):                    # host memory
        with pytest.raises(ValueError, match="out must be"):
            env.step_device(acts, out=bad)
        with pytest.raises(ValueError, match="out must be"):
            env.reset_device(out=bad)
    env.step_device(acts, out=torch.empty((4, 12, 12, 9), dtype=torch.uint8, device="cuda"))
    env.close()
    with pytest.raises(RuntimeError, match="handle"):
        env.step_device(torch.zeros((4, 2), dtype=torch.int32, device="cuda"))


def test_selfplay_ppo_rollouts_end_to_end():
    """BASELINE configs[4] shape at reduced batch: 2 snakes, 19x19, PyTorch policy on the same GPU
    driving the HIP env through step_device, a few PPO updates; checks plumbing, not learning."""
    import torch
    from msnake import selfplay
    env = _mk(num_envs=256, dim=19, n_snakes=2, rules="snake_env", seed=3)
    model, hist = selfplay.learn(env, nsteps=16, total_timesteps=256 * 16 * 3, nminibatches=4, log_fn=None)
    assert len(hist) == 3 and hist[-1]["total_timesteps"] == 256 * 16 * 3
    for k in ("eprewmean 100", "eplenmean", "policy_loss", "value_loss", "policy_entropy", "approxkl", "clipfrac",
              "explained_variance", "num_opponents", "fps"):
        assert k in hist[-1]
    assert np.isfinite(hist[-1]["policy_loss"]) and 0 < hist[-1]["policy_entropy"] <= np.log(5) + 1e-3
    st = env.stats()
    assert st["env_steps"] == 256 * 16 * 3 and st["errors"] == 0 and st["episodes"] > 0
    assert next(model.parameters()).is_cuda
    env.close()
    # the episode infos a rollout reports are exactly what the env's own totals say (Monitor's r and l)
    env = _mk(num_envs=128, dim=10, n_snakes=2, rules="snake_env", seed=4)
    pol = selfplay.CnnPolicy((12, 12, 3)).to(env.device)
    runner = selfplay.Runner(env, pol, [pol], 48, 0.99, 0.95)
    env.stats(reset=True)
    epinfos = runner.run()[-1]
    st = env.stats()
    assert st["episodes"] == len(epinfos) > 20
    assert st["ep_len_sum"] == sum(e["l"] for e in epinfos)
    assert st["ep_return_sum"] == round(sum(e["r"] for e in epinfos))
    env.close()


def test_vecenv_surface_like_the_reference_runner_uses_it():
    """ppo_multi_agent.Runner's view of the env (src/ppo_multi_agent.py:145-216): attributes, tuple
    actions of length 2/3 even with fewer snakes, NumPy returns, info dicts, reset/close."""
    import msnake
    env = msnake.make("snake-multiple-test-v0", 33, n_snakes=1, seed=4)
    assert env.num_envs == 33 and env.action_space.n == 5
    assert env.observation_space.shape == (21, 21, 6) and env.obs_shape == (21, 21, 9)
    assert env.unwrapped is env and env.rules == "snake_env"
    obs = env.reset()
    assert obs.dtype == np.uint8 and obs.shape == (33, 21, 21, 9)
    full_actions = list(zip([1] * 33, [1] * 33))  # MultiModel.multi_step with one snake: tuples of 2
    obs2, rews, dones, infos = env.step(full_actions)
    assert rews.dtype == np.float32 and dones.dtype == np.bool_ and len(infos) == 33
    assert (1.0 - dones).shape == (33,)  # ppo_multi_agent.py:207
    assert all(set(i) >= {"ale.lives", "num_snakes"} for i in infos)
    with pytest.raises(RuntimeError):
        env.step_wait()  # no step_async pending
    env.step_async(full_actions)
    env.step_wait()
    total_done = 0
    for _ in range(40):
        _, _, dones, infos = env.step(full_actions)
        total_done += int(dones.sum())
        for d, i in zip(dones, infos):
            assert ("episode" in i) == bool(d)
            if d:
                assert set(i["episode"]) == {"r", "l", "t"} and i["episode"]["l"] >= 1
    st = env.stats(reset=True)
    assert st["episodes"] >= total_done > 0 and st["env_steps"] == 33 * 42
    assert env.stats()["episodes"] == 0 and env.stats()["env_steps"] == 0
    assert np.array_equal(env.render(), env._obs.cpu().numpy())
    env.close(); env.close()
    # pinned zero-copy host views give the same numbers
    a = msnake.make("snake-multiple-test-v0", 50, n_snakes=3, seed=9)
    b = msnake.make("snake-multiple-test-v0", 50, n_snakes=3, seed=9, host_views=True)
    assert np.array_equal(a.reset(), b.reset())
    for t in range(15):
        act = np.random.default_rng(t).integers(0, 5, (50, 3))
        ra, rb = a.step(act), b.step(act)
        assert all(np.array_equal(x, y) for x, y in zip(ra[:3], rb[:3]))
        assert [i for i in ra[3]] == [i for i in rb[3]] or all(
            {k: v for k, v in i.items() if k != "episode"} == {k: v for k, v in j.items() if k != "episode"}
            for i, j in zip(ra[3], rb[3]))
    a.close(); b.close()
    for env_id, shape in (("snake-new-multiple-v0", (12, 12, 6)), ("snake-adversarial-v0", (12, 12, 9))):
        e2 = msnake.make(env_id, 8, n_snakes=2)
        assert e2.obs_shape == shape and e2.reset().shape == (8,) + shape
        e2.close()


def _serpentine(dim, length, start_row, rs):
    """A self-avoiding boustrophedon body of `length` cells starting in row `start_row` (head last
    visited, i.e. head first in the list), so bodies far longer than one 64-cell chunk are legal."""
    path = []
    for r in range(start_row, dim):
        cols = range(dim) if (r - start_row) % 2 == 0 else range(dim - 1, -1, -1)
        path += [[c, r] for c in cols]
    path = path[:length]
    return path[::-1]


@pytest.mark.parametrize("rules", ["snake_env", "new_world", "adversarial"])
def test_fuzz_long_bodies_against_oracle(rules):
    """Hand-built states with bodies of 50..250 cells (up to 4 chunks of 64): the ring path of the
    collision scan, the respawn occupancy and the painter, which random play almost never reaches."""
    from oracle.snake_oracle import Oracle
    rs = np.random.default_rng(99)
    dim, n, ns = 19, 96, 2
    nf = 2 if rules != "new_world" else 3
    env = _mk(num_envs=n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=13)
    ora = Oracle(n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=13)
    env.reset(); ora.reset()
    for e in range(n):
        la, lb = int(rs.integers(50, 250)), int(rs.integers(1, 60))
        a = _serpentine(dim, la, 0, rs)
        b = _serpentine(dim, lb, 15, rs)
        # head of a is at the end of its last (partial) row, moving along that row
        hr = a[0][1]
        va = [1, 0] if (hr % 2 == 0) else [-1, 0]
        st = {"snakes": [a, b], "fruits": [[int(rs.integers(0, dim)), int(rs.integers(0, dim))] for _ in range(nf)],
              "vels": [va, [0, 0]], "grow_to": [la + int(rs.integers(0, 3)), lb + 2], "t": int(rs.integers(0, 100)),
              "ctr": int(rs.integers(0, 1000)), "alive": [True, True], "in_dead": [False, False], "spare_fruits": int(rs.integers(0, 3))}
        _set_state(env, e, st); ora.set_state(e, st)
    assert np.array_equal(env.render(), ora.render())
    for t in range(40):
        act = rs.integers(0, 5, (n, ns)).astype(np.int32)
        act[:, 0] = np.where(rs.random(n) < 0.8, 0, act[:, 0])  # mostly keep going so the long snake survives a while
        obs, rew, done, infos = env.step(act)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(act)
        assert np.array_equal(rew, o_rew) and np.array_equal(done, o_done.astype(bool)), t
        assert np.array_equal(infos._ns, o_ns) and np.array_equal(infos._l, o_el), t
        assert np.array_equal(obs, o_obs), t
    for e in range(0, n, 7):
        assert _state(env, e) == ora.get_state(e), e
    assert env.stats()["errors"] == 0
    env.close()


@pytest.mark.parametrize("dim", [7, 19])
def test_respawn_with_and_without_shared_cells(dim):
    """snake_env respawns take the map-free path (k-th free cell by rank over the body registers) only while no two
    pieces share a cell; installed states whose bodies cross or stack must fall back to the occupancy map, and a
    head that runs into a body, into another head or out of the grid in the very step it eats must be counted once.
    Every env gets a fruit right in front of a moving head so that the first steps respawn."""
    from oracle.snake_oracle import Oracle
    rs = np.random.default_rng(4242 + dim)
    n, ns = 192, 3
    env = _mk(num_envs=n, dim=dim, n_snakes=ns, rules="snake_env", seed=5)
    ora = Oracle(n, dim=dim, n_snakes=ns, n_fruits=ns, rules="snake_env", seed=5)
    env.reset(); ora.reset()
    step = {1: (1, 0), 2: (0, 1), 3: (-1, 0), 4: (0, -1)}
    first = np.zeros((n, ns), np.int32)
    for e in range(n):
        kind = e % 4  # 0 disjoint, 1 bodies cross, 2 two snakes stacked on one cell, 3 a snake with one cell twice
        snakes, vels = [], []
        for j in range(ns):
            ln = int(rs.integers(1, 6))
            r = 1 + 2 * j if dim > 6 else j
            c0 = int(rs.integers(ln, dim - 1))
            snakes.append([[c0 - i, r] for i in range(ln)])  # head first, pointing to +c0
            vels.append([1, 0])
        if kind == 1:
            snakes[1] = [[snakes[0][-1][0], snakes[0][-1][1] + 1 - i] for i in range(3)][::-1]  # a column through snake 0's tail
            vels[1] = [0, 1]
        elif kind == 2:
            snakes[2] = [list(snakes[1][0])]
            vels[2] = [0, 0]
        elif kind == 3 and len(snakes[0]) > 1:
            snakes[0] = snakes[0] + [list(snakes[0][-1])]  # (tail cell twice; a body crossing its own HEAD cell cannot be rendered in the reference)
        fruits = []
        for j in range(ns):
            a = int(rs.integers(1, 5))
            first[e, j] = a if rs.random() < 0.8 else 0
            v = step[a] if first[e, j] else tuple(vels[j])
            if v == (-vels[j][0], -vels[j][1]) or v == (0, 0):
                v = tuple(vels[j])  # a reversal is ignored: the snake keeps going
            h = snakes[j][0]
            fc = [h[0] + v[0], h[1] + v[1]] if rs.random() < 0.85 else [int(rs.integers(0, dim)), int(rs.integers(0, dim))]
            fruits.append([min(max(fc[0], 0), dim - 1), min(max(fc[1], 0), dim - 1)])
        st = {"snakes": snakes, "fruits": fruits, "vels": vels, "grow_to": [len(b) + int(rs.integers(0, 3)) for b in snakes],
              "t": int(rs.integers(0, 50)), "ctr": int(rs.integers(0, 500)), "alive": [True] * ns, "in_dead": [False] * ns, "spare_fruits": 0}
        _set_state(env, e, st); ora.set_state(e, st)
    assert np.array_equal(env.render(), ora.render())
    ate = 0
    for t in range(24):
        act = first if t == 0 else rs.integers(0, 5, (n, ns)).astype(np.int32)
        obs, rew, done, infos = env.step(act)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(act)
        ate += int((o_rew > 0).sum())
        assert np.array_equal(rew, o_rew) and np.array_equal(done, o_done.astype(bool)), t
        assert np.array_equal(infos._ns, o_ns) and np.array_equal(infos._l, o_el), t
        assert np.array_equal(obs, o_obs), t
        if t in (0, 1, 5):
            for e in range(n):
                assert _state(env, e) == ora.get_state(e), (t, e)
    assert ate > n // 4  # the set-up did produce respawns
    assert env.stats()["errors"] == 0
    env.close()


@pytest.mark.parametrize("dim,ns,nf,rules", [(2, 1, 1, "snake_env"), (3, 2, 2, "snake_env"), (5, 3, 3, "adversarial"),
                                              (30, 3, 3, "snake_env"), (62, 3, 3, "snake_env"), (62, 4, 32, "new_world"),
                                              (45, 4, 0, "new_world"), (7, 1, 9, "new_world")])
def test_unusual_grid_sizes(dim, ns, nf, rules):
    """Grid sizes from 2x2 to the 62x62 maximum (LDS image up to 48 KB: one env per workgroup),
    fruit counts 0..32, against the oracle."""
    n = 37
    _run_vs_oracle(n, dim, ns, nf, rules, 60, seed=31, greedy=0.5)


@pytest.mark.parametrize("dim,ns", [(3, 2), (3, 3), (4, 3), (5, 2), (6, 3)])
def test_small_boards_stress_respawn_ordering(dim, ns):
    """Tiny boards: eaten fruits respawn next to (or under the next head of) the other snakes all the
    time, so the order "snake s moves, its fruits respawn, snake s+1 moves" ([S]:119-141) decides the
    outcome in a large share of the steps; several snakes eat in one step; boards fill up."""
    assert _run_vs_oracle(4096, dim, ns, ns, "snake_env", 150, seed=41 + dim, greedy=0.3) > 1000


@pytest.mark.parametrize("rules", ["snake_env", "adversarial", "new_world"])
def test_draw_counter_crosses_32_bits(rules):
    """The Philox draw counter is 64 bit; env e starts 3*e draws below 2**32, so the in-register
    draw cache, the draws parked in the record between launches and the counter's carry all cross
    the boundary at different phases (resets take 12 draws at once, respawns one)."""
    from oracle.snake_oracle import Oracle
    n, dim, ns = 96, 6, 3
    nf = 3 if rules != "new_world" else 4
    env = _mk(num_envs=n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=21)
    ora = Oracle(n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=21)
    env.reset(); ora.reset()
    for e in range(n):
        st = ora.get_state(e)
        st["ctr"] = (1 << 32) - (3 * e if rules != "new_world" else e % 10)  # new_world draws rarely
        _set_state(env, e, st); ora.set_state(e, st)
    rs = np.random.default_rng(5)
    for t in range(120):
        act = rs.integers(0, 5, (n, ns)).astype(np.int32)
        obs, rew, done, infos = env.step(act)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(act)
        assert np.array_equal(obs, o_obs) and np.array_equal(rew, o_rew) and np.array_equal(done, o_done.astype(bool)), t
    crossed = 0
    for e in range(n):
        got, want = _state(env, e), ora.get_state(e)
        assert got == want, e
        crossed += got["ctr"] >= (1 << 32)
    assert crossed > n // 4
    env.close()


def test_set_state_keeps_the_logging_totals():
    """Episode totals live in the env records; installing a canonical state (which does not carry
    them) must not lose what msnake_get_stats will report."""
    env = _mk(num_envs=32, dim=6, n_snakes=2, rules="snake_env", seed=3)
    env.reset()
    rs = np.random.default_rng(1)
    for _ in range(60):
        env.step(rs.integers(0, 5, (32, 2)).astype(np.int32))
    before = env.stats()
    assert before["episodes"] > 10
    for e in range(32):
        _set_state(env, e, _state(env, e))
    assert env.stats() == before
    env.close()


def _random_configs(k, seed):
    rs = np.random.default_rng(seed)
    out = []
    for i in range(k):
        rules = ["snake_env", "new_world", "adversarial"][i % 3]
        ns = int(rs.integers(1, 5 if rules == "new_world" else 4))
        nf = int(rs.integers(0, 12)) if rules == "new_world" else ns
        out.append(dict(rules=rules, dim=int(rs.integers(2, 24)), n_snakes=ns, n_fruits=nf,
                        max_steps=int(rs.choice([3, 17, 60, 2000])), auto_reset=bool(rs.integers(0, 2)),
                        seed=int(rs.integers(0, 2**31)), env_id_base=int(rs.integers(0, 2**40)),
                        num_envs=int(rs.integers(1, 90))))
    return out


@pytest.mark.parametrize("cfg", _random_configs(30, 2024), ids=lambda c: f"{c['rules']}-{c['dim']}x{c['n_snakes']}x{c['n_fruits']}-"
                                                                         f"T{c['max_steps']}-ar{int(c['auto_reset'])}-n{c['num_envs']}")
def test_random_configurations_against_oracle(cfg):
    """Thirty configurations drawn at random over every create() argument (rule set, grid size, snake
    and fruit counts, step cap, auto-reset on/off, seed, env-id base, ragged batch sizes): reset + 80
    steps, everything compared with the oracle, states at the end."""
    from oracle.snake_oracle import Oracle
    env = _mk(**cfg)
    ora = Oracle(**cfg)
    n, ns = cfg["num_envs"], cfg["n_snakes"]
    assert np.array_equal(env.reset(), ora.reset())
    rs = np.random.default_rng(cfg["seed"] % 1000)
    for t in range(80):
        act = rs.integers(-1, 6, (n, ns)).astype(np.int32)  # includes the invalid codes -1 and 5
        obs, rew, done, infos = env.step(act)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(act)
        assert np.array_equal(obs, o_obs), t
        assert np.array_equal(rew, o_rew) and np.array_equal(done, o_done.astype(bool)), t
        assert np.array_equal(infos._ns, o_ns) and np.array_equal(infos._r, o_er) and np.array_equal(infos._l, o_el), t
        if not cfg["auto_reset"] and t % 20 == 19:  # without auto-reset the caller resets, like a bare gym loop
            assert np.array_equal(env.reset(), ora.reset())
    for e in range(n):
        assert _state(env, e) == ora.get_state(e), e
    assert env.stats()["errors"] == 0
    env.close()


def test_step_is_hip_graph_capturable():
    """msnake_step does no allocation, copy or synchronisation, so a caller can capture it (with
    its policy) into a HIP graph; replays must equal direct launches."""
    import torch
    n = 512
    a = _mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=17)
    b = _mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=17)
    a.reset(); b.reset()
    acts = torch.zeros((n, 3), dtype=torch.int32, device=a.device)
    tape = torch.randint(0, 5, (25, n, 3), dtype=torch.int32, device=a.device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on the side stream, as graph capture wants
        a.step_device(acts); b.step_device(acts)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a.step_device(acts)
    # the capture itself did not execute the step
    for t in range(25):
        acts.copy_(tape[t])
        g.replay()
        o, r, d, i = b.step_device(tape[t])
        assert torch.equal(a._obs, o) and torch.equal(a._rew, r) and torch.equal(a._done, d) and torch.equal(a._info, i), t
    a.close(); b.close()


# ---------------------------------------------------------------------------------------------------
# round 2: large batches (BASELINE configs[3] sizes and beyond), bulk state I/O, configs[4] at full size
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("num_envs,base,steps", [(16384, 0, 40), (32768, 7 * 4096, 48), (65536, 0, 40)])
def test_large_batches_byte_for_byte(num_envs, base, steps):
    """> 8 192 envs switch to 4 envs per workgroup, the short (128-byte) record, many 64-workgroup
    groups of the XCD-aware env mapping, plain observation stores (62 / 124 MiB per step) and, at
    65 536 envs (248 MiB), streaming ones again.  32 768 envs = BASELINE configs[3]; its run owns the
    global ids of rank 7 of 8 (env_id_base 7 x 4 096).  Every observation byte is compared."""
    assert _run_vs_oracle(num_envs, 19, 3, 3, "snake_env", steps, seed=3, env_id_base=base) > num_envs


@pytest.mark.parametrize("rules,ns,nf", [("new_world", 3, 3), ("adversarial", 2, 2)])
def test_large_batches_other_rule_sets(rules, ns, nf):
    _run_vs_oracle(20000, 10, ns, nf, rules, 30, seed=14)


@pytest.mark.parametrize("policy", ["full", "short"])
def test_record_policy_is_invisible(policy):
    """snake_env / adversarial steps run on the full 256-byte record (Philox draws parked in its upper
    half) or on its first 128 bytes; the library picks by batch size, msnake_config.record_policy forces it."""
    assert _run_vs_oracle(3000, 19, 3, 3, "snake_env", 120, seed=15, record_policy=policy) > 1000
    _run_vs_oracle(1500, 10, 3, 3, "adversarial", 150, seed=16, greedy=0.5, record_policy=policy)
    _run_vs_oracle(700, 6, 3, 3, "snake_env", 120, seed=17, greedy=0.3, record_policy=policy, envs_per_block=4)


def test_config4_selfplay_at_full_size():
    """BASELINE configs[4]: 4 096 envs x 2 snakes x 19x19 driving self-play PPO rollouts end to end
    (PyTorch policy on the same GPU, HIP env through step_device): two updates of 16-step rollouts,
    the env's own episode totals cross-checked against what the rollouts reported; and the same env
    shape against the oracle.  Reference: ppo_multi_agent.py:170-216 (Runner.run), :231-404 (learn)."""
    from msnake import selfplay
    n, nsteps = 4096, 16
    env = _mk(num_envs=n, dim=19, n_snakes=2, rules="snake_env", seed=31)
    model, hist = selfplay.learn(env, nsteps=nsteps, total_timesteps=n * nsteps * 2, nminibatches=8, noptepochs=1, log_fn=None)
    assert len(hist) == 2 and hist[-1]["total_timesteps"] == n * nsteps * 2
    assert np.isfinite(hist[-1]["policy_loss"]) and np.isfinite(hist[-1]["value_loss"])
    st = env.stats()
    assert st["env_steps"] == n * nsteps * 2 and st["errors"] == 0 and st["episodes"] > 1000
    env.close()
    env = _mk(num_envs=n, dim=19, n_snakes=2, rules="snake_env", seed=32)
    pol = selfplay.CnnPolicy((21, 21, 3)).to(env.device)
    runner = selfplay.Runner(env, pol, [pol], nsteps, 0.99, 0.95)
    env.stats(reset=True)
    epinfos = runner.run()[-1]
    st = env.stats()
    assert st["episodes"] == len(epinfos) > 500
    assert st["ep_len_sum"] == sum(e["l"] for e in epinfos) and st["ep_return_sum"] == round(sum(e["r"] for e in epinfos))
    env.close()
    assert _run_vs_oracle(4096, 19, 2, 2, "snake_env", 150, seed=33) > 1000


@pytest.mark.parametrize("rules,dim,ns,nf", [("snake_env", 19, 3, 3), ("new_world", 10, 3, 5), ("adversarial", 10, 3, 3)])
def test_bulk_state_round_trip(rules, dim, ns, nf):
    """msnake_get_state_all / msnake_set_state_all: one packing kernel + one copy for all 4 096 envs;
    the blob must hold exactly what 4 096 msnake_get_state calls return, restore the handle it came
    from, and seed a second handle that then continues bit for bit."""
    n = 4096
    a = _mk(num_envs=n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=51)
    a.reset()
    rs = np.random.default_rng(8)
    for _ in range(70):
        a.step(np.where(rs.random((n, ns)) < 0.6, 0, rs.integers(0, 5, (n, ns))).astype(np.int32))
    blob = a.get_state_all()
    head = np.frombuffer(blob[:40].tobytes(), np.int32)
    assert bytes(blob[:4]) == b"MSST" and head[2] == n and head[3] == dim and head[4] == ns
    offs = np.frombuffer(blob[40:40 + 8 * (n + 1)].tobytes(), np.uint64)
    words = np.frombuffer(blob[40 + 8 * (n + 1):].tobytes(), np.int32)
    assert offs[0] == 0 and offs[-1] == len(words)
    for e in range(0, n, 29):
        assert np.array_equal(words[int(offs[e]):int(offs[e + 1])], a.get_state_words(e)), e
    ref_obs = a.render().copy()
    snap = [a.get_state_words(e).copy() for e in range(0, n, 101)]
    acts = rs.integers(0, 5, (25, n, ns)).astype(np.int32)
    cont = [a.step(acts[t]) for t in range(25)]           # a moves on ...
    cont = [(o.copy(), r.copy(), d.copy()) for o, r, d, _ in cont]
    b = _mk(num_envs=n, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=51)
    b.reset()
    b.set_state_all(blob)                                  # ... b starts from the checkpoint
    assert np.array_equal(b.render(), ref_obs)
    for t in range(25):
        o, r, d, _ = b.step(acts[t])
        assert np.array_equal(o, cont[t][0]) and np.array_equal(r, cont[t][1]) and np.array_equal(d, cont[t][2]), t
    a.set_state_all(blob)                                  # and a can be rolled back
    assert np.array_equal(a.render(), ref_obs)
    for k, e in enumerate(range(0, n, 101)):
        assert np.array_equal(a.get_state_words(e), snap[k]), e
    # malformed blobs are refused and say why
    bad = blob.copy(); bad[0] ^= 0xFF
    with pytest.raises(RuntimeError, match="blob"):
        a.set_state_all(bad)
    with pytest.raises(RuntimeError, match="truncated"):
        a.set_state_all(blob[:len(blob) - 4])
    c = _mk(num_envs=n // 2, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=51)
    with pytest.raises(RuntimeError, match="handle differs"):
        c.set_state_all(blob)
    # a cell off the board in one env's words: that env is named and left untouched
    w = words.copy()
    victim = None
    for e in range(n):
        pos = int(offs[e]) + 8 + 2 * int(w[int(offs[e]) + 6])   # snake 0's block: len, v0, v1, grow, alive, in_dead, cells
        if w[pos] > 0:
            w[pos + 6] = dim + 7
            victim = e
            break
    assert victim is not None
    bad = np.concatenate([blob[:40 + 8 * (n + 1)], np.frombuffer(w.tobytes(), np.uint8)])
    before = a.get_state_words(victim).copy()
    with pytest.raises(RuntimeError, match=f"env {victim}:"):
        a.set_state_all(bad)
    assert np.array_equal(a.get_state_words(victim), before)
    for env in (a, b, c):
        env.close()


def test_episode_totals_without_auto_reset_count_once():
    """auto_reset = 0: a finished env keeps returning done until the caller resets it, but the
    msnake_get_stats totals count that episode once."""
    n = 64
    env = _mk(num_envs=n, dim=6, n_snakes=2, rules="snake_env", seed=61, auto_reset=False, max_steps=12)
    env.reset()
    rs = np.random.default_rng(2)
    first_done = np.zeros(n, bool)
    for _ in range(30):
        _, _, done, _ = env.step(rs.integers(0, 5, (n, 2)).astype(np.int32))
        first_done |= done
    assert first_done.all()  # the 12-step cap has ended every episode, many steps ago
    assert env.stats()["episodes"] == n
    env.reset()
    for _ in range(30):
        env.step(rs.integers(0, 5, (n, 2)).astype(np.int32))
    assert env.stats()["episodes"] == 2 * n
    env.close()


def test_obs_alignment_is_checked_for_the_fused_upscale():
    import torch
    env = _mk(num_envs=8, dim=10, n_snakes=1, rules="snake_env", seed=1, obs_scale=7)
    H, W, C = env.obs_shape
    buf = torch.empty(8 * H * W * C + 8, dtype=torch.uint8, device=env.device)
    acts = torch.zeros((8, 1), dtype=torch.int32, device=env.device)
    rc = env._L.msnake_step(env._h, acts.data_ptr(), 1, buf.data_ptr() + 1, env._rew.data_ptr(), env._done.data_ptr(),
                            env._info.data_ptr(), env._stream())
    assert rc == -4 and b"aligned" in env._L.msnake_last_error()
    env.close()


def test_rollout_device_wrapper_equals_single_steps():
    """MultiSnakeVecEnv.rollout_device(): the Python face of msnake_rollout_tape / msnake_step_tape."""
    import torch
    n, T = 300, 20
    envs = [_mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=71) for _ in range(4)]
    for e in envs:
        e.reset()
    tape = torch.randint(0, 5, (T, n, 3), dtype=torch.int32, device=envs[0].device)
    ref = [tuple(x.clone() for x in envs[0].step_device(tape[t])) for t in range(T)]
    o1, r1, d1, i1 = envs[1].rollout_device(tape)                      # persistent launch, all observations
    o2, r2, d2, i2 = envs[2].rollout_device(tape, persistent=False)    # one launch per step
    o3, r3, d3, i3 = envs[3].rollout_device(tape, keep_obs=False)      # only the last observation
    for t in range(T):
        for got in ((o1[t], r1[t], d1[t], i1[t]), (o2[t], r2[t], d2[t], i2[t])):
            assert all(torch.equal(a, b) for a, b in zip(got, ref[t])), t
        assert torch.equal(r3[t], ref[t][1]) and torch.equal(d3[t], ref[t][2]) and torch.equal(i3[t], ref[t][3]), t
    assert torch.equal(o3, ref[-1][0])
    with pytest.raises(ValueError, match="tape must be"):
        envs[0].rollout_device(tape[:, :10])
    for e in envs:
        e.close()


# ---------------------------------------------------------------------------------------------------
# round 3: kernel-path combinations that had no oracle comparison (persistent tape x short record / 4 envs per
# workgroup, fused up-scale above 8 192 envs, persistent tape x fused up-scale), the finished flag in the
# canonical state, launch tuning through msnake_config
# ---------------------------------------------------------------------------------------------------
def _tape_vs_oracle_and_steps(num_envs, dim, ns, nf, rules, T, seed, obs_scale=1, chunk=None, **tuning):
    """rollout_device(tape) -- ONE persistent launch per chunk -- against (a) T step_device launches of a second
    handle, every output tensor, and (b) the oracle: reward / done / info every step and every observation byte
    (the oracle's native frame, replicated on the device when obs_scale > 1)."""
    import torch
    from oracle.snake_oracle import Oracle
    kw = dict(num_envs=num_envs, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=seed, obs_scale=obs_scale, **tuning)
    a, b = _mk(**kw), _mk(**kw)
    ora = Oracle(num_envs, dim=dim, n_snakes=ns, n_fruits=nf, rules=rules, seed=seed)
    up = (lambda o: o) if obs_scale == 1 else (lambda o: o.repeat_interleave(obs_scale, 1).repeat_interleave(obs_scale, 2))
    dev = a.device
    o0 = torch.from_numpy(ora.reset()).to(dev)
    assert torch.equal(a.reset_device(), up(o0)) and torch.equal(b.reset_device(), up(o0))
    rs = np.random.default_rng(seed + 5)
    acts = rs.integers(0, 5, (T, num_envs, ns)).astype(np.int32)
    tape = torch.from_numpy(acts).to(dev)
    chunk = chunk or T
    n_done = 0
    for t0 in range(0, T, chunk):
        obs, rew, done, info = a.rollout_device(tape[t0:t0 + chunk])
        for j in range(obs.shape[0]):
            t = t0 + j
            o, r, d, i = b.step_device(tape[t])
            assert torch.equal(o, obs[j]) and torch.equal(r, rew[j]) and torch.equal(d, done[j]) and torch.equal(i, info[j]), t
            o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(acts[t], threads=8)
            assert np.array_equal(rew[j].cpu().numpy(), o_rew) and np.array_equal(done[j].cpu().numpy(), o_done), t
            ih = info[j].cpu().numpy()
            assert np.array_equal(ih[:, 2], o_ns) and np.array_equal(ih[:, 1], o_el), t
            assert np.array_equal(ih[:, 0].copy().view(np.float32), o_er), t
            assert torch.equal(obs[j], up(torch.from_numpy(o_obs).to(dev))), t
            n_done += int(o_done.sum())
        del obs
    for e in range(0, num_envs, max(1, num_envs // 48)):
        assert _state(a, e) == _state(b, e) == ora.get_state(e), e
    sa, sb = a.stats(), b.stats()
    assert sa == sb and sa["errors"] == 0 and sa["episodes"] == n_done
    a.close(); b.close()
    return n_done


def test_persistent_tape_on_the_short_record_path_snake_env():
    """16 384 envs x 19x19x3: above 8 192 envs a handle runs 4 envs per workgroup on the 128-byte record (no parked
    Philox draws: every respawning / resetting wave evaluates Philox itself, the record's upper half is never
    written back).  MODE 3 on that shape, against per-step launches and the oracle, resets and respawns included."""
    assert _tape_vs_oracle_and_steps(16384, 19, 3, 3, "snake_env", 48, seed=81, chunk=24) > 16384


def test_persistent_tape_on_the_short_record_path_adversarial():
    assert _tape_vs_oracle_and_steps(20000, 10, 3, 3, "adversarial", 40, seed=82, chunk=20) > 20000


def test_persistent_tape_short_record_forced_small_batch():
    """The same path at a small batch, forced through msnake_config (record_policy short, 4 envs per workgroup)."""
    _tape_vs_oracle_and_steps(777, 19, 3, 3, "snake_env", 48, seed=83, record_policy="short", envs_per_block=4)
    _tape_vs_oracle_and_steps(500, 10, 2, 2, "adversarial", 48, seed=84, record_policy="short", envs_per_block=2)


def test_fused_warpframe_large_batch_against_oracle():
    """obs_scale = 4 at 16 384 envs (4 envs per workgroup, short record, 1 GB of 84x84x9 frames per step): every
    byte == the oracle's native frame replicated x4 on both axes.  (cv2 parity itself stays unpinned.)"""
    import torch
    from oracle.snake_oracle import Oracle
    n, steps = 16384, 20
    env = _mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=85, obs_scale=4)
    ora = Oracle(n, dim=19, n_snakes=3, rules="snake_env", seed=85)
    up = lambda o: torch.from_numpy(o).to(env.device).repeat_interleave(4, 1).repeat_interleave(4, 2)
    assert torch.equal(env.reset_device(), up(ora.reset()))
    rs = np.random.default_rng(6)
    for t in range(steps):
        act = rs.integers(0, 5, (n, 3)).astype(np.int32)
        obs, rew, done, info = env.step_device(torch.from_numpy(act).to(env.device))
        o_obs, o_rew, o_done, o_ns, _, _ = ora.step(act, threads=8)
        assert torch.equal(obs, up(o_obs)), t
        assert np.array_equal(rew.cpu().numpy(), o_rew) and np.array_equal(done.cpu().numpy(), o_done), t
        assert np.array_equal(info[:, 2].cpu().numpy(), o_ns), t
    assert env.stats()["errors"] == 0
    env.close()


def test_persistent_tape_with_fused_warpframe_at_4096_envs():
    """msnake_rollout_tape x obs_scale 4 at the bench batch size (260 MB of frames per step)."""
    assert _tape_vs_oracle_and_steps(4096, 19, 3, 3, "snake_env", 12, seed=86, obs_scale=4, chunk=6) > 1000
    _tape_vs_oracle_and_steps(2048, 10, 1, 1, "snake_env", 12, seed=87, obs_scale=7, chunk=12)


@pytest.mark.parametrize("rules,ns,nf", [("snake_env", 2, 2), ("new_world", 2, 3), ("adversarial", 2, 2)])
def test_finished_flag_survives_a_checkpoint(rules, ns, nf):
    """auto_reset = 0: an episode that has ended stays "finished" until the caller resets the env, and
    msnake_get_stats counts it once.  The canonical state carries that bit (word 7, bit 8), so restoring a
    checkpoint taken after the episode ended -- into the same handle or a fresh one -- does not count it again."""
    from oracle.snake_oracle import flat_finished
    n = 64
    kw = dict(num_envs=n, dim=6, n_snakes=ns, n_fruits=nf, rules=rules, seed=91, auto_reset=False, max_steps=9)
    a = _mk(**kw)
    a.reset()
    rs = np.random.default_rng(3)
    for _ in range(12):
        a.step(rs.integers(0, 5, (n, ns)).astype(np.int32))
    assert a.stats()["episodes"] == n                      # the 9-step cap has ended every episode
    assert all(flat_finished(a.get_state_words(e)) for e in range(n))
    blob = a.get_state_all()
    assert a.blob_info(blob)["version"] == 2 and a.blob_info(blob)["num_envs"] == n
    a.set_state_all(blob)                                  # roll back into the same handle ...
    b = _mk(**kw)
    b.reset()
    b.set_state_all(blob)                                  # ... and seed a fresh one
    for e in range(0, n, 5):
        b.set_state_words(e, a.get_state_words(e))         # the per-env path carries the bit too
    acts = rs.integers(0, 5, (6, n, ns)).astype(np.int32)
    for t in range(6):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        assert ra[2].all() and rb[2].all()                 # still done, still reported ...
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
    assert a.stats()["episodes"] == n and b.stats()["episodes"] == 0   # ... and not counted again
    a.reset(); b.reset()
    assert not any(flat_finished(a.get_state_words(e)) for e in range(n))
    for _ in range(12):
        act = rs.integers(0, 5, (n, ns)).astype(np.int32)
        a.step(act); b.step(act)
    assert a.stats()["episodes"] == 2 * n and b.stats()["episodes"] == n
    # a version-1 blob (no finished bit anywhere) is still accepted
    old = blob.copy()
    old[4:8] = np.frombuffer(np.uint32(1).tobytes(), np.uint8)
    b.set_state_all(old)
    a.close(); b.close()


def test_launch_tuning_fields_are_invisible():
    """msnake_config's ABI-3 tuning fields change the launch shape, never a result: every envs_per_block, both record
    policies and both store policies give the oracle's bytes (snake_env and adversarial; new_world ignores the record policy)."""
    for epb in (1, 2, 3, 4, 5, 8):
        assert _run_vs_oracle(530, 19, 3, 3, "snake_env", 40, seed=95, envs_per_block=epb) > 100
    _run_vs_oracle(300, 10, 3, 3, "adversarial", 60, seed=96, envs_per_block=2, record_policy="short", obs_store_policy="plain")
    _run_vs_oracle(300, 10, 3, 5, "new_world", 60, seed=97, envs_per_block=4, record_policy="short", obs_store_policy="stream")
    with pytest.raises(RuntimeError, match="envs_per_block"):
        _mk(num_envs=4, dim=10, n_snakes=2, rules="snake_env", envs_per_block=9)


@pytest.mark.parametrize("lead", [0, 4, 8, 16, 48])
def test_fused_warpframe_copy_out_variants(lead):
    """The fused copy-out writes 16 bytes per lane over the contiguous 84x84 frames when they are 16-byte aligned
    (whole 128-byte lines per wave instruction; the first line of a frame starts anywhere in it: lead 16 / 48) and
    falls back to dword stores when the caller's buffer is only 4-byte aligned (lead 4 / 8).  Same bytes either way,
    guard bytes untouched."""
    import torch
    from oracle.snake_oracle import Oracle
    n = 70
    env = _mk(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=44, obs_scale=4)
    ora = Oracle(n, dim=19, n_snakes=3, rules="snake_env", seed=44)
    H, W, C = env.obs_shape
    nbytes = n * H * W * C
    buf = torch.full((nbytes + 256,), 0xAB, dtype=torch.uint8, device=env.device)
    assert buf.data_ptr() % 256 == 0
    out = buf[lead:lead + nbytes].view(n, H, W, C)
    up = lambda o: np.repeat(np.repeat(o, 4, axis=1), 4, axis=2)
    assert np.array_equal(env.reset_device(out=out).cpu().numpy(), up(ora.reset()))
    rs = np.random.default_rng(lead)
    for t in range(25):
        act = rs.integers(0, 5, (n, 3)).astype(np.int32)
        obs, _, _, _ = env.step_device(torch.from_numpy(act).to(env.device), out=out)
        assert np.array_equal(obs.cpu().numpy(), up(ora.step(act)[0])), t
    host = buf.cpu().numpy()
    assert (host[:lead] == 0xAB).all() and (host[lead + nbytes:] == 0xAB).all()
    env.close()


@pytest.mark.parametrize("dim,n_snakes,rules,scale,lead", [(19, 3, "snake_env", 4, 0), (19, 3, "snake_env", 4, 48), (19, 3, "snake_env", 4, 4),
                                                           (10, 1, "snake_env", 7, 16), (10, 2, "new_world", 7, 0), (19, 1, "new_world", 4, 32),
                                                           (10, 2, "adversarial", 7, 112), (19, 4, "new_world", 4, 0)])
def test_fused_warpframe_streaming_flat_copy_out(dim, n_snakes, rules, scale, lead):
    """Fused frames beyond ~400 MiB per launch stream: the contiguous frame leaves as 16-byte-per-lane nt stores over whole
    128-byte lines, chunks re-cut across the (dword-, not 16-byte-multiple) rows, row-straddling chunks = wall pixels.
    Forced here at a small batch through obs_store_policy="stream", at several positions of the frame inside its first
    line (lead 0 / 16 / 32 / 48 / 112) and with a buffer that is only 4-byte aligned (lead 4: dword fallback)."""
    import torch
    from oracle.snake_oracle import Oracle
    n = 130
    env = _mk(num_envs=n, dim=dim, n_snakes=n_snakes, rules=rules, seed=45, obs_scale=scale, obs_store_policy="stream")
    ora = Oracle(n, dim=dim, n_snakes=n_snakes, rules=rules, seed=45)
    H, W, C = env.obs_shape
    nbytes = n * H * W * C
    buf = torch.full((nbytes + 256,), 0xAB, dtype=torch.uint8, device=env.device)
    assert buf.data_ptr() % 256 == 0
    out = buf[lead:lead + nbytes].view(n, H, W, C)
    up = lambda o: np.repeat(np.repeat(o, scale, axis=1), scale, axis=2)
    assert np.array_equal(env.reset_device(out=out).cpu().numpy(), up(ora.reset()))
    rs = np.random.default_rng(lead)
    for t in range(30):
        act = rs.integers(0, 5, (n, n_snakes)).astype(np.int32)
        obs, rew, done, _ = env.step_device(torch.from_numpy(act).to(env.device), out=out)
        o_obs, o_rew, o_done, _, _, _ = ora.step(act)
        assert np.array_equal(obs.cpu().numpy(), up(o_obs)), t
        assert np.array_equal(rew.cpu().numpy(), o_rew) and np.array_equal(done.cpu().numpy(), o_done), t
    assert np.array_equal(env.render_device(out=out).cpu().numpy(), up(ora.render()))
    host = buf.cpu().numpy()
    assert (host[:lead] == 0xAB).all() and (host[lead + nbytes:] == 0xAB).all()
    env.close()


@pytest.mark.parametrize("n,policy", [(512, "stream"), (512, "plain"), (520, "stream"), (777, "stream"), (4096, "auto")])
def test_persistent_tape_copy_out_shapes(n, policy):
    """msnake_rollout_tape copies out in the aligned whole-line shape when every step's observations keep one alignment
    (step stride = n * 3969 bytes a multiple of 16: n = 512, 4 096) and in the byte-aligned shape otherwise (520, 777), with
    plain or streaming stores (tape_store_policy; "auto" streams from 192 MiB per call: 4 096 envs x 16 steps = 260 MB).
    Same bytes as per-step launches either way, and as a re-used (stride 0) buffer's last step."""
    import torch
    T = 16
    kw = dict(num_envs=n, dim=19, n_snakes=3, rules="snake_env", seed=93)
    a, b = _mk(tape_store_policy=policy, **kw), _mk(**kw)
    a.reset(); b.reset()
    tape = torch.randint(0, 5, (2 * T, n, 3), dtype=torch.int32, device=a.device)
    o1, r1, d1, i1 = a.rollout_device(tape[:T])
    for t in range(T):
        o, r, d, i = b.step_device(tape[t])
        assert torch.equal(o, o1[t]) and torch.equal(r, r1[t]) and torch.equal(d, d1[t]) and torch.equal(i, i1[t]), t
    o2, r2, d2, i2 = a.rollout_device(tape[T:], keep_obs=False)
    for t in range(T, 2 * T):
        o, r, d, i = b.step_device(tape[t])
    assert torch.equal(o, o2) and torch.equal(r, r2[-1]) and torch.equal(d, d2[-1])
    assert a.stats() == b.stats()
    a.close(); b.close()
