"""The self-play driver's math against NumPy restatements of ppo_multi_agent.py (CPU, no env)."""
import numpy as np
import pandas as pd
import pytest
import torch

from msnake import selfplay


def test_gae_matches_reference_loop():
    rs = np.random.default_rng(0)
    T, N, gamma, lam = 9, 5, 0.99, 0.95
    rew, val = rs.normal(size=(T, N)).astype(np.float32), rs.normal(size=(T, N)).astype(np.float32)
    dones = (rs.random((T, N)) < 0.2).astype(np.float32)
    last_v, last_d = rs.normal(size=N).astype(np.float32), (rs.random(N) < 0.2).astype(np.float32)
    # ppo_multi_agent.py:205-214 restated
    advs = np.zeros_like(rew)
    lastgaelam = 0
    for t in reversed(range(T)):
        if t == T - 1:
            nnt, nv = 1.0 - last_d, last_v
        else:
            nnt, nv = 1.0 - dones[t + 1], val[t + 1]
        delta = rew[t] + gamma * nv * nnt - val[t]
        advs[t] = lastgaelam = delta + gamma * lam * nnt * lastgaelam
    ret, adv = selfplay.gae(*(torch.from_numpy(x) for x in (rew, val, dones, last_v, last_d)), gamma, lam)
    assert np.allclose(adv.numpy(), advs, atol=1e-5) and np.allclose(ret.numpy(), advs + val, atol=1e-5)


def test_ppo_loss_matches_reference_formula():
    rs = np.random.default_rng(1)
    B, A, clip = 64, 5, 0.1
    logits = rs.normal(size=(B, A)).astype(np.float32)
    vpred, ret, oldv = (rs.normal(size=B).astype(np.float32) for _ in range(3))
    act = rs.integers(0, A, B)
    oldnlp = rs.normal(size=B).astype(np.float32) * 0.1 + 1.6
    logp = logits - np.log(np.exp(logits).sum(1, keepdims=True))
    nlp = -logp[np.arange(B), act]
    p = np.exp(logp)
    ent = (-(p * logp).sum(1)).mean()
    advs = ret - oldv
    advs = (advs - advs.mean()) / (advs.std() + 1e-8)
    vclip = oldv + np.clip(vpred - oldv, -clip, clip)
    vf = 0.5 * np.maximum((vpred - ret) ** 2, (vclip - ret) ** 2).mean()
    ratio = np.exp(oldnlp - nlp)
    pg = np.maximum(-advs * ratio, -advs * np.clip(ratio, 1 - clip, 1 + clip)).mean()
    want = pg - ent * 0.01 + vf * 0.5
    loss, parts = selfplay.ppo_loss(*(torch.from_numpy(np.asarray(x)) for x in (logits, vpred, act, ret, oldv, oldnlp)), clip)
    assert abs(float(loss) - want) < 1e-4
    assert abs(float(parts["approxkl"]) - 0.5 * ((nlp - oldnlp) ** 2).mean()) < 1e-5
    assert abs(float(parts["clipfrac"]) - (np.abs(ratio - 1) > clip).mean()) < 1e-6


def test_policy_shapes_and_sampling():
    torch.manual_seed(0)
    for shape in ((21, 21, 3), (84, 84, 3), (12, 12, 3)):
        pol = selfplay.CnnPolicy(shape)
        ob = torch.randint(0, 256, (7,) + shape, dtype=torch.uint8)
        a, v, nlp = pol.step(ob)
        assert a.shape == (7,) and v.shape == (7,) and nlp.shape == (7,)
        assert int(a.min()) >= 0 and int(a.max()) < 5 and bool((nlp > 0).all())
    assert isinstance(selfplay.CnnPolicy((84, 84, 3)).convs[0], torch.nn.Conv2d)
    assert selfplay.CnnPolicy((84, 84, 3)).convs[0].kernel_size == (8, 8)   # nature_cnn
    assert selfplay.CnnPolicy((21, 21, 3)).convs[0].kernel_size == (3, 3)   # custom_cnn


def test_opponent_pool_ring_and_uniform_sampling():
    pol = selfplay.CnnPolicy((12, 12, 3))
    pool = selfplay.OpponentPool(max_saved=3)
    for _ in range(5):
        pool.save(pol)
    assert pool.num == 3 and pool.idx == 2 and sorted(pool.slots) == [0, 1, 2]
    other = selfplay.CnnPolicy((12, 12, 3))
    assert 0 <= pool.load_random(other) <= 2
    for a, b in zip(pol.state_dict().values(), other.state_dict().values()):
        assert torch.equal(a, b)


def test_csv_logger_is_readable_like_the_reference_plots(tmp_path):
    path = tmp_path / "ppo.csv"
    log = selfplay.CSVLogger(str(path))
    log.writekvs({"nupdates": 1, "total_timesteps": 100, "eprewmean 100": 0.5})
    log.writekvs({"nupdates": 2, "total_timesteps": 200, "eprewmean 100": 0.7, "num_opponents": 1})
    log.close()
    data = pd.read_csv(path)  # src/plotting.py:41-46 reads exactly these two columns
    assert list(data["total_timesteps"]) == [100, 200] and list(data["eprewmean 100"]) == [0.5, 0.7]
    mon = selfplay.MonitorCSV(str(tmp_path / "m.monitor.csv"), "snake")
    mon.write([{"r": 3.0, "l": 21, "t": 1.5}])
    mon.close()
    lines = open(tmp_path / "m.monitor.csv").read().splitlines()
    assert lines[0].startswith('#{"t_start"') and lines[1] == "r,l,t" and lines[2] == "3.0,21,1.5"


def test_json_and_tensorboard_sinks_round_trip(tmp_path):
    """logger.py:86-99 JSON lines and logger.py:136-170 TensorBoard events (written without TensorFlow):
    CRC32C known answers, TFRecord framing with masked CRCs, Event/Summary fields decoded back."""
    import json
    import struct
    from msnake import selfplay as sp
    assert sp.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    assert sp.crc32c(b"\x00" * 32) == 0x8A9136AA           # RFC 3720 B.4
    rows = [{"nupdates": 1, "fps": 1234, "eprewmean 100": -0.5, "policy_loss": 0.25},
            {"nupdates": 2, "fps": 1300, "eprewmean 100": 0.125, "policy_loss": -0.75}]
    js, tb = sp.JSONLogger(str(tmp_path / "progress.json")), sp.TensorBoardLogger(str(tmp_path / "tb"))
    for r in rows:
        js.writekvs(r); tb.writekvs(r)
    js.close(); tb.close()
    assert [json.loads(l) for l in open(tmp_path / "progress.json")] == rows

    def varint(b, i):
        v = s = 0
        while True:
            v |= (b[i] & 0x7F) << s
            s += 7
            i += 1
            if not b[i - 1] & 0x80:
                return v, i

    def fields(b):
        i, out = 0, []
        while i < len(b):
            key, i = varint(b, i)
            f, wt = key >> 3, key & 7
            if wt == 0:
                v, i = varint(b, i)
            elif wt == 1:
                v, i = struct.unpack_from("<d", b, i)[0], i + 8
            elif wt == 5:
                v, i = struct.unpack_from("<f", b, i)[0], i + 4
            else:
                n, i = varint(b, i)
                v, i = b[i:i + n], i + n
            out.append((f, v))
        return out

    data = open(tb.path, "rb").read()
    i, events = 0, []
    while i < len(data):
        (n,) = struct.unpack_from("<Q", data, i)
        assert struct.unpack_from("<I", data, i + 8)[0] == sp._masked_crc(data[i:i + 8])
        rec = data[i + 12:i + 12 + n]
        assert struct.unpack_from("<I", data, i + 12 + n)[0] == sp._masked_crc(rec)
        events.append(fields(rec))
        i += 16 + n
    assert len(events) == 3 and (3, b"brain.Event:2") in events[0]
    for step, (ev, row) in enumerate(zip(events[1:], rows), 1):
        ev = dict(ev)
        assert ev[2] == step and ev[1] > 1e9
        vals = {}
        for f, v in fields(ev[5]):
            assert f == 1
            d = dict(fields(v))
            vals[d[1].decode()] = d[2]
        assert vals == {k: float(np.float32(v)) for k, v in row.items()}


class _FakeEnv:
    """CPU stand-in with the device-side surface learn() uses (reset_device / step_device)."""

    def __init__(self, n=8, n_snakes=2, seed=0):
        self.num_envs, self.n_snakes, self.obs_shape, self.device = n, n_snakes, (12, 12, 9), torch.device("cpu")
        self.g = torch.Generator().manual_seed(seed)
        self.len = torch.zeros(n, dtype=torch.int32)

    def reset_device(self):
        return torch.randint(0, 256, (self.num_envs,) + self.obs_shape, dtype=torch.uint8, generator=self.g)

    def step_device(self, actions):
        assert actions.shape == (self.num_envs, self.n_snakes) and actions.dtype == torch.int32
        self.len += 1
        done = (torch.rand(self.num_envs, generator=self.g) < 0.3)
        rew = torch.randint(0, 2, (self.num_envs,), generator=self.g).float()
        info = torch.zeros((self.num_envs, 4), dtype=torch.int32)
        info[:, 0] = torch.full((self.num_envs,), 7.0).view(torch.int32)
        info[:, 1] = self.len
        self.len = torch.where(done, torch.zeros_like(self.len), self.len)
        return self.reset_device(), rew, done.to(torch.uint8), info


def test_checkpoints_save_load_and_resume(tmp_path):
    """ppo_multi_agent.py:112-133 (save/load), :296-306 and :349-364 (opponent files), :392-404 (periodic,
    highscore and final saves) -- as weights-only files -- plus resuming a run from trainer_state.pt."""
    import os
    d = str(tmp_path / "ckpt")
    kw = dict(nsteps=4, total_timesteps=8 * 4 * 6, nminibatches=2, noptepochs=1, opponent_save_interval=2, log_fn=None)
    model, hist = selfplay.learn(_FakeEnv(), save_dir=d, save_interval=2, **kw)
    files = set(os.listdir(d))
    assert {"opponent1_0.pt", "opponent1_1.pt", "opponent1_2.pt", "opponent1_3.pt", "snake_model_num2_0.pt",
            "snake_model_num2_1.pt", "snake_model_num2_2.pt", "snake_model_num2_3.pt", "snake_model_num2.pt",
            "trainer_state.pt"} <= files and "highscore_model.pt" not in files
    fresh = selfplay.CnnPolicy((12, 12, 3))
    selfplay.load_weights(fresh, os.path.join(d, "snake_model_num2.pt"))
    for a, b in zip(model.state_dict().values(), fresh.state_dict().values()):
        assert torch.equal(a, b)
    # the files hold tensors only: the safe loader is enough
    ts = torch.load(os.path.join(d, "trainer_state.pt"), weights_only=True)
    assert ts["update"] == 6 and ts["pools"] == [(4, 4)] and ts["model_idx"] == 4
    assert int(ts["optimizer"]["state"][0]["step"]) == 6 * 2

    # load_path: learner and opponent start from the file (the reference's baseline_file)
    m2, _ = selfplay.learn(_FakeEnv(), load_path=os.path.join(d, "snake_model_num2_0.pt"),
                           **dict(kw, total_timesteps=8 * 4, lr=lambda f: 0.0))
    first = torch.load(os.path.join(d, "snake_model_num2_0.pt"), weights_only=True)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, first[k]), k

    # an interrupted run resumes from its last trainer_state (update 2) and finishes updates 3..6
    d2 = str(tmp_path / "ckpt2")

    class Stop(Exception):
        pass

    seen = []

    def stop_at_3(line):
        seen.append(line)
        if "nupdates=3" in line:
            raise Stop

    try:
        selfplay.learn(_FakeEnv(), save_dir=d2, save_interval=2, **dict(kw, log_fn=stop_at_3))
        assert False, "should have been interrupted"
    except Stop:
        pass
    assert torch.load(os.path.join(d2, "trainer_state.pt"), weights_only=True)["update"] == 2
    _, hist2 = selfplay.learn(_FakeEnv(seed=1), save_dir=d2, save_interval=2, resume=True, **kw)
    assert [h["nupdates"] for h in hist2] == [3, 4, 5, 6] and hist2[-1]["num_opponents"] == 4
    ts2 = torch.load(os.path.join(d2, "trainer_state.pt"), weights_only=True)
    assert ts2["update"] == 6 and int(ts2["optimizer"]["state"][0]["step"]) == 6 * 2 and ts2["pools"] == [(4, 4)]

    # single snake: highscore_model.pt once the 100-episode mean (7.0 here) passes 5 (ppo_multi_agent.py:397-401)
    d3 = str(tmp_path / "ckpt3")
    selfplay.learn(_FakeEnv(n_snakes=1), save_dir=d3, **dict(kw, total_timesteps=8 * 4 * 2))
    assert "highscore_model.pt" in os.listdir(d3) and not any(f.startswith("opponent") for f in os.listdir(d3))


def test_resume_without_save_interval_and_refusal_without_state(tmp_path):
    """The trainer state is written with EVERY opponent-pool save (not only every --save-interval), so an
    interrupted `--save DIR --resume` run without a save interval continues instead of silently starting over;
    and a directory that holds opponent files but no state file is refused rather than overwritten."""
    import os
    d = str(tmp_path / "run")
    kw = dict(nsteps=4, total_timesteps=8 * 4 * 6, nminibatches=2, noptepochs=1, opponent_save_interval=2)

    class Stop(Exception):
        pass

    def stop_at_5(line):
        if "nupdates=5" in line:
            raise Stop

    with pytest.raises(Stop):
        selfplay.learn(_FakeEnv(), save_dir=d, log_fn=stop_at_5, **kw)   # save_interval = 0
    ts = torch.load(os.path.join(d, "trainer_state.pt"), weights_only=True)
    assert ts["update"] == 4 and ts["pools"] == [(3, 3)]                  # pool files 0 (start), 1 (update 2), 2 (update 4)
    assert {"opponent1_0.pt", "opponent1_1.pt", "opponent1_2.pt"} <= set(os.listdir(d))
    _, hist = selfplay.learn(_FakeEnv(seed=1), save_dir=d, resume=True, log_fn=None, **kw)
    assert [h["nupdates"] for h in hist] == [5, 6] and hist[-1]["num_opponents"] == 4
    # a fresh run writes its state with the very first pool file
    d2 = str(tmp_path / "run2")

    def stop_at_1(line):
        raise Stop

    with pytest.raises(Stop):
        selfplay.learn(_FakeEnv(), save_dir=d2, log_fn=stop_at_1, **kw)
    assert torch.load(os.path.join(d2, "trainer_state.pt"), weights_only=True)["update"] == 0
    # opponents but no state: refuse
    os.remove(os.path.join(d2, "trainer_state.pt"))
    with pytest.raises(RuntimeError, match="trainer_state"):
        selfplay.learn(_FakeEnv(), save_dir=d2, resume=True, log_fn=None, **kw)
    selfplay.learn(_FakeEnv(), save_dir=d2, resume=False, log_fn=None, **dict(kw, total_timesteps=8 * 4))  # starting over is allowed
