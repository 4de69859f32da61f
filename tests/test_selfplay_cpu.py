"""The self-play driver's math against NumPy restatements of ppo_multi_agent.py (CPU, no env)."""
import numpy as np
import pandas as pd
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
