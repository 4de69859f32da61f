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
