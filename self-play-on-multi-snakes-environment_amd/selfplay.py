"""Self-play PPO driver on PyTorch-ROCm over the HIP env (BASELINE.json configs[4]).

A caller of the hot path, kept deliberately thin: it reproduces the structure of the reference's
`src/ppo_multi_agent.py` so that the env API is exercised exactly as there, with observations,
actions, rewards and dones staying on the GPU end to end (the reference moves every frame through
n pipes and NumPy, and sleeps 0.1 s per rollout step, ppo_multi_agent.py:175).

  reference                                         here
  policies.py:12-30  nature_cnn / custom_cnn        CnnPolicy (same layers, orthogonal init, /255)
  policies.py:45-59  CategoricalPd sample / neglogp Gumbel-max sample, cross-entropy neglogp
  ppo_multi_agent.py:23-53  MultiModel.multi_step   Runner.multi_step (opponent None -> action 1)
  ppo_multi_agent.py:164-168 channel slicing        obs[..., 0:3] / [3:6] / [6:9]
  ppo_multi_agent.py:205-214 GAE                    gae()
  ppo_multi_agent.py:72-86, 98-99 clipped PPO loss  ppo_loss() (+ per-minibatch advantage norm)
  ppo_multi_agent.py:314-322, 349-364 opponent pool OpponentPool (save every 50 updates, <= 1000,
                                                    uniform sampling)
  ppo_multi_agent.py:366-390 log keys               same key names; CSVLogger writes them in
                                                    baselines' logger.CSVOutputFormat layout, so
                                                    src/plotting.py reads the file unchanged
  baselines/bench/monitor.py:29-32,66-75            MonitorCSV ('#{json}' line, then r,l,t rows)
Parity with the reference here is statistical (learning curves), not bit-exact.
"""
import json
import os
import random
import time
from collections import deque

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- policy
def _ortho(module, scale):
    nn.init.orthogonal_(module.weight, gain=scale)
    nn.init.zeros_(module.bias)
    return module


class CnnPolicy(nn.Module):
    """policies.py CnnPolicy: custom_cnn (4 x conv3x3 SAME, 32-32-64-64, fc512) for native frames,
    nature_cnn (8/4, 4/2, 3/1, fc512) for 84x84 frames; pi head init 0.01, value head init 1."""

    # MIOpen (ROCm 7.2, no tuning db in this image) falls back to a ~60x slower solver for these
    # small-image convolutions (a) when the input is NHWC-strided -- which a permuted view of the
    # env's HWC frames is, hence the .contiguous() below: backward 237 ms -> 4 ms at B=1024 on
    # MI355X -- and (b) once the batch exceeds a few thousand frames (1.9 s at B=8192), so the
    # trunk runs on plain NCHW batch chunks.
    CONV_CHUNK = 1024

    def __init__(self, ob_shape, n_actions=5, atari_size=None, amp_dtype=None):
        super().__init__()
        # opt-in mixed precision (torch.bfloat16): the trunk runs under autocast on MFMA, parameters,
        # heads' outputs, loss and optimizer stay fp32.  The reference trains in fp32, so None is the default.
        self.amp_dtype = amp_dtype
        h, w, c = ob_shape
        atari_size = (h == 84) if atari_size is None else atari_size
        g = float(np.sqrt(2))
        if atari_size:
            convs = [_ortho(nn.Conv2d(c, 32, 8, 4), g), nn.ReLU(), _ortho(nn.Conv2d(32, 64, 4, 2), g), nn.ReLU(),
                     _ortho(nn.Conv2d(64, 64, 3, 1), g), nn.ReLU()]
        else:
            convs = [_ortho(nn.Conv2d(c, 32, 3, 1, 1), g), nn.ReLU(), _ortho(nn.Conv2d(32, 32, 3, 1, 1), g), nn.ReLU(),
                     _ortho(nn.Conv2d(32, 64, 3, 1, 1), g), nn.ReLU(), _ortho(nn.Conv2d(64, 64, 3, 1, 1), g), nn.ReLU()]
        self.convs = nn.Sequential(*convs)
        with torch.no_grad():
            nflat = self.convs(torch.zeros(1, c, h, w)).numel()
        self.fc1 = _ortho(nn.Linear(nflat, 512), g)
        self.pi = _ortho(nn.Linear(512, n_actions), 0.01)
        self.v = _ortho(nn.Linear(512, 1), 1.0)

    def forward(self, ob_u8):
        """ob_u8: uint8 [B, H, W, 3] (NHWC like the env emits) -> logits [B, A], value [B]."""
        feats = []
        with torch.autocast(ob_u8.device.type, dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None):
            for i in range(0, ob_u8.shape[0], self.CONV_CHUNK):
                x = ob_u8[i:i + self.CONV_CHUNK].permute(0, 3, 1, 2).contiguous().float() / 255.0  # plain NCHW
                feats.append(self.convs(x).flatten(1))
            x = F.relu(self.fc1(torch.cat(feats) if len(feats) > 1 else feats[0]))
            logits, v = self.pi(x), self.v(x)[:, 0]
        return logits.float(), v.float()

    @torch.no_grad()
    def step(self, ob_u8):
        logits, v = self(ob_u8)
        u = torch.rand_like(logits).clamp_(1e-20, 1.0)
        a = torch.argmax(logits - torch.log(-torch.log(u)), dim=-1)  # CategoricalPd.sample (Gumbel-max)
        return a, v, neglogp(logits, a)

    @torch.no_grad()
    def value(self, ob_u8):
        return self(ob_u8)[1]


def neglogp(logits, actions):
    return F.cross_entropy(logits, actions, reduction="none")


def entropy(logits):
    a0 = logits - logits.max(dim=-1, keepdim=True).values
    ea0 = torch.exp(a0)
    z0 = ea0.sum(dim=-1, keepdim=True)
    return (ea0 / z0 * (torch.log(z0) - a0)).sum(dim=-1)


# ----------------------------------------------------------------------------- PPO math
def gae(rewards, values, dones, last_values, last_dones, gamma=0.99, lam=0.95):
    """ppo_multi_agent.py:205-214.  rewards/values/dones: [T, N] where dones[t] is the done flag
    that PRECEDED step t (mb_dones), last_dones the flags after the last step."""
    T = rewards.shape[0]
    advs = torch.zeros_like(rewards)
    lastgaelam = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        if t == T - 1:
            nextnonterminal, nextvalues = 1.0 - last_dones, last_values
        else:
            nextnonterminal, nextvalues = 1.0 - dones[t + 1], values[t + 1]
        delta = rewards[t] + gamma * nextvalues * nextnonterminal - values[t]
        advs[t] = lastgaelam = delta + gamma * lam * nextnonterminal * lastgaelam
    return advs + values, advs


def ppo_loss(logits, vpred, actions, returns, old_values, old_neglogp, cliprange, ent_coef=0.01, vf_coef=0.5):
    """ppo_multi_agent.py:72-86 with the advantage normalisation of :98-99."""
    advs = returns - old_values
    advs = (advs - advs.mean()) / (advs.std(unbiased=False) + 1e-8)
    nlp = neglogp(logits, actions)
    ent = entropy(logits).mean()
    vclipped = old_values + torch.clamp(vpred - old_values, -cliprange, cliprange)
    vf_loss = 0.5 * torch.max((vpred - returns) ** 2, (vclipped - returns) ** 2).mean()
    ratio = torch.exp(old_neglogp - nlp)
    pg_loss = torch.max(-advs * ratio, -advs * torch.clamp(ratio, 1.0 - cliprange, 1.0 + cliprange)).mean()
    approxkl = 0.5 * ((nlp - old_neglogp) ** 2).mean()
    clipfrac = ((ratio - 1.0).abs() > cliprange).float().mean()
    loss = pg_loss - ent * ent_coef + vf_loss * vf_coef
    return loss, {"policy_loss": pg_loss, "value_loss": vf_loss, "policy_entropy": ent, "approxkl": approxkl,
                  "clipfrac": clipfrac}


def explained_variance(ypred, y):
    vary = y.var(unbiased=False)
    return float("nan") if float(vary) == 0 else float(1 - (y - ypred).var(unbiased=False) / vary)


# ----------------------------------------------------------------------------- opponent pool, logs
def save_weights(model, path):
    """Model.save (ppo_multi_agent.py:112-114: joblib.dump of the parameter list): here a plain
    state_dict of tensors, written atomically, loadable with torch.load(weights_only=True)."""
    tmp = path + ".tmp"
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, tmp)
    os.replace(tmp, path)


def load_weights(model, path):
    """Model.load (ppo_multi_agent.py:124-132).  weights_only: nothing in the file is executed."""
    model.load_state_dict(torch.load(path, map_location=next(model.parameters()).device, weights_only=True))


class OpponentPool:
    """Ring of past policies (config.py:13-14: save every 50 updates, keep <= 1000).  Held in device
    memory; with `directory` every save is also written through to `opponent<which>_<idx>.pt`
    (utils.py:57-64 names opponent1_<x>.pkl / opponent2_<x>.pkl) so a later run can restore the pool."""

    def __init__(self, max_saved=1000, directory=None, which=1):
        self.max_saved, self.slots, self.idx, self.num = max_saved, {}, 0, 0
        self.directory, self.which = directory, which

    def file(self, idx):
        return os.path.join(self.directory, f"opponent{self.which}_{idx}.pt")

    def save(self, model):
        self.slots[self.idx] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        if self.directory:
            save_weights(model, self.file(self.idx))
        self.idx += 1
        self.num = max(self.idx, self.num)
        self.idx %= self.max_saved

    def restore(self, device, idx, num):
        """Reload slots 0..num-1 from `directory` and continue writing at `idx` (resume)."""
        for i in range(num):
            self.slots[i] = torch.load(self.file(i), map_location=device, weights_only=True)
        self.idx, self.num = idx, num

    def load_random(self, model, rng=random):
        sel = rng.randint(0, max(self.num - 1, 0))  # ppo_multi_agent.py:315
        model.load_state_dict(self.slots[sel])
        return sel


class CSVLogger:
    """Same file layout as baselines.logger.CSVOutputFormat (logger.py:101-135): a header of keys,
    one row per writekvs, the header rewritten when new keys appear."""

    def __init__(self, filename):
        self.file, self.keys = open(filename, "w+t"), []

    def writekvs(self, kvs):
        extra = [k for k in kvs if k not in self.keys]
        if extra:
            self.keys.extend(extra)
            self.file.seek(0)
            lines = self.file.readlines()
            self.file.seek(0)
            self.file.write(",".join(self.keys) + "\n")
            for line in lines[1:]:
                self.file.write(line[:-1] + "," * len(extra) + "\n")
        self.file.write(",".join("" if kvs.get(k) is None else str(kvs.get(k)) for k in self.keys) + "\n")
        self.file.flush()

    def close(self):
        self.file.close()


class JSONLogger:
    """baselines.logger.JSONOutputFormat (logger.py:86-99): one JSON object per writekvs, per line."""

    def __init__(self, filename):
        self.file = open(filename, "wt")

    def writekvs(self, kvs):
        self.file.write(json.dumps({k: (float(v) if hasattr(v, "dtype") else v) for k, v in kvs.items()}) + "\n")
        self.file.flush()

    def close(self):
        self.file.close()


def _crc32c_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC32C = _crc32c_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC32C[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


class TensorBoardLogger:
    """baselines.logger.TensorBoardOutputFormat (logger.py:136-170) without TensorFlow: an
    `events.out.tfevents.*` file of TFRecord-framed Event protos (wall_time, step, one
    Summary.Value{tag, simple_value} per key), step counting from 1 like the reference.  The few
    proto fields involved are encoded by hand; records carry the masked CRC32C TensorBoard checks."""

    def __init__(self, directory):
        import socket
        import struct
        self._struct = struct
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, "events.out.tfevents.%d.%s" % (int(time.time()), socket.gethostname()))
        self.file = open(self.path, "wb")
        self.step = 1
        ver = b"brain.Event:2"
        self._record(b"\x09" + struct.pack("<d", time.time()) + b"\x1a" + _varint(len(ver)) + ver)

    def _record(self, data):
        st = self._struct
        head = st.pack("<Q", len(data))
        self.file.write(head + st.pack("<I", _masked_crc(head)) + data + st.pack("<I", _masked_crc(data)))
        self.file.flush()

    def writekvs(self, kvs):
        st = self._struct
        summary = b""
        for k, v in kvs.items():
            tag = str(k).encode()
            value = b"\x0a" + _varint(len(tag)) + tag + b"\x15" + st.pack("<f", float(v))
            summary += b"\x0a" + _varint(len(value)) + value
        event = (b"\x09" + st.pack("<d", time.time()) + b"\x10" + _varint(self.step) +
                 b"\x2a" + _varint(len(summary)) + summary)
        self._record(event)
        self.step += 1

    def close(self):
        self.file.close()


class MonitorCSV:
    """baselines.bench.Monitor's file: '#{"t_start":..,"env_id":..}' then r,l,t rows."""

    def __init__(self, filename, env_id=None):
        self.f = open(filename, "wt")
        self.f.write("#%s\n" % json.dumps({"t_start": time.time(), "env_id": env_id}))
        self.f.write("r,l,t\n")

    def write(self, epinfos):
        for ep in epinfos:
            self.f.write(f"{ep['r']},{ep['l']},{ep['t']}\n")
        self.f.flush()

    def close(self):
        self.f.close()


# ----------------------------------------------------------------------------- rollouts
class Runner:
    """ppo_multi_agent.py:145-216 with everything on the device."""

    def __init__(self, env, model, opponents, nsteps, gamma, lam):
        self.env, self.model, self.opponents = env, model, opponents
        self.nsteps, self.gamma, self.lam = nsteps, gamma, lam
        self.obs = env.reset_device()
        self.dones = torch.zeros(env.num_envs, dtype=torch.float32, device=self.obs.device)
        self.tstart = time.time()

    def multi_step(self):
        a, v, nlp = self.model.step(self.obs[..., 0:3])
        acts = [a]
        for i, opp in enumerate(self.opponents):
            if opp is None:
                acts.append(torch.ones_like(a))  # ppo_multi_agent.py:32,37
            else:
                acts.append(opp.step(self.obs[..., 3 * (i + 1):3 * (i + 2)])[0])
        return a, v, nlp, torch.stack(acts, dim=1).to(torch.int32)

    def run(self):
        T, N = self.nsteps, self.env.num_envs
        dev = self.obs.device
        mb_obs = torch.empty((T, N) + tuple(self.obs.shape[1:3]) + (3,), dtype=torch.uint8, device=dev)
        mb_rew = torch.empty((T, N), device=dev); mb_val = torch.empty((T, N), device=dev)
        mb_nlp = torch.empty((T, N), device=dev); mb_done = torch.empty((T, N), device=dev)
        mb_act = torch.empty((T, N), dtype=torch.long, device=dev)
        ep_r, ep_l = [], []
        for t in range(T):
            a, v, nlp, full = self.multi_step()
            mb_obs[t] = self.obs[..., 0:3]
            mb_act[t], mb_val[t], mb_nlp[t], mb_done[t] = a, v, nlp, self.dones
            self.obs, rew, done, info = self.env.step_device(full)
            mb_rew[t] = rew
            self.dones = done.float()
            ep_r.append(torch.where(done.bool(), info[:, 0].view(torch.float32), torch.full_like(rew, float("nan"))))
            ep_l.append(info[:, 1].clone())  # `info` is the env's buffer: the next step overwrites it
        last_values = self.model.value(self.obs[..., 0:3])
        returns, _ = gae(mb_rew, mb_val, mb_done, last_values, self.dones, self.gamma, self.lam)
        # episode infos: one host sync per rollout instead of one per step
        r = torch.stack(ep_r).flatten(); l = torch.stack(ep_l).flatten()
        keep = ~torch.isnan(r)
        t_el = round(time.time() - self.tstart, 6)
        epinfos = [{"r": round(float(x), 6), "l": int(y), "t": t_el} for x, y in zip(r[keep].tolist(), l[keep].tolist())]
        flat = lambda x: x.transpose(0, 1).reshape((T * N,) + tuple(x.shape[2:]))  # sf01
        return flat(mb_obs), flat(returns), flat(mb_done), flat(mb_act), flat(mb_val), flat(mb_nlp), epinfos


def learn(env, nsteps=64, total_timesteps=int(1e6), ent_coef=0.01, lr=lambda f: f * 2.5e-4, vf_coef=0.5,
          max_grad_norm=0.5, gamma=0.99, lam=0.95, log_interval=1, nminibatches=8, noptepochs=4,
          cliprange=lambda f: f * 0.1, opponent_save_interval=50, max_saved_opponents=1000, csv_path=None,
          monitor_path=None, seed=0, log_fn=print, amp_dtype=None, json_path=None, tb_dir=None,
          save_dir=None, save_interval=0, load_path=None, resume=False):
    """ppo_multi_agent.py:231-404 (hyper-parameters of test/ppo1_single_test.py:42-47 as defaults).

    Checkpoints (ppo_multi_agent.py:112-133, 296-306, 349-364, 392-404), weights only, under `save_dir`
    (the reference's saved_models/<expr>/): the opponent pool as opponent<i>_<idx>.pt every
    `opponent_save_interval` updates, snake_model_num<N>_<k>.pt at update 1 and every `save_interval`
    updates, highscore_model.pt whenever the 100-episode mean passes the next integer above 5 (single
    snake only, :398-401), snake_model_num<N>.pt at the end.  `load_path` initialises the learner and
    its opponents from a weights file (the reference's `baseline_file`, :264-275).  `resume=True`
    continues a run from save_dir/trainer_state.pt (weights, Adam moments, update counter, pool
    positions -- the reference cannot do this; schedules continue where they stopped)."""
    torch.manual_seed(seed); random.seed(seed)
    dev = env.device
    n_snakes = env.n_snakes
    H, W, _ = env.obs_shape
    model = CnnPolicy((H, W, 3), amp_dtype=amp_dtype).to(dev)
    opponents = [CnnPolicy((H, W, 3), amp_dtype=amp_dtype).to(dev) for _ in range(n_snakes - 1)]
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    if load_path:
        for m in [model] + opponents:
            load_weights(m, load_path)
    pools = [OpponentPool(max_saved_opponents, save_dir, i + 1) for i in range(len(opponents))]
    opt = torch.optim.Adam(model.parameters(), lr=lr(1.0), eps=1e-5)
    first_update, model_idx, next_highscore = 1, 0, 5
    state_file = os.path.join(save_dir, "trainer_state.pt") if save_dir else None
    fresh = not (resume and state_file and os.path.exists(state_file))
    if resume and fresh and save_dir and any(f.startswith("opponent") and f.endswith(".pt") for f in os.listdir(save_dir)):
        # an interrupted run left its opponent pool but no trainer state: starting over would overwrite the pool
        # from slot 0 under a model that never saw it
        raise RuntimeError(f"resume=True: {save_dir} holds opponent files but no trainer_state.pt; "
                           "pass resume=False to start over (the pool is overwritten) or restore the state file")
    if not fresh:
        ts = torch.load(state_file, map_location=dev, weights_only=True)
        model.load_state_dict(ts["model"])
        opt.load_state_dict(ts["optimizer"])
        first_update, model_idx, next_highscore = ts["update"] + 1, ts["model_idx"], ts["next_highscore"]
        for pool, (idx, num) in zip(pools, ts["pools"]):
            pool.restore(dev, idx, num)

    def save_trainer_state(update):
        tmp = state_file + ".tmp"
        torch.save({"model": model.state_dict(), "optimizer": opt.state_dict(), "update": update,
                    "model_idx": model_idx, "next_highscore": next_highscore,
                    "pools": [(p.idx, p.num) for p in pools]}, tmp)
        os.replace(tmp, state_file)

    if fresh:
        for pool in pools:
            pool.save(model)
        if save_dir:  # from the first pool file on there is a state a resume can start from
            save_trainer_state(0)
    runner = Runner(env, model, opponents, nsteps, gamma, lam)
    nbatch = env.num_envs * nsteps
    nbatch_train = nbatch // nminibatches
    assert nbatch % nminibatches == 0
    nupdates = max(1, total_timesteps // nbatch)
    epinfobuf = deque(maxlen=100)
    sinks = [CSVLogger(csv_path)] if csv_path else []  # the reference's LOG_FORMAT sinks (logger.py:172-186)
    if json_path:
        sinks.append(JSONLogger(json_path))
    if tb_dir:
        sinks.append(TensorBoardLogger(tb_dir))
    mon = MonitorCSV(monitor_path, "msnake") if monitor_path else None
    history, tfirst = [], time.time()
    for update in range(first_update, nupdates + 1):
        for opp, pool in zip(opponents, pools):
            pool.load_random(opp)
        tstart = time.time()
        frac = 1.0 - (update - 1.0) / nupdates
        lrnow, clipnow = lr(frac), cliprange(frac)
        for g in opt.param_groups:
            g["lr"] = lrnow
        obs, returns, masks, actions, values, nlps, epinfos = runner.run()
        epinfobuf.extend(epinfos)
        if mon:
            mon.write(epinfos)
        stats = []
        for _ in range(noptepochs):
            inds = torch.randperm(nbatch, device=dev)
            for start in range(0, nbatch, nbatch_train):
                mb = inds[start:start + nbatch_train]
                logits, vpred = model(obs[mb])
                loss, parts = ppo_loss(logits, vpred, actions[mb], returns[mb], values[mb], nlps[mb], clipnow,
                                       ent_coef, vf_coef)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                if max_grad_norm is not None:
                    nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
                opt.step()
                stats.append(torch.stack([parts[k].detach() for k in
                                          ("policy_loss", "value_loss", "policy_entropy", "approxkl", "clipfrac")]))
        if update % opponent_save_interval == 0:
            for pool in pools:
                pool.save(model)
            if save_dir:  # the pool files and the state file move together: a restore never finds opponents newer
                save_trainer_state(update)  # than the model / optimizer / update counter it loads
        if update % log_interval == 0 or update == 1:
            lossvals = torch.stack(stats).mean(0).tolist()
            tnow = time.time()
            safemean = lambda xs: float("nan") if len(xs) == 0 else float(np.mean(xs))
            eprew = safemean([e["r"] for e in epinfobuf])
            kvs = {"serial_timesteps": update * nsteps, "nupdates": update, "total_timesteps": update * nbatch,
                   "fps": int(nbatch / (tnow - tstart)), "explained_variance": explained_variance(values, returns),
                   "eprewmean 100": eprew, "eplenmean": safemean([e["l"] for e in epinfobuf]),
                   "time_elapsed": tnow - tfirst, "ep_rew_mean": eprew}
            if n_snakes > 1:
                kvs["num_opponents"] = pools[0].num
            kvs.update(dict(zip(("policy_loss", "value_loss", "policy_entropy", "approxkl", "clipfrac"), lossvals)))
            history.append(kvs)
            for sink in sinks:
                sink.writekvs(kvs)
            if log_fn:
                log_fn(" ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in kvs.items()))
        if save_dir:
            if save_interval and (update % save_interval == 0 or update == 1):  # ppo_multi_agent.py:392-394
                save_weights(model, os.path.join(save_dir, f"snake_model_num{n_snakes}_{model_idx}.pt"))
                model_idx += 1
                save_trainer_state(update)
            eprew_now = float(np.mean([e["r"] for e in epinfobuf])) if epinfobuf else float("nan")
            if n_snakes == 1 and eprew_now > next_highscore:                    # ppo_multi_agent.py:397-401
                next_highscore += 1
                save_weights(model, os.path.join(save_dir, "highscore_model.pt"))
    if save_dir:
        save_weights(model, os.path.join(save_dir, f"snake_model_num{n_snakes}.pt"))  # ppo_multi_agent.py:402
        save_trainer_state(nupdates)
    for sink in sinks:
        sink.close()
    if mon:
        mon.close()
    return model, history
