"""MultiSnakeVecEnv -- the reference's VecEnv surface over the HIP batched step.

Replaces, for the snake path, what `utils.make_basic_env` builds in the reference
(src/utils.py:34-49): SubprocVecEnv([gym.make(id) + seed(seed+rank) + Monitor + WarpFrame] * n).
Same attributes and methods as baselines' VecEnv ABC (src/baselines/common/vec_env/__init__.py:22-88):
num_envs, observation_space, action_space, reset(), step_async(), step_wait(), step(), close(),
render(), unwrapped -- so `ppo_multi_agent.Runner` (src/ppo_multi_agent.py:145-216) can drive it
unchanged: env.reset() -> uint8[nenv,H,W,C]; env.step(list of per-env action tuples) ->
(obs, rews float32[nenv], dones bool[nenv], infos).

All env state and all step outputs live in HBM; `step_device()` is the zero-copy entry point for a
PyTorch-ROCm policy (device tensors in, device tensors out, asynchronous on the current stream),
`step()` is the NumPy-returning reference-compatible wrapper around it.
"""
import ctypes
import time

import numpy as np

from . import _capi
from .spaces import Box, Discrete

# gym ids of the reference (src/gym-snake/gym_snake/__init__.py:11-26) -> rule presets
GYM_IDS = {
    "snake-multiple-test-v0": dict(rules="snake_env", dim=19),
    "snake-new-multiple-v0": dict(rules="new_world", dim=10),
    "snake-adversarial-v0": dict(rules="adversarial", dim=10),
}


def normalize_actions(actions, num_envs, n_snakes):
    """list(zip(a0, a1[, a2])) / ndarray / scalar-per-env -> contiguous int32 [num_envs, stride].

    ppo_multi_agent.py:41-44 always sends tuples of length 2 or 3, even with one snake; surplus
    entries are ignored by the env (snake_multiple_test.py:174). A bare scalar per env is wrapped
    like snake_multiple_test.py:167-168 does.
    """
    a = np.asarray(actions)
    if a.ndim == 1:
        a = a[:, None]
    if a.ndim != 2 or a.shape[0] != num_envs:
        raise ValueError(f"expected {num_envs} action rows, got array of shape {a.shape}")
    if a.shape[1] < n_snakes:
        raise ValueError(f"each action row needs >= {n_snakes} entries, got {a.shape[1]}")
    return np.ascontiguousarray(a, dtype=np.int32)


class LazyInfos:
    """Sequence of per-env info dicts, materialised on access.

    Same keys as the reference: {"ale.lives": 1, "num_snakes": k} (snake_multiple_test.py:197) plus
    Monitor's info['episode'] = {'r','l','t'} on the step an episode ends (monitor.py:61-78).
    Building 4096 dicts per step would dominate the step time, so they are built on demand;
    `episodes()` is the fast path for the `info.get('episode')` scan at ppo_multi_agent.py:187-190.
    """

    def __init__(self, done, num_snakes, ep_return, ep_len, t_elapsed):
        self._done, self._ns, self._r, self._l, self._t = done, num_snakes, ep_return, ep_len, t_elapsed

    def __len__(self):
        return len(self._done)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        info = {"ale.lives": 1, "num_snakes": int(self._ns[i])}
        if self._done[i]:
            info["episode"] = {"r": round(float(self._r[i]), 6), "l": int(self._l[i]), "t": self._t}
        return info

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def episodes(self):
        idx = np.nonzero(self._done)[0]
        return [{"r": round(float(self._r[i]), 6), "l": int(self._l[i]), "t": self._t} for i in idx]


class MultiSnakeVecEnv:
    """num_envs independent multi-snake games stepped in lockstep by one HIP kernel launch."""

    metadata = {"render.modes": ["rgb_array"]}

    def __init__(self, num_envs, dim=19, n_snakes=3, n_fruits=None, rules="snake_env", seed=0,
                 env_id_base=0, device=None, max_steps=2000, auto_reset=True, obs_scale=1,
                 declared_channels=6, host_views=False, envs_per_block=0, record_policy="auto",
                 obs_store_policy="auto", tape_store_policy="auto"):
        import torch  # device memory and streams only

        if not torch.cuda.is_available():
            raise RuntimeError("MultiSnakeVecEnv needs an MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU path")
        self._torch = torch
        self._L = _capi.load()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError(f"device must be a cuda (HIP) device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        rules_id = _capi.RULES[rules] if isinstance(rules, str) else int(rules)
        if n_fruits is None:
            n_fruits = n_snakes
        self.cfg = _capi.MsnakeConfig(ctypes.sizeof(_capi.MsnakeConfig), self.device.index or 0, int(num_envs),
                                      int(dim), int(n_snakes), int(n_fruits), rules_id, int(max_steps),
                                      int(bool(auto_reset)), int(obs_scale), int(seed), int(env_id_base),
                                      # launch tuning (msnake_config, ABI 3): how the work is laid out, never a result
                                      int(envs_per_block), _capi.RECORD_POLICY[record_policy],
                                      _capi.STORE_POLICY[obs_store_policy], _capi.STORE_POLICY[tape_store_policy])
        self._h = ctypes.c_void_p()
        _capi.check(self._L.msnake_create(ctypes.byref(self.cfg), ctypes.byref(self._h)), "msnake_create")
        H, W, C = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _capi.check(self._L.msnake_obs_shape(self._h, ctypes.byref(H), ctypes.byref(W), ctypes.byref(C)))
        self.num_envs = int(num_envs)
        self.n_snakes = int(n_snakes)
        self.rules = _capi.RULE_NAMES[rules_id]
        self.obs_shape = (H.value, W.value, C.value)
        # WarpFrame declares 6 channels whatever the env emits (utils.py:25) and get_shape halves
        # that (utils.py:51-55): keep the declaration so ppo_multi_agent builds 3-channel policies.
        self.observation_space = Box(0, 255, (H.value, W.value, declared_channels or C.value), np.uint8)
        self.action_space = Discrete(5)
        with torch.cuda.device(self.device):
            self._obs = torch.empty((self.num_envs,) + self.obs_shape, dtype=torch.uint8, device=self.device)
            self._rew = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
            self._done = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
            self._info = torch.zeros((self.num_envs, 4), dtype=torch.int32, device=self.device)
        # NumPy-returning calls copy device -> host.  Default: fresh arrays every call, like the
        # reference's np.stack (pageable copy, ~10 GB/s).  host_views=True returns views of pinned
        # staging buffers instead (~5x faster, but OVERWRITTEN by the next call: copy what you keep,
        # as ppo_multi_agent.py:165-168 does anyway).
        self._host_views = bool(host_views)
        self._pinned = None
        if self._host_views:
            pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pinned = (pin(self._obs), pin(self._rew), pin(self._done), pin(self._info))
        self._pending = None
        self._tstart = time.time()
        self.closed = False
        # hot-path constants of step_device(): the handle's own output buffers never move
        self._p_obs, self._p_rew = self._obs.data_ptr(), self._rew.data_ptr()
        self._p_done, self._p_info = self._done.data_ptr(), self._info.data_ptr()
        self._step_fn = self._L.msnake_step
        self._cur_stream = torch.cuda.current_stream

    # ------------------------------------------------------------------ device-side API
    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _out(self, out):
        """The kernel writes num_envs*H*W*C bytes through a raw pointer: a caller-supplied buffer must
        be exactly that, on this device."""
        if out is None:
            return self._obs
        torch = self._torch
        want = (self.num_envs,) + self.obs_shape
        if (not isinstance(out, torch.Tensor) or out.dtype != torch.uint8 or out.device != self.device or
                tuple(out.shape) != want or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous uint8 tensor of shape {want} on {self.device}")
        return out

    def reset_device(self, out=None):
        obs = self._out(out)
        _capi.check(self._L.msnake_reset(self._h, obs.data_ptr(), self._stream()), "msnake_reset")
        return obs

    def step_device(self, actions, out=None):
        """actions: int32 cuda tensor [num_envs, >= n_snakes]. Returns device tensors
        (obs uint8[nenv,H,W,C], rew f32[nenv], done u8[nenv], info i32[nenv,4] = (ep_return bits,
        ep_len, num_snakes, done)); nothing is synchronised."""
        torch = self._torch
        if actions.dtype != torch.int32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        shape = actions.shape
        if len(shape) != 2 or shape[0] != self.num_envs or shape[1] < self.n_snakes:
            raise ValueError(f"actions must be [{self.num_envs}, >={self.n_snakes}], got {tuple(actions.shape)}")
        if out is None:
            obs, p_obs = self._obs, self._p_obs
        else:
            obs = self._out(out)
            p_obs = obs.data_ptr()
        rc = self._step_fn(self._h, actions.data_ptr(), shape[1], p_obs, self._p_rew, self._p_done, self._p_info,
                           self._cur_stream(self.device).cuda_stream)
        if rc < 0:
            _capi.check(rc, "msnake_step")
        return obs, self._rew, self._done, self._info

    def rollout_device(self, tape, persistent=True, keep_obs=True):
        """T lockstep steps from an action tape int32 cuda [T, num_envs, >= n_snakes] in ONE call.

        persistent=True: msnake_rollout_tape (one launch, env state kept in registers across the steps);
        False: msnake_step_tape (one launch per step, issued from C).  Same results either way, and the
        same as T step_device() calls.  Returns fresh device tensors (obs uint8 [T, n, H, W, C] -- or
        only the last step's [n, H, W, C] when keep_obs is False --, rew f32 [T, n], done u8 [T, n],
        info i32 [T, n, 4]); nothing is synchronised."""
        torch = self._torch
        if tape.dtype != torch.int32 or tape.device != self.device or not tape.is_contiguous():
            tape = tape.to(device=self.device, dtype=torch.int32).contiguous()
        if tape.dim() != 3 or tape.shape[1] != self.num_envs or tape.shape[2] < self.n_snakes or tape.shape[0] < 1:
            raise ValueError(f"tape must be [T >= 1, {self.num_envs}, >= {self.n_snakes}], got {tuple(tape.shape)}")
        T, n = int(tape.shape[0]), self.num_envs
        H, W, C = self.obs_shape
        with torch.cuda.device(self.device):
            obs = torch.empty(((T, n) if keep_obs else (n,)) + (H, W, C), dtype=torch.uint8, device=self.device)
            rew = torch.empty((T, n), dtype=torch.float32, device=self.device)
            done = torch.empty((T, n), dtype=torch.uint8, device=self.device)
            info = torch.empty((T, n, 4), dtype=torch.int32, device=self.device)
        fn = self._L.msnake_rollout_tape if persistent else self._L.msnake_step_tape
        _capi.check(fn(self._h, tape.data_ptr(), int(tape.shape[2]), T, obs.data_ptr(), n * H * W * C if keep_obs else 0,
                       rew.data_ptr(), done.data_ptr(), info.data_ptr(), n, self._stream()),
                    "msnake_rollout_tape" if persistent else "msnake_step_tape")
        return obs, rew, done, info

    def render_device(self, out=None):
        obs = self._out(out)
        _capi.check(self._L.msnake_render(self._h, obs.data_ptr(), self._stream()), "msnake_render")
        return obs

    # ------------------------------------------------------------------ VecEnv surface (NumPy out)
    def _obs_to_host(self, obs):
        if not self._host_views:
            return obs.cpu().numpy()
        self._pinned[0].copy_(obs, non_blocking=True)
        self._torch.cuda.current_stream(self.device).synchronize()
        return self._pinned[0].numpy()

    def reset(self):
        self._tstart = time.time()
        return self._obs_to_host(self.reset_device())

    def step_async(self, actions):
        torch = self._torch
        if isinstance(actions, torch.Tensor):
            a = actions
        else:
            a = torch.from_numpy(normalize_actions(actions, self.num_envs, self.n_snakes)).to(self.device)
        self._pending = self.step_device(a)

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError("step_wait() called without step_async()")  # NotSteppingError in baselines
        obs, rew, done, info = self._pending
        self._pending = None
        if self._host_views:
            for dst, src in zip(self._pinned, (obs, rew, done, info)):
                dst.copy_(src, non_blocking=True)
            self._torch.cuda.current_stream(self.device).synchronize()
            obs_h, rew_h, done_u8, info_h = (t.numpy() for t in self._pinned)
        else:
            obs_h, rew_h, done_u8, info_h = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy()
        done_h = done_u8.astype(bool)
        infos = LazyInfos(done_h, info_h[:, 2].copy(), info_h[:, 0].copy().view(np.float32), info_h[:, 1].copy(),
                          round(time.time() - self._tstart, 6))
        return obs_h, rew_h, done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def render(self, mode="rgb_array"):
        return self.render_device().cpu().numpy()  # always a fresh array

    def close(self):
        if not self.closed and self._h:
            self._L.msnake_destroy(self._h)
            self._h = ctypes.c_void_p()
        self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ state / stats (tests, logging)
    def get_state_words(self, env):
        n = _capi.check(self._L.msnake_get_state(self._h, env, None, 0), "msnake_get_state")
        buf = np.zeros(n, np.int32)
        _capi.check(self._L.msnake_get_state(self._h, env, buf.ctypes.data, n), "msnake_get_state")
        return buf

    def set_state_words(self, env, words):
        w = np.ascontiguousarray(words, dtype=np.int32)
        _capi.check(self._L.msnake_set_state(self._h, env, w.ctypes.data, len(w)), "msnake_set_state")

    def get_state_all(self):
        """Canonical state of every env as one bytes-like blob (numpy uint8; layout: include/msnake.h)."""
        n = _capi.check(self._L.msnake_get_state_all(self._h, None, 0), "msnake_get_state_all")
        buf = np.zeros(n, np.uint8)
        _capi.check(self._L.msnake_get_state_all(self._h, buf.ctypes.data, n), "msnake_get_state_all")
        return buf

    def set_state_all(self, blob):
        b = np.ascontiguousarray(np.frombuffer(blob, np.uint8) if not isinstance(blob, np.ndarray) else blob, dtype=np.uint8)
        _capi.check(self._L.msnake_set_state_all(self._h, b.ctypes.data, b.nbytes), "msnake_set_state_all")

    @staticmethod
    def blob_info(blob):
        """What a get_state_all() blob holds (host-only check, no handle needed): dict with version, num_envs,
        dim, n_snakes, n_fruits, rules (name), total_words.  Raises RuntimeError for a malformed blob."""
        b = np.ascontiguousarray(np.frombuffer(blob, np.uint8) if not isinstance(blob, np.ndarray) else blob, dtype=np.uint8)
        bi = _capi.MsnakeBlobInfo()
        _capi.check(_capi.load().msnake_state_blob_info(b.ctypes.data, b.nbytes, ctypes.byref(bi)), "msnake_state_blob_info")
        return {"version": bi.version, "num_envs": bi.num_envs, "dim": bi.dim, "n_snakes": bi.n_snakes,
                "n_fruits": bi.n_fruits, "rules": _capi.RULE_NAMES.get(bi.rules, bi.rules), "total_words": bi.total_words}

    def stats(self, reset=False):
        st = _capi.MsnakeStats()
        _capi.check(self._L.msnake_get_stats(self._h, ctypes.byref(st), int(reset)), "msnake_get_stats")
        return {"episodes": st.episodes, "ep_len_sum": st.ep_len_sum, "ep_return_sum": st.ep_return_sum,
                "env_steps": st.env_steps, "errors": st.errors}

    def kernel_name(self):
        return self._L.msnake_kernel_name(self._h).decode()

    def algorithmic_bytes_per_env_step(self):
        return int(self._L.msnake_algorithmic_bytes_per_env_step(self._h))


def make(env_id, num_envs, n_snakes=None, **kw):
    """make('snake-multiple-test-v0', 4096, n_snakes=3) -- the reference's gym ids as presets."""
    preset = dict(GYM_IDS[env_id])
    if n_snakes is not None:
        preset["n_snakes"] = n_snakes
    preset.update(kw)
    return MultiSnakeVecEnv(num_envs, **preset)
