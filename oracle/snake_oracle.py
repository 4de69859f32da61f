"""ctypes wrapper around oracle/snake_oracle.c (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY -- see the header of snake_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
SRC_PATH = os.path.join(HERE, "snake_oracle.c")

RULES = {"snake_env": 0, "new_world": 1, "adversarial": 2}


def build(force=False):
    """gcc the C restatement into oracle/liboracle.so (kept out of git, travels with gpurun)."""
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC_PATH)):
        return LIB_PATH
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB_PATH, SRC_PATH])
    return LIB_PATH


class _Config(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int32), ("dim", ctypes.c_int32), ("n_snakes", ctypes.c_int32),
                ("n_fruits", ctypes.c_int32), ("rules", ctypes.c_int32), ("max_steps", ctypes.c_int32),
                ("auto_reset", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("seed", ctypes.c_uint64), ("env_id_base", ctypes.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.POINTER(_Config)]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_obs_shape.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 3
        L.orc_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 6
        L.orc_step_mt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 6
        L.orc_rollout_mt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 6
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_export_state.restype = ctypes.c_int32
        L.orc_export_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32]
        L.orc_import_state.restype = ctypes.c_int32
        L.orc_import_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32]
        L.orc_philox_u32.restype = ctypes.c_uint32
        L.orc_philox_u32.argtypes = [ctypes.c_uint64] * 3
        L.orc_philox_block.argtypes = [ctypes.c_void_p] * 3
        _lib = L
    return _lib


def philox_u32(seed, env_id, draw):
    return int(lib().orc_philox_u32(seed, env_id, draw))


def philox_block(ctr, key):
    c = np.array(ctr, np.uint32)
    k = np.array(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().orc_philox_block(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return [int(x) for x in o]


def state_to_flat(st, n_snakes):
    """canonical dict (tools/gen_golden.py canon_state) -> flat int32 words (snake_oracle.c)."""
    ctr = int(st.get("ctr", 0))
    w = [int(st.get("t", 0)), ctr & 0xFFFFFFFF, ctr >> 32, int(st.get("spare_fruits", 0)),
         int(st.get("ep_len", 0)),
         int(np.array([st.get("ep_return", 0.0)], np.float32).view(np.uint32)[0]),
         len(st["fruits"]), n_snakes | (0x100 if st.get("finished") else 0)]
    for f in st["fruits"]:
        w += [int(f[0]), int(f[1])]
    for s in range(n_snakes):
        body = st["snakes"][s]
        alive = st["alive"][s] if "alive" in st else True
        in_dead = st["in_dead"][s] if "in_dead" in st else False
        w += [len(body), int(st["vels"][s][0]), int(st["vels"][s][1]), int(st["grow_to"][s]),
              int(bool(alive)), int(bool(in_dead))]
        for c in body:
            w += [int(c[0]), int(c[1])]
    return np.array([x - (1 << 32) if x >= (1 << 31) else x for x in w], dtype=np.int32)


def flat_finished(buf):
    """Bit 8 of word 7 of the PRODUCT's canonical words (include/msnake.h): the episode has ended and the env was
    not reset since.  A vec-layer bookkeeping bit of the product (episode totals count once); the reference and
    this oracle have no such state, so flat_to_state() leaves it out of the dict."""
    return bool(int(buf[7]) & 0x100)


def flat_to_state(buf):
    buf = [int(x) for x in buf]
    k = 0
    st = {"t": buf[0], "ctr": (buf[1] & 0xFFFFFFFF) | ((buf[2] & 0xFFFFFFFF) << 32),
          "spare_fruits": buf[3], "ep_len": buf[4],
          "ep_return": float(np.array([buf[5] & 0xFFFFFFFF], np.uint32).view(np.float32)[0])}
    nf, n = buf[6], buf[7] & 0xFF
    k = 8
    st["fruits"] = [[buf[k + 2 * i], buf[k + 2 * i + 1]] for i in range(nf)]
    k += 2 * nf
    st["snakes"], st["vels"], st["grow_to"], st["alive"], st["in_dead"] = [], [], [], [], []
    for _ in range(n):
        ln = buf[k]
        st["vels"].append([buf[k + 1], buf[k + 2]])
        st["grow_to"].append(buf[k + 3])
        st["alive"].append(bool(buf[k + 4]))
        st["in_dead"].append(bool(buf[k + 5]))
        k += 6
        st["snakes"].append([[buf[k + 2 * i], buf[k + 2 * i + 1]] for i in range(ln)])
        k += 2 * ln
    return st


class Oracle:
    """Batch of reference-rule envs behind the vec-layer contract (auto reset, episode stats)."""

    def __init__(self, num_envs, dim=19, n_snakes=3, n_fruits=None, rules="snake_env", seed=0,
                 env_id_base=0, max_steps=2000, auto_reset=True):
        self.L = lib()
        r = RULES[rules] if isinstance(rules, str) else int(rules)
        if n_fruits is None:
            n_fruits = n_snakes
        self.cfg = _Config(num_envs, dim, n_snakes, n_fruits, r, max_steps, int(auto_reset), 0, seed,
                           env_id_base)
        self.h = self.L.orc_create(ctypes.byref(self.cfg))
        if not self.h:
            raise ValueError("orc_create rejected the configuration")
        H, W, C = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self.L.orc_obs_shape(self.h, ctypes.byref(H), ctypes.byref(W), ctypes.byref(C))
        self.num_envs, self.n_snakes = num_envs, n_snakes
        self.obs_shape = (H.value, W.value, C.value)
        self.obs = np.zeros((num_envs,) + self.obs_shape, np.uint8)
        self.rew = np.zeros(num_envs, np.float32)
        self.done = np.zeros(num_envs, np.uint8)
        self.num_snakes = np.zeros(num_envs, np.int32)
        self.ep_return = np.zeros(num_envs, np.float32)
        self.ep_len = np.zeros(num_envs, np.int32)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.orc_reset(self.h, self.obs.ctypes.data)
        return self.obs

    def render(self):
        self.L.orc_render(self.h, self.obs.ctypes.data)
        return self.obs

    def step(self, actions, threads=1, want_obs=True):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.ndim == 2 and a.shape[0] == self.num_envs and a.shape[1] >= self.n_snakes
        obs_ptr = self.obs.ctypes.data if want_obs else None
        args = (a.ctypes.data, a.shape[1], obs_ptr, self.rew.ctypes.data, self.done.ctypes.data,
                self.num_snakes.ctypes.data, self.ep_return.ctypes.data, self.ep_len.ctypes.data)
        if threads > 1:
            self.L.orc_step_mt(self.h, threads, *args)
        else:
            self.L.orc_step(self.h, *args)
        return self.obs, self.rew, self.done, self.num_snakes, self.ep_return, self.ep_len

    def rollout(self, tape, threads):
        """T lockstep steps of tape int32 [T, num_envs, stride] with one persistent thread team."""
        a = np.ascontiguousarray(tape, dtype=np.int32)
        assert a.ndim == 3 and a.shape[1] == self.num_envs and a.shape[2] >= self.n_snakes
        self.L.orc_rollout_mt(self.h, threads, a.ctypes.data, a.shape[2], a.shape[0], self.obs.ctypes.data,
                              self.rew.ctypes.data, self.done.ctypes.data, self.num_snakes.ctypes.data,
                              self.ep_return.ctypes.data, self.ep_len.ctypes.data)
        return self.obs, self.rew, self.done, self.num_snakes, self.ep_return, self.ep_len

    def get_state(self, env):
        n = self.L.orc_export_state(self.h, env, None, 0)
        buf = np.zeros(n, np.int32)
        self.L.orc_export_state(self.h, env, buf.ctypes.data, n)
        return flat_to_state(buf)

    def set_state(self, env, st):
        buf = state_to_flat(st, self.n_snakes)
        rc = self.L.orc_import_state(self.h, env, buf.ctypes.data, len(buf))
        if rc != 0:
            raise ValueError(f"orc_import_state failed: {rc}")
