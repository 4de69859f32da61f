#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE in this container.

Runs only where /root/reference exists (the build container).  Nothing from the
reference is copied: its modules are loaded by file path, stepped, and only
inputs / observed outputs (data) are written to tests/golden/.

What is loaded:
  [S]  src/gym-snake/gym_snake/envs/snake_multiple_test.py   SnakeEnv
  [A]  src/gym-snake/gym_snake/envs/snake_adversarial_env.py SnakeAdversarial
  [N]  src/gym-snake/gym_snake/core/new_world.py             World / Snake
  [NE] src/gym-snake/gym_snake/envs/snake_multiple_env_new.py NewMultipleSnakes
`gym` is absent here, so a minimal in-memory stub supplies the base class and
the names those files import.  The reference only ever calls `.randint(n)` on
its RNG, so a Philox4x32-10 backed object is injected as `np_random`: draw i of
global env e is word (i & 3) of philox(counter={i>>2, 0, e, 0}, key=seed) and
randint(n) = (u32 * n) >> 32.  That is the RNG contract of the whole repo.

The vec layer (auto-reset returning the reset observation with the terminal
reward/done, and Monitor's episode return/length) is restated here in
`VecHarness` from src/baselines/common/vec_env/subproc_vec_env.py:13-16 and
src/baselines/bench/monitor.py:57-78, because those files need packages that are
not installed (TF / mpi4py / cloudpickle imports in baselines/__init__ chain).

Usage: python tools/gen_golden.py            (rewrites tests/golden/*)
"""
import contextlib

import io
import json
import os
import sys
import types
import zlib

import numpy as np

REF = "/root/reference"
REF_SRC = os.path.join(REF, "src")
ENVS = os.path.join(REF_SRC, "gym-snake", "gym_snake", "envs")
CORE = os.path.join(REF_SRC, "gym-snake", "gym_snake", "core")
OUT = os.environ.get("MSNAKE_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al. 2011), plain Python ints."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return (c0, c1, c2, c3)


def philox_u32(seed, env_id, draw):
    blk = draw >> 2
    out = philox4x32_10((blk & MASK, (blk >> 32) & MASK, env_id & MASK, (env_id >> 32) & MASK),
                        (seed & MASK, (seed >> 32) & MASK))
    return out[draw & 3]


class PhiloxRandint:
    """The only RNG surface the reference env code touches: .randint(n)."""

    def __init__(self, seed, env_id, ctr=0):
        self.seed, self.env_id, self.ctr = seed, env_id, ctr
        self.log = []

    def randint(self, n):
        u = philox_u32(self.seed, self.env_id, self.ctr)
        self.ctr += 1
        v = (u * int(n)) >> 32
        self.log.append((int(n), v))
        return v


# ----------------------------------------------------------------------------- gym stub
def install_gym_stub():
    gym = types.ModuleType("gym")

    class Env:
        pass

    class ObservationWrapper(Env):
        pass

    gym.Env = Env
    gym.ObservationWrapper = ObservationWrapper
    spaces = types.ModuleType("gym.spaces")

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low=0, high=255, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces.Discrete, spaces.Box = Discrete, Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(0), seed)
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    cc = types.ModuleType("gym.envs.classic_control")
    rendering = types.ModuleType("gym.envs.classic_control.rendering")
    cc.rendering = rendering
    envs.classic_control = cc
    gym.spaces, gym.utils, gym.envs = spaces, utils, envs
    for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding), ("gym.envs", envs),
                      ("gym.envs.classic_control", cc),
                      ("gym.envs.classic_control.rendering", rendering)]:
        sys.modules[name] = mod


def load_by_path(name, path):
    """Execute the reference SOURCE TEXT of `path` as module `name`.  compile() of the text that is
    read here, never a cached .pyc found beside it (the reference tree is untrusted and read-only:
    a stale or planted __pycache__ entry with a matching header must not be able to define the
    fixtures), and nothing is written back into that tree."""
    with open(path, "r", encoding="utf-8") as f:
        src = f.read()
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src, path, "exec", dont_inherit=True), mod.__dict__)
    return mod


def load_reference():
    sys.dont_write_bytecode = True
    install_gym_stub()
    # [S] / [A] do `from config import Config`: give them the by-path module instead of a sys.path import
    cfg = load_by_path("config", os.path.join(REF_SRC, "config.py"))
    S = load_by_path("ref_snake_multiple_test", os.path.join(ENVS, "snake_multiple_test.py"))
    A = load_by_path("ref_snake_adversarial", os.path.join(ENVS, "snake_adversarial_env.py"))
    N = load_by_path("ref_new_world", os.path.join(CORE, "new_world.py"))
    # [NE] does `from gym_snake.core.new_world import Snake, World`; give it the by-path module
    pkg = types.ModuleType("gym_snake")
    pkg.__path__ = []
    core = types.ModuleType("gym_snake.core")
    core.__path__ = []
    core.new_world = N
    pkg.core = core
    sys.modules["gym_snake"] = pkg
    sys.modules["gym_snake.core"] = core
    sys.modules["gym_snake.core.new_world"] = N
    NE = load_by_path("ref_snake_multiple_env_new", os.path.join(ENVS, "snake_multiple_env_new.py"))
    return S, A, N, NE, cfg.Config


S_MOD, A_MOD, N_MOD, NE_MOD, Config = load_reference()

RULES_SNAKE_ENV, RULES_NEW_WORLD, RULES_ADVERSARIAL = 0, 1, 2
RULE_NAMES = {0: "snake_env", 1: "new_world", 2: "adversarial"}


# ----------------------------------------------------------------------------- env factories
def make_ref_env(rules, dim, n_snakes, n_fruits, seed, env_id):
    """One reference env instance wired to the Philox shim. Not yet reset."""
    rng = PhiloxRandint(seed, env_id)
    if rules == RULES_SNAKE_ENV:
        Config.set_num_snakes(n_snakes)
        env = S_MOD.SnakeEnv()
        env.dim = dim
        env.np_random = rng
    elif rules == RULES_ADVERSARIAL:
        Config.set_num_snakes(n_snakes)
        env = A_MOD.SnakeAdversarial()
        env.dim = dim
        env.np_random = rng
    else:
        # [NE].__init__ seeds then builds a World (draws!).  The build's contract is that a
        # handle is created and then reset() is called: __init__'s draws must not be part of the
        # stream, so construct without running __init__ and let reset() build the first World.
        env = NE_MOD.NewMultipleSnakes.__new__(NE_MOD.NewMultipleSnakes)
        env.SIZE = (dim, dim)
        env.dim = dim
        env.current_step = 0
        env.n_snakes = n_snakes
        env.n_fruits = n_fruits
        env.screen_res = 300
        env.np_rand = rng
        env.viewer = None
    env._rng = rng
    env._rules = rules
    env._n_snakes = n_snakes
    return env


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def canon_state(env):
    """Canonical, JSON-able snapshot of everything the rules read."""
    r = env._rules
    if r in (RULES_SNAKE_ENV, RULES_ADVERSARIAL):
        snakes, fruits, vels, grow, t = env.state
        st = {
            "snakes": [[list(map(int, c)) for c in s] for s in snakes],
            "fruits": [list(map(int, f)) for f in fruits],
            "vels": [list(map(int, v)) for v in vels[:len(snakes)]],
            "grow_to": [int(g) for g in grow[:len(snakes)]],
            "t": int(t),
            "ctr": int(env._rng.ctr),
        }
        if r == RULES_ADVERSARIAL:
            st["spare_fruits"] = int(env.spare_fruits)
        return st
    w = env.world
    return {
        "snakes": [[list(map(int, c)) for c in s.snake_body] for s in w.snakes],
        "fruits": [list(map(int, f)) for f in w.fruits],
        "vels": [list(map(int, s.direction)) for s in w.snakes],
        "grow_to": [int(s.snake_length) for s in w.snakes],
        "alive": [bool(s.alive) for s in w.snakes],
        "in_dead": [bool(s in w.dead_snakes) for s in w.snakes],
        "t": int(env.current_step),
        "ctr": int(env._rng.ctr),
    }


def set_state(env, st):
    """Install a hand-built state into a reference env (used for edge cases)."""
    r = env._rules
    if r in (RULES_SNAKE_ENV, RULES_ADVERSARIAL):
        n = len(st["snakes"])
        vels = [tuple(v) for v in st["vels"]] + [(0, 0)] * (3 - n)
        grow = list(st["grow_to"]) + [3] * (3 - n)
        env.state = [[[tuple(c) for c in s] for s in st["snakes"]],
                     [tuple(f) for f in st["fruits"]], vels, grow, st["t"]]
        if r == RULES_ADVERSARIAL:
            env.spare_fruits = st.get("spare_fruits", 0)
    else:
        dim = env.dim
        w = N_MOD.World.__new__(N_MOD.World)
        w.size = (dim, dim)
        w.dim = dim
        w.world = np.zeros((dim, dim))
        w.np_rand = env._rng
        w.is_competitive = False
        w.snakes, w.dead_snakes, w.fruits, w.time_step = [], [], [], 0
        for i, body in enumerate(st["snakes"]):
            s = N_MOD.Snake(i, tuple(body[0]) if body else (0, 0), tuple(st["vels"][i]))
            s.snake_body = [tuple(c) for c in body]
            s.snake_length = st["grow_to"][i]
            s.alive = st["alive"][i]
            w.snakes.append(s)
        for i, s in enumerate(w.snakes):
            if st["in_dead"][i]:
                w.dead_snakes.append(s)
        w.fruits = [tuple(f) for f in st["fruits"]]
        env.world = w
        env.current_step = st["t"]
    env._rng.ctr = st["ctr"]


def current_obs(env):
    if env._rules == RULES_NEW_WORLD:
        return env.world.get_multi_snake_obs()
    return env.get_multi_snake_ob()


class VecHarness:
    """E reference envs + the auto-reset / episode-stat duties of the vec layer.

    subproc_vec_env.py:13-16  -> on done the returned ob is the reset ob, reward/done/info are
                                 the terminal step's.
    monitor.py:57-78          -> info['episode'] = {'r': sum(rewards), 'l': len(rewards)} on done.
    """

    def __init__(self, rules, dim, n_snakes, n_fruits, seed, num_envs, env_id_base=0, auto_reset=True):
        self.auto_reset = auto_reset
        self.envs = [make_ref_env(rules, dim, n_snakes, n_fruits, seed, env_id_base + e)
                     for e in range(num_envs)]
        self.ep_r = [0.0] * num_envs
        self.ep_l = [0] * num_envs

    def reset(self):
        self.ep_r = [0.0] * len(self.envs)
        self.ep_l = [0] * len(self.envs)
        return np.stack([quiet(e.reset) for e in self.envs])

    def step(self, actions):
        obs, rew, done, nsn, epr, epl = [], [], [], [], [], []
        for i, env in enumerate(self.envs):
            ob, r, d, info = quiet(env.step, [int(a) for a in actions[i]])
            self.ep_r[i] += r
            self.ep_l[i] += 1
            er, el = 0.0, 0
            if d:
                er, el = round(self.ep_r[i], 6), self.ep_l[i]
                if self.auto_reset:
                    self.ep_r[i], self.ep_l[i] = 0.0, 0
                    ob = quiet(env.reset)
                # auto_reset=False is the raw gym env: stepping continues on the terminal state and
                # the running totals keep accumulating (reported on every done step).
            obs.append(ob)
            rew.append(float(r))
            done.append(bool(d))
            nsn.append(int(info["num_snakes"]))
            epr.append(float(er))
            epl.append(int(el))
        return (np.stack(obs), np.array(rew, np.float32), np.array(done, np.uint8),
                np.array(nsn, np.int32), np.array(epr, np.float32), np.array(epl, np.int32))


# ----------------------------------------------------------------------------- action tapes
DIRS = {1: (1, 0), 2: (0, 1), 3: (-1, 0), 4: (0, -1)}


def policy_actions(env, n_snakes, rs, eps):
    """eps-greedy 'walk towards the nearest fruit' so that tapes contain eats, long bodies,
    self hits and mutual hits, not only the ~19-step random-walk episodes."""
    st = canon_state(env)
    acts = []
    for s in range(n_snakes):
        body = st["snakes"][s] if s < len(st["snakes"]) else []
        if not body or rs.random() < eps or not st["fruits"]:
            acts.append(int(rs.integers(0, 5)))
            continue
        h = body[0]
        f = min(st["fruits"], key=lambda f: abs(f[0] - h[0]) + abs(f[1] - h[1]))
        cand = []
        if f[0] > h[0]:
            cand.append(1)
        if f[0] < h[0]:
            cand.append(3)
        if f[1] > h[1]:
            cand.append(2)
        if f[1] < h[1]:
            cand.append(4)
        acts.append(int(cand[rs.integers(0, len(cand))]) if cand else int(rs.integers(0, 5)))
    return acts


def pack_states(states, n_snakes, n_fruit_cap):
    """states[t][e] -> padded arrays."""
    T, E = len(states), len(states[0])
    lmax = max(1, max(len(b) for row in states for st in row for b in st["snakes"]))
    fmax = max(1, max(len(st["fruits"]) for row in states for st in row))
    fmax = max(fmax, n_fruit_cap)
    body = np.full((T, E, n_snakes, lmax, 2), -128, np.int8)
    blen = np.zeros((T, E, n_snakes), np.int16)
    fruits = np.full((T, E, fmax, 2), -128, np.int8)
    nfr = np.zeros((T, E), np.int16)
    vels = np.zeros((T, E, n_snakes, 2), np.int8)
    grow = np.zeros((T, E, n_snakes), np.int32)
    tt = np.zeros((T, E), np.int32)
    ctr = np.zeros((T, E), np.int64)
    alive = np.zeros((T, E, n_snakes), np.uint8)
    in_dead = np.zeros((T, E, n_snakes), np.uint8)
    spare = np.zeros((T, E), np.int32)
    for t in range(T):
        for e in range(E):
            st = states[t][e]
            for s, b in enumerate(st["snakes"]):
                blen[t, e, s] = len(b)
                if b:
                    body[t, e, s, :len(b)] = np.array(b, np.int8)
                vels[t, e, s] = st["vels"][s]
                grow[t, e, s] = st["grow_to"][s]
                if "alive" in st:
                    alive[t, e, s] = st["alive"][s]
                    in_dead[t, e, s] = st["in_dead"][s]
            nfr[t, e] = len(st["fruits"])
            if st["fruits"]:
                fruits[t, e, :len(st["fruits"])] = np.array(st["fruits"], np.int8)
            tt[t, e] = st["t"]
            ctr[t, e] = st["ctr"]
            spare[t, e] = st.get("spare_fruits", 0)
    return dict(st_body=body, st_len=blen, st_fruits=fruits, st_nfruits=nfr, st_vels=vels,
                st_grow=grow, st_t=tt, st_ctr=ctr, st_alive=alive, st_in_dead=in_dead,
                st_spare=spare)


def gen_tape(name, rules, dim, n_snakes, n_fruits, seed, num_envs, steps, env_id_base=0,
             action_width=None, full_obs_every=32, auto_reset=True):
    """Random + eps-greedy tape through the vec harness; everything observable is stored."""
    rs = np.random.default_rng(zlib.crc32(name.encode()))
    vec = VecHarness(rules, dim, n_snakes, n_fruits, seed, num_envs, env_id_base, auto_reset)
    aw = action_width or n_snakes
    obs0 = vec.reset()
    states0 = [canon_state(e) for e in vec.envs]
    actions = np.zeros((steps, num_envs, aw), np.int32)
    rew = np.zeros((steps, num_envs), np.float32)
    done = np.zeros((steps, num_envs), np.uint8)
    nsn = np.zeros((steps, num_envs), np.int32)
    epr = np.zeros((steps, num_envs), np.float32)
    epl = np.zeros((steps, num_envs), np.int32)
    crc = np.zeros((steps, num_envs), np.uint32)
    full = []
    full_t = []
    states = []
    # per-env exploration rate: a third pure random, the rest increasingly greedy
    eps = np.where(np.arange(num_envs) % 3 == 0, 1.0, np.linspace(0.5, 0.02, num_envs))
    for t in range(steps):
        for e, env in enumerate(vec.envs):
            a = policy_actions(env, n_snakes, rs, eps[e])
            a = a + [int(rs.integers(0, 5)) for _ in range(aw - n_snakes)]  # surplus, ignored
            actions[t, e] = a
        ob, r, d, ns, er, el = vec.step(actions[t])
        rew[t], done[t], nsn[t], epr[t], epl[t] = r, d, ns, er, el
        crc[t] = [zlib.crc32(np.ascontiguousarray(ob[e]).tobytes()) for e in range(num_envs)]
        if t % full_obs_every == 0 or t == steps - 1:
            full.append(ob)
            full_t.append(t)
        states.append([canon_state(e) for e in vec.envs])
    out = dict(
        meta=np.array(json.dumps(dict(name=name, rules=rules, rules_name=RULE_NAMES[rules], dim=dim,
                                      n_snakes=n_snakes, n_fruits=n_fruits, seed=seed,
                                      num_envs=num_envs, steps=steps, env_id_base=env_id_base,
                                      action_width=aw, max_steps=2000,
                                      auto_reset=bool(auto_reset)))),
        actions=actions.astype(np.int8), reward=rew, done=done, num_snakes=nsn.astype(np.int8),
        ep_return=epr, ep_len=epl, obs_crc=crc, obs0=obs0,
        full_obs=np.stack(full), full_obs_t=np.array(full_t, np.int32))
    out.update({"s0_" + k[3:]: v[0] for k, v in pack_states([states0], n_snakes, n_fruits).items()})
    out.update(pack_states(states, n_snakes, n_fruits))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    ndone = int(done.sum())
    maxlen = int(out["st_len"].max())
    eats = float((rew > 0).sum())
    print(f"{name}: {num_envs} envs x {steps} steps, episodes={ndone}, max body={maxlen}, "
          f"eat-steps(main)={eats:.0f}, {os.path.getsize(path) / 1024:.0f} KiB")


# ----------------------------------------------------------------------------- edge cases
def obs_blob(ob):
    return {"shape": list(ob.shape), "crc32": int(zlib.crc32(np.ascontiguousarray(ob).tobytes())),
            "z": zlib.compress(np.ascontiguousarray(ob).tobytes(), 9).hex()}


def run_edge(name, rules, dim, n_snakes, n_fruits, state, action_steps, seed=7, env_id=3, note=""):
    env = make_ref_env(rules, dim, n_snakes, n_fruits, seed, env_id)
    if rules in (RULES_SNAKE_ENV, RULES_ADVERSARIAL):
        quiet(env.reset)  # so attributes exist; then overwrite
    set_state(env, state)
    rec = {"name": name, "note": note, "rules": rules, "rules_name": RULE_NAMES[rules], "dim": dim,
           "n_snakes": n_snakes, "n_fruits": n_fruits, "seed": seed, "env_id": env_id,
           "state0": canon_state(env), "obs0": obs_blob(current_obs(env)), "steps": []}
    for acts in action_steps:
        env._rng.log = []
        if acts == "reset":
            ob = quiet(env.reset)
            rec["steps"].append({"actions": "reset", "state": canon_state(env), "obs": obs_blob(ob),
                                 "draws": list(map(list, env._rng.log))})
            continue
        ob, r, d, info = quiet(env.step, list(acts))
        rec["steps"].append({"actions": list(acts), "reward": float(r), "done": bool(d),
                             "num_snakes": int(info["num_snakes"]), "state": canon_state(env),
                             "obs": obs_blob(ob), "draws": list(map(list, env._rng.log))})
    return rec


def S_state(snakes, fruits, vels=None, grow=None, t=0, ctr=0, spare=None):
    n = len(snakes)
    st = {"snakes": snakes, "fruits": fruits, "vels": vels or [[0, 0]] * n,
          "grow_to": grow or [3] * n, "t": t, "ctr": ctr}
    if spare is not None:
        st["spare_fruits"] = spare
    return st


def N_state(snakes, fruits, vels=None, grow=None, alive=None, in_dead=None, t=0, ctr=0):
    n = len(snakes)
    return {"snakes": snakes, "fruits": fruits, "vels": vels or [[0, 0]] * n, "grow_to": grow or [3] * n,
            "alive": alive or [True] * n, "in_dead": in_dead or [False] * n, "t": t, "ctr": ctr}


def gen_edges():
    E = []
    S, N, A = RULES_SNAKE_ENV, RULES_NEW_WORLD, RULES_ADVERSARIAL
    far = [[18, 18], [18, 17], [17, 18]]
    # --- [S] (SURVEY appendix A1-A6 and more)
    E.append(run_edge("S_head_to_head", S, 19, 3, 3,
                      S_state([[[5, 5]], [[7, 5]], [[0, 0]]], far), [[1, 3, 0]],
                      note="A1: both heads enter (6,5): both die"))
    E.append(run_edge("S_tail_follow", S, 19, 3, 3,
                      S_state([[[5, 5], [4, 5], [3, 5]], [[2, 5], [1, 5], [0, 5]], [[0, 0]]], far,
                              vels=[[1, 0], [1, 0], [0, 0]]), [[0, 0, 0], [0, 0, 0]],
                      note="A2: entering a just-vacated tail cell is legal"))
    E.append(run_edge("S_oob_alias_respawn", S, 19, 3, 3,
                      S_state([[[18, 3]], [[5, 5]], [[0, 0]]], [[6, 5], [18, 18], [18, 17]],
                              vels=[[1, 0], [1, 0], [0, 0]]), [[0, 0, 0]],
                      note="A3: out-of-grid head aliases in safe_choose_cell's used index"))
    E.append(run_edge("S_coincident_fruits", S, 19, 3, 3,
                      S_state([[[5, 5]], [[9, 9]], [[0, 0]]], [[6, 5], [6, 5], [18, 17]],
                              vels=[[1, 0], [0, 0], [0, 0]]), [[0, 0, 0], [0, 0, 0], [0, 2, 0]],
                      note="A4/A5: two fruits on one cell: reward 2, grow +4, two respawn draws"))
    E.append(run_edge("S_reset_draw_order", S, 19, 3, 3,
                      S_state([[[5, 5]], [[9, 9]], [[0, 0]]], far), ["reset", [1, 2, 3], "reset"],
                      note="A6: 12 draws interleaved snake cell / fruit cell"))
    E.append(run_edge("S_no_reverse", S, 19, 3, 3,
                      S_state([[[5, 5], [4, 5]], [[9, 9], [9, 8]], [[3, 3], [4, 3]]], far,
                              vels=[[1, 0], [0, 1], [-1, 0]]), [[3, 4, 1], [2, 1, 4], [0, 0, 0]],
                      note="180-degree turns are ignored; other turns apply"))
    E.append(run_edge("S_self_hit", S, 19, 3, 3,
                      S_state([[[5, 5], [5, 6], [6, 6], [6, 5], [6, 4]], [[12, 12]], [[0, 0]]], far,
                              vels=[[0, -1], [0, 0], [0, 0]], grow=[9, 3, 3]), [[1, 0, 0]],
                      note="head runs into own body piece (6,5)"))
    E.append(run_edge("S_own_tail_vacated", S, 19, 3, 3,
                      S_state([[[5, 5], [5, 6], [6, 6], [6, 5]], [[12, 12]], [[0, 0]]], far,
                              vels=[[0, -1], [0, 0], [0, 0]], grow=[4, 3, 3]), [[1, 0, 0]],
                      note="head enters own tail cell that is popped this step: legal"))
    E.append(run_edge("S_own_tail_growing", S, 19, 3, 3,
                      S_state([[[5, 5], [5, 6], [6, 6], [6, 5]], [[12, 12]], [[0, 0]]], far,
                              vels=[[0, -1], [0, 0], [0, 0]], grow=[6, 3, 3]), [[1, 0, 0]],
                      note="same but still growing: tail not popped, so it is a hit"))
    E.append(run_edge("S_hit_stationary", S, 19, 3, 3,
                      S_state([[[5, 5]], [[6, 5]], [[0, 0]]], far), [[1, 0, 0]],
                      note="moving into a zero-velocity snake kills only the mover"))
    E.append(run_edge("S_walls", S, 19, 3, 3,
                      S_state([[[0, 7]], [[7, 18]], [[18, 0]]], far,
                              vels=[[-1, 0], [0, 1], [0, -1]]), [[0, 0, 0]],
                      note="three different walls in one step"))
    E.append(run_edge("S_opponent_dies_main_lives", S, 19, 3, 3,
                      S_state([[[5, 5]], [[0, 9]], [[9, 9]]], far, vels=[[1, 0], [-1, 0], [0, 0]]),
                      [[0, 0, 0], [0, 0, 0], [0, 0, 4]], note="dead opponents stay empty; num_snakes"))
    E.append(run_edge("S_fruit_under_snake", S, 19, 3, 3,
                      S_state([[[5, 5], [4, 5]], [[9, 9]], [[0, 0]]], [[4, 5], [9, 9], [0, 0]],
                              vels=[[1, 0], [0, 0], [0, 0]]), [[0, 0, 0]],
                      note="render order: fruits first, snakes over them"))
    E.append(run_edge("S_eat_then_next_eats_respawn", S, 10, 2, 2,
                      S_state([[[3, 3]], [[7, 7]]], [[4, 3], [0, 0]], vels=[[1, 0], [0, 0]]),
                      [[0, 0], [0, 1], [2, 2]], seed=11, env_id=0,
                      note="10x10 two snakes; eat, respawn"))
    E.append(run_edge("S_step_cap", S, 19, 3, 3,
                      S_state([[[5, 5]], [[9, 9]], [[0, 0]]], far, t=1998), [[0, 0, 0], [0, 0, 0]],
                      note="done at t>=2000 while alive: reward stays 0"))
    E.append(run_edge("S_one_snake_views", S, 10, 1, 1,
                      S_state([[[3, 3], [2, 3]]], [[8, 8]], vels=[[1, 0]]), [[0], [2], [2]],
                      note="NUM_SNAKES=1: 9 channels, views 1,2 show the snake blue"))
    E.append(run_edge("S_two_snake_views", S, 19, 2, 2,
                      S_state([[[3, 3], [2, 3]], [[10, 10], [10, 11]]], [[8, 8], [1, 1]],
                              vels=[[1, 0], [0, -1]]), [[0, 0], [2, 3]],
                      note="NUM_SNAKES=2 (config 5 shape)"))
    E.append(run_edge("S_full_row_respawn", S, 10, 1, 1,
                      S_state([[[c, 0] for c in range(9, -1, -1)] + [[0, 1], [1, 1]]], [[9, 1]],
                              vels=[[1, 0]], grow=[12]),
                      [[2]], seed=5, env_id=9,
                      note="long body; k-th free cell must skip a fully used row"))
    # 3x3 board: the fruit sits on the last free cell -> no free cell left: x=0, NO draw
    E.append(run_edge("S_no_free_cell", S, 3, 1, 1,
                      S_state([[[1, 2], [0, 2], [0, 1], [1, 1], [2, 1], [2, 0], [1, 0], [0, 0]]],
                              [[2, 2]], vels=[[1, 0]], grow=[20]),
                      [[0], [0]], note="available empty: fruit -> (0,0) without consuming a draw"))
    # 10x10 boustrophedon of 99 cells about to eat the 100th cell (ring capacity = dim*dim + 1)
    path = []
    for r in range(10):
        cols = range(10) if r % 2 == 0 else range(9, -1, -1)
        path += [[c, r] for c in cols]
    body99 = path[:99][::-1]            # head is path[98] = (1,9); path[99] = (0,9)
    E.append(run_edge("S_full_board_10", S, 10, 1, 1,
                      S_state([body99], [[0, 9]], vels=[[-1, 0]], grow=[150]),
                      [[0], [4]], note="body reaches dim*dim cells, then self hit"))
    # a fruit respawned by snake 0 is eaten by snake 1 in the SAME step (sequential update order)
    tgt_k = 84  # cell (7,8) -> index 87, minus the 3 used indices {33,34,77} below it
    ctr = next(c for c in range(200000) if (philox_u32(11, 0, c) * 97) >> 32 == tgt_k)
    E.append(run_edge("S_respawn_eaten_same_step", S, 10, 2, 2,
                      S_state([[[3, 3]], [[7, 7]]], [[4, 3], [0, 0]], vels=[[1, 0], [0, 1]], ctr=ctr),
                      [[0, 0]], seed=11, env_id=0,
                      note="snake 0 eats, fruit respawns on (7,8), snake 1 steps onto it and eats it"))
    E.append(run_edge("S_invalid_action", S, 19, 3, 3,
                      S_state([[[5, 5]], [[9, 9]], [[0, 0]]], far, vels=[[1, 0], [0, 1], [0, 0]]),
                      [[7, -1, 5]], note="actions outside 1..4 keep the velocity"))
    # --- [S] ordering between one snake's respawn and the other snakes' moves ([S]:119-141, 171-176)
    E.append(run_edge("S_two_heads_one_fruit", S, 19, 3, 3,
                      S_state([[[5, 5]], [[7, 5]], [[0, 0]]], [[6, 5], [18, 17], [17, 18]]), [[1, 3, 0]],
                      note="both heads enter the fruit cell: only snake 0 eats (the fruit has moved on)"))
    E.append(run_edge("S_opponents_share_fruit", S, 19, 3, 3,
                      S_state([[[10, 10]], [[5, 5]], [[7, 5]]], [[18, 18], [6, 5], [17, 18]]),
                      [[0, 1, 3], [0, 0, 0]],
                      note="snakes 1 and 2 enter fruit 1's cell: snake 1 eats, both die, main plays on"))
    # 3x3: the eater is at full length, so its popped tail cell (0,0) is free again for the respawn
    for seed in range(64):
        rec = run_edge("S_respawn_on_vacated_tail", S, 3, 1, 1,
                       S_state([[[0, 2], [0, 1], [1, 1], [2, 1], [2, 0], [1, 0], [0, 0]]], [[1, 2]],
                               vels=[[1, 0]], grow=[5]), [[0]], seed=seed, env_id=0,
                       note="free cells after the move are the popped tail (0,0) and (2,2); lands on (0,0)")
        if rec["steps"][0]["state"]["fruits"][0] == [0, 0]:
            break
    else:
        raise SystemExit("no seed puts the fruit on the vacated tail")
    E.append(rec)
    # 3x3, two snakes: snake 0's respawn sees snake 1 UNMOVED (its tail (0,2) still there, its next
    # head cell (2,1) still free); the fruit lands exactly on (2,1) and snake 1 eats it this step
    for seed in range(256):
        rec = run_edge("S_respawn_sees_later_unmoved", S, 3, 2, 2,
                       S_state([[[0, 0]], [[2, 2], [1, 2], [0, 2]]], [[1, 0], [0, 0]],
                               vels=[[1, 0], [0, -1]]), [[0, 0], [0, 0]], seed=seed, env_id=0,
                       note="fruit 0 respawns on snake 1's next head cell and is eaten again at once")
        if rec["steps"][0]["state"]["grow_to"][1] == 5:
            break
    else:
        raise SystemExit("no seed puts the fruit on snake 1's next head cell")
    E.append(rec)
    # --- [N] (A7-A10)
    E.append(run_edge("N_init_draws", N, 19, 3, 3,
                      N_state([[[5, 5]], [[9, 9]], [[0, 0]]], far), ["reset"],
                      note="A7: 6 x randint(19) then 3 x randint(358)"))
    E.append(run_edge("N_zero_velocity_stack_and_revive", N, 19, 3, 3,
                      N_state([[[5, 5]], [[9, 9]], [[0, 0]]], far),
                      [[0, 0, 0], [1, 1, 1], [0, 0, 0], [0, 0, 0], [0, 0, 0]],
                      note="A8: duplicates, alive False then True again, inverted done"))
    E.append(run_edge("N_sequential_head_to_head", N, 19, 2, 2,
                      N_state([[[5, 5]], [[7, 5]]], far[:2]), [[1, 3], [0, 0]],
                      note="A9: only the lower index dies"))
    E.append(run_edge("N_second_fruit_pop", N, 19, 1, 2,
                      N_state([[[5, 5], [4, 5], [3, 5]]], [[0, 0], [6, 5]], vels=[[1, 0]]), [[0]],
                      note="A10a: eating fruit index 1 at full length pops first"))
    E.append(run_edge("N_first_fruit_no_pop", N, 19, 1, 2,
                      N_state([[[5, 5], [4, 5], [3, 5]]], [[6, 5], [0, 0]], vels=[[1, 0]]), [[0]],
                      note="A10b: eating fruit index 0 grows before the pop test"))
    E.append(run_edge("N_self_hit_keeps_body", N, 19, 2, 2,
                      N_state([[[5, 5], [5, 6], [6, 6], [6, 5], [6, 4]], [[12, 12]]], far[:2],
                              vels=[[0, -1], [0, 0]], grow=[9, 3]), [[1, 0], [0, 0], [0, 0]],
                      note="self hit: alive False, body kept, keeps moving, not rendered"))
    E.append(run_edge("N_wall_clears", N, 10, 2, 4,
                      N_state([[[9, 3], [8, 3]], [[2, 2]]], [[0, 0], [5, 5], [6, 6], [7, 7]],
                              vels=[[1, 0], [0, 0]]), [[0, 2], [0, 0]],
                      note="out of grid: body cleared at once; dead_snakes append-once"))
    E.append(run_edge("N_hit_dead_body", N, 19, 2, 2,
                      N_state([[[5, 5]], [[6, 5], [6, 6], [6, 7], [6, 6]]], far[:2],
                              vels=[[0, 0], [0, -1]], grow=[3, 9], alive=[True, False],
                              in_dead=[False, True]), [[1, 0]],
                      note="a not-alive snake's kept body still kills others"))
    E.append(run_edge("N_step_cap", N, 10, 1, 1,
                      N_state([[]], [[3, 3]], alive=[False], in_dead=[True], t=1998), [[0], [0]],
                      note="empty main snake: done False until current_step>=2000"))
    E.append(run_edge("N_four_snakes", N, 10, 4, 4,
                      N_state([[[1, 1]], [[3, 3]], [[5, 5]], [[7, 7]]],
                              [[2, 1], [0, 0], [9, 9], [9, 8]]), [[1, 2, 3, 4], [0, 0, 0, 0]],
                      note="four views = 12 channels"))
    # --- [A] (A11)
    E.append(run_edge("A_body_becomes_fruit", A, 10, 2, 2,
                      S_state([[[2, 2]], [[9, 3], [8, 3], [7, 3]]], [[0, 0], [5, 5]],
                              vels=[[0, 0], [1, 0]], spare=0),
                      [[0, 0], [0, 0], "reset", [0, 0]],
                      note="A11: wall death: body cells (incl. out-of-grid head) -> fruits, spare += len^2"))
    E.append(run_edge("A_eat_with_spare", A, 10, 2, 2,
                      S_state([[[8, 2]], []], [[0, 0], [5, 5], [10, 3], [9, 3], [8, 3]],
                              vels=[[0, 1], [0, 0]], spare=9),
                      [[0, 0], [1, 0], [0, 0]],
                      note="eating while spare>0 leaves the fruit in place (3 eats), then the snake "
                           "leaves the grid on a fruit cell and its body joins the fruit list"))
    E.append(run_edge("A_spare_runs_out", A, 10, 1, 1,
                      S_state([[[2, 2]]], [[3, 2], [4, 2], [5, 2]], vels=[[1, 0]], spare=1),
                      [[0], [0], [0]], seed=21, env_id=1,
                      note="spare 1 -> 0: first eat keeps the fruit, the next ones respawn"))
    with open(os.path.join(OUT, "edge_cases.json"), "w") as f:
        json.dump(E, f, separators=(",", ":"))
    print(f"edge_cases.json: {len(E)} cases, {os.path.getsize(os.path.join(OUT, 'edge_cases.json')) / 1024:.0f} KiB")


def gen_philox_kat():
    kat = {
        "random123_kat": [  # counter, key -> output (published Random123 known-answer vectors)
            {"ctr": [0, 0, 0, 0], "key": [0, 0]},
            {"ctr": [MASK] * 4, "key": [MASK] * 2},
            {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0]},
        ],
        "streams": [],
    }
    for k in kat["random123_kat"]:
        k["out"] = list(philox4x32_10(k["ctr"], k["key"]))
    for seed, env, off in [(0, 0, 0), (1234, 4095, 5), (0xDEADBEEFCAFEF00D, 32767, 0xFFFFFFFE)]:
        kat["streams"].append({"seed": seed, "env_id": env, "offset": off,
                               "u32": [philox_u32(seed, env, off + i) for i in range(16)]})
    with open(os.path.join(OUT, "philox_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("philox_kat.json written")


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_philox_kat()
    gen_edges()
    S, N, A = RULES_SNAKE_ENV, RULES_NEW_WORLD, RULES_ADVERSARIAL
    gen_tape("tape_S_10x10_1", S, 10, 1, 1, seed=0, num_envs=48, steps=256, action_width=2)
    gen_tape("tape_S_19x19_2", S, 19, 2, 2, seed=0, num_envs=48, steps=256, action_width=3)
    gen_tape("tape_S_19x19_3", S, 19, 3, 3, seed=0, num_envs=48, steps=256)
    gen_tape("tape_S_19x19_3_shard", S, 19, 3, 3, seed=99, num_envs=16, steps=64, env_id_base=4080)
    gen_tape("tape_N_10x10_1", N, 10, 1, 1, seed=0, num_envs=48, steps=256)
    gen_tape("tape_N_19x19_2", N, 19, 2, 4, seed=0, num_envs=48, steps=256)
    gen_tape("tape_N_19x19_3", N, 19, 3, 3, seed=0, num_envs=48, steps=256)
    gen_tape("tape_S_19x19_3_raw", S, 19, 3, 3, seed=5, num_envs=24, steps=128, auto_reset=False)
    gen_tape("tape_N_19x19_3_raw", N, 19, 3, 3, seed=5, num_envs=32, steps=256, auto_reset=False)
    gen_tape("tape_N_10x10_2_raw", N, 10, 2, 4, seed=6, num_envs=32, steps=256, auto_reset=False)
    gen_tape("tape_A_10x10_2", A, 10, 2, 2, seed=0, num_envs=32, steps=192)
    gen_tape("tape_A_10x10_3", A, 10, 3, 3, seed=3, num_envs=32, steps=192)
    # unusual shapes (the oracle must not be right only at the BASELINE sizes)
    gen_tape("tape_S_5x5_3", S, 5, 3, 3, seed=8, num_envs=12, steps=96)
    gen_tape("tape_S_3x3_2", S, 3, 2, 2, seed=9, num_envs=12, steps=96)
    gen_tape("tape_S_23x23_1", S, 23, 1, 1, seed=10, num_envs=8, steps=64, action_width=3)
    gen_tape("tape_N_6x6_4_f9", N, 6, 4, 9, seed=11, num_envs=12, steps=96)
    gen_tape("tape_N_13x13_2_f0", N, 13, 2, 0, seed=12, num_envs=8, steps=64)
    gen_tape("tape_A_6x6_3", A, 6, 3, 3, seed=13, num_envs=12, steps=96)
    gen_tape("tape_A_19x19_1", A, 19, 1, 1, seed=14, num_envs=8, steps=64)


if __name__ == "__main__":
    main()
