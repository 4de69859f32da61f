"""The oracle against the RUNNING reference, beyond the committed fixtures: configurations and action streams drawn at
random (fixed seeds), every observation byte, reward, done, num_snakes, episode return / length and the full state after
every step.  Runs only where /root/reference exists (the build container; the reference cannot travel to the GPU box):
there it re-pins oracle/snake_oracle.c to the reference itself on inputs no fixture holds.  The reference is loaded like
tools/gen_golden.py loads it (source text by path, gym stubbed in memory, the Philox shim injected as np_random); nothing
of it is copied or written."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "gym-snake")),
                                reason="the reference tree is not present on this machine")


@pytest.fixture(scope="module")
def gg():
    spec = importlib.util.spec_from_file_location("gen_golden_live", os.path.join(ROOT, "tools", "gen_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (loads the reference modules; writes nothing: only main() writes fixtures)
    return mod


def _configs(k, seed):
    rs = np.random.default_rng(seed)
    out = []
    for i in range(k):
        rules = i % 3
        ns = int(rs.integers(1, 5 if rules == 1 else 4))
        nf = int(rs.integers(0, 10)) if rules == 1 else ns
        out.append(dict(rules=rules, dim=int(rs.integers(3, 24)), n_snakes=ns, n_fruits=nf, seed=int(rs.integers(0, 2**31)),
                        env_id_base=int(rs.integers(0, 2**40)), num_envs=int(rs.integers(2, 7)), auto_reset=bool(rs.integers(0, 2)),
                        eps=float(rs.choice([0.1, 0.5, 1.0]))))
    return out


@pytest.mark.parametrize("cfg", _configs(150, 777), ids=lambda c: f"{['S', 'N', 'A'][c['rules']]}-{c['dim']}x{c['n_snakes']}x{c['n_fruits']}-"
                                                                    f"ar{int(c['auto_reset'])}-eps{c['eps']}")
def test_oracle_equals_the_running_reference(gg, cfg):
    from oracle.snake_oracle import Oracle
    rules, E, ns = cfg["rules"], cfg["num_envs"], cfg["n_snakes"]
    ref = gg.VecHarness(rules, cfg["dim"], ns, cfg["n_fruits"], cfg["seed"], E, env_id_base=cfg["env_id_base"],
                        auto_reset=cfg["auto_reset"])
    ora = Oracle(E, dim=cfg["dim"], n_snakes=ns, n_fruits=cfg["n_fruits"], rules=rules, seed=cfg["seed"],
                 env_id_base=cfg["env_id_base"], auto_reset=cfg["auto_reset"])
    assert np.array_equal(ref.reset(), ora.reset())
    rs = np.random.default_rng(cfg["seed"] % 9973)
    keys = ["snakes", "fruits", "vels", "grow_to", "t", "ctr"] + (["alive", "in_dead"] if rules == 1 else []) + (["spare_fruits"] if rules == 2 else [])
    for t in range(70):
        # eps-greedy walk towards the nearest fruit (eats, long bodies, collisions), with invalid action codes mixed in
        acts = np.array([gg.policy_actions(e, ns, rs, cfg["eps"]) for e in ref.envs], np.int32)
        if t % 11 == 5:
            acts[rs.integers(0, E), rs.integers(0, ns)] = int(rs.choice([-1, 5, 7]))
        r_obs, r_rew, r_done, r_ns, r_er, r_el = ref.step(acts)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(acts)
        assert np.array_equal(o_rew, r_rew) and np.array_equal(o_done, r_done), t
        assert np.array_equal(o_ns, r_ns) and np.array_equal(o_er, r_er) and np.array_equal(o_el, r_el), t
        assert np.array_equal(o_obs, r_obs), t
        for e in range(E):
            want = gg.canon_state(ref.envs[e])
            got = ora.get_state(e)
            assert {k: got[k] for k in keys} == {k: want[k] for k in keys}, (t, e)
        if not cfg["auto_reset"] and t % 25 == 24:  # a bare gym loop resets by hand
            assert np.array_equal(ref.reset(), ora.reset())
