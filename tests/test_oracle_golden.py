"""The CPU oracle (oracle/snake_oracle.c) against fixtures recorded from the running reference.

This is what pins the oracle: every edge case and every step of every tape must agree bit for
bit (state, reward, done, num_snakes, episode stats, observation bytes / CRC32).
"""
import json
import os

import numpy as np
import pytest

from golden_util import (GOLDEN, blob_to_obs, crc_rows, edge_cases, load_tape, state_view, tape_names,
                         unpack_state)
from oracle import snake_oracle as so


def test_philox_kat():
    kat = json.load(open(os.path.join(GOLDEN, "philox_kat.json")))
    # published Random123 known answers for philox4x32-10 (kat_vectors)
    published = [[0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8],
                 [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd],
                 [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]]
    for k, pub in zip(kat["random123_kat"], published):
        assert k["out"] == pub
        assert so.philox_block(k["ctr"], k["key"]) == pub
    for s in kat["streams"]:
        got = [so.philox_u32(s["seed"], s["env_id"], s["offset"] + i) for i in range(16)]
        assert got == s["u32"]


@pytest.mark.parametrize("case", edge_cases(), ids=lambda c: c["name"])
def test_edge_case(case):
    rules = case["rules"]
    o = so.Oracle(1, dim=case["dim"], n_snakes=case["n_snakes"], n_fruits=case["n_fruits"], rules=rules,
                  seed=case["seed"], env_id_base=case["env_id"], auto_reset=False)
    o.reset()
    o.set_state(0, case["state0"])
    assert state_view(o.get_state(0), rules) == state_view(case["state0"], rules)
    assert np.array_equal(o.render()[0], blob_to_obs(case["obs0"]))
    for i, step in enumerate(case["steps"]):
        if step["actions"] == "reset":
            keep = o.get_state(0)
            obs = o.reset()
            if rules == 2:  # [A]: spare_fruits survives reset
                assert o.get_state(0)["spare_fruits"] == keep["spare_fruits"]
        else:
            obs, rew, done, ns, _, _ = o.step(np.array([step["actions"]], np.int32))
            assert float(rew[0]) == step["reward"], (case["name"], i)
            assert bool(done[0]) == step["done"], (case["name"], i)
            assert int(ns[0]) == step["num_snakes"], (case["name"], i)
        assert state_view(o.get_state(0), rules) == state_view(step["state"], rules), (case["name"], i)
        assert np.array_equal(obs[0], blob_to_obs(step["obs"])), (case["name"], i)


@pytest.mark.parametrize("name", tape_names())
def test_tape(name):
    meta, z = load_tape(name)
    rules, E, T = meta["rules"], meta["num_envs"], meta["steps"]
    o = so.Oracle(E, dim=meta["dim"], n_snakes=meta["n_snakes"], n_fruits=meta["n_fruits"], rules=rules,
                  seed=meta["seed"], env_id_base=meta["env_id_base"], max_steps=meta["max_steps"],
                  auto_reset=meta["auto_reset"])
    obs = o.reset()
    assert np.array_equal(obs, z["obs0"])
    for e in range(E):
        assert state_view(o.get_state(e), rules) == unpack_state(z, "s0_", 0, e, rules)
    full_t = {int(t): i for i, t in enumerate(z["full_obs_t"])}
    actions = z["actions"].astype(np.int32)
    for t in range(T):
        obs, rew, done, ns, epr, epl = o.step(actions[t])
        assert np.array_equal(rew, z["reward"][t]), (name, t)
        assert np.array_equal(done, z["done"][t]), (name, t)
        assert np.array_equal(ns, z["num_snakes"][t].astype(np.int32)), (name, t)
        assert np.array_equal(epr, z["ep_return"][t]), (name, t)
        assert np.array_equal(epl, z["ep_len"][t]), (name, t)
        assert np.array_equal(crc_rows(obs), z["obs_crc"][t]), (name, t)
        if t in full_t:
            assert np.array_equal(obs, z["full_obs"][full_t[t]]), (name, t)
        if t % 8 == 0 or t == T - 1:
            for e in range(E):
                assert state_view(o.get_state(e), rules) == unpack_state(z, "st_", t, e, rules), (name, t, e)


def test_multithreaded_step_matches_single():
    rs = np.random.default_rng(0)
    a = so.Oracle(64, dim=19, n_snakes=3, seed=3)
    b = so.Oracle(64, dim=19, n_snakes=3, seed=3)
    assert np.array_equal(a.reset(), b.reset())
    for _ in range(50):
        act = rs.integers(0, 5, (64, 3)).astype(np.int32)
        ra = [x.copy() for x in a.step(act)]
        rb = [x.copy() for x in b.step(act, threads=4)]
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)


def test_philox_streams_equal_rocrand_engine(tmp_path):
    """the (seed, subsequence = global env id, offset = draw index) convention of
    include/msnake.h IS rocRAND's: the first 16 outputs of rocrand_device::philox4x32_10_engine,
    compiled for the host from the installed rocRAND header (oracle/rocrand_kat.cpp), equal the
    committed streams of tests/golden/philox_kat.json -- and so, by test_philox_kat above, the
    oracle's Philox and, by the GPU parity tests, the kernel's."""
    import shutil
    import subprocess
    import pytest
    header = "/opt/rocm/include/rocrand/rocrand_philox4x32_10.h"
    if not os.path.exists(header) or not shutil.which("g++"):
        pytest.skip("rocRAND header or g++ not present")
    exe = str(tmp_path / "rocrand_kat")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "rocrand_kat.cpp")
    subprocess.check_call(["g++", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe, src])
    streams = json.load(open(os.path.join(GOLDEN, "philox_kat.json")))["streams"]
    assert len(streams) >= 3
    args = [str(x) for s in streams for x in (s["seed"], s["env_id"], s["offset"])]
    out = subprocess.run([exe] + args, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(streams)
    for line, s in zip(out, streams):
        assert [int(x) for x in line.split()] == s["u32"], s
