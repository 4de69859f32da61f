"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the host
mirror of the VecEnv interface behaves like the reference's, and the product refuses to run
without its HIP extension / a GPU instead of silently falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

import msnake

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "msnake.h")).read()
    declared = set(re.findall(r"\b(msnake_[a-z_]+)\s*\(", header))
    declared -= {"msnake_create_fn"}
    assert declared == set(msnake._capi.SYMBOLS), declared ^ set(msnake._capi.SYMBOLS)
    lib = msnake._capi.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.msnake_abi_version() == msnake._capi.ABI_VERSION == 3


def test_struct_layouts_match_header():
    assert ctypes.sizeof(msnake._capi.MsnakeConfig) == 72  # ABI 3 = the 56-byte ABI-2 struct + four tuning fields
    assert msnake._capi.MsnakeConfig.envs_per_block.offset == msnake._capi.CONFIG_SIZE_V2 == 56
    header = open(os.path.join(ROOT, "include", "msnake.h")).read()
    assert "#define MSNAKE_CONFIG_SIZE_V2 56u" in header and "#define MSNAKE_ABI_VERSION 3" in header
    assert ctypes.sizeof(msnake._capi.MsnakeStats) == 64


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        msnake.MultiSnakeVecEnv(4)
    lib = msnake._capi.load()
    cfg = msnake._capi.MsnakeConfig(ctypes.sizeof(msnake._capi.MsnakeConfig), 0, 4, 19, 3, 3, 0, 2000, 1, 1, 0, 0)
    h = ctypes.c_void_p()
    rc = lib.msnake_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -6 and not h.value
    with pytest.raises(RuntimeError, match="no CPU path"):
        msnake._capi.check(rc, "msnake_create")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "self-play-on-multi-snakes-environment_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "snake_oracle" not in src, f


def test_create_argument_checks_before_touching_the_gpu():
    lib = msnake._capi.load()
    h = ctypes.c_void_p()
    bad = msnake._capi.MsnakeConfig(12, 0, 4, 19, 3, 3, 0, 2000, 1, 1, 0, 0)  # wrong struct_size
    assert lib.msnake_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert b"struct_size" in lib.msnake_last_error()
    sz = ctypes.sizeof(msnake._capi.MsnakeConfig)
    for kw, needle in [(dict(num_envs=0), b"num_envs"), (dict(dim=1), b"dim"), (dict(dim=63), b"dim"),
                       (dict(n_snakes=4), b"n_snakes"), (dict(n_fruits=2), b"n_fruits"), (dict(rules=7), b"rules"), (dict(rules=2, n_snakes=4, n_fruits=4), b"n_snakes"),
                       (dict(max_steps=0), b"max_steps"), (dict(obs_scale=3), b"obs_scale"),
                       (dict(envs_per_block=9), b"envs_per_block"), (dict(envs_per_block=-1), b"envs_per_block"),
                       (dict(record_policy=3), b"record_policy"), (dict(obs_store_policy=5), b"store_policy"),
                       (dict(tape_store_policy=-2), b"store_policy")]:
        f = dict(struct_size=sz, device=0, num_envs=4, dim=19, n_snakes=3, n_fruits=3, rules=0, max_steps=2000,
                 auto_reset=1, obs_scale=1, seed=0, env_id_base=0)
        f.update(kw)
        cfg = msnake._capi.MsnakeConfig(**f)
        assert lib.msnake_create(ctypes.byref(cfg), ctypes.byref(h)) == -1, kw
        assert needle in lib.msnake_last_error(), (kw, lib.msnake_last_error())
    assert lib.msnake_step(None, None, 3, None, None, None, None, None) == -3  # NULL handle
    # an ABI-2 caller (56-byte struct, no tuning fields) is still understood: garbage behind the prefix is not read
    old = msnake._capi.MsnakeConfig(msnake._capi.CONFIG_SIZE_V2, 0, 0, 19, 3, 3, 0, 2000, 1, 1, 0, 0, 99, 99, 99, 99)
    assert lib.msnake_create(ctypes.byref(old), ctypes.byref(h)) == -1 and b"num_envs" in lib.msnake_last_error()


def test_shipped_library_reads_no_environment_knobs():
    """Launch tuning is part of msnake_config (ABI 3); getenv exists only in the -DMSNAKE_DBG_STAGES block."""
    src = open(os.path.join(ROOT, "self-play-on-multi-snakes-environment_amd", "csrc", "msnake_capi.hip")).read()
    outside = re.sub(r"#if(def MSNAKE_DBG_STAGES|\s+defined\(MSNAKE_DBG_STAGES\) \|\| defined\(MSNAKE_SPAN_LIGHT\)).*?#endif", "", src, flags=re.S)
    assert "getenv" not in outside
    assert "getenv" not in open(os.path.join(ROOT, "self-play-on-multi-snakes-environment_amd", "csrc", "msnake_kernels.hip")).read()


def test_normalize_actions_matches_reference_call_shapes():
    # ppo_multi_agent.py:41-44: list(zip(actions, opp1[, opp2])) of length 2 or 3, even with 1 snake
    acts = list(zip([1, 2, 3, 4], [0, 0, 1, 1]))
    a = msnake.normalize_actions(acts, 4, 1)
    assert a.dtype == np.int32 and a.shape == (4, 2) and a.flags.c_contiguous
    assert msnake.normalize_actions(np.array([1, 2, 3]), 3, 1).shape == (3, 1)  # bare scalar per env
    with pytest.raises(ValueError):
        msnake.normalize_actions(acts, 4, 3)
    with pytest.raises(ValueError):
        msnake.normalize_actions(acts, 5, 2)


def test_lazy_infos_have_the_reference_keys():
    done = np.array([False, True, False])
    infos = msnake.LazyInfos(done, np.array([3, 1, 2]), np.array([0.0, 4.0, 0.0], np.float32), np.array([0, 17, 0]), 1.5)
    assert len(infos) == 3
    assert infos[0] == {"ale.lives": 1, "num_snakes": 3}
    assert infos[1] == {"ale.lives": 1, "num_snakes": 1, "episode": {"r": 4.0, "l": 17, "t": 1.5}}
    assert [i.get("episode") for i in infos] == [None, {"r": 4.0, "l": 17, "t": 1.5}, None]
    assert infos.episodes() == [{"r": 4.0, "l": 17, "t": 1.5}]
    assert infos[-1]["num_snakes"] == 2
    with pytest.raises(IndexError):
        infos[3]


def test_spaces_and_gym_id_presets():
    assert msnake.Discrete(5).n == 5
    b = msnake.Box(0, 255, (21, 21, 6), np.uint8)
    assert b.shape == (21, 21, 6) and b.dtype == np.uint8
    assert set(msnake.GYM_IDS) == {"snake-multiple-test-v0", "snake-new-multiple-v0", "snake-adversarial-v0"}
    assert msnake.GYM_IDS["snake-multiple-test-v0"]["rules"] == "snake_env"


def test_shard_range_partitions_exactly():
    for total in (1, 7, 4096, 32768, 1000):
        for world in (1, 2, 3, 8):
            parts = [msnake.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        msnake.shard_range(8, 2, 2)


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """build() loads libmsnake.so before smoke() imports torch: the library must still bind to the
    HIP runtime torch brings (same SONAME), not pull /opt/rocm's copy in as a second runtime -- with
    two runtimes in one process the second one finds no device (seen on the GPU box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import msnake; msnake._capi.load()\n"
            "import torch\n"
            "paths = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
            "print(len(paths), sorted(paths))\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout


def test_make_sharded_defaults_to_the_local_rank_device(monkeypatch):
    """One process per GPU under torchrun: rank r's shard must land on cuda:LOCAL_RANK without the caller
    having called torch.cuda.set_device, and keep global env ids."""
    seen = {}

    class FakeEnv:
        def __init__(self, num_envs, **kw):
            seen.update(num_envs=num_envs, **kw)

    monkeypatch.setattr(msnake.vec_env, "MultiSnakeVecEnv", FakeEnv)
    monkeypatch.setenv("RANK", "5")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    msnake.make_sharded(32768, dim=19, n_snakes=3)
    assert seen == dict(num_envs=4096, env_id_base=5 * 4096, device="cuda:5", dim=19, n_snakes=3)
    msnake.make_sharded(10, rank=1, world=3, device="cuda:0")
    assert seen["num_envs"] == 3 and seen["env_id_base"] == 4 and seen["device"] == "cuda:0"


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment starts the N ranks itself, before
    importing torch, and hands the child's exit code back."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
