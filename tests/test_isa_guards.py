"""Guards on the generated gfx950 ISA for two things the compiler is free to break silently (both happened once):
* the scalar loads of the early Philox are issued by one asm statement and waited for by another -- nothing may touch their
  destination SGPRs in between (a register copy there would copy registers the loads have not written yet);
* the streaming copy-out must really emit `nt` stores (written as `if (nt) nt-store else store` the two were merged into a
  plain store: the nontemporal hint is droppable metadata).
CPU only: hipcc cross-compiles to assembly, nothing runs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "self-play-on-multi-snakes-environment_amd", "csrc", "msnake_kernels.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "k.s"
    r = subprocess.run([HIPCC, "-Os", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S",
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(out), SRC], check=True,
                       capture_output=True, text=True)
    text = out.read_text()
    funcs = re.split(r"\n(?=_ZN6msnake18msnake_step_kernel\w+:)", text)[1:]
    # register use per kernel from the resource remarks: {mangled name: {"VGPRs": n, "SGPRs Spill": n, ...}}
    res, cur = {}, None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = res.setdefault(v, {})
        else:
            cur[k] = int(v)
    funcs_and_resources.resources = res
    return funcs


class funcs_and_resources:  # (the resource table of the module-scoped compile above)
    resources = {}


def _regs(tok):
    if tok.startswith("s["):
        a, b = map(int, re.findall(r"\d+", tok))
        return range(a, b + 1)
    return range(int(tok[1:]), int(tok[1:]) + 1)


def test_early_scalar_loads_are_not_touched_before_their_wait(isa):
    checked = 0
    for f in isa:
        lines = f.split("\n")
        for i, l in enumerate(lines):
            m = re.match(r"\s*s_load_dwordx2 s\[(\d+):(\d+)\], s\[\d+:\d+\], (52|0x34|0x8c|140)\b", l)
            if not m:
                continue
            checked += 1
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for t in lines[i + 1:]:
                if "s_waitcnt" in t and "lgkmcnt(0)" in t:
                    break
                if not t.strip() or t.strip().startswith((";", ".")):
                    continue
                touched = {r for tok in re.findall(r"s\[\d+:\d+\]|\bs\d+\b", t) for r in _regs(tok)} & dst
                assert not touched, (lines[0][:70], t.strip())
    assert checked >= 2  # the headline instantiation at least: the draw counter and the state of the parked draws


def test_streaming_copy_out_emits_nt_stores(isa):
    head = [f for f in isa if f.startswith("_ZN6msnake18msnake_step_kernelILi0ELi3ELi0ELi1E")][0]
    nt = len(re.findall(r"global_store_dwordx4 .* nt\b", head))
    plain = len([l for l in re.findall(r"global_store_dwordx4 [^\n]*", head) if not l.rstrip().endswith("nt")])
    assert nt >= 4 and plain >= 4, (nt, plain)  # both store kinds of the aligned copy-out, four 1 KiB instructions each


def test_snake_env_kernels_do_not_spill(isa):
    """The headline instantiation stays within 48 VGPRs (the allocation granule it was tuned for) and no snake_env kernel -- per-step,
    reset, render or persistent tape -- spills a register or uses scratch (msnake_step_kernel<RULES = 0, NS, MODE, K>)."""
    res = funcs_and_resources.resources
    seen = 0
    for name, r in res.items():
        m = re.match(r"_ZN6msnake18msnake_step_kernelILi0ELi(\d)ELi([0123])ELi(\d)E", name)
        if not m:
            continue
        seen += 1
        assert r["SGPRs Spill"] == 0 and r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, (name[:60], r)
        if m.groups() == ("3", "0", "1"):
            assert r["VGPRs"] <= 48, r
    assert seen >= 36
