"""AddressSanitizer + UBSan over the CPU oracle (the checker must itself be memory-safe)."""
import os
import shutil
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "orc_asan")
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                            os.path.join(ORACLE, "asan_check.c")], capture_output=True, text=True, cwd=ORACLE)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ASAN/UBSAN run clean" in run.stdout, run.stdout + run.stderr
