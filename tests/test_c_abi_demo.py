"""examples/abi_demo.c — a plain C program (gcc, no Python, no HIP headers) against include/mmiss.h + libmmiss.so:
the drop-in boundary really is a C ABI. On the CPU box it must build, link and fail loudly at the first compute
entry point; on the MI355X it must run the whole load -> embed -> add -> query -> blend sequence and exit 0."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-image-similarity-search_amd")


def _build(tmp_path):
    exe = str(tmp_path / "abi_demo")
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "abi_demo.c"), "-L", PKG, "-lmmiss", f"-Wl,-rpath,{PKG}", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_consumer_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_consumer_runs_end_to_end(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok:")
