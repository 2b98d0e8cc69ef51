"""CPU: the partition / merge logic of the multi-GPU stream path (decompress_amd/csrc/multi.h), checked in C++ with
made-up per-device results (tests/cpp/test_multi.cpp); no GPU, no HIP."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_and_merge_in_cpp(tmp_path):
    exe = tmp_path / "test_multi"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "test_multi.cpp")],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "ok" in out
