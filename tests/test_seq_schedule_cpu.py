"""The ordering logic of the multi-GPU sequence driver (csrc/seq_schedule.hpp, run by bbme_seq over HIP / RCCL) on the CPU:
tests/cpp/seq_schedule_test.cpp runs the same template over a mock backend whose streams, copy stream and writer thread are
queues executed in random interleavings, checks every file against its pair, and shows that a missing wait is caught."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sequence_pipeline_ordering_under_random_interleavings(tmp_path):
    exe = str(tmp_path / "seq_schedule_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "blockbasedmotionestimation_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "seq_schedule_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "seq_schedule ok" in r.stdout
