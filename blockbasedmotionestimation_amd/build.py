"""Builds libbbme.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

    python -m blockbasedmotionestimation_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libbbme.so")
CLI = os.path.join(PKG, "bbme_cli")
RCCL_LIB = os.path.join(PKG, "libbbme_rccl.so")
SEQ = os.path.join(PKG, "bbme_seq")
SOURCES = ["bbme_host.cpp", "bbme_device.hip"]
HEADERS = ["bbme_internal.hpp", "bbme_kernels.hpp", "motion_framework.hpp", "rw_flow.hpp", "bbme_main.cpp", "seq_schedule.hpp",
           "bbme_rccl.cpp", "bbme_seq_main.cpp", os.path.join(ROOT, "include", "bbme.h"), os.path.join(ROOT, "include", "bbme_rccl.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def needs_build():
    if not all(os.path.exists(f) for f in (LIB, CLI, RCCL_LIB, SEQ)):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + \
           [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-x", "hip"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    # the C++ host side (MF / Flow classes + the reference's driver as a CLI) links only the C-ABI
    cli = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           os.path.join(CSRC, "bbme_main.cpp"), "-o", CLI + ".tmp", "-L", PKG, "-lbbme",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib"]
    if verbose:
        print(" ".join(cli))
    subprocess.check_call(cli)
    os.replace(CLI + ".tmp", CLI)
    # the multi-GPU sequence without torch: the gather over RCCL as a small C-ABI library on top of libbbme.so, and its driver
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    host = ["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
            "-I", os.path.join(rocm, "include")]
    link = ["-L", PKG, "-lbbme", "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath,$ORIGIN",
            "-Wl,-rpath," + os.path.join(rocm, "lib")]
    for cmd, dst in ((host + ["-fPIC", "-shared", os.path.join(CSRC, "bbme_rccl.cpp"), "-o", RCCL_LIB + ".tmp"] + link, RCCL_LIB),
                     (host + [os.path.join(CSRC, "bbme_seq_main.cpp"), "-o", SEQ + ".tmp", "-lbbme_rccl", "-lpthread"] + link, SEQ)):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(dst + ".tmp", dst)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
