"""In-tree builds (no pip, no JIT cache): libkmc_hip.so (hipcc, gfx950), libkmc_synth.so (g++), the oracle and —
when /root/reference is present — oracle/_ref (the real reference + the two plug-in builds)."""
from __future__ import annotations

import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kmc_amd")
CSRC = os.path.join(PKG, "csrc")
LIB_HIP = os.path.join(PKG, "libkmc_hip.so")
LIB_SYNTH = os.path.join(PKG, "libkmc_synth.so")


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "kmc_hip.h")]  # kmc_hip.hip + everything it includes
    if force or _newer(LIB_HIP, srcs):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
               os.path.join(CSRC, "kmc_hip.hip"), "-o", LIB_HIP, "-lrccl"]
        subprocess.check_call(cmd)
    return LIB_HIP


def build_synth(force: bool = False) -> str:
    src = os.path.join(CSRC, "synth_bins.cpp")
    if force or _newer(LIB_SYNTH, [src]):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", LIB_SYNTH])
    return LIB_SYNTH


def build_oracle(with_reference: bool = True) -> None:
    """Builds the checker (oracle/liboracle_stage2.so) and, when the reference tree is mounted, oracle/_ref."""
    od = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", od, "oracle"])
    if with_reference and os.path.exists("/root/reference/kmc_core/kmc_runner.cpp"):
        subprocess.check_call(["make", "-s", "-j8", "-C", od, "ref"])


BIN_DIR = os.path.join(PKG, "bin")  # the product drop-in binaries: kmc_hip, kmc_hip_s1, kmc_hip_sr, kmc_hipsort (kmc_amd/host/Makefile)


def build_dropin() -> None:
    """kmc_amd/bin/: the reference's own pipeline linked with the plug-ins of kmc_amd/host/ over libkmc_hip.so (needs the reference tree)."""
    if os.path.exists("/root/reference/kmc_core/kmc_runner.cpp"):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(PKG, "host")])


def build_all() -> None:
    build_hip()
    build_synth()
    build_oracle()
    build_dropin()
