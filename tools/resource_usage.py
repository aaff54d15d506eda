"""Register / scratch / occupancy table of every kernel of the shipped build: `hipcc -Rpass-analysis=kernel-resource-usage` over kmc_amd/csrc/kmc_hip.hip with the
flags of kmc_amd/build.py, parsed into one line per kernel. `python tools/resource_usage.py > profiles/r04/resource_usage.txt`; tests/test_resource_usage.py
holds the hot kernels to their scratch budget (a spill in k_onesweep<1> or k_compact<1> is HBM traffic the roofline pays for)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def short(name):
    m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def collect(extra_flags=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *extra_flags,
               os.path.join(ROOT, "kmc_amd", "csrc", "kmc_hip.hip"), "-o", os.path.join(td, "ru.so"), "-lrccl"]
        err = subprocess.run(cmd, capture_output=True, text=True, cwd=td).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "Function Name":
            cur = {"mangled": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = int(val)
    names = demangle([r["mangled"] for r in rows])
    for r in rows:
        r["name"] = short(names.get(r["mangled"], r["mangled"]))
    return rows


def main():
    rows = collect(sys.argv[1:])
    print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage kmc_amd/csrc/kmc_hip.hip %s" % " ".join(sys.argv[1:]))
    print("%-44s %6s %6s %10s %10s %12s %10s" % ("kernel", "VGPRs", "SGPRs", "VGPR spill", "SGPR spill", "scratch B/ln", "waves/SIMD"))
    for r in sorted(rows, key=lambda r: r["name"]):
        print("%-44s %6d %6d %10d %10d %12d %10d" % (r["name"], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                                                      r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
