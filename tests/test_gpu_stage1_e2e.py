"""-m gpu: the stage-1 worker plug-in over kmc_hip_split_part inside the reference pipeline (kmc_amd/bin/kmc_hip_s1: stage 1 AND stage 2 on the
GPU). The kernels and the launch sequence also run on the CPU under emulation inside the reference pipeline (tests/test_stage1_plugin.py), the
product binary over the product's host library compiled against an emulated HIP runtime (tests/test_hostlib_emulated.py). First met a real GPU in
the driver's round-2 run (GPUTEST_r02: 4 xpassed); ordinary tests since round 3. Last file of the -m gpu collection."""
import hashlib
import re
import os
import subprocess

import pytest

from kmc_amd import synth

pytestmark = [pytest.mark.gpu]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _exe(name):
    """oracle/_ref/<name> for the reference and the oracle builds (checkers), kmc_amd/bin/<name> for the product drop-in binaries"""
    return os.path.join(ROOT, "kmc_amd", "bin", name) if name.startswith("kmc_hip") else os.path.join(REF, name)


_state = {"broken": False}  # one failed or hung run is enough: the other parameter sets do not spend GPU time on the same problem


def _run(exe, flags, inp, tmp_path, tag, env=None):
    t = tmp_path / ("tmp_" + tag)
    t.mkdir(exist_ok=True)
    db = str(tmp_path / ("db_" + tag))
    e = dict(os.environ, KMC_HIP_LIB=os.path.join(ROOT, "kmc_amd", "libkmc_hip.so"), **(env or {}))
    try:
        r = subprocess.run([_exe(exe), *flags, inp, db, str(t)], capture_output=True, text=True, env=e, timeout=120)
    except subprocess.TimeoutExpired:
        _state["broken"] = True
        raise
    if r.returncode != 0:
        _state["broken"] = True
    assert r.returncode == 0, (exe, flags, (r.stdout + r.stderr)[-1500:])
    md5 = tuple(hashlib.md5(open(db + x, "rb").read()).hexdigest() for x in (".kmc_pre", ".kmc_suf"))
    stats = [ln.split(":")[1].strip() for ln in r.stdout.splitlines() if "No. of" in ln or "Total no." in ln]
    return md5, stats, r.stderr


@pytest.mark.parametrize("flags", [["-k27", "-ci1"], ["-k27", "-b"], ["-k55"], ["-k21", "-ci1"]], ids=lambda f: "".join(f))
def test_kmc_with_hip_stage1_and_stage2_writes_the_reference_database(flags, tmp_path):
    if _state["broken"]:
        pytest.fail("an earlier run of kmc_hip_s1 failed or hung")
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=31, genome_len=2_000_000, n_reads=300_000, read_len=150)
    want = _run("kmc", flags + ["-m4", "-sf1", "-sp1", "-sr1"], fq, tmp_path, "ref")
    got = _run("kmc_hip_s1", flags + ["-m4", "-sf2", "-sp4", "-sr4"], fq, tmp_path, "hip", env={"KMC_HIP_VERBOSE": "1"})
    assert got[:2] == want[:2]
    # every worker reports; over ALL of them: parts went through the HIP engine. The binary has no path into the reference splitter (kb_splitter_plugin.h fails
    # closed; the hand-over code is compiled out of every shipped binary): an uncovered part would have stopped the run above
    rep = re.findall(r"\[kmc_hip stage 1\] worker: (\d+) parts through the engine .*?, (\d+) uncovered parts", got[2])
    assert len(rep) >= 1, got[2][-2000:]
    assert sum(int(a) for a, _ in rep) > 0 and sum(int(c) for _, c in rep) == 0, rep
    sym = subprocess.run(["nm", "-C", _exe("kmc_hip_s1")], capture_output=True, text=True).stdout
    # the hand-over is not in the binary. (CWSplitter_ref itself — the reference's worker under its renamed class — is still linked: it lives in the reference's
    # splitter.o next to CSplitter, which stage 0 and the small-k path use; nothing in this worker refers to it.)
    assert "to_reference" not in sym and "CWSplitter::operator()" in sym


def test_long_read_parts_and_long_lines_on_the_device(tmp_path):
    """a FASTA with single-line records of 0.6, 3 and 40 Mbp between short ones: lines beyond mem_part_pmm_reads inside ordinary parts (S1_PIECE_MARK) and the
    reader's ReadType::long_read parts (the 40 Mbp record exceeds the reader's 32 MB buffer), through kmc_hip_split_part on the GPU: database and statistics
    (reads, super-k-mers) equal the reference's"""
    if _state["broken"]:
        pytest.fail("an earlier run of kmc_hip_s1 failed or hung")
    import numpy as np

    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    fa = str(tmp_path / "in.fa")
    with open(fa, "wb") as f:
        for i, n in enumerate([200, 600_000, 150, 3_000_000, 90, 40_000_000, 530_000, 100]):
            f.write(b">r%d\n" % i + acgt[rng.integers(0, 4, size=n)].tobytes() + b"\n")
    flags = ["-k27", "-ci1", "-fa"]
    want = _run("kmc", flags + ["-m8", "-sf1", "-sp1", "-sr1"], fa, tmp_path, "ref")
    got = _run("kmc_hip_s1", flags + ["-m8", "-sf1", "-sp2", "-sr2"], fa, tmp_path, "hip", env={"KMC_HIP_VERBOSE": "1"})
    assert got[:2] == want[:2]
    rep = re.findall(r"\((?:[0-9.]+) s inside; (\d+) of them long-read parts\)", got[2])
    assert sum(int(x) for x in rep) >= 2, got[2][-2000:]
