"""-m gpu, OPT-IN: the stage-1 worker plug-in over kmc_hip_split_part inside the reference pipeline (oracle/_ref/kmc_hip_s1). The host glue behind
that entry point (kmc_hip.hip S1HipBackend, hip_split_loader.cpp) was written after round 2's GPU budget was spent and has never run on a GPU;
its launch sequence is the one tests/test_stage1_plugin.py proves under emulation. These tests therefore run only with KMC_TEST_UNVALIDATED=1,
so that an unproven path cannot stop the validated suite; the first GPU session of the next round switches them on for good."""
import hashlib
import os
import subprocess

import pytest

from kmc_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("KMC_TEST_UNVALIDATED") != "1", reason="opt-in: set KMC_TEST_UNVALIDATED=1 (never run on a GPU yet)")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _run(exe, flags, inp, tmp_path, tag, env=None):
    t = tmp_path / ("tmp_" + tag)
    t.mkdir(exist_ok=True)
    db = str(tmp_path / ("db_" + tag))
    e = dict(os.environ, KMC_HIP_LIB=os.path.join(ROOT, "kmc_amd", "libkmc_hip.so"), **(env or {}))
    r = subprocess.run([os.path.join(REF, exe), *flags, inp, db, str(t)], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, (exe, flags, (r.stdout + r.stderr)[-1500:])
    md5 = tuple(hashlib.md5(open(db + x, "rb").read()).hexdigest() for x in (".kmc_pre", ".kmc_suf"))
    stats = [ln.split(":")[1].strip() for ln in r.stdout.splitlines() if "No. of" in ln or "Total no." in ln]
    return md5, stats, r.stderr


@pytest.mark.parametrize("flags", [["-k27", "-ci1"], ["-k27", "-b"], ["-k55"], ["-k21", "-ci1"]], ids=lambda f: "".join(f))
def test_kmc_with_hip_stage1_and_stage2_writes_the_reference_database(flags, tmp_path):
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=31, genome_len=2_000_000, n_reads=300_000, read_len=150)
    want = _run("kmc", flags + ["-m4", "-sf1", "-sp1", "-sr1"], fq, tmp_path, "ref")
    got = _run("kmc_hip_s1", flags + ["-m4", "-sf2", "-sp4", "-sr4"], fq, tmp_path, "hip", env={"KMC_HIP_VERBOSE": "1"})
    assert got[:2] == want[:2]
    assert "parts through the engine" in got[2] and " 0 parts through the engine" not in got[2].split("[kmc_hip stage 1]")[1]
