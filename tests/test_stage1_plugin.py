"""CPU: the stage-1 splitter worker plug-in (kmc_amd/host/kb_splitter_plugin.h — the drop-in boundary of a GPU stage 1)
inside the REAL reference pipeline, with the stage-1 oracle as its per-part engine (oracle/_ref/kmc_oracle_s1, oracle/Makefile). The database
must be byte-identical to the unmodified reference's: that pins the worker's protocol towards the storer and the bin descriptors, the oracle's
text parser, and the k+x-mer sums the reference's stage 2 sizes its arrays with. kmc_oracle_all = every plug-in of this repo at once
(splitter + stage-2 worker + bin reader + completer), both engines the oracles. Skipped where oracle/_ref is not built."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from kmc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _exe(name):
    """oracle/_ref/<name> for the reference and the oracle builds (checkers), kmc_amd/bin/<name> for the product drop-in binaries"""
    return os.path.join(ROOT, "kmc_amd", "bin", name) if name.startswith("kmc_hip") else os.path.join(REF, name)
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "kmc_oracle_s1")), reason="oracle/_ref not built (needs /root/reference)")


def _run(exe, flags, inp, tmp_path, tag, env=None):
    t = tmp_path / ("tmp_" + tag)
    t.mkdir(exist_ok=True)
    db = str(tmp_path / ("db_" + tag))
    r = subprocess.run([_exe(exe), *flags, inp, db, str(t)], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (exe, flags, (r.stdout + r.stderr)[-800:])
    md5 = tuple(hashlib.md5(open(db + e, "rb").read()).hexdigest() for e in (".kmc_pre", ".kmc_suf"))
    stats = [ln.split(":")[1].strip() for ln in r.stdout.splitlines() if "No. of" in ln or "Total no." in ln]
    _run.report = [ln for ln in r.stderr.splitlines() if ln.startswith("[kmc_hip stage 1]")]
    return md5, stats


def _report_sum(what):
    """sum over the workers' KMC_HIP_VERBOSE lines of the number in front of `what`"""
    import re

    return sum(int(m.group(1)) for ln in _run.report for m in [re.search(r"(\d+) " + re.escape(what), ln)] if m)


def _rnd(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()


@needs_ref
@pytest.mark.parametrize("exe", ["kmc_oracle_s1", "kmc_oracle_all"])
@pytest.mark.parametrize("flags", [["-k27", "-ci1", "-sp4"], ["-k21", "-sp1"], ["-k55", "-ci2", "-sp8"], ["-k27", "-b", "-ci1", "-sp3"], ["-k32", "-sp2"],
                                   ["-k127", "-ci1", "-sp2"], ["-k27", "-p5", "-n64", "-sp2"]], ids=lambda f: "".join(f))
def test_splitter_plugin_writes_the_reference_database(exe, flags, tmp_path):
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=11, genome_len=300_000, n_reads=40_000, read_len=150)
    common = ["-m2", "-sf1", "-sr1"]
    want = _run("kmc", [f for f in flags if not f.startswith("-sp")] + common + ["-sp1"], fq, tmp_path, "ref")
    got = _run(exe, flags + common, fq, tmp_path, "plug")
    assert got == want


def _edge_case_records(k):
    rng = np.random.default_rng(5)
    per = _rnd(rng, 11)
    return [_rnd(rng, 150) + b"N" + _rnd(rng, 80) + b"NN" + _rnd(rng, k - 1) + b"n" + _rnd(rng, 200), (per * 80)[:700], _rnd(rng, k), _rnd(rng, k - 1), b"", b"N" * 40,
            _rnd(rng, 3000), b"A" * 300, b"AC" * 200, _rnd(rng, 120).lower(), b"ACGTRYKM" * 20, _rnd(rng, 1)]


@needs_ref
@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("fmt", ["fq", "fa"])
def test_splitter_plugin_on_text_edge_cases(fmt, eol, tmp_path):
    """empty reads, reads shorter than k, N runs, lower case, IUPAC codes, CRLF line ends, FASTA — every way the text parser can be wrong"""
    k = 27
    rng = np.random.default_rng(3)
    genome = _rnd(rng, 50_000)
    recs = _edge_case_records(k) + [genome[a:a + 100] for a in rng.integers(0, len(genome) - 100, size=3000)]
    path = str(tmp_path / ("in." + fmt))
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            if fmt == "fq":
                f.write(b"@r%d some text" % i + eol + r + eol + b"+" + eol + b"I" * len(r) + eol)
            else:
                f.write(b">r%d some text" % i + eol + r + eol)
    flags = ["-k%d" % k, "-ci1", "-m2", "-sf1", "-sr1"] + (["-fa"] if fmt == "fa" else [])
    want = _run("kmc", flags + ["-sp1"], path, tmp_path, "ref")
    got = _run("kmc_oracle_s1", flags + ["-sp3"], path, tmp_path, "plug")
    assert got == want


@needs_ref
def test_splitter_plugin_when_one_bin_takes_whole_parts(tmp_path):
    """reads without any allowed m-mer all carry the special signature: one bin receives far more than a 64 KB buffer from every part, so the
    worker cuts the piece record by record and counts the k+x-mer records itself (kmc_record_plus_x) — the reference's stage 2 then sizes and
    fills its arrays with those sums"""
    rng = np.random.default_rng(8)
    path = str(tmp_path / "in.fq")
    with open(path, "wb") as f:
        for i in range(30_000):
            r = [b"A" * 150, b"AC" * 75, _rnd(rng, 150), b"T" * 149 + b"G"][i % 4]
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    for flags in (["-k27", "-ci1"], ["-k27", "-b", "-ci1"], ["-k40", "-ci1"]):
        common = flags + ["-m2", "-sf1", "-sr1"]
        want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
        assert _run("kmc_oracle_s1", common + ["-sp2"], path, tmp_path, "plug", env={"KMC_HIP_VERBOSE": "1"}) == want, flags
        assert _report_sum("cut record by record") > 0


@needs_ref
def test_splitter_plugin_takes_long_read_parts_through_its_engine(tmp_path):
    """a record larger than a part of the reader arrives as ReadType::long_read pieces (queues.h:40): since round 5 they go through the engine like every other
    part (here the oracle's restatement of CSplitter::GetSeqLongRead): database and statistics — the super-k-mer count among them: the pieces of
    mem_part_pmm_reads symbols are cut where the reference cuts them — equal the unmodified reference's; short reads before and after likewise"""
    rng = np.random.default_rng(9)
    path = str(tmp_path / "in.fq")
    with open(path, "wb") as f:
        for i in range(2000):
            r = _rnd(rng, 150)
            f.write(b"@s%d\n" % i + r + b"\n+\n" + b"I" * 150 + b"\n")
        big = _rnd(rng, 40_000_000)
        f.write(b"@long\n" + big + b"\n+\n" + b"I" * len(big) + b"\n")
        for i in range(2000):
            r = _rnd(rng, 150)
            f.write(b"@t%d\n" % i + r + b"\n+\n" + b"I" * 150 + b"\n")
    common = ["-k27", "-ci1", "-m2", "-sf1", "-sr1"]
    want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
    assert _run("kmc_oracle_s1", common + ["-sp2"], path, tmp_path, "plug", env={"KMC_HIP_VERBOSE": "1"}) == want
    assert _report_sum("of them long-read parts") > 0 and _report_sum("parts through the engine") > 0 and _report_sum("uncovered parts") == 0


@needs_ref
def test_splitter_plugin_refuses_jobs_it_does_not_cover(tmp_path):
    """-hc (homopolymer compression) is not what the engine computes: the worker stops the run and names the reason (it used to run the reference's worker
    instead: a silent path into the reference). -e alone is the reference's estimate-only worker, not this class: unchanged."""
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=4, genome_len=50_000, n_reads=5_000, read_len=150)
    t = tmp_path / "tmp_hc"
    t.mkdir()
    r = subprocess.run([_exe("kmc_oracle_s1"), "-k27", "-hc", "-ci1", "-m2", "-sf1", "-sr1", "-sp2", fq, str(tmp_path / "db_hc"), str(t)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not cover homopolymer compression" in r.stdout + r.stderr, (r.stdout + r.stderr)[-800:]
    t2 = tmp_path / "tmp_e"
    t2.mkdir()
    r = subprocess.run([_exe("kmc_oracle_s1"), "-k27", "-e", "-m2", "-sf1", "-sp2", fq, str(tmp_path / "db_e"), str(t2)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-800:]  # estimation only: CWEstimateOnlySplitter, the reference's own class


@needs_ref
@pytest.mark.parametrize("skewed", [False, True], ids=["reads", "one-bin-takes-parts"])
@pytest.mark.parametrize("flags", [["-k27", "-ci1"], ["-k27", "-b"], ["-k21"], ["-k55"], ["-k28", "-ci1"]], ids=lambda f: "".join(f))
def test_bin_descriptors_of_the_plugin_equal_the_reference_collectors(flags, skewed, tmp_path):
    """per bin, what stage 1 leaves in CBinDesc — bytes, k-mers and n_plus_x_recs (the (k+x)-mer records stage 2 will expand the bin into) —
    from the reference's collectors (kmc_oracle: reference stage 1) and from the plug-in over the oracle engine (kmc_oracle_all): the direct
    pin of oracle_s1_kxmer_recs and of the worker's sums (canonical and -b counting take different branches, kb_collector.cpp:85-98). The skewed
    input sends most of every part to one bin: there the worker cuts pieces record by record and counts with kmc_record_plus_x."""
    fq = str(tmp_path / "in.fq")
    if skewed:
        rng = np.random.default_rng(8)
        with open(fq, "wb") as f:
            for i in range(20_000):
                r = [b"A" * 150, b"AC" * 75, _rnd(rng, 150), b"T" * 149 + b"G", b"ACGT" * 37][i % 5]
                f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    else:
        synth.make_fastq(fq, seed=21, genome_len=200_000, n_reads=20_000, read_len=150)
    out = {}
    for exe in ("kmc_oracle", "kmc_oracle_all"):
        dump = str(tmp_path / (exe + ".desc"))
        _run(exe, flags + ["-m2", "-sf1", "-sp2", "-sr1"], fq, tmp_path, exe, env={"KMC_HIP_BINDESC_DUMP": dump, "KMC_HIP_VERBOSE": "1"})
        rows = sorted(tuple(int(x) for x in ln.split()) for ln in open(dump))
        out[exe] = rows
    if skewed:
        assert _report_sum("cut record by record") > 0
    assert out["kmc_oracle"] == out["kmc_oracle_all"]
    assert sum(r[2] for r in out["kmc_oracle"]) > 1_000_000 and (skewed or len(out["kmc_oracle"]) >= 64)
    assert any(r[3] for r in out["kmc_oracle"]), "n_plus_x_recs is zero everywhere: the k+x-mer path was not exercised"


# ---- the stage-1 KERNELS inside the real pipeline: kmc_emu_s1 = reference KMC + the splitter worker + an engine that runs
# kmc_amd/csrc/stage1_kernels.hip.h under the CPU emulation of tests/hipemu, in the order a HIP engine will launch them
needs_emu = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "kmc_emu_s1")), reason="oracle/_ref/kmc_emu_s1 not built (needs /root/reference)")


def _small_text(fmt, eol, k, n_reads=1500):
    rng = np.random.default_rng(13)
    genome = _rnd(rng, 20_000)
    recs = _edge_case_records(k) + [genome[a:a + 100] for a in rng.integers(0, len(genome) - 100, size=n_reads)]
    out = []
    for i, r in enumerate(recs):
        if fmt == "fq":
            out.append(b"@r%d some text" % i + eol + r + eol + b"+" + eol + b"@" * len(r) + eol)  # quality lines that start like titles
        else:
            out.append(b">r%d some text" % i + eol + r + eol)
    return b"".join(out)


@needs_emu
@pytest.mark.parametrize("fmt,eol,flags", [("fq", b"\n", ["-k27", "-ci1"]), ("fq", b"\r\n", ["-k27", "-ci1", "-b"]), ("fa", b"\n", ["-k55", "-ci1"]),
                                           ("fa", b"\r\n", ["-k21"]), ("fq", b"\n", ["-k127", "-ci1"])], ids=lambda v: v if isinstance(v, str) else None)
def test_emulated_stage1_kernels_inside_the_reference_pipeline(fmt, eol, flags, tmp_path):
    """text of the reader's parts -> k_s1_text_to_codes -> k_s1_check_records -> k_s1_cut -> k_s1_bin_totals + k_s1_bin_plus_x -> k_s1_bin_layout
    -> k_s1_emit -> the worker's buffers -> reference storer, bin files, stage 2: the database and the statistics (reads, super-k-mers, k-mers)
    must be the unmodified reference's"""
    k = int(flags[0][2:])
    path = str(tmp_path / ("in." + fmt))
    with open(path, "wb") as f:
        f.write(_small_text(fmt, eol, k))
    common = flags + ["-m2", "-sf1", "-sr1"] + (["-fa"] if fmt == "fa" else [])
    want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
    got = _run("kmc_emu_s1", common + ["-sp2"], path, tmp_path, "emu", env={"KMC_HIP_VERBOSE": "1"})
    assert got == want
    assert _report_sum("parts through the engine") > 0 and _report_sum("uncovered parts") == 0 and _report_sum("bin pieces") > 0


def _run_raw(exe, flags, inp, tmp_path, tag, env=None):
    t = tmp_path / ("tmp_" + tag)
    t.mkdir(exist_ok=True)
    return subprocess.run([_exe(exe), *flags, inp, str(tmp_path / ("db_" + tag)), str(t)], capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=600)


@needs_emu
def test_text_the_kernels_do_not_cover_stops_the_run(tmp_path):
    """FAIL CLOSED (round 5). A blank line between records is fine with CSplitter::GetSeq (splitter.cpp:293-298) but not with the kernels' line model: the
    record check flags the part and the worker must stop the run with a message that names the cause — it has no path into the reference splitter. Only the
    test build with -DKMC_HIP_S1_REFERENCE_FALLBACK, and only under KMC_HIP_S1_FALLBACK=1, hands the part over (and the database then does not change)."""
    path = str(tmp_path / "in.fq")
    text = _small_text("fq", b"\n", 27, n_reads=600)
    cut = text.index(b"@r300 ")
    with open(path, "wb") as f:
        f.write(text[:cut] + b"\n" + text[cut:])
    common = ["-k27", "-ci1", "-m2", "-sf1", "-sr1"]
    r = _run_raw("kmc_emu_s1", common + ["-sp1"], path, tmp_path, "emu", env={"KMC_HIP_VERBOSE": "1", "KMC_HIP_S1_FALLBACK": "1"})  # the switch means nothing to the default build
    assert r.returncode != 0 and "malformed FASTA / FASTQ" in r.stdout + r.stderr, (r.stdout + r.stderr)[-800:]
    if not os.path.exists(_exe("kmc_emu_s1_fb")):
        pytest.skip("oracle/_ref/kmc_emu_s1_fb not built")
    r = _run_raw("kmc_emu_s1_fb", common + ["-sp1"], path, tmp_path, "fb0")  # compiled in, not switched on: still an error
    assert r.returncode != 0 and "malformed FASTA / FASTQ" in r.stdout + r.stderr
    want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
    got = _run("kmc_emu_s1_fb", common + ["-sp1"], path, tmp_path, "fb1", env={"KMC_HIP_VERBOSE": "1", "KMC_HIP_S1_FALLBACK": "1"})
    assert got == want and _report_sum("uncovered parts") > 0


@needs_emu
@pytest.mark.parametrize("flags,what", [(["-k27", "-hc"], "homopolymer compression"), (["-k27", "-fm"], "other than FASTA / FASTQ"), (["-k27", "--opt-out-size"], "histogram estimation")],
                         ids=lambda v: "".join(v) if isinstance(v, list) else None)
def test_jobs_the_device_splitter_does_not_cover_are_refused(flags, what, tmp_path):
    """-hc, multi-line FASTA / BAM / KMC input and --opt-out-size: the stage-1 worker refuses the job by name and points at kmc_hip (reference stage 1 + device stage 2);
    it does not run the reference's CWSplitter behind the user's back"""
    path = str(tmp_path / ("in.fa" if "-fm" in flags else "in.fq"))
    with open(path, "wb") as f:
        f.write(_small_text("fa" if "-fm" in flags else "fq", b"\n", 27, n_reads=200))
    r = _run_raw("kmc_emu_s1", flags + ["-ci1", "-m2", "-sf1", "-sr1", "-sp1"], path, tmp_path, "emu")
    assert r.returncode != 0 and "does not cover" in r.stdout + r.stderr and what in r.stdout + r.stderr and "kmc_hip" in r.stdout + r.stderr, (r.stdout + r.stderr)[-800:]


@needs_emu
@pytest.mark.parametrize("fmt", ["fa", "fq"])
def test_long_lines_and_long_read_parts_inside_the_reference_pipeline(fmt, tmp_path):
    """the reader's own long-read parts (a record longer than its buffer: fastq_reader.cpp:704-721, :843-860) and lines of mem_part_pmm_reads = 524 296 symbols or
    more inside ordinary parts, through the emulated kernels inside the reference pipeline: database AND statistics (reads, super-k-mers) equal the reference's,
    no part refused. -m2 makes the reader's buffer small (kmc.h:390-399) so that a few Mbp are enough."""
    rng = np.random.default_rng(3)
    k = 27
    recs = [_rnd(rng, 200), _rnd(rng, 600_000), _rnd(rng, 150), _rnd(rng, int(os.environ.get("KMC_TEST_LONG_READ_BP", "3400000"))), _rnd(rng, 90), _rnd(rng, 530_000)]
    path = str(tmp_path / ("in." + fmt))
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write((b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n") if fmt == "fq" else (b">r%d\n" % i + r + b"\n"))
    common = ["-k%d" % k, "-ci1", "-m2", "-sf1", "-sr1"] + (["-fa"] if fmt == "fa" else [])
    want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
    got = _run("kmc_emu_s1", common + ["-sp2"], path, tmp_path, "emu", env={"KMC_HIP_VERBOSE": "1"})
    assert got == want
    assert _report_sum("parts through the engine") > 0 and _report_sum("uncovered parts") == 0
    assert _report_sum("of them long-read parts") >= 2  # the 3.4 Mbp record does not fit the reader's 3 MB buffer ("Input buffer size" under -m2)


@needs_emu
def test_emulated_chain_retries_the_cut_when_its_first_guess_is_short(tmp_path):
    """the number of super-k-mers is known only after the cutting kernel ran: the chain guesses, and cuts again with the exact number when the
    guess was short (kmc_amd/csrc/stage1_chain.h). A guess of 4096 against ~7 000 super-k-mers forces that path."""
    path = str(tmp_path / "in.fq")
    with open(path, "wb") as f:
        f.write(_small_text("fq", b"\n", 27))
    common = ["-k27", "-ci1", "-m2", "-sf1", "-sr1"]
    want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
    n_sk = int([ln for ln in subprocess.run([os.path.join(REF, "kmc"), *common, path, str(tmp_path / "x"), str(tmp_path)], capture_output=True, text=True).stdout.splitlines()
                if "super-k-mers" in ln][0].split(":")[1])
    assert n_sk > 4096
    assert _run("kmc_emu_s1", common + ["-sp1"], path, tmp_path, "emu", env={"KMC_EMU_SK_GUESS_DIV": "1000000000000"}) == want


def test_hip_stage1_binary_fails_loudly_without_a_gpu(tmp_path):
    """kmc_hip_s1 (every plug-in over libkmc_hip.so, the splitter over kmc_hip_split_part) has no CPU fallback either: without a GPU the first
    part must stop the run with the engine's error, not be split on the host"""
    exe = _exe("kmc_hip_s1")
    if not os.path.exists(exe):
        pytest.skip("kmc_amd/bin/kmc_hip_s1 not built")
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=7, genome_len=20_000, n_reads=500)
    for eager in ("1", "0"):
        tmp = tmp_path / ("tmp" + eager)
        tmp.mkdir()
        env = dict(os.environ, KMC_HIP_LIB=os.path.join(ROOT, "kmc_amd", "libkmc_hip.so"), KMC_HIP_EAGER_INIT=eager)
        r = subprocess.run([exe, "-k27", "-t2", fq, str(tmp_path / ("out" + eager)), str(tmp)], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0, r.stdout + r.stderr
        assert "split engine" in r.stdout + r.stderr and ("no ROCm-capable device" in r.stdout + r.stderr or "HIP" in r.stdout + r.stderr), r.stdout + r.stderr
    # a job the device splitter does not cover is refused by name before any device is needed (no silent hand-over to the reference's CWSplitter)
    tmp = tmp_path / "tmphc"
    tmp.mkdir()
    r = subprocess.run([exe, "-k27", "-t2", "-hc", fq, str(tmp_path / "outhc"), str(tmp)], env=dict(os.environ, KMC_HIP_LIB=os.path.join(ROOT, "kmc_amd", "libkmc_hip.so")),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not cover homopolymer compression" in r.stdout + r.stderr, r.stdout + r.stderr


def _random_text(seed):
    """a FASTA or FASTQ file of random records: reads of a small genome and of a skewed alphabet (N, IUPAC, lower case), empty / homopolymer /
    periodic reads, titles and quality strings of random printable characters ('@', '+', '>' anywhere), LF and CRLF mixed line by line"""
    rng = np.random.default_rng(seed)
    fmt = ["fq", "fa"][int(rng.integers(0, 2))]
    k = int(rng.choice([11, 21, 27, 28, 32, 40, 55]))
    alpha = np.frombuffer(b"ACGTacgtNnRY", dtype=np.uint8)
    p = np.array([20, 20, 20, 20, 2, 2, 2, 2, 1, 0.5, 0.3, 0.2])
    p = p / p.sum()
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=5000)]
    out = []
    for _ in range(int(rng.integers(1, 800))):
        mode = rng.random()
        if mode < 0.6:
            a = int(rng.integers(0, 4800))
            r = genome[a:a + int(rng.integers(0, 200))].tobytes()
        elif mode < 0.9:
            r = alpha[rng.choice(alpha.size, size=int(rng.integers(0, 300)), p=p)].tobytes()
        else:
            r = [b"", b"A" * int(rng.integers(1, 400)), b"ACGT" * int(rng.integers(1, 60)), b"N" * 5][int(rng.integers(0, 4))]
        eol = b"\r\n" if rng.random() < 0.3 else b"\n"
        title = (b"@" if fmt == "fq" else b">") + bytes(rng.integers(33, 127, size=int(rng.integers(0, 40)), dtype=np.uint8))
        if fmt == "fq":
            q = bytes(rng.integers(33, 127, size=len(r), dtype=np.uint8))
            out.append(title + eol + r + eol + b"+" + (title[1:] if rng.random() < 0.2 else b"") + eol + q + eol)
        else:
            out.append(title + eol + r + eol)
    return fmt, k, b"".join(out)


@needs_emu
@pytest.mark.parametrize("seed", range(8))
def test_emulated_stage1_kernels_on_random_text(seed, tmp_path):
    """seeded sweep (130 more seeds were run once while this was written: no difference): database and statistics of the reference pipeline
    with the emulated stage-1 kernels vs the unmodified reference"""
    fmt, k, text = _random_text(seed)
    path = str(tmp_path / ("in." + fmt))
    with open(path, "wb") as f:
        f.write(text)
    flags = ["-k%d" % k, "-ci1", "-m2", "-sf1", "-sr1"] + (["-fa"] if fmt == "fa" else [])
    assert _run("kmc_emu_s1", flags + ["-sp1"], path, tmp_path, "emu") == _run("kmc", flags + ["-sp1"], path, tmp_path, "ref")


# ---- the PRODUCT binaries on the CPU: kmc_hip / kmc_hip_s1 (plug-ins + the dlopen loaders hip_loader.cpp / hip_split_loader.cpp) bound to a
# mock of libkmc_hip.so (tests/hipemu/mock_hip_lib.cpp: stage-2 oracle per bin, emulated stage-1 chain per part) instead of the GPU library
@pytest.mark.parametrize("exe,flags", [("kmc_hip_s1", ["-k27", "-ci1", "-sp3", "-sr3"]), ("kmc_hip_s1", ["-k55", "-b", "-sp2", "-sr1"]), ("kmc_hip", ["-k27", "-ci1", "-sp2", "-sr2"])],
                         ids=lambda v: v if isinstance(v, str) else "".join(v))
def test_product_binaries_over_a_mock_library_write_the_reference_database(exe, flags, tmp_path):
    """everything between the reference pipeline and the C-ABI — worker plug-ins, ordered emission with several workers, loaders, argument
    marshalling of kmc_hip_process_bin_submit/_wait and kmc_hip_split_set_map/_split_part, two configured devices — runs here as shipped; only the
    library behind the C-ABI is a stand-in"""
    if not os.path.exists(_exe(exe)):
        pytest.skip("oracle/_ref/%s not built" % exe)
    import emu

    mock = emu.build_mock()
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=11, genome_len=50_000, n_reads=3_000, read_len=150)
    common = [f for f in flags if not f.startswith(("-sp", "-sr"))] + ["-m2", "-sf1"]
    want = _run("kmc", common + ["-sp1", "-sr1"], fq, tmp_path, "ref")
    got = _run(exe, flags + ["-m2", "-sf1"], fq, tmp_path, "mock", env={"KMC_HIP_LIB": mock, "KMC_HIP_DEVICES": "0,1", "KMC_HIP_VERBOSE": "1"})
    assert got == want
    if exe == "kmc_hip_s1":
        assert _report_sum("parts through the engine") > 0 and _report_sum("bin pieces") > 0


@needs_emu
@pytest.mark.parametrize("fmt,flags", [("fq", ["-k27", "-ci1"]), ("fa", ["-k55", "-b"]), ("fq", ["-k21", "-n64"])], ids=lambda v: v if isinstance(v, str) else "".join(v))
def test_emit_through_a_sort_gives_the_same_database_and_bins_in_read_order(fmt, flags, tmp_path):
    """the alternative emit (k_s1_sort_keys -> sort by bin -> k_s1_emit_sorted, KMC_HIP_S1_SORTED_EMIT=1): same database, and — because the sort is
    stable — every bin holds its super-k-mers in read order, i.e. the bin DESCRIPTORS and the bin images equal those of the reference's splitter
    run with one thread (checked through the database and through the per-bin descriptors)"""
    k = int(flags[0][2:])
    path = str(tmp_path / ("in." + fmt))
    with open(path, "wb") as f:
        f.write(_small_text(fmt, b"\n", k))
    common = flags + ["-m2", "-sf1", "-sr1"] + (["-fa"] if fmt == "fa" else [])
    want = _run("kmc", common + ["-sp1"], path, tmp_path, "ref")
    got = _run("kmc_emu_s1", common + ["-sp2"], path, tmp_path, "emu", env={"KMC_HIP_S1_SORTED_EMIT": "1", "KMC_HIP_VERBOSE": "1"})
    assert got == want and _report_sum("parts through the engine") > 0


def test_bin_dump_of_the_stage2_worker_is_what_bench_reads(tmp_path):
    """bench.py's e2e_large leg runs the bins of a REAL run device-resident: kmc_hip's worker writes what it received from the reference's stage 1
    ($KMC_HIP_BIN_DUMP_DIR: image + pack list per bin) and bench.DumpedBins reads it back. Here on the mock library: the dumped bins, counted by the oracle,
    must add up to the statistics the run printed, and the pack lists must tile the images."""
    if not os.path.exists(_exe("kmc_hip")):
        pytest.skip("kmc_amd/bin/kmc_hip not built")
    import sys

    import emu
    import oracle_py as O

    sys.path.insert(0, ROOT)
    import bench

    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=5, genome_len=40_000, n_reads=3_000, read_len=150)
    dump = tmp_path / "dump"
    dump.mkdir()
    md5, stats = _run("kmc_hip", ["-k27", "-m2", "-sf1", "-sp2", "-sr2"], fq, tmp_path, "hip", env={"KMC_HIP_LIB": emu.build_mock(), "KMC_HIP_BIN_DUMP_DIR": str(dump)})
    sb = bench.DumpedBins(str(dump))
    assert len(sb.meta) == 512 and 0 < len(sb.own) <= 512
    tot = np.zeros(4, dtype=np.uint64)
    for b in sb.own:
        img = sb.image(b)
        assert img.size == sb.size[b] == int(sb.packs(b).sum())
        _, _, st = O.process_bin(O.make_params(27), img, int(sb.n_rec[b]))
        tot += np.asarray(st, dtype=np.uint64)
    below, above, unique, counted, total = (int(x) for x in stats[:5])
    assert [int(x) for x in tot] == [unique, below, above, total] and counted == unique - below - above
