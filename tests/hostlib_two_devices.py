"""Child process of tests/test_hostlib_emulated.py (needs KMC_HIP_LIB = the emulated host library and HIPEMU_DEVICES=2 before kmc_amd.capi loads):
the multi-device paths of the host library on two emulated devices — a context over (0, 1), a wide-record bin (k = 200, SIZE 7: the kernels
that ask for more than 64 KiB of dynamic LDS, per-device function attribute) on the SECOND device, kmc_hip_allreduce_stats over both, and
kmc_hip_process_bin_multi: one bin over both devices (expand a share each, exchange by the top key byte through ncclSend / ncclRecv, sort + compact a
key range each, ordered emission) against the oracle — k = 27 / 55 / 200, KFF records, tiny bins (fewer k-mers than devices), a bin whose records all
share the top byte (one device owns everything), the error returns."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binsynth  # noqa: E402
import oracle_py as O  # noqa: E402
from kmc_amd import capi  # noqa: E402

assert capi.load().kmc_hip_device_count() == 2
c2 = capi.Context((0, 1))
rng = np.random.default_rng(200)
img, nk, packs = binsynth.random_bin(rng, 200, 120, max_extra=40)
p = capi.make_params(200, lut_prefix_len=4)
po = O.make_params(200, lut_prefix_len=4)
w = O.process_bin(po, img, nk)
for dev in (1, 0):
    out, lut, st = c2.process_bin(p, img, nk, packs, dev=dev)
    assert np.array_equal(out, w[0]) and np.array_equal(lut, w[1]) and np.array_equal(st, w[2]), dev
a = np.array([[1, 2, 3, 2**40], [10, 20, 30, 5]], dtype=np.uint64)
r = c2.allreduce_stats(a)
assert np.array_equal(r[0], a.sum(axis=0)) and np.array_equal(r[1], a.sum(axis=0))


def multi_equals_oracle(ctx, k, img, nk, packs, **kw):
    p = capi.make_params(k, **kw)
    op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
    want = O.process_bin(op, img, nk)
    got = ctx.process_bin(p, img, nk, packs, multi=True)
    assert all(np.array_equal(a, b) for a, b in zip(got, want)), (k, kw, nk, got[2], want[2])


for k, kw in ((27, {"lut_prefix_len": 3}), (55, {"lut_prefix_len": 3, "cutoff_min": 1}), (27, {"output_type": 1}), (200, {"lut_prefix_len": 4}), (27, {"without_output": 1})):
    (img, nk, packs, _), = capi.synth_bins(seed=7, genome_len=4000, n_reads=250, k=k, n_bins=1, n_threads=1, read_len=300 if k > 150 else 150)
    multi_equals_oracle(c2, k, img, nk, packs, **kw)
    multi_equals_oracle(c2, k, img, nk, None, **kw)  # the library finds the pack boundaries itself
for n_sk, mx in ((1, 0), (1, 5), (2, 0), (3, 1), (40, 10)):
    img, nk, packs = binsynth.random_bin(rng, 27, n_sk, max_extra=mx)
    multi_equals_oracle(c2, 27, img, nk, packs, lut_prefix_len=3, cutoff_min=1)
(img, nk, packs, _), = capi.synth_bins(seed=5, genome_len=300, n_reads=400, k=27, n_bins=1, err=0.0, n_threads=1)
multi_equals_oracle(c2, 27, img, nk, packs)
out, lut, st = c2.process_bin(capi.make_params(27), np.zeros(0, np.uint8), 0, None, multi=True)
assert out.size == 0 and not lut.any() and not st.any()
img, nk, packs = binsynth.random_bin(rng, 27, 50, max_extra=20)
for bad, code in (({"n_rec": nk + 1}, -4), ({"n_rec": nk, "out_capacity": 8}, -5)):  # KMC_HIP_ECORRUPT, KMC_HIP_ECAPACITY
    try:
        c2.process_bin(capi.make_params(27, cutoff_min=1), img, bad["n_rec"], packs, out_capacity=bad.get("out_capacity"), multi=True)
        raise SystemExit("kmc_hip_process_bin_multi accepted %r" % (bad,))
    except capi.KmcHipError as e:
        assert e.code == code, e
c2.close()
# the same over a context that names one device three times: the exchange goes through copies instead of RCCL
c3 = capi.Context((1, 1, 1))
(img, nk, packs, _), = capi.synth_bins(seed=9, genome_len=4000, n_reads=250, k=27, n_bins=1, n_threads=1)
multi_equals_oracle(c3, 27, img, nk, packs, lut_prefix_len=3)
c3.close()
print("two devices ok")
