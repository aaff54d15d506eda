"""Child process of tests/test_hostlib_emulated.py (needs KMC_HIP_LIB = the emulated host library and HIPEMU_DEVICES=2 before kmc_amd.capi loads):
the multi-device paths of the host library on two emulated devices — a context over (0, 1), a wide-record bin (k = 200, SIZE 7: the kernels
that ask for more than 64 KiB of dynamic LDS, per-device function attribute) on the SECOND device, kmc_hip_allreduce_stats over both."""
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
c2.close()
print("two devices ok")
