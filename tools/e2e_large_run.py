"""bench.py's e2e_large leg on its own (the drop-in against the reference on ONE large FASTQ + the reference's own bins device-resident): python tools/e2e_large_run.py [Gbp] [k]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kmc_amd import capi  # noqa: E402

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 27
capi.require_gpu_backend()
ctx = capi.Context((0,))
print(json.dumps(bench.e2e_large_leg(ctx, k, gbp, budget_s=1500.0), indent=1))
ctx.close()
