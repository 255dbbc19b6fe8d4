"""One small call of m4g_bargain_find_blocks per variant with progress lines (for runs under a short time limit)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from brotli_amd import m4  # noqa: E402
from oracle_lib import OracleBargain  # noqa: E402

data = open(os.path.join(ROOT, "tests", "golden", "Isaac.Newton-Opticks.txt"), "rb").read()
for v in (1, 2, 3):
    print(f"variant {v}: calling", flush=True)
    t0 = time.time()
    got, off = m4.bargain_find_blocks(data, v, 1 << 20, False, 1 << 16)
    want, woff = OracleBargain(v, 1 << 20, False).find_blocks(data, 1 << 16)
    print(f"variant {v}: {time.time() - t0:.2f} s, equal: {np.array_equal(got, want) and np.array_equal(off, woff)}", flush=True)
