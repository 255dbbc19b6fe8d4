"""Match records per KiB and per sub-block (oracle, 1 MiB per input kind): the measurement behind the compact record
slots of a batch plan (brotli_amd/csrc/m4g.hip spec_alloc).  Uses the oracle: analysis tool only."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import OracleM4
from brotli_amd import synth
rng = np.random.default_rng(1)
def kinds(n):
    yield "text", synth.text(n)
    yield "log", synth.log(n)
    for a in (2, 3, 4, 8, 16):
        yield f"alpha{a}", rng.integers(0, a, n, dtype=np.uint8).tobytes()
    # short phrases from a small dictionary, no separators
    words = [bytes(rng.integers(97, 123, int(l), dtype=np.uint8)) for l in rng.integers(4, 7, 200)]
    yield "words4-6", b"".join(words[i] for i in rng.integers(0, 200, n // 5))[:n]
    words = [bytes(rng.integers(97, 123, 4, dtype=np.uint8)) for _ in range(64)]
    yield "words4", b"".join(words[i] for i in rng.integers(0, 64, n // 4))[:n]
    yield "digits", ("".join(str(x) for x in rng.integers(0, 10**6, n // 5))).encode()[:n]
for params in ((0,0,0,0,0,0), (1<<20, 4, 5, 17, 8, 0)):
  for name, data in kinds(1 << 20):
    m = OracleM4(*params).FindMatches(data)
    pos = np.cumsum(m[:, 0].astype(np.int64) + m[:, 1]) - m[:, 1]   # match start
    for sub in (1024, 2048):
        cnt = np.bincount(pos // sub, minlength=len(data) // sub)
        print(params[4], name, sub, "avg/KiB %.1f" % (len(m) / 1024), "max per sub", cnt.max(), "slot", sub // 8 + 2,
              "over", int((cnt > sub // 8 + 2).sum()), "of", len(cnt))
