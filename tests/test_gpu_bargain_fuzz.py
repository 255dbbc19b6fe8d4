"""Randomised parity of m4g_bargain_find_blocks against the restated finders: mixtures of text, noise, zero pages,
short periods and paragraphs repeated at short and long range; block sizes, MaxDistance, variant and Skip drawn at
random.  Match for Match."""
import numpy as np
import pytest

from brotli_amd import m4
from oracle_lib import OracleBargain
from test_bargain_oracle import text

pytestmark = pytest.mark.gpu


def mixture(rng, n):
    out = bytearray()
    para = [text(int(rng.integers(200, 5000)), int(rng.integers(1, 10 ** 6))) for _ in range(6)]
    while len(out) < n:
        k = int(rng.integers(0, 7))
        if k == 0:
            out += text(int(rng.integers(10, 20000)), int(rng.integers(1, 10 ** 6)))
        elif k == 1:
            out += bytes(rng.integers(0, 256, size=int(rng.integers(1, 6000)), dtype=np.uint8))
        elif k == 2:
            out += bytes(int(rng.integers(1, 9000)))
        elif k == 3:
            p = bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8))
            out += p * int(rng.integers(2, 900))
        elif k == 4:
            out += para[int(rng.integers(0, len(para)))]
        elif k == 5 and len(out) > 100:
            a = int(rng.integers(0, len(out) - 50))
            out += out[a:a + int(rng.integers(4, 3000))]
        else:
            out += bytes(rng.integers(0, 4, size=int(rng.integers(1, 4000)), dtype=np.uint8))
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(36))
def test_fuzz(seed):
    rng = np.random.default_rng(1000 + seed)
    variant = 1 + seed % 3
    n = int(rng.integers(1, 400000))
    bs = int(rng.integers(3851, 150000))
    md = int(rng.choice([0, 1 << 10, 4097, 1 << 14, 1 << 16, 1 << 20]))
    skip = bool(rng.integers(0, 2))
    if skip and ((bs | (md or 1 << 16)) & 1) and n > 2 * (md or 1 << 16):
        bs += bs & 1                      # Skip needs even slides (include/m4g.h)
        if md & 1:
            md += 1
    data = mixture(rng, n)
    got, goff = m4.bargain_find_blocks(data, variant, md, skip, bs)
    want, woff = OracleBargain(variant, md, skip).find_blocks(data, bs)
    assert np.array_equal(goff, woff)
    bad = np.nonzero((got != want).any(axis=1))[0] if len(got) == len(want) else [min(len(got), len(want))]
    assert len(bad) == 0, (variant, n, bs, md, skip, int(bad[0]))


def test_skip_with_odd_slides_is_refused():
    with pytest.raises(m4.M4GError):
        m4.bargain_find_blocks(text(20000, 1), 1, 1001, True, 4096)   # odd MaxDistance, stream longer than 2 x 1001
    with pytest.raises(m4.M4GError):
        m4.bargain_find_blocks(text(20000, 1), 2, 1024, True, 4097)   # odd block size
    same_len = m4.bargain_find_blocks(text(2000, 1), 1, 1001, True, 4096)  # no slide can occur: allowed
    want, _ = OracleBargain(1, 1001, True).find_blocks(text(2000, 1), 4096)
    assert np.array_equal(same_len[0], want)
