"""oracle/m4_py.py -- second, independently structured restatement of
matchfinder.M4 (reference: matchfinder/m4.go:63-339, emitter.go:27-34).

TEST INFRASTRUCTURE ONLY (same rules as oracle/m4_ref.h).  It exists because
the reference ships no golden Match lists and Go is not installed: the
line-faithful C restatement (m4_ref.c) is cross-checked against this one,
which is deliberately built differently -- on the decomposition of
SURVEY.md section 3.2:

  * phase 1 (bytes only): hash every inserted position, then derive
    ``chain[i] = i - (latest earlier inserted position with the same hash)``
    with a dictionary sweep.  No hash table indexed by position order is kept
    during the parse.
  * phase 2 (serial parse): the 3-deep pending-match state machine, which
    *jumps* over positions that m4.go:149-151 would skip, instead of visiting
    every byte.

This is also the algorithmic shape of the GPU kernels (K1 chain build, K3
parse), so agreement between the two oracles validates that shape.
Pure Python: use on inputs of at most a few hundred KiB.
"""
from __future__ import annotations

import numpy as np

HASH_MUL64 = 0x1E35A7BD1E35A7BD
MASK64 = (1 << 64) - 1


class M4Py:
    """Mirror of matchfinder.M4's exported fields (m4.go:15-43)."""

    def __init__(self, MaxDistance=0, MinLength=0, HashLen=0, TableBits=0,
                 ChainLength=0, DistanceBitCost=0):
        self.MaxDistance = MaxDistance
        self.MinLength = MinLength
        self.HashLen = HashLen
        self.TableBits = TableBits
        self.ChainLength = ChainLength
        self.DistanceBitCost = DistanceBitCost
        self.Reset()

    def Reset(self):
        self.history = bytearray()
        self.chain = []          # delta to previous same-hash position, 0 = none
        self.last = {}           # hash -> latest inserted history index

    # -- phase 1 ----------------------------------------------------------
    def _hashes(self, buf: bytes, lo: int, hi: int) -> np.ndarray:
        """hash of every position in [lo, hi) (m4.go:141)."""
        if hi <= lo:
            return np.zeros(0, dtype=np.uint64)
        a = np.frombuffer(buf, dtype=np.uint8)
        v = np.zeros(hi - lo, dtype=np.uint64)
        for k in range(min(self.HashLen, 8)):
            v |= a[lo + k:hi + k].astype(np.uint64) << np.uint64(8 * k)
        with np.errstate(over="ignore"):
            v = v * np.uint64(HASH_MUL64)
        return v >> np.uint64(64 - self.TableBits)

    def _slide(self):
        """m4.go:82-99 expressed on the dictionary: entries that would be
        clamped to 0 become 'empty'."""
        if len(self.history) > 2 * self.MaxDistance:
            delta = len(self.history) - self.MaxDistance
            del self.history[:delta]
            del self.chain[:delta]
            new = {}
            for h, p in self.last.items():
                if p - delta > 0:
                    new[h] = p - delta
            self.last = new

    # -- phase 2 ----------------------------------------------------------
    def _score(self, m):
        if m is None:
            return 0
        s, e, c = m
        return (e - s) * 256 - (s - c).bit_length() * self.DistanceBitCost

    def _extend2(self, src, n, start, cand, lo):
        """m4.go:328-339: forward from +4, backward not below lo / source 0."""
        i, j = cand + 4, start + 4
        while j < n and src[i] == src[j]:
            i += 1
            j += 1
        while start > lo and cand > 0 and src[start - 1] == src[cand - 1]:
            start -= 1
            cand -= 1
        return (start, j, cand)

    def FindMatches(self, dst: list, block: bytes) -> list:
        if self.MaxDistance == 0:
            self.MaxDistance = 65535
        if self.MinLength == 0:
            self.MinLength = 4
        if self.HashLen == 0:
            self.HashLen = 6
        if self.TableBits == 0:
            self.TableBits = 17
        self._slide()
        next_emit = len(self.history)
        first = next_emit
        self.history += block
        src = bytes(self.history)
        n = len(src)
        self.chain.extend([0] * len(block))
        chain = self.chain

        # phase 1: chain deltas for the newly inserted positions [first, n-7)
        hs = self._hashes(src, first, n - 7)
        last = self.last
        for k, h in enumerate(hs.tolist()):
            i = first + k
            p = last.get(h, 0)
            if p != 0:
                chain[i] = i - p
            last[h] = i
        # position 0 is indistinguishable from "empty" (table value 0)
        # -> last.get(h) == 0 is treated as none above.

        min_len = self.MinLength
        max_dist = self.MaxDistance
        out = dst
        m0 = m1 = None

        def emit(m):
            nonlocal next_emit
            s, e, c = m
            out.append((s - next_emit, e - s, s - c))
            next_emit = e

        def flush():
            nonlocal m0, m1
            if m1 is not None:
                s, e, c = m1
                if e > m0[0]:
                    e = m0[0]
                m1 = (s, e, c)
                if e - s >= min_len and self._score(m1) > 0:
                    emit(m1)
            emit(m0)
            m0 = m1 = None

        i = first
        end = n - 7
        while i < end:
            if m0 is not None and i >= m0[1]:
                flush()
            if m0 is None:
                if out:
                    pd = out[-1][2]
                    if i + 1 - pd < 0:
                        raise IndexError("repeat distance out of range")
                    if src[i + 1:i + 5] == src[i + 1 - pd:i + 5 - pd]:
                        m = self._extend2(src, n, i + 1, i + 1 - pd, next_emit + 1)
                        if m[1] - m[0] >= min_len:
                            m0 = m
            if m0 is not None and i < m0[1]:
                probe = m0[1] + 2 - self.HashLen
                if i != probe:
                    # jump: nothing happens until the probe position or the end
                    i = probe if i < probe < m0[1] else m0[1]
                    continue
            d = chain[i]
            if d == 0 or d > max_dist:
                i += 1
                continue
            cand = i - d
            cur = None
            if src[cand:cand + 4] == src[i:i + 4]:
                m = self._extend2(src, n, i, cand, next_emit)
                if m[1] - m[0] > min_len and self._score(m) > 0:
                    cur = m
            for _ in range(self.ChainLength):
                d = chain[cand]
                if d == 0:
                    break
                cand -= d
                if cand <= 0 or i - cand > max_dist:
                    break
                if src[cand:cand + 4] == src[i:i + 4]:
                    m = self._extend2(src, n, i, cand, next_emit)
                    if m[1] - m[0] > min_len and self._score(m) > self._score(cur):
                        cur = m
            if cur is None or cur[1] - cur[0] < min_len:
                i += 1
                continue
            penalty = 0
            if m0 is not None:
                penalty = 275
                if cur[0] <= (m1[1] if m1 is not None else 0):
                    penalty = 0
            if self._score(cur) <= self._score(m0) + penalty:
                i += 1
                continue
            m2, m1, m0 = m1, m0, cur
            if m2 is not None:
                if m0[0] < m2[1]:
                    m1 = m2
                elif m0[0] < m2[1] + min_len:
                    emit(m2)
                    m1 = None
                else:
                    s, e, c = m2
                    if e > m1[0]:
                        e = m1[0]
                        if self.ChainLength > 0 and e - s >= min_len:
                            pos = s
                            while True:
                                dd = chain[pos]
                                if dd == 0:
                                    break
                                pos -= dd
                                if pos <= c:
                                    break
                                if src[s:e] == src[pos:pos + e - s]:
                                    c = pos
                                    break
                    m2 = (s, e, c)
                    if e - s >= min_len and self._score(m2) > 0:
                        emit(m2)
            i += 1

        if m1 is not None:
            s, e, c = m1
            if e > m0[0]:
                e = m0[0]
            m1 = (s, e, c)
            if e - s >= min_len and self._score(m1) > 0:
                emit(m1)
        if m0 is not None:
            emit(m0)
        if next_emit < n:
            out.append((n - next_emit, 0, 0))
        return out
