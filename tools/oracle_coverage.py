#!/usr/bin/env python3
"""Branch coverage of the CPU oracles (oracle/m4_ref.c, bargain_ref.c, flate_ref.c, encoder_ref.c) under the CPU
test suite.

VERDICT r1 1(d): the two restatements were written by the same reader, so
agreement alone does not show that the agreeing inputs reach every arm of the
parse.  This script builds the oracle with gcov instrumentation (same sources,
-O0), runs tests/test_oracle.py against that build (M4G_ORACLE_SO), and writes
profiles/oracle_coverage.txt: the gcov summary plus every branch of
m4ref_find_matches / extend_match / extend_match2 with its taken count.  It
exits non-zero when a branch of those functions was never taken.

Run:  python tools/oracle_coverage.py      (CPU only, ~1 minute; --check: verdict only, no report)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Arms no input can reach, with the reason (source text of the line -> branch index -> why).
# A match enters the queue only at a position i < matches[0].End (the one searched position
# End+2-HashLen inside a pending match, m4.go:149; at i >= End the flush of m4.go:113 runs first), and
# its Start <= i.  So whenever two matches are pending the newer one starts before the older one ends:
# the three "older.End > newer.Start" tests are always true.
UNREACHABLE = {
    "if (matches[1].End > matches[0].Start)": "newer match always starts inside the older one (m4.go:117)",
    "if (matches[2].End > matches[1].Start) {": "same, for the oldest of three (m4.go:233)",
    "if (matches[1].End > matches[0].Start) matches[1].End = matches[0].Start;": "same, after the loop (m4.go:264)",
}


# The three newer oracles: every function is in scope; arms that no input can reach are listed with the reason
# (source text of the line -> {branch index: why}; "*" = every branch of that line not taken).
ALLOC = "allocation failure / NULL-argument guard of the C harness (not part of the restated algorithm)"
API = "calling convention of the test harness (optional output argument, capacity check), not the algorithm"
MORE = {
    "bargain_ref.c": {
        "if (isnan(x) || (isinf(x) && x > 0)) return x;": {"*": "math.Log's special cases: the argument is len / count, finite and >= 1"},
        "if (x < 0) return NAN;": {"*": "same"},
        "if (x == 0) return -INFINITY;": {"*": "same"},
        "if (isnan(x) || isinf(x) || x <= 0) return go_log(x);": {"*": "same"},
        "if (variant < 1 || variant > 3) return NULL;": {"*": API},
        "if (!z) return NULL;": {"*": ALLOC}, "if (!z->table[t]) {": {"*": ALLOC}, "if (!z) return;": {"*": ALLOC},
        "if (!nv) return -1;": {"*": ALLOC}, "z->arrivals = malloc((n ? n : 1) * sizeof(arrival));": {"*": ALLOC},
        "if (!z->arrivals) return -1;": {"*": ALLOC}, "if (!nh) return -1;": {"*": ALLOC},
        "if (repIndex >= 0 && load32(src + repIndex) == (uint32_t)cv) {":
            {"1": "repIndex = i - distance of an arrival at a position <= i, whose source was inside the buffer: never negative "
                  "(bargain1.go:150 keeps the test because a slide can move the history)"},
        "if (vec_push(dst, m)) return -1;": {"*": ALLOC},
        "if (bargainref_find_matches(z, &v, src, n)) {": {"*": ALLOC}, "if (bargainref_find_matches(z, &v, src + p, m)) {": {"*": ALLOC},
        "if (v.len > out_cap) r = -r;": {"*": API}, "else if (v.len) memcpy(out, v.v, v.len * sizeof *out);": {"*": API},
    },
    "flate_ref.c": {
        "if ((off >> 7) < 256) return g_offset_codes[off >> 7] + 14;":
            {"1": "offsets are checked to be <= 32768 before (token.go:66-71 panics above): (off - 1) >> 7 < 256 always"},
        "while (numCodegens > 4 && w->codegenFreq[kCodegenOrder[numCodegens - 1]] == 0) numCodegens--;":
            {"1": "every block has the end-of-block symbol, so some code length 1..15 is in use and the scan stops above 4"},
        "if (!e) return;": {"*": ALLOC}, "if (!dest.p) dest.p = (uint8_t *)malloc(1);": {"*": ALLOC},
    },
    "encoder_ref.c": {
        "if (dist_cache) memset(&dist_cache[i], 0, sizeof(dist_cache[i]));": {"*": API}, "if (dist_cache) dist_cache[i] = dc;": {"*": API},
        "else if (len > (1u << 24)) abort(); /* Go: panic(\"metablock too long\") */":
            {"*": "the Go code panics (compress_fragment_two_pass.go:232-234); the harness never hands over 16 MiB blocks"},
        "for (int k = 0; k < 18 && space < 32; k++) {": {"*": "constant table: the loop always ends on space == 32"},
        "if (l) space += 32 >> l;": {"*": "constant table"},
        "return x->symbol < y->symbol ? -1 : (x->symbol > y->symbol);":
            {"0": "ties: glibc's qsort is a merge sort and the array starts in ascending symbol order, so on equal counts x always "
                  "precedes y; the arm is there for a qsort that is not stable"},
        "if (!e) return;": {"*": ALLOC}, "if (nm_total) *nm_total += mv->len;": {"*": API},
        "if (nmatches_out) *nmatches_out = nm;": {"*": API}, "if (!dest.p) dest.p = (uint8_t *)malloc(1);": {"*": ALLOC},
    },
}
SUITES = ["test_oracle.py", "test_bargain_oracle.py", "test_flate_oracle.py"]


def report_file(tmp, name):
    """gcov of one oracle file: (summary text, lines of the report, missed count, total, unreachable)."""
    g = subprocess.run(["gcov", "-b", "-c", "-o", tmp, os.path.join(tmp, name[:-2] + ".gcda")],
                       cwd=tmp, capture_output=True, text=True, check=True)
    summary = g.stdout.split("\n\n")[0]
    allow = MORE[name]
    out, missed, total, unreachable = [], 0, 0, 0
    cur = None
    for ln in open(os.path.join(tmp, name + ".gcov")).read().splitlines():
        m = re.match(r"\s*([0-9#=\-]+\*?):\s*(\d+):(.*)", ln)
        if m:
            cur = (int(m.group(2)), m.group(3).strip())
            continue
        m = re.match(r"branch\s+(\d+)\s+(taken (\d+)|never executed)", ln)
        if m and cur:
            total += 1
            taken = int(m.group(3)) if m.group(3) else 0
            if taken:
                continue
            why = None
            for text, arms in allow.items():
                if cur[1].startswith(text) or text.startswith(cur[1]):
                    why = arms.get(m.group(1), arms.get("*"))
            if why:
                unreachable += 1
            else:
                missed += 1
            out.append(f"{name}:{cur[0]:<4d} branch {m.group(1)} never taken | {cur[1][:100]}"
                       + (f"   <- {why}" if why else "   <- MISSED"))
    return summary, out, missed, total, unreachable


def main():
    tmp = tempfile.mkdtemp(prefix="m4cov_")
    try:
        so = os.path.join(tmp, "libm4oracle_cov.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in ("m4_ref.c", "encoder_ref.c", "flate_ref.c", "bargain_ref.c")]
        objs = []
        for s in srcs:
            o = os.path.join(tmp, os.path.basename(s)[:-2] + ".o")
            subprocess.run(["gcc", "-O0", "-g", "-fPIC", "--coverage", "-std=gnu11", "-ffp-contract=off", "-c", s, "-o", o], check=True)
            objs.append(o)
        subprocess.run(["gcc", "-shared", "--coverage", "-o", so] + objs + ["-lm"], check=True)
        env = dict(os.environ, M4G_ORACLE_SO=so)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x"] + [os.path.join(ROOT, "tests", t) for t in SUITES],
                           env=env, cwd=ROOT, capture_output=True, text=True)
        tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            return 2
        g = subprocess.run(["gcov", "-b", "-c", "-o", tmp, os.path.join(tmp, "m4_ref.gcda")],
                           cwd=tmp, capture_output=True, text=True, check=True)
        summary = g.stdout.split("\n\n")[0]
        lines = open(os.path.join(tmp, "m4_ref.c.gcov")).read().splitlines()
        out, missed, total, unreachable = [], 0, 0, 0
        cur_src, in_scope = None, False
        for ln in lines:
            m = re.match(r"\s*([0-9#=\-]+\*?):\s*(\d+):(.*)", ln)
            if m:
                cur_src = (int(m.group(2)), m.group(3).rstrip(), m.group(1).strip())
                continue
            if ln.startswith("function "):
                name = ln.split()[1]
                in_scope = name in ("m4ref_find_matches", "extend_match", "extend_match2", "score", "emit")
                continue
            m = re.match(r"branch\s+(\d+)\s+(taken (\d+)|never executed)", ln)
            if m and in_scope and cur_src:
                taken = int(m.group(3)) if m.group(3) else 0
                why = UNREACHABLE.get(cur_src[1].strip()) if (taken == 0 and m.group(1) == "1") else None
                total += 1
                missed += taken == 0 and why is None
                unreachable += why is not None
                out.append(f"m4_ref.c:{cur_src[0]:<4d} branch {m.group(1)}  taken {taken:>10d}   | {cur_src[1].strip()[:90]}"
                           + (f"   <- unreachable: {why}" if why else ""))
        more = {name: report_file(tmp, name) for name in MORE}
        path = os.path.join(ROOT, "profiles", "oracle_coverage.txt")
        if "--check" in sys.argv:  # tests: verdict only, leave the committed report alone
            path = os.path.join(tmp, "oracle_coverage.txt")
        with open(path, "w") as f:
            f.write("Branch coverage of the CPU oracles (gcc --coverage -O0) under tests/" + ", tests/".join(SUITES) + "\n")
            f.write("generated by tools/oracle_coverage.py; pytest: " + tail + "\n\n")
            for name, (summ, o, mi, tot, unr) in more.items():
                f.write(summ + "\n")
                f.write(f"{name}: {tot - mi - unr} of {tot} branch outcomes taken, {unr} that no input of the harness can reach "
                        f"(listed with the reason), {mi} missed\n")
                f.write("\n".join(o) + "\n\n")
            f.write(summary + "\n\n")
            f.write(f"branches of the parse (m4ref_find_matches, extend_match, extend_match2, score, emit): "
                    f"{total - missed - unreachable} of {total} taken at least once, {unreachable} structurally "
                    f"unreachable (marked below), {missed} missed\n\n")
            f.write("\n".join(out) + "\n")
        print(f"{path}: m4_ref.c {total - missed - unreachable}/{total} branches taken, {unreachable} unreachable, {missed} missed")
        for o in out:
            if re.search(r"taken\s+0\s", o) and "unreachable" not in o:
                print("NEVER TAKEN:", o)
        bad = missed
        for name, (summ, o, mi, tot, unr) in more.items():
            print(f"  {name}: {tot - mi - unr}/{tot} taken, {unr} unreachable, {mi} missed")
            for x in o:
                if x.endswith("MISSED"):
                    print("NEVER TAKEN:", x)
            bad += mi
        return 1 if bad else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
