"""CPU: the exact FP64 modular arithmetic the keyswitch kernels use on gfx950 (hexl-fpga_amd/csrc/f64_arith.hpp)
is plain IEEE-754 double mul/add/fma/rint, so the very same source is validated on the host against exact
128-bit integer arithmetic and the oracle's canonical transforms (tests/cpp/f64_selftest.cpp)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_f64_arith_selftest(orc):
    d = ROOT / "tests" / "cpp"
    subprocess.run(["make", "-C", str(d), "f64_selftest"], check=True)
    out = subprocess.run([str(d / "f64_selftest")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout


def test_xsched_table_is_the_generators_output():
    """The X-schedule table of f64_arith.hpp is data: tools/gen_xsched.py (brute force over N / X / F per stage) must reproduce it
    line for line, and an independent replay of the bound recurrence here must accept every entry."""
    import re
    gen = subprocess.run(["python3", str(ROOT / "tools" / "gen_xsched.py")], capture_output=True, text=True, check=True).stdout
    rows = [l.split("//")[0].strip() for l in gen.splitlines() if l.startswith("    {")]
    hdr = (ROOT / "hexl-fpga_amd" / "csrc" / "f64_arith.hpp").read_text()
    have = [l.split("//")[0].strip() for l in hdr.splitlines() if re.match(r"\s+\{\s*\d+, \d, \d,\s+\d+, 0x", l)]
    assert rows and rows == have
    top = {3: 2.0 ** 51 * (1 + 2.0 ** -7), 6: 2.0 ** 50, 12: 2.0 ** 49}
    for r in rows:
        period, shift, down, stages, mask = [int(x, 0) for x in re.findall(r"0x[0-9a-f]+|\d+", r)]
        a = top[period] / 2.0 ** 53
        c = 1.25 if shift else 0.625
        assert mask >> 31 == 1 and stages <= 15
        for s in range(stages):
            op = (mask >> (2 * s)) & 3
            c = {0: (1 + 1.5 * a) * c + 0.5, 1: 1.0 + 1e-9 + 1.5 * a * c, 2: 1.0 + 1e-9 + 1.5 * a * (0.5 + 1e-9)}[op]
            assert c < 1.0 / a, r
        if down:
            assert 1.7 + c < 1.0 / a, r
        assert c * top[period] ** 2 / 2 < 2.0 ** 103, r


def test_isched_table_replays_and_matches_the_generator():
    """The I-schedule table (inverse transforms) is data too: an independent replay of its recurrence accepts every entry, and
    tools/gen_isched.py reproduces the entries cheap enough to regenerate here (the period-12 tier and one of each other tier)."""
    import re
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import gen_isched
    hdr = (ROOT / "hexl-fpga_amd" / "csrc" / "f64_arith.hpp").read_text()
    have = {}
    for l in hdr.splitlines():
        m = re.match(r"\s+\{\s*(\d+), (\d+), (\d), (0x[0-9a-f]{16})ull\}", l)
        if m:
            have[(int(m.group(1)), int(m.group(2)), int(m.group(3)))] = (int(m.group(4), 16), l.split("//")[0].strip())
    assert len(have) == 3 * len(gen_isched.GEOMETRIES)
    for (period, logn, loge), (mask, _) in have.items():
        a = gen_isched.TIER_TOP[period] / 2.0 ** 53
        first = gen_isched.pass_first_stages(logn, loge)
        bs = bp = 1.25
        assert mask >> 63 == 1
        for s in range(1, logn):
            bits = (mask >> (4 * (s - 1))) & 15
            if s in first:
                assert bits & 3 == bits >> 2, (period, logn, loge, s)
            assert 2 * bs < 1 / a and 2 * bp < 1 / a
            r = 0.5 + 1e-9
            ss, ps = (r if bits & 1 else 2 * bs), (r if bits & 2 else 0.5 + 1.5 * a * 2 * bs)
            sp, pp = (r if bits & 4 else 2 * bp), (r if bits & 8 else 0.5 + 1.5 * a * 2 * bp)
            bs, bp = max(ss, sp), max(ps, pp)
        assert 2 * bs < 1 / a and 2 * bp < 1 / a
    for key in [(12, logn, loge) for logn, loge in gen_isched.GEOMETRIES] + [(6, 10, 4), (3, 10, 4)]:
        assert gen_isched.entry(*key).split("//")[0].strip() == have[key][1], key
