"""CPU: a model check of the barrier-light LDS protocol of the workgroup transforms (hexl-fpga_amd/csrc/ntt_core.hpp: redeal_x, redeal_pass,
WgNtt / WgNttF64 forward and inverse, ReadersGate). The ownership maps of Geom (idxA, idxB, idxF<LO>) and the synchronisation each re-deal
carries are restated here; for every geometry the library instantiates and every pair of consecutive transforms in one workgroup (forward or
inverse, then forward or inverse) every (access by wave A, write by wave B) pair on overlapping words must be ordered -- by a barrier between
them or by the readers' gate. Round 6: the inverse-after-inverse pair was NOT (a one-in-10^4 wrong result at N = 2048, found on the GPU by
tools/soak_ks_random.py); the check below shows that hole without the gate and none with it."""
import itertools

import pytest


class Geom:
    def __init__(self, logn, loge):
        self.LOGN, self.LOGE = logn, loge
        self.N, self.E, self.T = 1 << logn, 1 << loge, 1 << (logn - loge)
        self.P = (logn + loge - 1) // loge
        self.KL = logn - (self.P - 1) * loge
        self.LOGT = logn - loge
        self.WB = min(self.LOGT, 6)

    def idxB(self, r, tid):
        grp = ((tid >> self.WB) << (self.LOGE - self.KL + self.WB)) + ((r >> self.KL) << self.WB) + (tid & ((1 << self.WB) - 1))
        return (grp << self.KL) + (r & ((1 << self.KL) - 1))

    def idxF(self, lo, r, tid):
        return ((tid >> lo) << (lo + self.LOGE)) + (r << lo) + (tid & ((1 << lo) - 1))

    def private(self, lo):
        return self.LOGT <= 6 or lo <= 6

    def words(self, fn, wave):
        return frozenset(fn(r, tid) for tid in range(wave * 64, min(self.T, wave * 64 + 64)) for r in range(self.E))


def forward_events(g, wave, fresh=False):
    """(kind, words) per LDS phase of one wave: W / R with the words touched, BAR, as WgNtt::fwd_pass + redeal_x issue them"""
    ev = []
    for p in range(g.P - 1):
        lo = g.LOGN - (p + 1) * g.LOGE
        upper = lambda r, t, lo=lo: g.idxF(lo, r, t)
        lower = (lambda r, t: g.idxB(r, t)) if p + 1 == g.P - 1 else (lambda r, t, lo=lo: g.idxF(lo - g.LOGE, r, t))
        if g.private(lo):
            ev += [("W", g.words(upper, wave)), ("R", g.words(lower, wave))]
        else:
            if not (fresh and p == 0):
                ev.append(("BAR", None))
            ev += [("W", g.words(upper, wave)), ("BAR", None), ("R", g.words(lower, wave))]
    return ev


def inverse_events(g, wave, gate, fresh=False):
    ev = []
    for p in range(g.P - 1):
        lo = g.KL + p * g.LOGE
        lower = (lambda r, t: g.idxB(r, t)) if p == 0 else (lambda r, t, lo=lo: g.idxF(lo - g.LOGE, r, t))
        upper = lambda r, t, lo=lo: g.idxF(lo, r, t)
        if gate and p == 0:
            ev.append(("WAIT", None))
        if g.private(lo):
            ev += [("W", g.words(lower, wave)), ("R", g.words(upper, wave))]
        else:
            if not (fresh and p == 0):
                ev.append(("BAR", None))
            ev += [("W", g.words(lower, wave)), ("BAR", None), ("R", g.words(upper, wave))]
            if gate and p == g.P - 2:
                ev.append(("ARRIVE", None))
    return ev


def staging_events(g, wave, kind):
    """the other LDS users of the keyswitch kernels at rings whose partial pass leaves a lane more than four words (KL > 2):
    `load` = load_natural_to_B / load_product_to_B (redeal_x<false, LEAD>: A positions in, B positions out) in front of an inverse transform;
    `down` = ksx_down_round's read-modify-write of `result` through LDS behind a forward transform"""
    A = g.words(lambda r, t: r * g.T + t, wave)
    B = g.words(lambda r, t: g.idxB(r, t), wave)
    if kind == "load":
        return [("BAR", None), ("W", A), ("BAR", None), ("R", B)]
    return [("BAR", None), ("W", A), ("BAR", None), ("R", B), ("W", B), ("BAR", None), ("R", A)]


def races(g, kinds, gate):
    """unordered conflicting pairs over the transforms `kinds` run back to back by one workgroup"""
    waves = (g.T + 63) // 64
    prog = []
    for w in range(waves):
        ev = []
        for k in kinds:
            ev += forward_events(g, w) if k == "fwd" else inverse_events(g, w, gate) if k == "inv" else staging_events(g, w, k)
        # annotate: barriers passed, own arrivals made, arrivals of everybody this wave has waited for
        out, bars, arrived, waited = [], 0, 0, 0
        for kind, words in ev:
            if kind == "BAR":
                bars += 1
            elif kind == "ARRIVE":
                arrived += 1
            elif kind == "WAIT":
                waited = max(waited, arrived)
            else:
                out.append((kind, words, bars, arrived, waited, len(out)))
        prog.append(out)
    found = []
    for a, b in itertools.permutations(range(waves), 2):
        for (ka, wa, bar_a, arr_a, _, ia) in prog[a]:
            for (kb, wb, bar_b, _, wait_b, ib) in prog[b]:
                if kb != "W" or not (wa & wb):
                    continue
                if bar_a != bar_b:
                    continue                                      # a barrier lies between the two
                if arr_a < wait_b:
                    continue                                      # A arrived after its access, B waited for that arrival before its write
                if ka == "W" and ia == ib:
                    continue                                      # (the same phase of both waves writing the same word would be a bug of the maps)
                found.append((a, ka, ia, b, ib))
    return found


GEOMS = [(10, 4), (11, 4), (12, 4), (13, 4), (14, 4), (11, 5), (12, 5), (13, 5), (14, 5)]


@pytest.mark.parametrize("logn,loge", GEOMS)
def test_ownership_maps_are_bijections(logn, loge):
    g = Geom(logn, loge)
    every = [lambda r, t: g.idxB(r, t)] + [lambda r, t, lo=lo: g.idxF(lo, r, t) for lo in range(0, g.LOGT + 1)]
    for fn in every:
        assert sorted(fn(r, t) for t in range(g.T) for r in range(g.E)) == list(range(g.N))


@pytest.mark.parametrize("logn,loge", GEOMS)
@pytest.mark.parametrize("kinds", [("fwd", "fwd"), ("fwd", "inv"), ("inv", "fwd"), ("inv", "inv"), ("inv", "inv", "inv"),
                                   ("fwd", "inv", "inv", "fwd", "inv")])
def test_consecutive_transforms_are_ordered_with_the_gate(logn, loge, kinds):
    assert races(Geom(logn, loge), kinds, gate=True) == []


@pytest.mark.parametrize("logn,loge", [(11, 4), (12, 4)])
@pytest.mark.parametrize("kinds", [("fwd", "down", "fwd", "down", "fwd"),             # k_ksx_main at N = 2048 / 4096: rounds, then the two mod-down rounds
                                   ("load", "inv", "load", "inv"),                    # k_ksx_intt's item loop there
                                   ("fwd", "fwd", "inv", "inv", "fwd")])              # k_ksx_special: rounds, the two inverse transforms, the next instance
@pytest.mark.parametrize("gate", [True, False])
def test_kernel_sequences_with_the_staging_phases(logn, loge, kinds, gate):
    found = races(Geom(logn, loge), kinds, gate)
    if gate or kinds[0] == "load" or "down" in kinds:
        assert found == []                                            # (the staging phases bring their own barriers)
    else:
        assert found                                                  # k_ksx_special without the gate: the round-6 race


@pytest.mark.parametrize("logn,loge", GEOMS)
def test_without_the_gate_only_inverse_after_inverse_is_open(logn, loge):
    g = Geom(logn, loge)
    for kinds in (("fwd", "fwd"), ("fwd", "inv"), ("inv", "fwd")):
        assert races(g, kinds, gate=False) == [], kinds
    open_pairs = races(g, ("inv", "inv"), gate=False)
    multi_wave = g.T > 64
    assert bool(open_pairs) == multi_wave                         # one-wave workgroups (N = 1024) have nothing to order
    if multi_wave:
        # every open pair: a READ of the first transform's last (cross-wave) exchange against a WRITE of one of the second transform's
        # wave-private re-deals (the ones in front of ITS cross-wave exchange, whose LEAD barrier orders everything behind it)
        per_wave = len([e for e in inverse_events(g, 0, False) if e[0] in "WR"])
        assert all(ka == "R" and ia == per_wave - 1 and per_wave <= ib < 2 * per_wave - 2 for (_, ka, ia, _, ib) in open_pairs)
        assert any(ib == per_wave for (_, _, _, _, ib) in open_pairs)              # ... the first of them among these
