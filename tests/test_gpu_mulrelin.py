"""GPU parity: hexl_multiply_relinearize (ciphertext multiply + relinearize in one pass, SURVEY 8f.4 -- the use-case of
the reference's combined image, device/dyadic_multiply_keyswitch.cpp:4-5) against the COMPOSITION of the two oracle
calls the reference's flow makes: DyadicMultiply (tests/test_dyadic_multiply.cpp:59-82 semantics), then KeySwitch with
result = product components 0..1 and t_target = component 2 (tests/test_dyadic_multiply_keyswitch.cpp:295-313 runs the
two primitives of that image back to back)."""
import numpy as np
import pytest

from ks_util import KsCase, primes_below, seal_chain, tier_ladder

pytestmark = pytest.mark.gpu


def operands(orc, case, b, which):
    n, L = case.n, case.L
    return np.concatenate([orc.splitmix(n, case.seed * 7 + b * 131 + which * 17 + p * 5 + i, int(case.moduli[i]))
                           for p in range(2) for i in range(L)])


def composed(orc, case, a, b):
    n, L = case.n, case.L
    prod = orc.dyadic(a, b, n, case.moduli[:L], exact=True)              # [3][L][n]
    out = prod[:2 * L * n].copy()
    orc.keyswitch(out, prod[2 * L * n:].copy(), n, L, case.K, L + 1, case.moduli, case.keys, case.modswitch)
    return out


@pytest.mark.parametrize("n,L,K,nb,strict", [(16384, 6, 7, 2, False), (16384, 7, 8, 96, False), (16384, 3, 4, 5, True),
                                             (8192, 3, 4, 70, False), (1024, 2, 3, 300, False), (8192, 2, 3, 3, True),
                                             (2048, 2, 3, 150, False), (4096, 3, 4, 120, False), (4096, 2, 3, 4, True),
                                             # round 5: limbs of different tiers (one launch per tier group, keyswitch_x.hip run_chunk_x)
                                             (16384, 6, 7, 40, "seal"), (16384, 3, 4, 5, "ladder"), (4096, 3, 4, 60, "seal"),
                                             # ... and N = 32768: every transform as two 16384-point halves (k_ksh_*<..., FUSED>)
                                             (32768, 3, 4, 5, False), (32768, 2, 3, 130, False), (32768, 3, 4, 4, True), (32768, 3, 4, 4, "seal")])
def test_vs_composition_of_the_oracles(hx, ctx, dev, orc, n, L, K, nb, strict):
    moduli = primes_below(orc, K, 1 << 52, n) if strict is True else None      # just below 2^52: the strict FP64 kernels
    if strict == "seal":
        moduli = seal_chain(orc, K, n)
    elif strict == "ladder":
        moduli = tier_ladder(orc, K, n)
    case = KsCase(orc, n, L, K, seed=40 + L, moduli=moduli)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    distinct = min(nb, 3)
    A = [operands(orc, case, b, 0) for b in range(distinct)]
    B = [operands(orc, case, b, 1) for b in range(distinct)]
    d_a = hx.as_i64(np.concatenate([A[b % distinct] for b in range(nb)])).to(dev)
    d_b = hx.as_i64(np.concatenate([B[b % distinct] for b in range(nb)])).to(dev)
    import torch
    d_out = torch.full((nb * 2 * L * n,), -1, dtype=torch.int64, device=dev)   # written, not accumulated into
    plan.multiply_relinearize(d_out, d_a, d_b, nb)
    ctx.sync()
    out = hx.to_u64(d_out).reshape(nb, -1)
    want = [composed(orc, case, A[b], B[b]) for b in range(distinct)]
    for b in range(nb):
        assert np.array_equal(out[b], want[b % distinct]), f"instance {b}"
    plan.close()


def test_rejects_what_it_does_not_cover(hx, ctx, dev, orc):
    n = 4096                                                     # moduli >= 2^52: the integer kernels have no fused pass
    case = KsCase(orc, n, 2, 3, seed=1, bits=55)
    plan = hx.KeySwitchPlan(ctx, n, 2, 3, 3, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    import torch
    z = torch.zeros(2 * 2 * n, dtype=torch.int64, device=dev)
    z2 = torch.zeros(2 * 2 * n, dtype=torch.int64, device=dev)
    z3 = torch.zeros(2 * 2 * n, dtype=torch.int64, device=dev)
    with pytest.raises(hx.HexlError):
        plan.multiply_relinearize(z, z2, z3, 1)
    plan.close()


def test_rejects_output_aliasing_an_operand(hx, ctx, dev, orc):
    """d_out is written while the operands are still being read (component 0 is stored before component 1's operands are
    loaded): overlap with d_a or d_b is refused instead of silently producing a wrong ciphertext"""
    n, L, K = 16384, 2, 3
    case = KsCase(orc, n, L, K, seed=3)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    import torch
    a = torch.zeros(2 * 2 * L * n, dtype=torch.int64, device=dev)
    b = torch.zeros(2 * L * n, dtype=torch.int64, device=dev)
    for out, x, y in ((a[:2 * L * n], a[:2 * L * n], b), (a[:2 * L * n], b, a[:2 * L * n]), (a[n:n + 2 * L * n], a[:2 * L * n], b)):
        with pytest.raises(hx.HexlError):
            plan.multiply_relinearize(out, x, y, 1)
    plan.multiply_relinearize(a[2 * L * n:], a[:2 * L * n], b, 1)       # adjacent, not overlapping: accepted
    ctx.sync()
    plan.close()
