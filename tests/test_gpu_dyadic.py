"""GPU parity: hexl_dyadic_multiply vs the oracle and vs the reference test's inline model
(tests/test_dyadic_multiply.cpp:32-85: toy non-prime moduli (b+m+1)*10, operands >> 4q)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ref_style_io(num, nm, n):
    mod, a, b = [], [], []
    for bb in range(num):
        mod += [(bb + m + 1) * 10 for m in range(nm)]
        i = np.arange(n, dtype=np.uint64)
        a += [bb + i + 1 + m * n for m in range(nm)] + [bb + i + 11 + m * n for m in range(nm)]
        b += [bb + i + 2 + m * n for m in range(nm)] + [bb + i + 22 + m * n for m in range(nm)]
    return (np.array(mod, dtype=np.uint64), np.concatenate(a).astype(np.uint64), np.concatenate(b).astype(np.uint64))


def gpu_dyadic(hx, ctx, dev, a, b, mod, n, nm):
    batch = len(a) // (2 * nm * n)
    import torch
    out = torch.empty(batch * 3 * nm * n, dtype=torch.int64, device=dev)
    ctx.dyadic_multiply(out, hx.as_i64(a).to(dev), hx.as_i64(b).to(dev), hx.as_i64(mod).to(dev), n, nm)
    ctx.sync()
    return hx.to_u64(out)


@pytest.mark.parametrize("n,nm,num", [(256, 1, 16), (255, 2, 3), (512, 1, 2), (1024, 2, 16), (4096, 7, 3), (16384, 14, 2), (32768, 4, 2)])
def test_reference_toy_moduli(hx, ctx, dev, orc, n, nm, num):
    mod, a, b = ref_style_io(num, nm, n)
    got = gpu_dyadic(hx, ctx, dev, a, b, mod, n, nm).reshape(num, 3, nm, n)
    A, B = a.reshape(num, 2, nm, n).astype(object), b.reshape(num, 2, nm, n).astype(object)
    M = mod.reshape(num, nm).astype(object)[:, :, None]
    assert np.array_equal(got[:, 0].astype(object), (A[:, 0] * B[:, 0]) % M)
    assert np.array_equal(got[:, 1].astype(object), (A[:, 0] * B[:, 1] + A[:, 1] * B[:, 0]) % M)
    assert np.array_equal(got[:, 2].astype(object), (A[:, 1] * B[:, 1]) % M)


@pytest.mark.parametrize("bits", [30, 52, 61])
def test_prime_moduli_vs_oracle(hx, ctx, dev, orc, bits):
    n, nm, batch = 8192, 4, 3
    mod1 = np.array(orc.primes(nm, bits, n), dtype=np.uint64)
    rng = np.random.default_rng(bits)
    a = np.concatenate([rng.integers(0, int(m), size=n, dtype=np.uint64) for _ in range(batch) for _p in range(2) for m in mod1])
    b = np.concatenate([rng.integers(0, int(m), size=n, dtype=np.uint64) for _ in range(batch) for _p in range(2) for m in mod1])
    mod = np.tile(mod1, batch)
    got = gpu_dyadic(hx, ctx, dev, a, b, mod, n, nm).reshape(batch, -1)
    for k in range(batch):
        sl = slice(k * 2 * nm * n, (k + 1) * 2 * nm * n)
        exact = orc.dyadic(a[sl], b[sl], n, mod1, exact=True)
        assert np.array_equal(got[k], exact)
        assert np.array_equal(exact, orc.dyadic(a[sl], b[sl], n, mod1, exact=False))   # reference MultMod, in domain


def test_moduli_between_2p61_and_2p62(hx, ctx, dev, orc):
    """Round 6 (tools/soak_dyadic_random.py): the Barrett quotient estimate of mod_ops.hpp:49-83 can fall short by TWO for a modulus of 62
    bits -- one operand pair in ~10^4 came back in [q, 2q) with the reference's single conditional subtraction. Moduli all over
    [2^61, 2^62) (primes, even values, 2^62 - 1), operands uniform below q, at q - 1 and anywhere in 64 bits: the mathematical result."""
    n, nm = 16384, 6                                              # (~10 words of these 300 k differed before the fix)
    rng = np.random.default_rng(62)
    mod = np.array([(1 << 62) - 1, (1 << 61), (1 << 61) + 1, 4475467519117804091, 3503029193451124011, orc.primes(1, 62, n)[0]], dtype=np.uint64)
    for kind in range(3):
        if kind == 0:
            a = np.concatenate([rng.integers(0, int(m), size=n, dtype=np.uint64) for _p in range(2) for m in mod])
            b = np.concatenate([rng.integers(0, int(m), size=n, dtype=np.uint64) for _p in range(2) for m in mod])
        elif kind == 1:
            a = np.concatenate([np.full(n, int(m) - 1, dtype=np.uint64) - rng.integers(0, 3, size=n, dtype=np.uint64) for _p in range(2) for m in mod])
            b = np.concatenate([np.full(n, int(m) - 1, dtype=np.uint64) - rng.integers(0, 3, size=n, dtype=np.uint64) for _p in range(2) for m in mod])
        else:
            a = rng.integers(0, 2**64 - 1, size=2 * nm * n, dtype=np.uint64)
            b = rng.integers(0, 2**64 - 1, size=2 * nm * n, dtype=np.uint64)
        got = gpu_dyadic(hx, ctx, dev, a, b, mod, n, nm).reshape(3, nm, n).astype(object)
        A, B, M = a.reshape(2, nm, n).astype(object), b.reshape(2, nm, n).astype(object), mod.astype(object)[:, None]
        assert np.array_equal(got[0], (A[0] * B[0]) % M) and np.array_equal(got[2], (A[1] * B[1]) % M), f"operand kind {kind}"
        assert np.array_equal(got[1], (A[0] * B[1] + A[1] * B[0]) % M), f"operand kind {kind}"
        assert np.array_equal(got.astype(np.uint64).reshape(-1), orc.dyadic(a, b, n, mod, exact=True))


def test_arbitrary_64bit_operands(hx, ctx, dev, orc):
    n, nm = 2048, 3
    mod = np.array([3, 2**61 - 1, 1000003], dtype=np.uint64)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**64 - 1, size=2 * nm * n, dtype=np.uint64)
    b = rng.integers(0, 2**64 - 1, size=2 * nm * n, dtype=np.uint64)
    assert np.array_equal(gpu_dyadic(hx, ctx, dev, a, b, mod, n, nm), orc.dyadic(a, b, n, mod, exact=True))


def test_full_baseline_config3(hx, ctx, dev, orc):
    """BASELINE config 3 at full size: n = 8192, 4 RNS moduli (GeneratePrimes(4, 52, 8192)), batch 4096 ciphertext pairs
    generated on the device. Spot items against the oracle, every output word below its modulus, and the symmetric
    components agree when the operands are swapped (out0, out2 unchanged; out1 unchanged)."""
    import torch
    n, nm, batch = 8192, 4, 4096
    mod1 = np.array(orc.primes(nm, 52, n), dtype=np.uint64)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    a = torch.empty((batch, 2, nm, n), dtype=torch.int64, device=dev)
    b = torch.empty((batch, 2, nm, n), dtype=torch.int64, device=dev)
    for m in range(nm):
        a[:, :, m].random_(0, int(mod1[m]), generator=g)
        b[:, :, m].random_(0, int(mod1[m]), generator=g)
    mod = hx.as_i64(np.tile(mod1, batch)).to(dev)
    out = torch.empty((batch, 3, nm, n), dtype=torch.int64, device=dev)
    ctx.dyadic_multiply(out.reshape(-1), a.reshape(-1), b.reshape(-1), mod, n, nm)
    ctx.sync()
    for k in (0, 1, 2047, 4095):
        want = orc.dyadic(hx.to_u64(a[k]).reshape(-1).copy(), hx.to_u64(b[k]).reshape(-1).copy(), n, mod1, exact=True)
        assert np.array_equal(hx.to_u64(out[k]).reshape(-1), want), f"item {k}"
    q = torch.tensor([int(v) for v in mod1], dtype=torch.int64, device=dev).view(1, 1, nm, 1)
    assert bool(((out >= 0) & (out < q)).all())
    out2 = torch.empty_like(out)
    ctx.dyadic_multiply(out2.reshape(-1), b.reshape(-1), a.reshape(-1), mod, n, nm)
    ctx.sync()
    assert torch.equal(out, out2)
