"""The kernels' division by a reciprocal (div_by_reciprocal in ray_tracing_amd/csrc/mrt_device.hip: camera
coordinates, camera.rs:88-106; sphere texture coordinates, sphere.rs:111-115) must give what the reference's `/`
gives: the correctly rounded quotient. The sequence is restated here with exact rational arithmetic (a fused
multiply-add is one rounding of the exact a*b + c) and checked against IEEE division on random operands and on the
operands that are hardest for it: divisors whose significand is nearly all ones or has few bits, numerators at
the top of a binade."""
import math
import random
import struct
from fractions import Fraction


def fma(a: float, b: float, c: float) -> float:
    if math.isnan(a) or math.isnan(b) or math.isnan(c):
        return math.nan
    return float(Fraction(a) * Fraction(b) + Fraction(c))  # Fraction -> float rounds to nearest even


def div_by_reciprocal(a: float, d: float, inv: float, corrections: int = 2) -> float:
    q = a * inv
    for _ in range(corrections):
        q = fma(fma(-d, q, a), inv, q)
    return q


def reciprocal_for_division(d: float) -> float:
    """The condition under which the sequence is the division (a normal divisor of moderate size whose significand
    is not all ones). The library uses it inside that condition only: whole-number divisors from 1 to 1e15
    (reciprocal_for_div_by, NaN otherwise, which sends the kernel to its `/`) and the constants pi and 2 pi."""
    m = abs(d)
    bits = struct.unpack("<Q", struct.pack("<d", m))[0]
    all_ones = (bits & 0xFFFFFFFFFFFFF) == 0xFFFFFFFFFFFFF
    return 1.0 / d if (2.0 ** -100 <= m <= 2.0 ** 100 and not all_ones) else math.nan


def from_parts(sign: int, exponent: int, significand: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", (sign << 63) | ((exponent + 1023) << 52) | significand))[0]


def operands(rng: random.Random, n: int):
    full = (1 << 52) - 1
    for i in range(n):
        kind_a, kind_d = rng.randrange(4), rng.randrange(4)

        def sig(kind):
            if kind == 0:
                return rng.getrandbits(52)
            if kind == 1:
                return full - rng.getrandbits(rng.randrange(1, 20))  # a run of leading ones
            if kind == 2:
                return rng.getrandbits(rng.randrange(1, 12))  # few significant bits
            return rng.getrandbits(52) | (full << 30) & full  # top of the binade
        yield (from_parts(rng.randrange(2), rng.randrange(-60, 60), sig(kind_a)),
               from_parts(rng.randrange(2), rng.randrange(-60, 60), sig(kind_d)))


def test_two_corrections_give_the_ieee_quotient():
    rng = random.Random(20240607)
    checked = 0
    for a, d in operands(rng, 60000):
        inv = reciprocal_for_division(d)
        if math.isnan(inv):
            continue
        checked += 1
        assert div_by_reciprocal(a, d, inv) == a / d, (a.hex(), d.hex())
    assert checked > 50000


def test_the_divisors_the_kernels_use():
    rng = random.Random(7)
    # camera: (pixel + jitter) / (extent - 1); texture coordinates: angle / pi, angle / (2 pi)
    for extent in (2, 3, 4, 100, 450, 800, 1080, 1920, 2160, 3840, 65536):
        d = float(extent - 1)
        inv = reciprocal_for_division(d)
        assert not math.isnan(inv)
        for _ in range(300):
            a = float(rng.randrange(extent)) + rng.random()
            assert div_by_reciprocal(a, d, inv) == a / d
        assert div_by_reciprocal(0.0, d, inv) == 0.0
    for d in (math.pi, 2.0 * math.pi):
        inv = reciprocal_for_division(d)
        assert inv == 1.0 / d
        for _ in range(3000):
            a = rng.random() * d
            assert div_by_reciprocal(a, d, inv) == a / d
        for a in (0.0, d, math.nextafter(d, 0.0), 5e-324 * 2 ** 200, 1e-9):
            assert div_by_reciprocal(a, d, inv) == a / d
    assert math.isnan(div_by_reciprocal(math.nan, math.pi, 1.0 / math.pi))  # NaN stays NaN


def test_divisors_outside_the_condition_get_no_reciprocal():
    for d in (0.0, -0.0, math.inf, math.nan, 5e-324, 2.0 ** -101, 2.0 ** 101, from_parts(0, 0, (1 << 52) - 1),
              from_parts(1, 7, (1 << 52) - 1)):
        assert math.isnan(reciprocal_for_division(d))
    for d in (1.0, -3.0, 2.0 ** -100, 2.0 ** 100, from_parts(0, 0, (1 << 52) - 2)):
        assert reciprocal_for_division(d) == 1.0 / d


def test_the_first_product_alone_is_not_the_quotient():
    """Why there are corrections at all: a * (1 / d) differs from a / d in a large share of the cases."""
    rng = random.Random(3)
    wrong = 0
    for a, d in operands(rng, 4000):
        inv = reciprocal_for_division(d)
        if not math.isnan(inv) and a * inv != a / d:
            wrong += 1
    assert wrong > 200
