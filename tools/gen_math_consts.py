#!/usr/bin/env python3
"""Derive the double constants used by the bit-reproducible transcendental
kernels (oracle/rt_oracle_math.h and the device copy in csrc/rt_math.h).

Everything is computed from first principles with exact rationals / 60-digit
decimals so the constants do not depend on any libm:
  * 1/n! Taylor coefficients for sin/cos,
  * 1/(2n+1) Taylor coefficients for atan,
  * pi/2 split into a 33-bit head and a tail (Cody-Waite),
  * atan(k/8), k=0..8.
Prints C hex-float literals.
"""
from fractions import Fraction
from decimal import Decimal, getcontext
import math, struct

getcontext().prec = 70

def dec_pi():
    # Machin: pi = 16 atan(1/5) - 4 atan(1/239)
    def atan_inv(n):
        x = Decimal(1) / n
        x2 = x * x
        term, s, k = x, x, 1
        while abs(term) > Decimal(10) ** -68:
            term = -term * x2
            k += 2
            s += term / k
        return s
    return 16 * atan_inv(5) - 4 * atan_inv(239)

def dec_atan(q):  # q Decimal in [0,1]; argument halving for convergence
    # atan(q) = 2 atan(q / (1 + sqrt(1+q^2)))
    n = 0
    while q > Decimal("0.1"):
        q = q / (1 + (1 + q * q).sqrt())
        n += 1
    x2 = q * q
    term, s, k = q, q, 1
    while abs(term) > Decimal(10) ** -68:
        term = -term * x2
        k += 2
        s += term / k
    return s * (2 ** n)

def to_double(d):  # correctly rounded Decimal/Fraction -> double
    if isinstance(d, Decimal):
        d = Fraction(d)
    return float(d)   # Fraction -> float is correctly rounded

def hexf(x):
    return float.hex(x)

pi = dec_pi()
pio2 = pi / 2
d = to_double(pio2)
bits = struct.unpack("<Q", struct.pack("<d", d))[0]
head = struct.unpack("<d", struct.pack("<Q", bits & ~((1 << 20) - 1)))[0]  # 33 significant bits
tail = to_double(pio2 - Decimal(head))
print("PIO2_HEAD ", hexf(head), repr(head))
print("PIO2_TAIL ", hexf(tail), repr(tail))
print("PI        ", hexf(to_double(pi)))
print("PIO2      ", hexf(to_double(pio2)))
print("PIO4      ", hexf(to_double(pi / 4)))
print("3PIO4     ", hexf(to_double(3 * pi / 4)))
print("TWO_OVER_PI", hexf(to_double(2 / pi)))
print("SIN coeffs (x^3..x^15):")
for n in range(3, 17, 2):
    c = Fraction((-1) ** (n // 2), math.factorial(n))
    print("  ", hexf(float(c)))
print("COS coeffs (x^2..x^16):")
for n in range(2, 18, 2):
    c = Fraction((-1) ** (n // 2), math.factorial(n))
    print("  ", hexf(float(c)))
print("ATAN coeffs (z^3..z^15):")
for n in range(3, 17, 2):
    c = Fraction((-1) ** (n // 2), n)
    print("  ", hexf(float(c)))
print("ATAN(k/8):")
for k in range(9):
    a = dec_atan(Decimal(k) / 8) if k else Decimal(0)
    v = to_double(a)
    assert abs(v - math.atan(k / 8)) <= 2.3e-16, (k, v, math.atan(k / 8))
    print("  ", hexf(v))
