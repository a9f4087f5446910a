"""Prints the coefficient tables of the portable acos / cos used by FastEigen3x3 in BOTH the oracle (oracle/o3d_oracle.c)
and the device code (open3d_slam_amd/csrc/det_math.hpp): exact rationals rounded to the nearest double, as C hex floats.
asin(s) = s + s*z*P(z), z = s^2, P(z) = sum_{k>=1} c_k z^(k-1), c_k = (2k)! / (4^k (k!)^2 (2k+1));  cos / sin: Taylor in w = y^2."""
from fractions import Fraction
from math import factorial


def nearest(fr: Fraction) -> float:
    return fr.numerator / fr.denominator  # int / int is correctly rounded in CPython


def table(name, vals):
    print(f"static const double {name}[{len(vals)}] = {{")
    for v in vals:
        print(f"    {float.hex(nearest(v))},  /* {nearest(v):.17g} */")
    print("};")


asin_c = [Fraction(factorial(2 * k), 4**k * factorial(k) ** 2 * (2 * k + 1)) for k in range(1, 28)]
cos_c = [Fraction((-1) ** k, factorial(2 * k)) for k in range(1, 12)]
sin_c = [Fraction((-1) ** k, factorial(2 * k + 1)) for k in range(1, 11)]
table("kAsinC", asin_c)
table("kCosC", cos_c)
table("kSinC", sin_c)
# pi/2 and pi as hi + lo (hi = nearest double, lo = nearest double of the remainder), from a 60-digit pi
PI = Fraction("3.141592653589793238462643383279502884197169399375105820974944")
for nm, v in (("PIO2", PI / 2), ("PI", PI), ("PIO4", PI / 4), ("PI3O4", 3 * PI / 4)):
    hi = nearest(v)
    lo = nearest(v - Fraction(hi))
    print(f"/* {nm} */ hi = {float.hex(hi)}, lo = {float.hex(lo)}")
