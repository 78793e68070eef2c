"""The 4-column table of glibc's double sin / cos (sysdeps/ieee754/dbl-64/sincostab.c: for x_k = k / 128 the values
sin x_k and cos x_k, each as a double-double {high, low}), recomputed from the Taylor series in exact rational arithmetic.
Prints the rows as C initialisers; hprt_math.h (det_sincos_glibc_d) and oracle/orc_math.h (det::sincos_glibc_d) carry the
output, and tools/debug/sin_cos_double_exhaustive.cpp checks the functions built on it against the libm of the machine."""
from fractions import Fraction


def sincos(x):
    s = Fraction(0); c = Fraction(0); term = Fraction(1)
    for n in range(60):                      # term = x^n / n!
        sign = 1 if (n // 2) % 2 == 0 else -1
        if n % 2 == 0: c += sign * term
        else: s += sign * term
        term = term * x / (n + 1)
    return s, c


for k in range(112):                         # |x| < 0.8555 after the reductions: k <= 110
    s, c = sincos(Fraction(k, 128))
    sh = float(s); sl = float(s - Fraction(sh)); ch = float(c); cl = float(c - Fraction(ch))
    print("    {" + ", ".join(v.hex() for v in (sh, sl, ch, cl)) + "},")
