"""Generates the constants of raytracer_amd/csrc/rtx_math.h (the fp32 math routines shared by the HIP kernels and the
CPU oracle) with 60-digit decimal arithmetic:  python tools/gen_math_tables.py > /tmp/tables.txt"""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60
LN2 = Decimal(2).ln()
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")


def f32(x):
    return struct.unpack("<f", struct.pack("<f", float(x)))[0]


def hexd(x):
    return float(x).hex()


def log2(x):
    return Decimal(x).ln() / LN2


OFF = 0x3F330000
print("// log2 table: interval i covers float bit patterns [OFF + i*2^19, OFF + (i+1)*2^19), OFF = 0x3f330000")
print("static RTX_MATH_CONST double rtx_pow_invc[16] = {")
rows_inv, rows_log = [], []
for i in range(16):
    lo = struct.unpack("<f", struct.pack("<I", OFF + i * (1 << 19)))[0]
    hi = struct.unpack("<f", struct.pack("<I", OFF + (i + 1) * (1 << 19)))[0]
    c = Decimal(lo + hi) / 2
    if lo <= 1.0 < hi:
        c = Decimal(1)
    rows_inv.append(hexd(Decimal(1) / c))
    rows_log.append(hexd(log2(c)))
print("    " + ", ".join(rows_inv) + "};")
print("static RTX_MATH_CONST double rtx_pow_logc[16] = {")
print("    " + ", ".join(rows_log) + "};")
print("// log2(1+z) = z*(L1 + z*(L2 + ...)): Taylor coefficients (-1)^(n+1) / (n ln 2)")
print("L:", ", ".join(hexd(Decimal((-1) ** (n + 1)) / (Decimal(n) * LN2)) for n in range(1, 10)))
print("// 2^r = 1 + r*(E1 + r*(E2 + ...)): (ln 2)^n / n!")
fact = Decimal(1)
es = []
for n in range(1, 12):
    fact *= n
    es.append(hexd(LN2 ** n / fact))
print("E:", ", ".join(es))
print("log2e (double):", hexd(Decimal(1) / LN2))
# pi/2 in three float pieces for Cody-Waite reduction with fmaf
p1 = f32(PI / 2)
p2 = f32(PI / 2 - Decimal(p1))
p3 = f32(PI / 2 - Decimal(p1) - Decimal(p2))
print("pio2 pieces:", float(p1).hex(), float(p2).hex(), float(p3).hex(), " 2/pi:", float(f32(Decimal(2) / PI)).hex())
print("pi f32:", float(f32(PI)).hex(), "pi/2:", float(f32(PI / 2)).hex(), "pi/4:", float(f32(PI / 4)).hex())
