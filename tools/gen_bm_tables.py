"""Writes the Box-Muller tables into the device header and the oracle's math header, between their
`BEGIN BM TABLES` / `END BM TABLES` markers:  python tools/gen_bm_tables.py [--check]

  log table, 64 entries over the offset mantissa of Cephes' reduction (bit patterns 0x3f3504f3 + k * 2^17 ...):
      c_k = the float at the interval's middle pattern (1.0 in the interval that holds it),
      entry = (fl(1 / c_k), fl(-log(fl(1 / c_k))))      -- log c_k consistent with the ROUNDED reciprocal
  angle table, 256 entries: (fl(cos A_k), fl(sin A_k)), A_k = 2 pi (k + 1/2) / 256
All values are computed in float64 and rounded once to float32; the headers hold their bit patterns."""
import math
import os
import re
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGETS = [os.path.join(ROOT, "genjax-chi_amd", "csrc", "gjx_device.hpp"), os.path.join(ROOT, "oracle", "gjx_oracle_math.h")]
C0 = 0x3F3504F3


def f32(x: float) -> float:
    return struct.unpack("<f", struct.pack("<f", x))[0]


def bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def from_bits(b: int) -> float:
    return struct.unpack("<f", struct.pack("<I", b))[0]


def log_table():
    out = []
    for k in range(64):
        lo = C0 + (k << 17)
        hi = lo + (1 << 17) - 1
        c = 1.0 if lo <= 0x3F800000 <= hi else from_bits(lo + (1 << 16))
        inv = f32(1.0 / c)
        out.append((bits(inv), bits(f32(-math.log(inv)))))
    return out


def angle_table():
    out = []
    for k in range(256):
        a = 2.0 * math.pi * (k + 0.5) / 256.0
        out.append((bits(f32(math.cos(a))), bits(f32(math.sin(a)))))
    return out


def block() -> str:
    def init(name, rows):
        lines = [f"#define {name} {{ \\"]
        for i in range(0, len(rows), 4):
            lines.append("  " + " ".join(f"{{0x{a:08x}u, 0x{b:08x}u}}," for a, b in rows[i:i + 4]) + " \\")
        lines.append("}")
        return "\n".join(lines)
    return init("GJX_BM_LG_INIT", log_table()) + "\n" + init("GJX_BM_CS_INIT", angle_table()) + "\n"


def main():
    check = "--check" in sys.argv
    body = block()
    rc = 0
    for path in TARGETS:
        src = open(path).read()
        m = re.search(r"(BEGIN BM TABLES[^\n]*\n)(.*?)([^\n]*END BM TABLES)", src, re.S)
        if not m:
            raise SystemExit(f"{path}: no BM TABLES markers")
        if m.group(2) != body:
            if check:
                print(f"{path}: tables differ from the generator's")
                rc = 1
            else:
                open(path, "w").write(src[:m.start(2)] + body + src[m.end(2):])
                print(f"{path}: tables written")
    return rc


if __name__ == "__main__":
    sys.exit(main())
