#!/usr/bin/env python3
"""make_ref_literal.py -- fills the GENERATED region of tools/ref_literal_jpeg.py: DCT_Upsample.P_Q!(R,C).calc and
R_S!(R,C).calc (jpegload.d:914-1072), transliterated MECHANICALLY, statement for statement, from the D text.

    python tools/make_ref_literal.py           rewrite the region (needs /root/reference)
    python tools/make_ref_literal.py --check   exit 1 if the committed region differs from a fresh transliteration

The transliteration is a handful of regular expressions (no arithmetic is re-derived, nothing is re-ordered):
    immutable Temp_Type X010 = EXPR;   ->   X010 = EXPR
    mixin(AT!(c, r))                   ->   AT(c, r)
    F!(0.415735f)                      ->   F(0.415735)
    P.at(r, c) = EXPR;                 ->   P.set(r, c, EXPR)
    // comment                         ->   # comment
Only /root/reference is read; only the region between the two marker lines is written.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/source/gamut/codecs/jpegload.d"
DST = os.path.join(ROOT, "tools", "ref_literal_jpeg.py")
BEGIN, END = "# BEGIN GENERATED (tools/make_ref_literal.py from jpegload.d)\n", "# END GENERATED\n"


def body_of(lines, header):
    """the statements of `static void calc(...) {` following the line that contains `header`"""
    i = next(k for k, l in enumerate(lines) if header in l)
    i = next(k for k in range(i, len(lines)) if "static void calc" in lines[k])
    first = i + 1
    depth, k = 1, first
    while depth:
        depth += lines[k].count("{") - lines[k].count("}")
        k += 1
    return first + 1, lines[first:k - 1]                       # 1-based number of the first body line


def translate(stmt_lines, first_no):
    out = []
    for n, l in enumerate(stmt_lines):
        s = l.strip()
        if not s:
            continue
        if s.startswith("//"):
            out.append("    #" + s[2:]); continue
        if s.startswith("template AT") or s.startswith("static if (c >= NUM_COLS") or s == "}":
            continue                                           # the AT template itself: _AT() in the hand-written part
        s = re.sub(r"mixin\(AT!\((\d+), (\d+)\)\)", r"AT(\1, \2)", s)
        s = re.sub(r"F!\((-?[0-9.]+)f\)", r"F(\1)", s)
        m = re.fullmatch(r"immutable Temp_Type (X\d+) = (.*);", s)
        if m:
            out.append(f"    {m.group(1)} = {m.group(2)}"); continue
        m = re.fullmatch(r"([PQRS])\.at\((\d), (\d)\) = (.*);", s)
        if m:
            out.append(f"    {m.group(1)}.set({m.group(2)}, {m.group(3)}, {m.group(4)})"); continue
        raise SystemExit(f"jpegload.d:{first_no + n}: statement not understood: {s}")
    return out


def generate():
    lines = open(SRC).read().split("\n")
    text = [BEGIN]
    for name, header, mats in (("P_Q_calc", "static struct P_Q(int NUM_ROWS, int NUM_COLS)", "P, Q"),
                               ("R_S_calc", "static struct R_S(int NUM_ROWS, int NUM_COLS)", "R, S")):
        first_no, body = body_of(lines, header)
        text.append(f"def {name}(NUM_ROWS, NUM_COLS, {mats}, pSrc):          # jpegload.d:{first_no}-{first_no + len(body) - 1}\n")
        text.append("    AT = _AT(pSrc, NUM_ROWS, NUM_COLS)\n")
        text += [l + "\n" for l in translate(body, first_no)]
        text.append("\n\n")
    text[-1] = "\n"
    text.append(END)
    return "".join(text)


def main():
    cur = open(DST).read()
    a, b = cur.index(BEGIN), cur.index(END) + len(END)
    new = generate()
    if "--check" in sys.argv:
        if cur[a:b] != new:
            raise SystemExit("tools/ref_literal_jpeg.py: GENERATED region differs from a fresh transliteration of jpegload.d")
        print("ref_literal_jpeg.py: generated region is current")
        return
    open(DST, "w").write(cur[:a] + new + cur[b:])
    print(f"wrote {new.count(chr(10))} lines")


if __name__ == "__main__":
    main()
