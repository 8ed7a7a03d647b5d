#!/usr/bin/env python3
"""make_ref_literal.py -- fills the GENERATED regions of tools/ref_literal_jpeg.py: DCT_Upsample.P_Q!(R,C).calc and
R_S!(R,C).calc (jpegload.d:914-1072), and Row!(N).idct / Col!(N).idct (:156-292) -- every statement of every `static if`
branch -- transliterated MECHANICALLY, statement for statement, from the D text.

    python tools/make_ref_literal.py           rewrite the region (needs /root/reference)
    python tools/make_ref_literal.py --check   exit 1 if the committed region differs from a fresh transliteration

The transliteration is a handful of regular expressions (no arithmetic is re-derived, nothing is re-ordered):
    immutable Temp_Type X010 = EXPR;   ->   X010 = EXPR
    mixin(AT!(c, r))                   ->   AT(c, r)
    F!(0.415735f)                      ->   F(0.415735)
    P.at(r, c) = EXPR;                 ->   P.set(r, c, EXPR)
    // comment                         ->   # comment
Row / Col keep the D expression text as it stands (operators, parentheses, constants); what changes is syntax only:
    static if (C) { ... } else static if (C) { ... } else { ... }   ->   if C: ... elif C: ... else: ...
    immutable int a = E, b = F;  /  int i = E;  /  immutable ubyte v = E;   ->   a = E (one line per declarator)
    mixin(ACCESS_COL!2)                ->   ACCESS_COL(2)           (the template itself: a closure, see the preamble emitted below)
    cast(ubyte)CLAMP(i)                ->   to_ubyte(CLAMP(i))
pSrc / pTemp / pDst_ptr are pointer objects of the hand-written part (p[k] reads or writes element k of N blocks at once; a
short read is promoted to int as D does); int arithmetic is numpy int32: wrap-around, arithmetic >>.
Only /root/reference is read; only the regions between the marker lines are written.
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


BEGIN_RC, END_RC = "# BEGIN GENERATED ROWCOL (tools/make_ref_literal.py from jpegload.d)\n", "# END GENERATED ROWCOL\n"


def split_top(s):
    """split at commas outside parentheses / brackets"""
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "([" ; depth -= ch in ")]"
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def translate_idct(lines, first_no):
    """the body of `static void idct(...) { ... }` of Row / Col -> python statements (indented one level)"""
    out, ind, k = [], 1, 0
    def emit(t):
        out.append("    " * ind + t)
    while k < len(lines):
        s = lines[k].strip(); no = first_no + k; k += 1
        if not s:
            continue
        if s.startswith("//"):
            emit("#" + s[2:]); continue
        if s.startswith("template ACCESS_"):                    # the template: emitted as a closure in the preamble
            depth = s.count("{") - s.count("}")
            while depth:
                depth += lines[k].count("{") - lines[k].count("}"); k += 1
            continue
        m = re.fullmatch(r"static if \((.*)\) \{", s)
        if m:
            emit(f"if {m.group(1)}:"); ind += 1; emit("pass"); continue
        m = re.fullmatch(r"\} else static if \((.*)\) \{", s)
        if m:
            ind -= 1; emit(f"elif {m.group(1)}:"); ind += 1; emit("pass"); continue
        if s == "} else {":
            ind -= 1; emit("else:"); ind += 1; emit("pass"); continue
        if s == "}":
            ind -= 1; continue
        m = re.fullmatch(r"static assert\((.*)\);", s)
        if m:
            emit(f"assert {m.group(1)}"); continue
        if not s.endswith(";"):
            raise SystemExit(f"jpegload.d:{no}: statement not understood: {s}")
        s = s[:-1]
        s = re.sub(r"mixin\((ACCESS_(?:COL|ROW))!(\d)\)", r"\1(\2)", s)
        s = re.sub(r"cast\(ubyte\)(\w+\([^()]*\))", r"to_ubyte(\1)", s)
        if "cast(" in s or "mixin(" in s:
            raise SystemExit(f"jpegload.d:{no}: statement not understood: {s}")
        m = re.fullmatch(r"(?:immutable )?(?:int|ubyte) (.*)", s)
        for part in (split_top(m.group(1)) if m else [s]):
            if not re.fullmatch(r"[A-Za-z_]\w*(\[[^\]]*\])? = .*", part):
                raise SystemExit(f"jpegload.d:{no}: statement not understood: {part}")
            emit(part)
    if ind != 1:
        raise SystemExit(f"jpegload.d:{first_no}: unbalanced braces")
    return out


def idct_body(lines, header):
    i = next(k for k, l in enumerate(lines) if header in l)
    i = next(k for k in range(i, len(lines)) if "static void idct(" in lines[k])
    first = i + 1
    depth, k = 1, first
    while depth:
        depth += lines[k].count("{") - lines[k].count("}")
        k += 1
    return first + 1, lines[first:k - 1]


def generate_rowcol():
    lines = open(SRC).read().split("\n")
    text = [BEGIN_RC]
    for name, header, args, access in (
            ("Row_idct_d", "struct Row(int NONZERO_COLS) {", "NONZERO_COLS, pTemp, pSrc",
             "    ACCESS_COL = lambda x: pSrc[x] if x < NONZERO_COLS else 0      # template ACCESS_COL: \"cast(int)pSrc[x]\" or \"0\"\n"),
            ("Col_idct_d", "struct Col (int NONZERO_ROWS) {", "NONZERO_ROWS, pDst_ptr, pTemp",
             "    ACCESS_ROW = lambda x: pTemp[x * 8] if x < NONZERO_ROWS else 0  # template ACCESS_ROW: \"pTemp[x*8]\" or \"0\"\n")):
        first_no, body = idct_body(lines, header)
        text.append(f"def {name}({args}):          # jpegload.d:{first_no}-{first_no + len(body) - 1}\n")
        text.append(access)
        text += [l + "\n" for l in translate_idct(body, first_no)]
        text.append("\n\n")
    text[-1] = "\n"
    text.append(END_RC)
    return "".join(text)


def main():
    cur = open(DST).read()
    regions = ((BEGIN, END, generate()), (BEGIN_RC, END_RC, generate_rowcol()))
    if "--check" in sys.argv:
        for b, e, new in regions:
            if cur[cur.index(b):cur.index(e) + len(e)] != new:
                raise SystemExit(f"tools/ref_literal_jpeg.py: the region after {b.strip()!r} differs from a fresh transliteration of jpegload.d")
        print("ref_literal_jpeg.py: generated regions are current")
        return
    for b, e, new in regions:
        cur = cur[:cur.index(b)] + new + cur[cur.index(e) + len(e):]
        print(f"{b.split()[2]}: wrote {new.count(chr(10))} lines")
    open(DST, "w").write(cur)


if __name__ == "__main__":
    main()
