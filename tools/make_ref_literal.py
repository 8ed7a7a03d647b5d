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


BEGIN_MISC, END_MISC = "# BEGIN GENERATED MISC (tools/make_ref_literal.py from jpegload.d)\n", "# END GENERATED MISC\n"


def region(lines, start_pat, end_pat, after=0):
    """1-based line numbers [a, b] of the first line matching start_pat at or after `after` .. the next line matching end_pat"""
    a = next(k for k in range(after, len(lines)) if re.search(start_pat, lines[k]))
    b = next(k for k in range(a, len(lines)) if re.search(end_pat, lines[k]))
    return a, b


def table(lines, name):
    """`static immutable T[N] name = [ ... ];` -> python list text (numbers only)"""
    a, b = region(lines, r"static immutable \w+\[\d+\] " + name + r" = \[", r"\];")
    text = " ".join(lines[a:b + 1])
    body = text[text.index("= [") + 3:text.rindex("]")]
    vals = [int(v) for v in body.replace("\n", " ").split(",") if v.strip()]
    return a + 1, b + 1, vals


def expr(s):
    """D expression text -> python: casts and template instantiations only; operators, parentheses and constants stay"""
    s = re.sub(r"FIX!\((-?[0-9.]+)f\)", r"FIX(\1)", s)
    s = re.sub(r"cast\(jpgd_block_t\)", "to_short", s)
    s = re.sub(r"cast\(__m128i?\*?\)\s*", "", s)
    s = re.sub(r"\.ptr\b", "", s)
    return s


def generate_misc():
    L = open(SRC).read().split("\n")
    out = [BEGIN_MISC]
    def emit(t=""):
        out.append(t + "\n")
    # ---- enums and tables: data, copied by the machine
    a, b = region(L, r"^enum CONST_BITS", r"^enum FIX_3_072711026")
    emit(f"# jpegload.d:{a + 1}-{b + 1}")
    for l in L[a:b + 1]:
        m = re.fullmatch(r"enum (\w+) = (?:cast\(int\))?(\d+);.*", l.strip())
        if m:
            emit(f"{m.group(1)} = {m.group(2)}")
        elif l.strip():
            raise SystemExit(f"enum line not understood: {l}")
    for name in ("g_ZAG", "s_idct_row_table", "s_idct_col_table", "s_max_rc"):
        a, b, vals = table(L, name)
        emit(f"{name} = {vals}          # jpegload.d:{a}-{b}")
    a, b = region(L, r"enum SCALEBITS = 16;", r"enum ONE_HALF")
    emit(f"SCALEBITS = 16          # jpegload.d:{a + 1}")
    m = re.fullmatch(r"enum ONE_HALF = \(cast\(int\) 1 << \(SCALEBITS-1\)\);", L[b].strip())
    if not m:
        raise SystemExit("ONE_HALF not understood")
    emit(f"ONE_HALF = 1 << (SCALEBITS-1)          # jpegload.d:{b + 1}")
    emit()
    # ---- idct(): the DC-only shortcut (the four statements on k)
    a, b = region(L, r"^void idct\(\)", r"for \(int i = 8; i > 0; i--\)")
    emit(f"def idct_dc_d(pSrc_ptr):          # jpegload.d:{a + 1}-{b + 1}")
    n = 0
    for l in L[a:b]:
        t = l.strip()
        m = re.fullmatch(r"(?:int )?k = (.*);", t)
        if m:
            emit(f"    k = to_int({m.group(1)})"); n += 1                # k is declared `int`
    if n != 4:
        raise SystemExit("idct DC shortcut: expected 4 statements on k")
    emit("    return k")
    emit()
    # ---- Matrix44: the six operator / store bodies
    for dname, pyname, sig in ((r'opOpAssign\(string op:"\+"\)', "Matrix44_iadd_d", "this, a"), (r'opOpAssign\(string op:"-"\)', "Matrix44_isub_d", "this, a"),
                               (r'opBinary\(string op:"\+"\)', "Matrix44_add_d", "this, b, ret"), (r'opBinary\(string op:"-"\)', "Matrix44_sub_d", "this, b, ret"),
                               (r"static void add_and_store\(\)", "add_and_store_d", "pDst, a, b"), (r"static void sub_and_store\(\)", "sub_and_store_d", "pDst, a, b")):
        a, _ = region(L, dname, dname)
        depth, k = L[a].count("{") - L[a].count("}"), a + 1
        while depth:
            depth += L[k].count("{") - L[k].count("}"); k += 1
        emit(f"def {pyname}({sig}):          # jpegload.d:{a + 1}-{k}")
        ind = 1
        for no in range(a + 1, k - 1):
            t = L[no].strip()
            if not t or t in ("alias a = this;", "Matrix44 ret;", "return this;", "return ret;"):
                if t == "alias a = this;":
                    emit("    a = this")
                continue
            m = re.fullmatch(r"foreach \(int r; 0\.\.(\w+)\) \{", t)
            if m:
                emit("    " * ind + f"for r in range({m.group(1)}):"); ind += 1; continue
            if t == "}":
                ind -= 1; continue
            m = re.fullmatch(r"at\((\w), (\d)\) ([+-])= (.*);", t)
            if m:
                emit("    " * ind + f"this.set({m.group(1)}, {m.group(2)}, this.at({m.group(1)}, {m.group(2)}) {m.group(3)} {m.group(4)})"); continue
            m = re.fullmatch(r"ret\.at\((\w), (\d)\) = (.*);", t)
            if m:
                emit("    " * ind + f"ret.set({m.group(1)}, {m.group(2)}, {m.group(3)})"); continue
            m = re.fullmatch(r"(pDst\[[^\]]*\]) = (.*);", t)
            if m:
                emit("    " * ind + f"{m.group(1)} = {expr(m.group(2))}"); continue
            raise SystemExit(f"jpegload.d:{no + 1}: Matrix44 statement not understood: {t}")
        emit()
    # ---- create_look_ups: the loop body
    a, b = region(L, r"void create_look_ups \(\)", r"^  \}")
    emit(f"def create_look_ups_body_d(m_crr, m_cbb, m_crg, m_cbg, i):          # jpegload.d:{a + 1}-{b + 1}")
    n = 0
    for no in range(a, b):
        t = L[no].strip()
        m = re.fullmatch(r"int k = (.*);", t)
        if m:
            emit(f"    k = {m.group(1)}"); n += 1; continue
        m = re.fullmatch(r"(m_\w+)\.ptr\[i\] = (.*);", t)
        if m:
            emit(f"    {m.group(1)}[i] = {expr(m.group(2))}"); n += 1
    if n != 5:
        raise SystemExit("create_look_ups: expected 5 statements")
    emit()
    # ---- transform_mcu_expand: from `auto a = ...` to the last idct_4x4
    t0, _ = region(L, r"void transform_mcu_expand \(int mcu_row\)", r".")
    a, _ = region(L, r"auto a = DCT_Upsample\.Matrix44\(P \+ Q\);", r".", t0)
    b, _ = region(L, r"pSrc_ptr \+= 64;", r".", a)
    emit(f"def mcu_expand_tail_d(P, Q, R, S, temp_block, pDst_ptr, idct_4x4):          # jpegload.d:{a + 1}-{b}")
    for no in range(a, b):
        t = L[no].strip()
        if not t:
            continue
        t = t.rstrip(";")
        t = t.replace("DCT_Upsample.Matrix44.", "Matrix44.").replace("DCT_Upsample.Matrix44(", "Matrix44(").replace("temp_block.ptr", "temp_block")
        t = re.sub(r"^auto (\w) = ", r"\1 = ", t)
        m = re.fullmatch(r"DCT_Upsample\.Matrix44\* (\w) = &(\w)", t)
        if m:
            t = f"{m.group(1)} = {m.group(2)}"
        t = t.replace("*b", "b").replace("*d", "d")
        if not re.fullmatch(r"(\w = Matrix44\(\w \+ \w\)|\w -= \w|\w = \w|Matrix44\.(add|sub)_and_store\(temp_block, \w, \w\)|idct_4x4\(temp_block, pDst_ptr\)|pDst_ptr \+= 64)", t):
            raise SystemExit(f"jpegload.d:{no + 1}: transform_mcu_expand statement not understood: {t}")
        emit("    " + t)
    emit()
    # ---- expanded_convert: the body of the `for (int j ...)` loop (the SSE sequence)
    t0, _ = region(L, r"void expanded_convert \(\)", r".")
    a, _ = region(L, r"for \(int j = 0; j \+ 3 < 8; j \+= 4\)", r".", t0)
    b, _ = region(L, r"d \+= 16;", r".", a)
    emit(f"def expanded_convert_simd_d(Py, Y_ofs, Cb_ofs, Cr_ofs, j, d):          # jpegload.d:{a + 3}-{b + 1}")
    for no in range(a + 2, b + 1):
        t = L[no].strip()
        if not t or t.startswith("//"):
            continue
        t = expr(t.rstrip(";"))
        t = re.sub(r"^__m128i? (\w+)\s*= ", r"\1 = ", t)
        t = re.sub(r"&(\w+)\[([^\]]*)\]", r"\1, \2", t)                     # &Py[Y_ofs + j] -> Py, Y_ofs + j
        t = re.sub(r"\s+\(", "(", t)
        t = re.sub(r"\s+=", " =", t)
        if t.startswith("_MM_TRANSPOSE4_PS("):
            args = t[len("_MM_TRANSPOSE4_PS("):-1]
            t = f"{args} = _MM_TRANSPOSE4_PS({args})"
        if not re.fullmatch(r"(\w+ = .*|\w+ \+= \w+|_mm_storeu_si128\(.*\)|[\w, ]+ = _MM_TRANSPOSE4_PS\(.*\))", t) or "cast(" in t:
            raise SystemExit(f"jpegload.d:{no + 1}: expanded_convert statement not understood: {t}")
        emit("    " + t)
    emit("    return d")
    out.append(END_MISC)
    return "".join(out)


def main():
    cur = open(DST).read()
    regions = ((BEGIN, END, generate()), (BEGIN_RC, END_RC, generate_rowcol()), (BEGIN_MISC, END_MISC, generate_misc()))
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
