#!/usr/bin/env python3
"""d_scanline_exec.py -- runs the reference's OWN scanline conversion functions: a transpiler for the small D subset that
source/gamut/scanline.d:139-836 is written in (pointer declarations, one `for` over x, `*p++` loads / stores, indexed
loads / stores, float and integer casts, one `if (a != 0) b /= a;`, memcpy).  TEST INFRASTRUCTURE ONLY.

Every statement of every `scanline_convert_A_to_B` is translated mechanically -- no arithmetic is restated by hand -- into
numpy code that evaluates it for all x at once: the k-th `*p++` of a loop body that advances p n times is p[k::n];
`inp[4*x+3]` is inp[3::4]; float literals, variables and pixels are IEEE binary32 and every operation rounds on its own
(numpy float32 array arithmetic), which is what LDC / DMD emit for x86-64 (SSE scalar float code); `cast(ubyte)(e)` is
truncation toward zero followed by the low 8 bits (cvttss2si + byte store).  The expression text itself (operator order,
parentheses) is the reference's.

Used (a) by tools/make_scanline_vectors.py to produce tests/golden/scanline_ref.npz -- input pixels and the outputs THE
REFERENCE'S TEXT gives for them, the golden vectors that pin oracle/oracle_convert.c -- and (b) by
tests/test_oracle_pinning.py, when /root/reference is present, to re-check those vectors and run larger sweeps.
"""
import re

import numpy as np

SRC = "/root/reference/source/gamut/scanline.d"
DT = {"ubyte": np.uint8, "ushort": np.uint16, "float": np.float32}


class Unsupported(Exception):
    pass


def _functions(text):
    """name -> list of body lines (without the outer braces)"""
    out = {}
    for m in re.finditer(r"^void (scanline_convert_\w+)\(const\(ubyte\)\* inScan, ubyte\* outScan, int width, void\* userData = null\)\s*\n\{\n", text, re.M):
        i, depth = m.end(), 1
        j = i
        while depth:
            c = text[j]
            depth += (c == "{") - (c == "}")
            j += 1
        out[m.group(1)] = [l.strip() for l in text[i:j - 1].split("\n")]
    return out


def _expr(e, ptrs, nload, counters):
    """one D expression -> numpy expression text"""
    def lit(m):
        return f"F32({m.group(1)})"
    e = re.sub(r"(?<![\w.])(\d+\.\d+)f", lit, e)

    def index(m):                                                # p[4*x+3], p[x*2+1], p[x], p[4*x]
        p, a = m.group(1), m.group(2).replace(" ", "")
        mm = re.fullmatch(r"(?:(\d+)\*x|x\*(\d+)|x)(?:\+(\d+))?", a)
        if not mm:
            raise Unsupported(f"index {a}")
        stride = int(mm.group(1) or mm.group(2) or 1); off = int(mm.group(3) or 0)
        return f"{p}[{off}::{stride}][:width]"
    e = re.sub(r"\b(\w+)\[([^\]]+)\]", index, e)
    def postinc(m):                                              # *p++ as an rvalue
        p = m.group(1)
        k = counters.setdefault(p, 0); counters[p] += 1
        return f"{p}[{k}::{nload[p]}][:width]"
    e = re.sub(r"\*(\w+)\+\+", postinc, e)

    return e


def transpile(name, body):
    """-> python source of def name(inScan, outScan, width) operating on uint8 numpy buffers"""
    ptrs = {"inScan": np.uint8, "outScan": np.uint8}
    src = [f"def {name}(inScan, outScan, width):"]
    lines = [l for l in body if l and not l.startswith("//") and not l.startswith("version(")]
    # ---- declarations before the loop
    k = 0
    while k < len(lines) and not lines[k].startswith("for "):
        l = lines[k]
        m = re.fullmatch(r"(?:const\((\w+)\)|(\w+))\s*\*\s*(\w+) = (?:cast\((?:const\(\w+\)|\w+)\s*\*\)\s*)?(inScan|outScan);", l)
        if m:
            t = m.group(1) or m.group(2)
            ptrs[m.group(3)] = DT[t]
            src.append(f"    {m.group(3)} = {m.group(4)}.view(np.{np.dtype(DT[t]).name})")
        elif l.startswith("memcpy(outScan, inScan,"):
            n = l[len("memcpy(outScan, inScan,"):].rstrip(");").strip().replace("float.sizeof", "4").replace("ubyte.sizeof", "1")
            src.append(f"    outScan[:{n}] = inScan[:{n}]")
        else:
            raise Unsupported(f"{name}: declaration `{l}`")
        k += 1
    if k == len(lines):
        return "\n".join(src) + "\n"
    assert lines[k] == "for (int x = 0; x < width; ++x)" and lines[k + 1] == "{" and lines[-1] == "}", (name, lines[k:k + 2], lines[-1])
    stmts = lines[k + 2:-1]
    # ---- how many times does the loop body advance each pointer?
    nload = {}
    for l in stmts:
        for p in re.findall(r"\*(\w+)\+\+", l):
            nload[p] = nload.get(p, 0) + 1
    counters = {}
    pending_if = None
    in_block = False
    for l in stmts:
        if pending_if is not None:                                # the statement(s) governed by `if (a != 0)`: `x /= a;`, braced or not
            if l == "{":
                in_block = True; continue
            if l == "}" and in_block:
                in_block = False; pending_if = None; continue
            m = re.fullmatch(r"(\w+) /= (\w+);", l)
            if not m:
                raise Unsupported(f"{name}: `{l}` under if")
            src.append(f"    {m.group(1)} = np.where({pending_if}, ({m.group(1)} / {m.group(2)}).astype(np.float32), {m.group(1)})")
            if not in_block:
                pending_if = None
            continue
        m = re.fullmatch(r"if \((\w+) != 0\)", l)
        if m:
            pending_if = f"{m.group(1)} != 0"; continue
        m = re.fullmatch(r"float (\w+) = (.*);", l)
        if m:
            src.append(f"    {m.group(1)} = f32({_expr(m.group(2), ptrs, nload, counters)})"); continue
        m = re.fullmatch(r"(ubyte|ushort) (\w+) = cast\(\1\)\((.*)\);", l)
        if m:
            src.append(f"    {m.group(2)} = cast_{m.group(1)}({_expr(m.group(3), ptrs, nload, counters)})"); continue
        m = re.fullmatch(r"ubyte (\w+) = (.*);", l)                   # ubyte b = inScan[x];
        if m:
            src.append(f"    {m.group(1)} = {_expr(m.group(2), ptrs, nload, counters)}"); continue
        m = re.fullmatch(r"\*(\w+)\+\+ = (.*);", l)
        if m:
            p = m.group(1)
            kk = counters.setdefault(p, 0); counters[p] += 1
            src.append(f"    {p}[{kk}::{nload[p]}][:width] = {_expr(m.group(2), ptrs, nload, counters)}"); continue
        m = re.fullmatch(r"(\w+\[[^\]]+\]) = (.*);", l)
        if m:
            src.append(f"    {_expr(m.group(1), ptrs, nload, counters)} = {_expr(m.group(2), ptrs, nload, counters)}"); continue
        raise Unsupported(f"{name}: statement `{l}`")
    for p, n in nload.items():
        assert counters.get(p, 0) == n, (name, p)
    return "\n".join(src) + "\n"


def F32(x):
    return np.float32(x)


def f32(x):
    return np.asarray(x, dtype=np.float32) if not isinstance(x, np.ndarray) else x.astype(np.float32)


def _cvttss2si(x):
    """float -> int as x86-64 compilers emit it for D's cast(ubyte) / cast(ushort) of a float: cvttss2si (truncation toward
    zero to int32; NaN and |x| >= 2^31 give the "integer indefinite" 0x80000000), then the low 8 / 16 bits are stored.
    In-range values (everything a [0, 1] image produces) do not depend on this choice."""
    x = np.asarray(x, np.float32)
    ok = np.isfinite(x) & (np.abs(x) < np.float32(2147483648.0))
    return np.where(ok, np.trunc(np.where(ok, x, 0)).astype(np.int64), -2147483648)


def cast_ubyte(x):
    return (_cvttss2si(x) & 0xFF).astype(np.uint8)


def cast_ushort(x):
    return (_cvttss2si(x) & 0xFFFF).astype(np.uint16)


class Reference:
    """the reference's scanline functions, executable"""

    def __init__(self, path=SRC):
        text = open(path).read()
        self.source = {}
        self.fn = {}
        env = {"np": np, "F32": F32, "f32": f32, "cast_ubyte": cast_ubyte, "cast_ushort": cast_ushort}
        for name, body in _functions(text).items():
            py = transpile(name, body)
            self.source[name] = py
            exec(compile(py, f"<{name}>", "exec"), env)
            self.fn[name] = env[name]
        # the two dispatch switches (:841-930): PixelType -> function, parsed from the `case` lines
        self.to_inter = {(m.group(1), m.group(3)): m.group(2) + "_to_" + m.group(3)
                         for m in re.finditer(r"case (\w+):\s*(scanline_convert_\w+?)_to_(rgba8|rgbaf32)\s*\(src, dest, width\)", text)}
        self.from_inter = {(m.group(3), m.group(1)): m.group(2)
                           for m in re.finditer(r"case (\w+):\s*(scanline_convert_(rgba8|rgbaf32)_to_\w+?)\s*\(src, dest, width\)", text)}

    def run(self, name, src, width, out_bytes):
        out = np.zeros(out_bytes, np.uint8)
        with np.errstate(all="ignore"):
            self.fn[name](np.ascontiguousarray(src).view(np.uint8).reshape(-1), out, width)
        return out

    def convert_row(self, src_type, dst_type, src, width, size):
        """scanlinesConvert (:70-121) for one row: through the intermediate type scanlinesInterType (:25-31) picks"""
        plain8 = ("l8", "la8", "rgb8", "rgba8")
        inter = "rgba8" if (src_type in plain8 and dst_type in plain8) else "rgbaf32"
        buf = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
        if src_type != inter:
            buf = self.run(self.to_inter[(src_type, inter)], buf, width, width * size[inter])
        if dst_type != inter:
            buf = self.run(self.from_inter[(inter, dst_type)], buf, width, width * size[dst_type])
        return buf


if __name__ == "__main__":
    R = Reference()
    print(len(R.fn), "functions;", len(R.to_inter), "+", len(R.from_inter), "dispatch cases")
    print(R.source["scanline_convert_lap8_to_rgbaf32"])
    print(R.source["scanline_convert_rgbaf32_to_lap16"])
    print(R.source["scanline_convert_rgba8_to_la8"])
