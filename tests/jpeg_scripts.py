"""Progressive (SOF2) JPEG files with ANY scan script, for tests: the quantised coefficients of a baseline file (decoded by the oracle's
feeder) are written again as the scans the script lists -- spectral selection and successive approximation as ITU T.81 Annex G describes
them (DC first / refinement, AC first with EOB runs, AC refinement with buffered correction bits), optional restart intervals, one
Huffman table per scan built from the symbols the scan uses.  Pillow / libjpeg only ever write libjpeg's default script; the reference's
decoder (jpegload.d:3296-3664) takes any legal one, and scripts whose scans do not line up (bands split four ways, DC scans per
component followed by an interleaved refinement, ...) are what the GPU path's fallback ordering exists for.

    script: list of (components, Ss, Se, Ah, Al), components = tuple of component indices (0 = Y, 1 = Cb, 2 = Cr)
"""
import numpy as np

import oracle_lib as O

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50,
          43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _segments(data):
    """marker segments of the header part: [(marker, payload)] up to the first SOS"""
    out, p = [], 2
    while p + 4 <= len(data):
        assert data[p] == 0xFF
        m = data[p + 1]
        n = (data[p + 2] << 8) | data[p + 3]
        out.append((m, data[p + 4:p + 2 + n]))
        if m == 0xDA:
            break
        p += 2 + n
    return out


class _Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, value, nbits):
        if nbits == 0:
            return
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1)); self.n += nbits
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _scan_symbols(blocks_of, comps, ss, se, ah, al, units, restart):
    """-> list of restart segments, each a list of ('s', symbol) / ('b', value, nbits) events.  blocks_of(u) -> [(component slot, block)] of unit u"""
    segs, ev = [], []
    last_dc = [0] * len(comps)
    eobrun, pending = 0, []                       # AC: the run of all-zero bands, the correction bits of its blocks (refinement)

    def emit_eobrun():
        nonlocal eobrun, pending
        if eobrun > 0:
            nb = eobrun.bit_length() - 1
            ev.append(("s", nb << 4))
            if nb:
                ev.append(("b", eobrun & ((1 << nb) - 1), nb))
            eobrun = 0
            for b in pending:
                ev.append(("b", b, 1))
            pending = []

    for u in range(units):
        if restart and u and u % restart == 0:
            emit_eobrun()
            segs.append(ev); ev = []
            last_dc = [0] * len(comps)
        for slot, blk in blocks_of(u):
            if ss == 0:
                if ah == 0:                                                   # DC first: difference of the point-transformed values
                    v = int(blk[0]) >> al
                    d = v - last_dc[slot]; last_dc[slot] = v
                    nb = abs(d).bit_length()
                    ev.append(("s", nb))
                    if nb:
                        ev.append(("b", d if d >= 0 else d - 1, nb))
                else:                                                          # DC refinement: the next bit
                    ev.append(("b", (int(blk[0]) >> al) & 1, 1))
                continue
            if ah == 0:                                                        # AC first
                r = 0
                for k in range(ss, se + 1):
                    c = int(blk[ZIGZAG[k]])
                    t = abs(c) >> al
                    if t == 0:
                        r += 1; continue
                    emit_eobrun()
                    while r > 15:
                        ev.append(("s", 0xF0)); r -= 16
                    nb = t.bit_length()
                    ev.append(("s", (r << 4) | nb)); ev.append(("b", t if c > 0 else ~t, nb)); r = 0
                if r > 0:
                    eobrun += 1
                    if eobrun == 0x7FFF:
                        emit_eobrun()
                continue
            # AC refinement
            absv = [abs(int(blk[ZIGZAG[k]])) >> al for k in range(ss, se + 1)]
            eob = max([i for i, t in enumerate(absv) if t == 1], default=-1)
            r, mine = 0, []
            for i, t in enumerate(absv):
                if t == 0:
                    r += 1; continue
                while r > 15 and i <= eob:
                    emit_eobrun()
                    ev.append(("s", 0xF0)); r -= 16
                    for b in mine:
                        ev.append(("b", b, 1))
                    mine = []
                if t > 1:
                    mine.append(t & 1); continue
                emit_eobrun()
                ev.append(("s", (r << 4) | 1)); ev.append(("b", 0 if int(blk[ZIGZAG[ss + i]]) < 0 else 1, 1))
                for b in mine:
                    ev.append(("b", b, 1))
                mine = []; r = 0
            if r > 0 or mine:
                eobrun += 1; pending += mine
                if eobrun == 0x7FFF or len(pending) > 900:
                    emit_eobrun()
    emit_eobrun()
    segs.append(ev)
    return segs


def progressive_with_script(baseline, script, restart=0):
    """baseline: bytes of a baseline JPEG (any sampling mode the oracle's feeder takes).  -> bytes of a progressive file with the given scans."""
    d = O.DecodedJpeg(baseline)
    hdr = _segments(baseline)
    qt = {}
    for m, pl in hdr:
        if m == 0xDB:
            p = 0
            while p < len(pl):
                pq, tq = pl[p] >> 4, pl[p] & 15
                assert pq == 0
                qt[tq] = np.array(list(pl[p + 1:p + 65]), np.int32); p += 65
    sof = next(pl for m, pl in hdr if m == 0xC0)
    ncomp = sof[5]
    hs = [sof[6 + 3 * c + 1] >> 4 for c in range(ncomp)]; vs = [sof[6 + 3 * c + 1] & 15 for c in range(ncomp)]
    tq = [sof[6 + 3 * c + 2] for c in range(ncomp)]
    cid = [sof[6 + 3 * c] for c in range(ncomp)]
    hmax, vmax = max(hs), max(vs)
    nb = sum(h * v for h, v in zip(hs, vs))
    off = [sum(hs[i] * vs[i] for i in range(c)) for c in range(ncomp)]
    co = d.coeffs.reshape(d.mcus_per_col * d.mcus_per_row, nb, 64).astype(np.int32)
    for c in range(ncomp):                                                     # back to quantised values (natural order, like the table after un-zigzag)
        q = np.zeros(64, np.int32); q[ZIGZAG] = qt[tq[c]]
        for b in range(hs[c] * vs[c]):
            assert not (co[:, off[c] + b] % q).any()
            co[:, off[c] + b] //= q
    out = bytearray(b"\xff\xd8")
    for m, pl in hdr:
        if m in (0xE0, 0xDB):
            out += bytes([0xFF, m]) + (len(pl) + 2).to_bytes(2, "big") + bytes(pl)
    out += b"\xff\xc2" + (len(sof) + 2).to_bytes(2, "big") + bytes(sof)
    if restart:
        out += b"\xff\xdd\x00\x04" + int(restart).to_bytes(2, "big")
    for comps, ss, se, ah, al in script:
        comps = tuple(comps)
        if len(comps) == 1:
            c = comps[0]
            nbx = ((d.width * hs[c] + hmax - 1) // hmax + 7) // 8; nby = ((d.height * vs[c] + vmax - 1) // vmax + 7) // 8
            units = nbx * nby

            def blocks_of(u, c=c, nbx=nbx):
                by, bx = divmod(u, nbx)
                mcu = (by // vs[c]) * d.mcus_per_row + bx // hs[c]
                return [(0, co[mcu, off[c] + (by % vs[c]) * hs[c] + bx % hs[c]])]
        else:
            units = d.mcus_per_row * d.mcus_per_col

            def blocks_of(u, comps=comps):
                return [(slot, co[u, off[c] + b]) for slot, c in enumerate(comps) for b in range(hs[c] * vs[c])]
        segs = _scan_symbols(blocks_of, comps, ss, se, ah, al, units, restart)
        dc_refine = ss == 0 and ah != 0
        if not dc_refine:
            syms = sorted({e[1] for sg in segs for e in sg if e[0] == "s"}) or [0]
            assert len(syms) <= 255
            code = {s: i for i, s in enumerate(syms)}                          # every symbol an 8-bit code, in order: canonical
            counts = [0] * 16; counts[7] = len(syms)
            out += b"\xff\xc4" + (2 + 1 + 16 + len(syms)).to_bytes(2, "big") + bytes([(0x10 if ss else 0x00)]) + bytes(counts) + bytes(syms)
        out += b"\xff\xda" + (6 + 2 * len(comps)).to_bytes(2, "big") + bytes([len(comps)])
        for c in comps:
            out += bytes([cid[c], 0x00])
        out += bytes([ss, se, (ah << 4) | al])
        for i, sg in enumerate(segs):
            if i:
                out += bytes([0xFF, 0xD0 + ((i - 1) & 7)])
            bw = _Bits()
            for e in sg:
                if e[0] == "s":
                    bw.put(code[e[1]], 8)
                else:
                    bw.put(e[1], e[2])
            bw.flush()
            out += bw.out
    out += b"\xff\xd9"
    return bytes(out)


# scripts -------------------------------------------------------------------------------------------------------------------------
LIBJPEG_DEFAULT = [((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((2,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((0,), 6, 63, 0, 2),
                   ((0,), 1, 63, 2, 1), ((0, 1, 2), 0, 0, 1, 0), ((2,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((0,), 1, 63, 1, 0)]
# the luma band cut four ways before it is refined (four scans under one refinement scan: more than an item's three dependency slots),
# DC scans per component under an interleaved DC refinement (units that do not line up), two refinement passes
FOUR_BANDS = [((0,), 0, 0, 0, 1), ((1,), 0, 0, 0, 1), ((2,), 0, 0, 0, 1),
              ((0,), 1, 2, 0, 2), ((0,), 3, 5, 0, 2), ((0,), 6, 20, 0, 2), ((0,), 21, 63, 0, 2),
              ((1,), 1, 63, 0, 1), ((2,), 1, 63, 0, 1),
              ((0,), 1, 63, 2, 1), ((0, 1, 2), 0, 0, 1, 0), ((0,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((2,), 1, 63, 1, 0)]
# spectral selection only, interleaved DC, chroma before luma
SPECTRAL_ONLY = [((0, 1, 2), 0, 0, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0), ((0,), 1, 9, 0, 0), ((0,), 10, 63, 0, 0)]
# refinement scans over PARTS of a band (each stands on one first scan, lined up) and three bit planes
DEEP = [((0, 1, 2), 0, 0, 0, 2), ((0,), 1, 63, 0, 3), ((1,), 1, 63, 0, 2), ((2,), 1, 63, 0, 2),
        ((0,), 1, 10, 3, 2), ((0,), 11, 63, 3, 2), ((0, 1, 2), 0, 0, 2, 1), ((0,), 1, 63, 2, 1), ((1,), 1, 63, 2, 1), ((2,), 1, 63, 2, 1),
        ((0, 1, 2), 0, 0, 1, 0), ((0,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((2,), 1, 63, 1, 0)]
GREY = [((0,), 0, 0, 0, 1), ((0,), 1, 63, 0, 1), ((0,), 0, 0, 1, 0), ((0,), 1, 63, 1, 0)]
SCRIPTS = {"libjpeg": LIBJPEG_DEFAULT, "four_bands": FOUR_BANDS, "spectral_only": SPECTRAL_ONLY, "deep": DEEP}
