"""A standalone JPEG ENTROPY decoder (ITU-T T.81 Annexes B, F and G), written from the standard only: it imports
nothing from oracle/ or from the library and shares no code or structure with either.  It recovers, from the bytes of
a file, the QUANTISED zig-zag coefficients the entropy-coded scans carry (no dequantisation, no IDCT) - which is
exactly what the block path of the encoder produced - so a test can compare

    coefficients the GPU computed   (jpegenc_blocks_host)      with
    coefficients the GPU's FILE carries (Encoder::encode -> bytes -> this decoder)

without the oracle in the loop, and likewise check the oracle's files against the oracle's blocks on CPU.

Scope: baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman files with spectral selection and
without successive approximation (Ah = Al = 0, what vstroebel/jpeg-encoder emits, encoder.rs:885-972), interleaved and
non-interleaved scans, restart intervals, 1-4 components, any sampling factors 1..4, EOBn runs.
"""
import numpy as np


class JpegError(ValueError):
    pass


class _Bits:
    """Entropy-coded segment reader: removes the stuffed 0x00 after 0xFF (B.1.1.5, F.1.2.3), stops at markers."""

    def __init__(self, data, pos):
        self.d, self.pos, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        d = self.d
        self.acc &= (1 << self.n) - 1               # drop the consumed bits (Python ints are unbounded)
        while self.n <= 24:
            if self.pos >= len(d):
                b = 0
            else:
                b = d[self.pos]
                if b == 0xFF:
                    nxt = d[self.pos + 1] if self.pos + 1 < len(d) else 0xD9
                    if nxt == 0x00:
                        self.pos += 2
                    else:                       # a marker: feed zeros, do not advance (F.2.2.5)
                        b = 0
                        self.acc = (self.acc << 8) | b
                        self.n += 8
                        continue
                else:
                    self.pos += 1
            self.acc = (self.acc << 8) | b
            self.n += 8

    def bit(self):
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def peek16(self):
        if self.n < 16:
            self._fill()
        return (self.acc >> (self.n - 16)) & 0xFFFF

    def skip(self, k):
        self.n -= k

    def align_and_expect_rst(self, index):
        """Discard the padding bits, then the RSTm marker must follow (F.2.2.5 / E.2.4)."""
        # bytes already pulled into the accumulator beyond the current byte boundary were real data only if n >= 8
        # cannot happen here: _fill never crosses a marker, and padding is < 8 bits
        self.acc, self.n = 0, 0
        d = self.d
        if self.pos + 1 >= len(d) or d[self.pos] != 0xFF or d[self.pos + 1] != 0xD0 + (index & 7):
            raise JpegError(f"expected RST{index & 7} at {self.pos}, found {d[self.pos:self.pos + 2].hex()}")
        self.pos += 2


class _Huff:
    """Decoder tables of Figure F.15 / F.16 from the BITS and HUFFVAL lists of a DHT segment (C.2, F.2.2.3)."""

    def __init__(self, counts, values):
        self.mincode, self.maxcode, self.valptr, self.values, self.fast = [0] * 17, [-1] * 18, [0] * 17, values, None
        code, k = 0, 0
        for length in range(1, 17):
            self.valptr[length] = k
            self.mincode[length] = code
            code += counts[length - 1]
            k += counts[length - 1]
            self.maxcode[length] = code - 1 if counts[length - 1] else -1
            code <<= 1
        if k != len(values):
            raise JpegError("DHT: counts and values disagree")

    def decode(self, br):
        """Figure F.16, through a table indexed by the next 16 bits (built on first use): entry = length << 8 | value."""
        if self.fast is None:
            fast = [0] * 65536
            k = 0
            for length in range(1, 17):
                n = 0 if self.maxcode[length] < 0 else self.maxcode[length] - self.mincode[length] + 1
                for i in range(n):
                    code = self.mincode[length] + i
                    lo = code << (16 - length)
                    entry = (length << 8) | self.values[k]
                    fast[lo:lo + (1 << (16 - length))] = [entry] * (1 << (16 - length))
                    k += 1
            self.fast = fast
        e = self.fast[br.peek16()]
        if e == 0:
            raise JpegError("invalid Huffman code")
        br.skip(e >> 8)
        return e & 0xFF


def _extend(v, t):                                   # Figure F.12
    return v if t == 0 or v >= (1 << (t - 1)) else v - (1 << t) + 1


def _ceil_div(a, b):
    return -(-a // b)


def decode_coefficients(data):
    """bytes of a JPEG file -> dict with
         'width', 'height', 'progressive', 'components': [{'id', 'h', 'v', 'tq'}], 'qtables': {tq: 64 zig-zag ints},
         'blocks': per component an int16 array (block_rows, block_cols, 64) of quantised zig-zag coefficients covering
                   the component's MCU-padded block grid (A.2.4), 'grid': per component (rows, cols) a non-interleaved scan covers
                   (A.2.3), 'restart_interval', 'scans': [(component ids, Ss, Se)]."""
    d = bytes(data)
    if d[:2] != b"\xff\xd8":
        raise JpegError("no SOI")
    pos = 2
    qt, dc_tab, ac_tab = {}, {}, {}
    frame = None
    ri = 0
    scans = []
    seen_eoi = False
    while pos < len(d):
        if d[pos] != 0xFF:
            raise JpegError(f"marker expected at {pos}")
        m = d[pos + 1]
        pos += 2
        if m == 0xD9:
            seen_eoi = True
            break
        if m == 0xFF:                                  # fill byte
            pos -= 1
            continue
        seglen = (d[pos] << 8) | d[pos + 1]
        seg = d[pos + 2:pos + seglen]
        pos += seglen
        if m == 0xDB:                                  # DQT (B.2.4.1)
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    qt[tq] = [(seg[i + 2 * k] << 8) | seg[i + 2 * k + 1] for k in range(64)]
                    i += 128
                else:
                    qt[tq] = list(seg[i:i + 64])
                    i += 64
        elif m == 0xC4:                                # DHT (B.2.4.2)
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                n = sum(counts)
                vals = list(seg[i + 17:i + 17 + n])
                (ac_tab if tc else dc_tab)[th] = _Huff(counts, vals)
                i += 17 + n
        elif m in (0xC0, 0xC1, 0xC2):                  # SOF0/1/2 (B.2.2)
            if seg[0] != 8:
                raise JpegError("only 8-bit precision")
            height, width, nf = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [{"id": seg[6 + 3 * i], "h": seg[7 + 3 * i] >> 4, "v": seg[7 + 3 * i] & 15, "tq": seg[8 + 3 * i]} for i in range(nf)]
            hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
            mcus_x, mcus_y = _ceil_div(width, 8 * hmax), _ceil_div(height, 8 * vmax)
            for c in comps:
                c["blocks"] = np.zeros((mcus_y * c["v"], mcus_x * c["h"], 64), dtype=np.int16)
                # A.2.3: a non-interleaved scan covers ceil(x_i / 8) x ceil(y_i / 8) blocks, x_i = ceil(X * H_i / Hmax)
                c["grid"] = (_ceil_div(_ceil_div(height * c["v"], vmax), 8), _ceil_div(_ceil_div(width * c["h"], hmax), 8))
            frame = {"width": width, "height": height, "progressive": m == 0xC2, "comps": comps, "hmax": hmax, "vmax": vmax,
                     "mcus_x": mcus_x, "mcus_y": mcus_y}
        elif m == 0xDD:                                # DRI (B.2.4.4)
            ri = (seg[0] << 8) | seg[1]
        elif m == 0xDA:                                # SOS (B.2.3) + entropy-coded data
            if frame is None:
                raise JpegError("SOS before SOF")
            ns = seg[0]
            sel = []
            for i in range(ns):
                cid, tables = seg[1 + 2 * i], seg[2 + 2 * i]
                comp = next(c for c in frame["comps"] if c["id"] == cid)
                sel.append((comp, tables >> 4, tables & 15))
            ss, se, ahl = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]
            if ahl != 0:
                raise JpegError("successive approximation is outside this decoder's scope")
            scans.append(([c["id"] for c, _, _ in sel], ss, se))
            pos = _decode_scan(d, pos, frame, sel, ss, se, ri, dc_tab, ac_tab)
        elif m in (0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError("unsupported frame type")
        # APPn, COM and anything else: skipped
    if not seen_eoi:
        raise JpegError("no EOI")
    if frame is None:
        raise JpegError("no frame")
    return {"width": frame["width"], "height": frame["height"], "progressive": frame["progressive"],
            "components": [{k: c[k] for k in ("id", "h", "v", "tq")} for c in frame["comps"]],
            "qtables": qt, "blocks": [c["blocks"] for c in frame["comps"]], "grid": [c["grid"] for c in frame["comps"]],
            "restart_interval": ri, "scans": scans, "mcus": (frame["mcus_y"], frame["mcus_x"])}


def _decode_scan(d, pos, frame, sel, ss, se, ri, dc_tab, ac_tab):
    br = _Bits(d, pos)
    interleaved = len(sel) > 1
    if frame["progressive"]:
        if ss == 0 and se != 0:
            raise JpegError("progressive DC scan must have Se = 0")
        if ss > 0 and interleaved:
            raise JpegError("progressive AC scans are non-interleaved (G.1.1.1.1)")
    elif (ss, se) != (0, 63):
        raise JpegError("sequential scan must cover 0..63")
    if interleaved:
        units_y, units_x = frame["mcus_y"], frame["mcus_x"]
    else:
        units_y, units_x = sel[0][0]["grid"]
    pred = [0] * len(sel)
    eobrun = 0
    restarts = 0
    to_go = ri
    for uy in range(units_y):
        for ux in range(units_x):
            if ri and to_go == 0:                      # E.2.4: restart interval boundary
                br.align_and_expect_rst(restarts)
                restarts += 1
                to_go = ri
                pred = [0] * len(sel)
                eobrun = 0
            for si, (comp, td, ta) in enumerate(sel):
                if interleaved:
                    cells = [(uy * comp["v"] + v, ux * comp["h"] + h) for v in range(comp["v"]) for h in range(comp["h"])]
                else:
                    cells = [(uy, ux)]
                for by, bx in cells:
                    blk = comp["blocks"][by, bx]
                    k = ss
                    if ss == 0:                        # DC (F.2.2.1)
                        t = dc_tab[td].decode(br)
                        diff = _extend(br.bits(t), t)
                        pred[si] += diff
                        blk[0] = pred[si]
                        k = 1
                    if se >= k:                        # AC (F.2.2.2, with the EOBn runs of G.1.2.2)
                        if eobrun > 0:
                            eobrun -= 1
                            continue
                        while k <= se:
                            rs = ac_tab[ta].decode(br)
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                eobrun = (1 << r) - 1
                                if r:
                                    eobrun += br.bits(r)
                                break
                            k += r
                            if k > se:
                                raise JpegError("AC run past the end of the band")
                            blk[k] = _extend(br.bits(s), s)
                            k += 1
            if ri:
                to_go -= 1
    # the segment ends at the next marker: padding bits (1s) then 0xFF + non-zero
    p = br.pos
    while p + 1 < len(d) and not (d[p] == 0xFF and d[p + 1] != 0x00 and not 0xD0 <= d[p + 1] <= 0xD7):
        p += 1
    if p - br.pos > 2:                                 # at most the last (possibly stuffed) byte was not pulled in yet
        raise JpegError(f"{p - br.pos} undecoded bytes at the end of a scan")
    return p


def blocks_in_mcu_order(dec):
    """(total_blocks, 64): the decoded blocks in the order an interleaved scan codes them (A.2.3)."""
    my, mx = dec["mcus"]
    out = []
    for uy in range(my):
        for ux in range(mx):
            for c, blk in zip(dec["components"], dec["blocks"]):
                for v in range(c["v"]):
                    for h in range(c["h"]):
                        out.append(blk[uy * c["v"] + v, ux * c["h"] + h])
    return np.stack(out)


def blocks_in_planar_order(dec):
    """(total_blocks, 64): component-major, each component's A.2.3 grid row by row."""
    out = []
    for (rows, cols), blk in zip(dec["grid"], dec["blocks"]):
        out.append(blk[:rows, :cols].reshape(-1, 64))
    return np.concatenate(out)
