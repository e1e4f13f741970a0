"""TEST INFRASTRUCTURE — pure Python / NumPy restatement of the JPEG decode that
``tf.image.decode_jpeg(channels=3)`` (dataset/dataset.py:28) performs, i.e. of libjpeg-turbo's Huffman
paths: baseline / sequential entropy decoding (ITU-T T.81 Annex F / jdhuff.c), progressive entropy decoding (T.81
Annex G / jdphuff.c: spectral selection + successive approximation over several scans), dequantisation + ``jpeg_idct_islow``
(jidctint.c), ``h2v1/h2v2/h1v2_fancy_upsample`` (jdsample.c) and ``ycc_rgb_convert`` (jdcolor.c).

The algorithm lives in a third-party dependency that is NOT vendored in the reference (TensorFlow bundles
libjpeg-turbo; version unpinned).  PINNING: Pillow in this image links libjpeg-turbo with the same defaults
(JDCT_ISLOW, fancy upsampling); tests/test_oracle_jpeg.py checks this restatement bit-for-bit against
Pillow on the three JPEG images embedded in the reference
(models/keras_cv_attention_models/test_images.py:6-15, committed as tests/golden/*.jpg) and on the
synthetic set, so the uint8 stage of the oracle is pinned to the real library.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
                   13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52,
                   45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class _Bits:
    def __init__(self, data):
        self.d, self.pos, self.acc, self.n = data, 0, 0, 0
        self.marker = False

    def _fill(self):
        while self.n <= 24:
            b = 0
            if not self.marker and self.pos < len(self.d):
                b = self.d[self.pos]
                if b == 0xFF:
                    if self.pos + 1 < len(self.d) and self.d[self.pos + 1] == 0:
                        self.pos += 2
                    else:
                        self.marker, b = True, 0
                else:
                    self.pos += 1
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFF
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        self._fill()
        v = (self.acc >> (self.n - k)) & ((1 << k) - 1)
        self.n -= k
        return v

    def restart(self):
        self.acc, self.n, self.marker = 0, 0, False
        if self.pos + 1 < len(self.d) and self.d[self.pos] == 0xFF and 0xD0 <= self.d[self.pos + 1] <= 0xD7:
            self.pos += 2


def _build(bits, vals):
    """code -> symbol dictionary keyed by (length, code) (T.81 Annex C)"""
    table, code, k = {}, 0, 0
    for l in range(1, 17):
        for _ in range(bits[l - 1]):
            table[(l, code)] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return table


def _sym(br, table):
    code = 0
    for l in range(1, 17):
        code = (code << 1) | br.get(1)
        s = table.get((l, code))
        if s is not None:
            return s
    raise ValueError("corrupt huffman code")


def _extend(v, s):
    return v - (1 << s) + 1 if s and v < (1 << (s - 1)) else v


def parse(data: bytes):
    d = data
    assert d[0] == 0xFF and d[1] == 0xD8, "no SOI"
    pos = 2
    P = dict(qt={}, dc={}, ac={}, ri=0)
    while pos < len(d):
        assert d[pos] == 0xFF
        while d[pos] == 0xFF:
            pos += 1
        m = d[pos]
        pos += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            break
        ln = (d[pos] << 8) | d[pos + 1]
        s = d[pos + 2:pos + ln]
        if m == 0xE0 and bytes(s[:5]) == b"JFIF\x00":
            P["jfif"] = True
        if m == 0xEE and len(s) >= 12 and bytes(s[:5]) == b"Adobe":
            P["adobe_transform"] = s[11]
        if m in (0xC0, 0xC1, 0xC2):
            P["progressive"] = (m == 0xC2)
            assert s[0] == 8
            P["h"], P["w"], nc = (s[1] << 8) | s[2], (s[3] << 8) | s[4], s[5]
            P["comp"] = [dict(id=s[6 + 3 * i], h=s[7 + 3 * i] >> 4, v=s[7 + 3 * i] & 15, tq=s[8 + 3 * i]) for i in range(nc)]
            if nc == 1:
                P["comp"][0]["h"] = P["comp"][0]["v"] = 1
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise ValueError("unsupported SOF")
        elif m == 0xDB:
            o = 0
            while o < len(s):
                pq, tq = s[o] >> 4, s[o] & 15
                o += 1
                q = np.zeros(64, np.int64)
                for i in range(64):
                    q[ZIGZAG[i]] = ((s[o + 2 * i] << 8) | s[o + 2 * i + 1]) if pq else s[o + i]
                P["qt"][tq] = q
                o += 128 if pq else 64
        elif m == 0xC4:
            o = 0
            while o + 17 <= len(s):
                tc, th = s[o] >> 4, s[o] & 15
                bits = list(s[o + 1:o + 17])
                n = sum(bits)
                vals = list(s[o + 17:o + 17 + n])
                P["ac" if tc else "dc"][th] = _build(bits, vals)
                o += 17 + n
        elif m == 0xDD:
            P["ri"] = (s[0] << 8) | s[1]
        elif m == 0xDA:
            ns = s[0]
            idx = []
            for i in range(ns):
                for ci, c in enumerate(P["comp"]):
                    if c["id"] == s[1 + 2 * i]:
                        c["td"], c["ta"] = s[2 + 2 * i] >> 4, s[2 + 2 * i] & 15
                        idx.append(ci)
            P["scan"] = d[pos + ln:]
            if not P.get("progressive") and ns == len(P["comp"]):
                return P                                  # the single interleaved scan of a sequential file
            # progressive (or multi-scan sequential): remember the scan with the tables in force, then skip its data
            ss, se, ah, al = s[1 + 2 * ns], s[2 + 2 * ns], s[3 + 2 * ns] >> 4, s[3 + 2 * ns] & 15
            P.setdefault("scans", []).append(dict(
                comps=idx, td=[P["comp"][i]["td"] for i in idx], ta=[P["comp"][i]["ta"] for i in idx], ss=ss, se=se, ah=ah,
                al=al, data=d[pos + ln:], dc=dict(P["dc"]), ac=dict(P["ac"]), ri=P["ri"]))
            pos += ln
            while pos + 1 < len(d) and not (d[pos] == 0xFF and d[pos + 1] != 0 and not 0xD0 <= d[pos + 1] <= 0xD7):
                pos += 1
            continue
        pos += ln
    if P.get("scans"):
        return P
    raise ValueError("no SOS")


def _entropy_decode_scans(P):
    """Progressive (and multi-scan sequential) entropy decoding: T.81 Annex G as libjpeg's jdphuff.c implements it.
    Every scan refines the same coefficient arrays; non-interleaved scans walk ceil(component_size / 8) blocks."""
    comps = P["comp"]
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    mx = -(-P["w"] // (8 * hmax))
    my = -(-P["h"] // (8 * vmax))
    coefs = [np.zeros((my * c["v"], mx * c["h"], 64), np.int64) for c in comps]
    for sc in P["scans"]:
        br = _Bits(sc["data"])
        ss, se, ah, al = sc["ss"], sc["se"], sc["ah"], sc["al"]
        state = dict(eobrun=0, pred=[0] * len(comps))

        def block(ci, k_td, k_ta, blk):
            if ss == 0 and not P.get("progressive"):          # sequential scan: a whole block
                s_ = _sym(br, sc["dc"][k_td])
                state["pred"][ci] += _extend(br.get(s_), s_)
                blk[0] = state["pred"][ci]
                k = 1
                while k < 64:
                    rs = _sym(br, sc["ac"][k_ta])
                    r, sz = rs >> 4, rs & 15
                    if sz == 0:
                        if r != 15:
                            break
                        k += 16
                        continue
                    k += r
                    blk[ZIGZAG[k]] = _extend(br.get(sz), sz)
                    k += 1
                return
            if ss == 0:                                       # DC scan
                if ah == 0:
                    s_ = _sym(br, sc["dc"][k_td])
                    state["pred"][ci] += _extend(br.get(s_), s_)
                    blk[0] = state["pred"][ci] * (1 << al)
                elif br.get(1):
                    blk[0] |= 1 << al
                return
            if ah == 0:                                       # AC first pass (decode_mcu_AC_first)
                if state["eobrun"] > 0:
                    state["eobrun"] -= 1
                    return
                k = ss
                while k <= se:
                    rs = _sym(br, sc["ac"][k_ta])
                    r, sz = rs >> 4, rs & 15
                    if sz:
                        k += r
                        blk[ZIGZAG[k]] = _extend(br.get(sz), sz) * (1 << al)
                    elif r == 15:
                        k += 15
                    else:
                        state["eobrun"] = (1 << r) + (br.get(r) if r else 0) - 1
                        break
                    k += 1
                return
            p1, m1 = 1 << al, -(1 << al)                      # AC refinement (decode_mcu_AC_refine)

            def refine(pos):
                if blk[pos] != 0 and br.get(1) and (blk[pos] & p1) == 0:
                    blk[pos] += p1 if blk[pos] >= 0 else m1

            k = ss
            if state["eobrun"] == 0:
                while k <= se:
                    rs = _sym(br, sc["ac"][k_ta])
                    r, sz = rs >> 4, rs & 15
                    val = 0
                    if sz:
                        val = p1 if br.get(1) else m1
                    elif r != 15:
                        state["eobrun"] = (1 << r) + (br.get(r) if r else 0)
                        break
                    while k <= se:
                        pos = ZIGZAG[k]
                        if blk[pos] != 0:
                            refine(pos)
                        else:
                            r -= 1
                            if r < 0:
                                break
                        k += 1
                    if val:
                        blk[ZIGZAG[k]] = val
                    k += 1
            if state["eobrun"] > 0:
                while k <= se:
                    refine(ZIGZAG[k])
                    k += 1
                state["eobrun"] -= 1

        left = sc["ri"]
        if len(sc["comps"]) > 1:                              # interleaved: MCU order
            units = [(y, x) for y in range(my) for x in range(mx)]
        else:                                                 # one component: its own block raster
            c = comps[sc["comps"][0]]
            bw = -(-(-(-P["w"] * c["h"] // hmax)) // 8)
            bh = -(-(-(-P["h"] * c["v"] // vmax)) // 8)
            units = [(y, x) for y in range(bh) for x in range(bw)]
        for (y, x) in units:
            if sc["ri"] and left == 0:
                br.restart()
                state["pred"] = [0] * len(comps)
                state["eobrun"] = 0
                left = sc["ri"]
            if len(sc["comps"]) > 1:
                for j, ci in enumerate(sc["comps"]):
                    c = comps[ci]
                    for by in range(c["v"]):
                        for bx in range(c["h"]):
                            block(ci, sc["td"][j], sc["ta"][j], coefs[ci][y * c["v"] + by, x * c["h"] + bx])
            else:
                ci = sc["comps"][0]
                block(ci, sc["td"][0], sc["ta"][0], coefs[ci][y, x])
            if sc["ri"]:
                left -= 1
    return coefs, (hmax, vmax)


def entropy_decode(P):
    """-> list of int64 coefficient arrays [blocks_h, blocks_w, 64] (natural order, quantised)"""
    if P.get("scans"):
        return _entropy_decode_scans(P)
    comps = P["comp"]
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    mx = -(-P["w"] // (8 * hmax))
    my = -(-P["h"] // (8 * vmax))
    coefs = [np.zeros((my * c["v"], mx * c["h"], 64), np.int64) for c in comps]
    br = _Bits(P["scan"])
    pred = [0] * len(comps)
    left = P["ri"]
    for y in range(my):
        for x in range(mx):
            if P["ri"] and left == 0:
                br.restart()
                pred = [0] * len(comps)
                left = P["ri"]
            for ci, c in enumerate(comps):
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = coefs[ci][y * c["v"] + by, x * c["h"] + bx]
                        s = _sym(br, P["dc"][c["td"]])
                        pred[ci] += _extend(br.get(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = _sym(br, P["ac"][c["ta"]])
                            r, sz = rs >> 4, rs & 15
                            if sz == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            blk[ZIGZAG[k]] = _extend(br.get(sz), sz)
                            k += 1
            if P["ri"]:
                left -= 1
    return coefs, (hmax, vmax)


# ---- jidctint.c ----
_C = dict(F_0_298631336=2446, F_0_390180644=3196, F_0_541196100=4433, F_0_765366865=6270, F_0_899976223=7373,
          F_1_175875602=9633, F_1_501321110=12299, F_1_847759065=15137, F_1_961570560=16069, F_2_053119869=16819,
          F_2_562915447=20995, F_3_072711026=25172)


def _idct_1d(i, shift):
    """i: [..., 8] int64 along the last axis -> [..., 8]"""
    c = _C
    z2, z3 = i[..., 2], i[..., 6]
    z1 = (z2 + z3) * c["F_0_541196100"]
    tmp2 = z1 + z3 * (-c["F_1_847759065"])
    tmp3 = z1 + z2 * c["F_0_765366865"]
    z2, z3 = i[..., 0], i[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = i[..., 7], i[..., 5], i[..., 3], i[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * c["F_1_175875602"]
    tmp0 = tmp0 * c["F_0_298631336"]
    tmp1 = tmp1 * c["F_2_053119869"]
    tmp2 = tmp2 * c["F_3_072711026"]
    tmp3 = tmp3 * c["F_1_501321110"]
    z1 = z1 * -c["F_0_899976223"]
    z2 = z2 * -c["F_2_562915447"]
    z3 = z3 * -c["F_1_961570560"] + z5
    z4 = z4 * -c["F_0_390180644"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    o = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return np.stack([(v + (1 << (shift - 1))) >> shift for v in o], axis=-1)


def idct_planes(coefs, qts):
    """dequantise + ISLOW IDCT + range limit -> uint8 planes [blocks_h*8, blocks_w*8]"""
    planes = []
    for cf, q in zip(coefs, qts):
        bh, bw, _ = cf.shape
        blk = (cf * q).reshape(bh, bw, 8, 8)                     # [r][c]
        ws = _idct_1d(blk.transpose(0, 1, 3, 2), 13 - 2)         # pass 1 over columns -> [bh,bw,col,row]
        ws = ws.transpose(0, 1, 3, 2)                            # [row][col]
        out = _idct_1d(ws, 13 + 2 + 3)                           # pass 2 over rows
        x = out & 1023                                           # idct_range_limit (jdmaster.c)
        px = np.where(x < 128, x + 128, np.where(x < 512, 255, np.where(x < 896, 0, x - 896)))
        planes.append(px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8).astype(np.int64))
    return planes


def _h2v1(p, dw):
    p = p[:, :dw]
    out = np.empty((p.shape[0], 2 * dw), np.int64)
    out[:, 0] = p[:, 0]
    out[:, 2::2] = (p[:, 1:] * 3 + p[:, :-1] + 1) >> 2
    out[:, 1:-1:2] = (p[:, :-1] * 3 + p[:, 1:] + 2) >> 2
    out[:, -1] = p[:, -1]
    return out


def _vert_sums(p, dh):
    """for every output row y: 3*row(y>>1) + nearer neighbour row (edge rows replicate, jdmainct.c)"""
    p = p[:dh]
    up = np.concatenate([p[:1], p[:-1]], 0)
    dn = np.concatenate([p[1:], p[-1:]], 0)
    s = np.empty((2 * dh, p.shape[1]), np.int64)
    s[0::2] = p * 3 + up
    s[1::2] = p * 3 + dn
    return s


def _h2v2(p, dw, dh):
    s = _vert_sums(p[:, :dw], dh)
    out = np.empty((2 * dh, 2 * dw), np.int64)
    out[:, 0] = (s[:, 0] * 4 + 8) >> 4
    out[:, 2::2] = (s[:, 1:] * 3 + s[:, :-1] + 8) >> 4
    out[:, 1:-1:2] = (s[:, :-1] * 3 + s[:, 1:] + 7) >> 4
    out[:, -1] = (s[:, -1] * 4 + 7) >> 4
    return out


def _h1v2(p, dw, dh):
    s = _vert_sums(p[:, :dw], dh)
    out = np.empty_like(s)
    out[0::2] = (s[0::2] + 1) >> 2
    out[1::2] = (s[1::2] + 2) >> 2
    return out


def decode_rgb(data: bytes) -> np.ndarray:
    """JPEG bytes -> uint8 [H, W, 3] (== PIL / libjpeg-turbo / tf.image.decode_jpeg(channels=3))"""
    P = parse(data)
    coefs, (hmax, vmax) = entropy_decode(P)
    planes = idct_planes(coefs, [P["qt"][c["tq"]] for c in P["comp"]])
    H, W = P["h"], P["w"]
    Y = planes[0][:H, :W]
    if len(planes) == 1:
        return np.stack([Y, Y, Y], -1).astype(np.uint8)
    ch = []
    for c, pl in zip(P["comp"][1:], planes[1:]):
        hs, vs = hmax // c["h"], vmax // c["v"]
        dw, dh = -(-W * c["h"] // hmax), -(-H * c["v"] // vmax)
        if hs == 1 and vs == 1:
            u = pl
        elif hs == 2 and dw <= 2:
            # jdsample.c jinit_upsampler: the fancy (triangle) filters are only chosen when downsampled_width > 2;
            # narrower planes get h2v1_upsample / h2v2_upsample, plain replication
            u = np.repeat(np.repeat(pl[:dh, :dw], 2, axis=1), vs, axis=0)
        elif hs == 2 and vs == 1:
            u = _h2v1(pl, dw)
        elif hs == 2 and vs == 2:
            u = _h2v2(pl, dw, dh)
        else:
            u = _h1v2(pl, dw, dh)
        ch.append(u[:H, :W] - 128)
    # colour space of a 3-component file as libjpeg guesses it (jdapimin.c default_decompress_parms)
    rgb_coded = False
    if not P.get("jfif"):
        if "adobe_transform" in P:
            rgb_coded = P["adobe_transform"] == 0
        else:
            rgb_coded = [c["id"] for c in P["comp"]] == [ord("R"), ord("G"), ord("B")]
    if rgb_coded:                                       # null conversion (jdcolor.c rgb_rgb_convert)
        return np.stack([Y, ch[0] + 128, ch[1] + 128], -1).astype(np.uint8)
    cb, cr = ch
    R = np.clip(Y + ((91881 * cr + 32768) >> 16), 0, 255)
    B = np.clip(Y + ((116130 * cb + 32768) >> 16), 0, 255)
    G = np.clip(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16), 0, 255)
    return np.stack([R, G, B], -1).astype(np.uint8)
