"""TEST INFRASTRUCTURE — CPU restatement of the baseline JPEG codec the reference's file boundaries go through.
Only tests/ may import this; the product path is csrc/jpeg.hip behind ffp_jpeg_*.

The reference writes crops and enhanced crops with `cv2.imwrite(path, img)` / `cv2.imwrite(path, img, [cv2.IMWRITE_JPEG_QUALITY, 95])`
(/root/reference/utils/visualization.py:218-221, utils/enhancer.py:273-278) and reads images with `cv2.imread` (utils/enhancer.py:254,
utils/visualization.py:200). OpenCV 4.11 (requirements.txt:98) does that with its bundled libjpeg-turbo: baseline sequential DCT,
YCbCr 4:2:0 (OpenCV's default sampling factor), quality 95 (OpenCV's default when none is given), the Annex K Huffman tables, the
accurate integer DCT ("islow"). That codec is not in /root/reference; what is restated here is the published IJG / libjpeg-turbo
algorithm: jccolor.c rgb_ycc_convert, jcsample.c h2v2_downsample (+ edge expansion), jfdctint.c, jcdctmgr.c quantisation, jcparam.c
quality scaling, jchuff.c encode_one_block, jcmarker.c headers; and for decoding jdhuff.c, jidctint.c, jdsample.c h2v2_fancy_upsample,
jdcolor.c ycc_rgb_convert.

Pinning: Pillow on this image is built on libjpeg-turbo (the same code base OpenCV bundles); tests/test_jpeg_oracle.py requires
`encode()` to be BYTE-IDENTICAL to `PIL.Image.save(format="JPEG", quality=q)` and `decode()` to be pixel-identical to `PIL.Image.open`
for seeded and real images. That pins the restatement to the codec family the reference uses, not to the reference's own files
(none of its outputs carry both an image and its source pixels).
"""
import numpy as np

ZIGZAG = np.asarray([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])

STD_LUMA_Q = np.asarray([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                         18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
STD_CHROMA_Q = np.asarray([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99])

# Annex K.3 Huffman tables: (bits[1..16], values)
DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
           [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
            0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
            0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
            0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
            0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
            0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
            0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
             [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
              0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
              0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
              0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
              0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
              0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
              0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


def quant_tables(quality: int):
    """jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline): natural (row-major) order."""
    q = max(1, min(100, int(quality)))
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for base in (STD_LUMA_Q, STD_CHROMA_Q):
        t = (base.astype(np.int64) * scale + 50) // 100
        out.append(np.clip(t, 1, 255).astype(np.int32))
    return out


def huff_codes(bits, vals):
    """jchuff.c jpeg_make_c_derived_tbl: symbol -> (code, length)."""
    code, k, table = 0, 0, {}
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


def fix(x):
    return int(x * 65536 + 0.5)


def rgb_to_ycc(rgb: np.ndarray):
    """jccolor.c rgb_ycc_convert, 16-bit fixed point."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (fix(0.29900) * r + fix(0.58700) * g + fix(0.11400) * b + half) >> 16
    cb = (-fix(0.16874) * r - fix(0.33126) * g + fix(0.50000) * b + off + half - 1) >> 16
    cr = (fix(0.50000) * r - fix(0.41869) * g - fix(0.08131) * b + off + half - 1) >> 16
    return y.astype(np.int32), cb.astype(np.int32), cr.astype(np.int32)


def pad_edge(p: np.ndarray, hp: int, wp: int) -> np.ndarray:
    """jcprepct.c expand_bottom_edge / jcsample.c expand_right_edge: replicate the last row / column."""
    return np.pad(p, ((0, hp - p.shape[0]), (0, wp - p.shape[1])), mode="edge")


def h2v2_downsample(p: np.ndarray) -> np.ndarray:
    """jcsample.c h2v2_downsample: 2x2 box with the alternating 1, 2 rounding bias along a row."""
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)[None, :]
    return ((s + bias) >> 2).astype(np.int32)


C = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137, f1_961=16069, f2_053=16819,
         f2_562=20995, f3_072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_1d(d, shift_even, shift_odd, first_pass):
    d = [d[..., i].astype(np.int64) for i in range(8)]
    t0, t7, t1, t6, t2, t5, t3, t4 = d[0] + d[7], d[0] - d[7], d[1] + d[6], d[1] - d[6], d[2] + d[5], d[2] - d[5], d[3] + d[4], d[3] - d[4]
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o = [None] * 8
    if first_pass:
        o[0], o[4] = (t10 + t11) << 2, (t10 - t11) << 2
    else:
        o[0], o[4] = _descale(t10 + t11, 2), _descale(t10 - t11, 2)
    z1 = (t12 + t13) * C["f0_541"]
    o[2] = _descale(z1 + t13 * C["f0_765"], shift_odd)
    o[6] = _descale(z1 - t12 * C["f1_847"], shift_odd)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * C["f1_175"]
    t4, t5, t6, t7 = t4 * C["f0_298"], t5 * C["f2_053"], t6 * C["f3_072"], t7 * C["f1_501"]
    z1, z2, z3, z4 = -z1 * C["f0_899"], -z2 * C["f2_562"], -z3 * C["f1_961"] + z5, -z4 * C["f0_390"] + z5
    o[7], o[5], o[3], o[1] = _descale(t4 + z1 + z3, shift_odd), _descale(t5 + z2 + z4, shift_odd), _descale(t6 + z2 + z3, shift_odd), _descale(t7 + z1 + z4, shift_odd)
    return np.stack(o, -1)


def fdct_islow(block: np.ndarray) -> np.ndarray:
    """jfdctint.c jpeg_fdct_islow on [..., 8, 8] samples already centred (-128): rows then columns; output scaled by 8."""
    rows = _fdct_1d(block, 2, 11, True)
    cols = _fdct_1d(np.swapaxes(rows, -1, -2), 2, 15, False)
    return np.swapaxes(cols, -1, -2)


def quantize(coef: np.ndarray, qtbl: np.ndarray) -> np.ndarray:
    """jcdctmgr.c forward_DCT: symmetric round-half-up division by 8 * q."""
    q = (qtbl.reshape(8, 8).astype(np.int64)) << 3
    a = np.abs(coef) + (q >> 1)
    return (np.sign(coef) * (a // q)).astype(np.int32)


def blocks_of(plane: np.ndarray) -> np.ndarray:
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).swapaxes(1, 2)


def encode_coefficients(rgb: np.ndarray, quality: int):
    """-> (Y [by][bx][8][8], Cb, Cr) quantised coefficients in natural order for an H x W x 3 RGB image, 4:2:0."""
    h, w = rgb.shape[:2]
    hp, wp = -(-h // 16) * 16, -(-w // 16) * 16
    y, cb, cr = rgb_to_ycc(rgb)
    ql, qc = quant_tables(quality)
    # Edge expansion as the library orders it: columns are replicated at FULL resolution before downsampling (jcsample.c
    # expand_right_edge inside h2v2_downsample), rows only up to a whole row group (2 rows: jcprepct.c expand_bottom_edge of the
    # colour buffer); the rest of the last iMCU row is filled by replicating DOWNSAMPLED rows (jcprepct.c pre_process_data tail).
    h2 = h + (h & 1)
    planes = [pad_edge(y, hp, wp)] + [pad_edge(h2v2_downsample(pad_edge(c, h2, wp)), hp // 2, wp // 2) for c in (cb, cr)]
    out = []
    for plane, q in zip(planes, (ql, qc, qc)):
        out.append(quantize(fdct_islow(blocks_of(plane) - 128), q))
    # jccoefct.c compress_data: luma blocks of an edge MCU that lie wholly outside the image's own 8x8 block grid are DUMMY blocks —
    # AC zero, DC copied from the previous block of the MCU (right edge: the block to the left; bottom edge: the last block of the
    # MCU row above) — so that they cost almost nothing. (Chroma has one block per MCU at 4:2:0: never a dummy.)
    yq = out[0]
    hb, wb = -(-h // 8), -(-w // 8)
    for by in range(yq.shape[0]):
        for bx in range(yq.shape[1]):
            if by < hb and bx < wb:
                continue
            if by < hb:                                  # right-edge dummy (bx is odd: second block of its MCU row)
                dc = yq[by, bx - 1, 0, 0]
            else:                                        # bottom-edge dummy row (by is odd): DC of the MCU's block (by - 1, right column)
                dc = yq[by - 1, (bx | 1), 0, 0]
            yq[by, bx] = 0
            yq[by, bx, 0, 0] = dc
    return out


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put(0x7F, 8 - self.n)       # jchuff.c flush_bits: pad with ones


def encode_block(bw, blk_zz, last_dc, dc_tab, ac_tab):
    """jchuff.c encode_one_block on 64 coefficients in zigzag order; returns the block's DC."""
    diff = int(blk_zz[0]) - last_dc
    t, t2 = (-diff, diff - 1) if diff < 0 else (diff, diff)
    nbits = t.bit_length()
    bw.put(*dc_tab[nbits])
    if nbits:
        bw.put(t2, nbits)
    r = 0
    for k in range(1, 64):
        v = int(blk_zz[k])
        if v == 0:
            r += 1
            continue
        while r > 15:
            bw.put(*ac_tab[0xF0])
            r -= 16
        t, t2 = (-v, v - 1) if v < 0 else (v, v)
        nbits = t.bit_length()
        bw.put(*ac_tab[(r << 4) + nbits])
        bw.put(t2, nbits)
        r = 0
    if r > 0:
        bw.put(*ac_tab[0])
    return int(blk_zz[0])


def header(h: int, w: int, quality: int) -> bytes:
    """jcmarker.c: SOI, JFIF APP0 (1.01, no units, 1:1), two DQT, SOF0 (2x2, 1x1, 1x1), four DHT, SOS."""
    def seg(marker, payload):
        return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)
    ql, qc = quant_tables(quality)
    out = bytes([0xFF, 0xD8]) + seg(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    out += seg(0xDB, bytes([0]) + bytes(int(v) for v in ql[ZIGZAG])) + seg(0xDB, bytes([1]) + bytes(int(v) for v in qc[ZIGZAG]))
    out += seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1]))
    for tc_th, (bits, vals) in ((0x00, DC_LUMA), (0x10, AC_LUMA), (0x01, DC_CHROMA), (0x11, AC_CHROMA)):
        out += seg(0xC4, bytes([tc_th]) + bytes(bits) + bytes(vals))
    return out + seg(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))


def encode(rgb: np.ndarray, quality: int = 95) -> bytes:
    """Baseline JFIF file of an H x W x 3 uint8 RGB image, 4:2:0, standard Huffman tables."""
    rgb = np.asarray(rgb, np.uint8)
    yq, cbq, crq = encode_coefficients(rgb, quality)
    dcl, acl, dcc, acc = huff_codes(*DC_LUMA), huff_codes(*AC_LUMA), huff_codes(*DC_CHROMA), huff_codes(*AC_CHROMA)
    bw = BitWriter()
    last = [0, 0, 0]
    for my in range(cbq.shape[0]):
        for mx in range(cbq.shape[1]):
            for dy in range(2):
                for dx in range(2):
                    last[0] = encode_block(bw, yq[2 * my + dy, 2 * mx + dx].reshape(64)[ZIGZAG], last[0], dcl, acl)
            last[1] = encode_block(bw, cbq[my, mx].reshape(64)[ZIGZAG], last[1], dcc, acc)
            last[2] = encode_block(bw, crq[my, mx].reshape(64)[ZIGZAG], last[2], dcc, acc)
    bw.flush()
    return header(rgb.shape[0], rgb.shape[1], quality) + bytes(bw.out) + bytes([0xFF, 0xD9])


# ---- decoding (cv2.imread's default path: islow IDCT, fancy upsampling, YCbCr -> RGB) -------------------------------------------------
def parse(data: bytes) -> dict:
    """Markers of a baseline / extended-sequential 8-bit JFIF file with one interleaved scan (what cv2.imwrite and cameras write;
    progressive files are a different entropy coder and are rejected)."""
    assert data[:2] == b"\xff\xd8", "not a JPEG"
    i, out = 2, {"qt": {}, "ht": {}, "dri": 0}
    while True:
        assert data[i] == 0xFF, "marker expected"
        while data[i + 1] == 0xFF:
            i += 1
        m = data[i + 1]
        L = (data[i + 2] << 8) | data[i + 3]
        seg = data[i + 4:i + 2 + L]
        if m == 0xDB:
            k = 0
            while k < len(seg):
                pq, tq = seg[k] >> 4, seg[k] & 15
                assert pq == 0, "16-bit quantisation tables"
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[k + 1:k + 65], np.uint8)
                out["qt"][tq] = t
                k += 65
        elif m == 0xC4:
            k = 0
            while k < len(seg):
                bits = list(seg[k + 1:k + 17])
                n = sum(bits)
                out["ht"][seg[k]] = (bits, list(seg[k + 17:k + 17 + n]))
                k += 17 + n
        elif m in (0xC0, 0xC1):
            assert seg[0] == 8
            out["h"], out["w"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            out["comps"] = [{"id": seg[6 + 3 * c], "hs": seg[7 + 3 * c] >> 4, "vs": seg[7 + 3 * c] & 15, "tq": seg[8 + 3 * c]} for c in range(seg[5])]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise ValueError("unsupported JPEG process (progressive / lossless / arithmetic)")
        elif m == 0xDD:
            out["dri"] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            assert ns == len(out["comps"]), "non-interleaved scans"
            for c in range(ns):
                comp = next(x for x in out["comps"] if x["id"] == seg[1 + 2 * c])
                comp["td"], comp["ta"] = seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15
            out["scan"] = i + 2 + L
            return out
        i += 2 + L


def decode_coefficients(data: bytes, info: dict):
    """jdhuff.c decode_mcu over the whole scan -> per component [blocks_y][blocks_x][64] coefficients in natural order (not yet
    dequantised), block grids padded to whole MCUs."""
    hmax, vmax = max(c["hs"] for c in info["comps"]), max(c["vs"] for c in info["comps"])
    mx, my = -(-info["w"] // (8 * hmax)), -(-info["h"] // (8 * vmax))
    look = {}
    for key, (bits, vals) in info["ht"].items():
        d, code, k = {}, 0, 0
        for l in range(1, 17):
            for _ in range(bits[l - 1]):
                d[(l, code)] = vals[k]
                code += 1
                k += 1
            code <<= 1
        look[key] = d
    coefs = [np.zeros((my * c["vs"], mx * c["hs"], 64), np.int32) for c in info["comps"]]
    pos, acc, nb = info["scan"], 0, 0

    def bit():
        nonlocal pos, acc, nb
        if nb == 0:
            b = data[pos]
            pos += 1
            if b == 0xFF:
                assert data[pos] == 0, "marker inside entropy-coded data"
                pos += 1
            acc, nb = b, 8
        nb -= 1
        return (acc >> nb) & 1

    def sym(tab):
        code, l = 0, 0
        while True:
            code = (code << 1) | bit()
            l += 1
            if (l, code) in tab:
                return tab[(l, code)]
            assert l < 16, "bad Huffman code"

    def receive(n):
        v = 0
        for _ in range(n):
            v = (v << 1) | bit()
        return v if n == 0 or v >= (1 << (n - 1)) else v - (1 << n) + 1

    last = [0] * len(info["comps"])
    count = 0
    for m_y in range(my):
        for m_x in range(mx):
            if info["dri"] and count and count % info["dri"] == 0:
                nb = 0                                              # byte-align, skip RSTn, reset predictions
                assert data[pos] == 0xFF and 0xD0 <= data[pos + 1] <= 0xD7
                pos += 2
                last = [0] * len(info["comps"])
            count += 1
            for ci, c in enumerate(info["comps"]):
                for dy in range(c["vs"]):
                    for dx in range(c["hs"]):
                        blk = coefs[ci][m_y * c["vs"] + dy, m_x * c["hs"] + dx]
                        s = sym(look[c["td"]])
                        last[ci] += receive(s)
                        blk[0] = last[ci]
                        k = 1
                        while k < 64:
                            rs = sym(look[0x10 | c["ta"]])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            blk[ZIGZAG[k]] = receive(s)
                            k += 1
    return coefs


def _idct_1d(v, first):
    """jidctint.c one 8-point pass on [..., 8] (already dequantised); first: columns (keeps 2 extra bits), second: rows (+ range shift)."""
    v = [v[..., i].astype(np.int64) for i in range(8)]
    z2, z3 = v[2], v[6]
    z1 = (z2 + z3) * C["f0_541"]
    t2, t3 = z1 - z3 * C["f1_847"], z1 + z2 * C["f0_765"]
    t0, t1 = (v[0] + v[4]) << 13, (v[0] - v[4]) << 13
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = v[7], v[5], v[3], v[1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * C["f1_175"]
    t0, t1, t2, t3 = t0 * C["f0_298"], t1 * C["f2_053"], t2 * C["f3_072"], t3 * C["f1_501"]
    z1, z2, z3, z4 = -z1 * C["f0_899"], -z2 * C["f2_562"], -z3 * C["f1_961"] + z5, -z4 * C["f0_390"] + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    n = 11 if first else 18
    return np.stack([_descale(t10 + t3, n), _descale(t11 + t2, n), _descale(t12 + t1, n), _descale(t13 + t0, n),
                     _descale(t13 - t0, n), _descale(t12 - t1, n), _descale(t11 - t2, n), _descale(t10 - t3, n)], -1)


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """[by][bx][64] quantised coefficients -> sample plane (uint8 range), jidctint.c jpeg_idct_islow."""
    d = (coef * qt[None, None, :]).reshape(coef.shape[0], coef.shape[1], 8, 8)
    cols = np.swapaxes(_idct_1d(np.swapaxes(d, -1, -2), True), -1, -2)
    rows = _idct_1d(cols, False)
    out = np.clip(rows + 128, 0, 255)
    return out.swapaxes(1, 2).reshape(coef.shape[0] * 8, coef.shape[1] * 8).astype(np.int32)


def h2v1_fancy(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample: 3/4 nearer + 1/4 further sample, rounding 1 / 2 alternately, edge columns copied."""
    w = p.shape[1]
    out = np.zeros((p.shape[0], 2 * w), np.int32)
    left, right = np.concatenate([p[:, :1], p[:, :-1]], 1), np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0], out[:, -1] = p[:, 0], p[:, -1]
    return out


def h2v2_fancy(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: triangle filter in both directions (9/16, 3/16, 3/16, 1/16), vertical neighbours from the row
    above / below with the image's first / last row replicated (jdmainct.c context rows)."""
    up, dn = np.concatenate([p[:1], p[:-1]], 0), np.concatenate([p[1:], p[-1:]], 0)
    out = np.zeros((2 * p.shape[0], 2 * p.shape[1]), np.int32)
    for v, near_far in enumerate((3 * p + up, 3 * p + dn)):
        s = near_far.astype(np.int64)
        left, right = np.concatenate([s[:, :1], s[:, :-1]], 1), np.concatenate([s[:, 1:], s[:, -1:]], 1)
        row = np.zeros((s.shape[0], 2 * s.shape[1]), np.int64)
        row[:, 0::2] = (3 * s + left + 8) >> 4
        row[:, 1::2] = (3 * s + right + 7) >> 4
        row[:, 0], row[:, -1] = (s[:, 0] * 4 + 8) >> 4, (s[:, -1] * 4 + 7) >> 4
        out[v::2] = row
    return out


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert (16-bit fixed-point tables)."""
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((fix(1.40200) * cr + 32768) >> 16)
    g = y + ((-fix(0.34414) * cb + 32768 - fix(0.71414) * cr) >> 16)
    b = y + ((fix(1.77200) * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(data: bytes) -> np.ndarray:
    """JFIF file -> H x W x 3 uint8 RGB as libjpeg(-turbo) decodes it by default (grayscale files come back with three equal channels,
    like cv2.imread's default flag)."""
    info = parse(data)
    coefs = decode_coefficients(data, info)
    h, w = info["h"], info["w"]
    hmax, vmax = max(c["hs"] for c in info["comps"]), max(c["vs"] for c in info["comps"])
    planes = []
    for c, co in zip(info["comps"], coefs):
        p = idct_islow(co, info["qt"][c["tq"]])
        ch, cw = -(-h * c["vs"] // vmax), -(-w * c["hs"] // hmax)          # the component's own size (jdmaster.c downsampled_height / width)
        p = p[:ch, :cw]
        fancy = cw > 2                                   # jdsample.c jinit_upsampler: narrower components are replicated, not filtered
        if c["hs"] == hmax and c["vs"] == vmax:
            pass
        elif c["hs"] * 2 == hmax and c["vs"] * 2 == vmax:
            p = h2v2_fancy(p) if fancy else np.repeat(np.repeat(p, 2, 0), 2, 1)
        elif c["hs"] * 2 == hmax and c["vs"] == vmax:
            p = h2v1_fancy(p) if fancy else np.repeat(p, 2, 1)
        else:
            raise ValueError("unsupported sampling factors")
        planes.append(p[:h, :w])
    if len(planes) == 1:
        return np.repeat(planes[0][..., None], 3, -1).astype(np.uint8)
    return ycc_to_rgb(*planes)
