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
