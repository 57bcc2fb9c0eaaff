// jpeg_gray.cpp -- dependency-free JPEG decoder of the drop-in host: the LUMA plane as 8-bit grey (DecodeJpegGray) or
// interleaved BGR (DecodeJpegBGR), byte for byte what libjpeg -- i.e. cv::imread -- returns.
//
// The reference reads `images/%08d.jpg` with cv::imread(IMREAD_GRAYSCALE) (APD.cpp:410-427), i.e.
// libjpeg decoding straight to one channel: the Y component of a YCbCr file, no RGB round trip
// (SURVEY.md Appendix E), and with IMREAD_COLOR for the fusion's point colours (APD.cpp:859).  Neither OpenCV nor libjpeg
// headers exist in this image, so this file restates the published algorithms: ITU-T T.81 Huffman decoding, sequential
// (Annex F) and progressive (Annex G: spectral selection + successive approximation), the "islow" accurate integer inverse
// DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants) that libjpeg uses by default, its fancy chroma upsampling and its
// fixed-point YCbCr -> RGB conversion.  For grey output the chroma blocks of a sequential file are entropy-decoded (to keep
// the bit stream in step) and dropped.
// Supported: SOF0/SOF1/SOF2 8-bit, 1 or 3 components, sampling factors 1..4, restart intervals.
// Not supported (returns false): arithmetic coding, lossless / hierarchical frames, 12-bit samples, a second SOF marker,
// and INCOMPLETE progressive files (no EOI, or low-frequency coefficients not refined to full precision): libjpeg
// would show those with inter-block smoothing, i.e. other pixels than a plain IDCT of what was read.
// Every table index that comes from the file is bounds-checked; tests/test_host_io.py fuzzes the decoder under ASan + UBSan.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct HuffTable {
    // canonical Huffman decoding tables (T.81 Annex F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    uint16_t look[512];  // 9-bit lookahead: (code length << 8) | symbol, 0 = longer code
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;

    void fill()
    {
        while (bits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int b2 = (p < end) ? *p : 0xD9;
                    if (b2 == 0x00) {
                        ++p;  // stuffed zero
                    } else {
                        --p;  // a marker: feed zeros from here on
                        hit_marker = true;
                        b = 0;
                    }
                }
            }
            acc |= (uint32_t)b << (24 - bits);
            bits += 8;
        }
    }
    int get_bit()
    {
        if (bits < 1) {
            fill();
        }
        const int v = (int)(acc >> 31);
        acc <<= 1;
        --bits;
        return v;
    }
    int get_bits(int n)
    {
        if (n == 0) {
            return 0;
        }
        if (bits < n) {
            fill();
        }
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        bits -= n;
        return v;
    }
    void reset()
    {
        acc = 0;
        bits = 0;
        hit_marker = false;
    }
};

const int kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool build_huff(HuffTable &t, const uint8_t counts[16], const uint8_t *vals, int nvals)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t.valptr[l] = k;
        t.mincode[l] = code;
        code += counts[l - 1];
        k += counts[l - 1];
        t.maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    if (k != nvals || nvals > 256) {
        return false;
    }
    memcpy(t.vals, vals, (size_t)nvals);
    memset(t.look, 0, sizeof(t.look));
    for (int l = 1; l <= 9; ++l) {
        for (int i = 0; i < counts[l - 1]; ++i) {
            const int c = t.mincode[l] + i;  // the l-bit code of symbol vals[valptr[l] + i]
            const int first = c << (9 - l);
            for (int f = 0; f < (1 << (9 - l)); ++f) {
                t.look[first + f] = (uint16_t)((l << 8) | t.vals[t.valptr[l] + i]);
            }
        }
    }
    t.present = true;
    return true;
}

int decode_symbol(BitReader &br, const HuffTable &t)
{
    if (br.bits < 16) {
        br.fill();
    }
    const uint16_t e = t.look[br.acc >> 23];
    if (e) {
        const int l = e >> 8;
        br.acc <<= l;
        br.bits -= l;
        return e & 0xFF;
    }
    const int peek = (int)(br.acc >> 16);
    for (int l = 10; l <= 16; ++l) {
        const int code = peek >> (16 - l);
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l]) {
            br.acc <<= l;
            br.bits -= l;
            return t.vals[t.valptr[l] + code - t.mincode[l]];
        }
    }
    return -1;
}

inline int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

// libjpeg "islow" inverse DCT: CONST_BITS = 13, PASS1_BITS = 2; coefficients already dequantised.
void idct_islow(const int *coef, uint8_t *out, int stride)
{
    const int FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270,
              FIX_0_899976223 = 7373, FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137,
              FIX_1_961570560 = 16069, FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;
    const int CONST_BITS = 13, PASS1_BITS = 2;
    long ws[64];
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    for (int c = 0; c < 8; ++c) {  // columns
        const int *in = coef + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            const long dc = (long)in[0] * (1L << PASS1_BITS);
            for (int r = 0; r < 8; ++r) {
                ws[r * 8 + c] = dc;
            }
            continue;
        }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * FIX_0_541196100;
        long tmp2 = z1 + z3 * (-FIX_1_847759065);
        long tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in[0];
        z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CONST_BITS);
        long tmp1 = (z2 - z3) * (1L << CONST_BITS);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56];
        tmp1 = in[40];
        tmp2 = in[24];
        tmp3 = in[8];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336;
        tmp1 *= FIX_2_053119869;
        tmp2 *= FIX_3_072711026;
        tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223;
        z2 *= -FIX_2_562915447;
        z3 *= -FIX_1_961570560;
        z4 *= -FIX_0_390180644;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        ws[0 * 8 + c] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        ws[7 * 8 + c] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[1 * 8 + c] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        ws[6 * 8 + c] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[2 * 8 + c] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        ws[5 * 8 + c] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[3 * 8 + c] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        ws[4 * 8 + c] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {  // rows
        const long *w = ws + r * 8;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * FIX_0_541196100;
        long tmp2 = z1 + z3 * (-FIX_1_847759065);
        long tmp3 = z1 + z2 * FIX_0_765366865;
        long tmp0 = (w[0] + w[4]) * (1L << CONST_BITS);
        long tmp1 = (w[0] - w[4]) * (1L << CONST_BITS);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7];
        tmp1 = w[5];
        tmp2 = w[3];
        tmp3 = w[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336;
        tmp1 *= FIX_2_053119869;
        tmp2 *= FIX_3_072711026;
        tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223;
        z2 *= -FIX_2_562915447;
        z3 *= -FIX_1_961570560;
        z4 *= -FIX_0_390180644;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        const int sh = CONST_BITS + PASS1_BITS + 3;
        auto put = [&](int c, long v) {
            long s = descale(v, sh) + 128;
            out[r * stride + c] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        };
        put(0, tmp10 + tmp3);
        put(7, tmp10 - tmp3);
        put(1, tmp11 + tmp2);
        put(6, tmp11 - tmp2);
        put(2, tmp12 + tmp1);
        put(5, tmp12 - tmp1);
        put(3, tmp13 + tmp0);
        put(4, tmp13 - tmp0);
    }
}

inline int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

}  // namespace

namespace {

// libjpeg's "fancy" (triangle filter) chroma upsampling, jdsample.c: h2v1_fancy_upsample / h2v2_fancy_upsample, and
// plain replication for every other integral factor (int_upsample).  `in` is the decoded component plane (pitch in_w,
// real size cw x ch: the columns / rows beyond belong to the encoder's block padding and are not used; rows above the
// first and below the last real row are copies of them, jdmainct.c), `out` the full-resolution plane.
void upsample_component(const std::vector<uint8_t> &in, int in_w, int cw, int ch, int hexp, int vexp, int width, int height,
                        std::vector<uint8_t> &out)
{
    out.assign((size_t)width * height, 0);
    auto row = [&](int r) { return &in[(size_t)(r < 0 ? 0 : (r >= ch ? ch - 1 : r)) * in_w]; };
    if (hexp == 1 && vexp == 1) {
        for (int y = 0; y < height; ++y) {
            memcpy(&out[(size_t)y * width], row(y), (size_t)width);
        }
        return;
    }
    const bool fancy_h2v1 = hexp == 2 && vexp == 1 && cw > 2;
    const bool fancy_h2v2 = hexp == 2 && vexp == 2 && cw > 2;
    if (!fancy_h2v1 && !fancy_h2v2) {  // replication
        for (int y = 0; y < height; ++y) {
            const uint8_t *r = row(y / vexp);
            for (int x = 0; x < width; ++x) {
                out[(size_t)y * width + x] = r[x / hexp];
            }
        }
        return;
    }
    std::vector<uint8_t> line((size_t)2 * cw);
    if (fancy_h2v1) {
        for (int y = 0; y < height; ++y) {
            const uint8_t *r = row(y);
            line[0] = r[0];
            line[1] = (uint8_t)((r[0] * 3 + r[1] + 2) >> 2);
            for (int i = 1; i < cw - 1; ++i) {
                const int v = r[i] * 3;
                line[2 * i] = (uint8_t)((v + r[i - 1] + 1) >> 2);
                line[2 * i + 1] = (uint8_t)((v + r[i + 1] + 2) >> 2);
            }
            line[2 * cw - 2] = (uint8_t)((r[cw - 1] * 3 + r[cw - 2] + 1) >> 2);
            line[2 * cw - 1] = r[cw - 1];
            memcpy(&out[(size_t)y * width], line.data(), (size_t)width);
        }
        return;
    }
    for (int y = 0; y < height; ++y) {  // h2v2: nearer input row 3/4, farther 1/4, then the same filter along the row
        const int r0 = y >> 1;
        const uint8_t *near = row(r0), *far = row((y & 1) ? r0 + 1 : r0 - 1);
        int thiscol = near[0] * 3 + far[0], nextcol = near[1] * 3 + far[1], lastcol;
        line[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
        line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
        for (int i = 1; i < cw - 1; ++i) {
            nextcol = near[i + 1] * 3 + far[i + 1];
            line[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            line[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
        }
        line[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        line[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        memcpy(&out[(size_t)y * width], line.data(), (size_t)width);
    }
}

inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Whole-file decoder behind DecodeJpegGray / DecodeJpegBGR.
bool decode_jpeg(const uint8_t *data, size_t size, bool want_colour, std::vector<uint8_t> &pixels, int &width, int &height)
{
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) {
        return false;
    }
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[4];
    int ncomp = 0, restart_interval = 0;
    int hmax = 1, vmax = 1;
    bool have_sof = false;
    bool progressive = false;  // SOF2: coefficients are collected over several scans, the IDCT runs after the last one
    std::vector<int16_t> coefs[4];          // progressive only: [block][64] in natural order, blocks_w x blocks_h per component
    int8_t coef_al[4][64];                  // progressive only: successive-approximation bit each zig-zag position was last coded at (-1 = never)
    bool saw_eoi = false;
    int blocks_w[4] = {0, 0, 0, 0}, blocks_h[4] = {0, 0, 0, 0};
    uint16_t latched_qt[4][64];             // quantisation table of a component as of its first scan (jdinput.c latch_quant_tables)
    bool qt_latched[4] = {false, false, false, false};
    bool scanned[4] = {false, false, false, false};
    int plane_w[4] = {0, 0, 0, 0}, plane_h[4] = {0, 0, 0, 0};
    static thread_local std::vector<uint8_t> planes[4];
    // grey or BGR pixels from the decoded component planes (shared by the sequential and the progressive path)
    const int yc = 0;  // component 0 is Y by JFIF convention
    auto finish = [&]() -> bool {
        if (!want_colour || ncomp == 1) {
            const int ch = want_colour ? 3 : 1;
            pixels.resize((size_t)width * height * ch);
            for (int y = 0; y < height; ++y) {
                const uint8_t *src = &planes[yc][(size_t)y * plane_w[yc]];
                if (ch == 1) {
                    memcpy(&pixels[(size_t)y * width], src, (size_t)width);
                } else {
                    for (int x = 0; x < width; ++x) {  // grey file read as colour: B = G = R = Y
                        pixels[((size_t)y * width + x) * 3 + 0] = pixels[((size_t)y * width + x) * 3 + 1] =
                            pixels[((size_t)y * width + x) * 3 + 2] = src[x];
                    }
                }
            }
            return true;
        }
        // YCbCr -> BGR: upsample the chroma planes (jdsample.c), then libjpeg's 16-bit fixed-point conversion
        // (jdcolor.c: build_ycc_rgb_table / ycc_rgb_convert), stored blue first like cv::imread(IMREAD_COLOR)
        std::vector<uint8_t> full[3];
        for (int c = 0; c < 3; ++c) {
            if (hmax % comp[c].h != 0 || vmax % comp[c].v != 0) {
                return false;  // fractional sampling ratios: not built
            }
            const int cw = (width * comp[c].h + hmax - 1) / hmax, chh = (height * comp[c].v + vmax - 1) / vmax;
            upsample_component(planes[c], plane_w[c], cw, chh, hmax / comp[c].h, vmax / comp[c].v, width, height, full[c]);
        }
        pixels.resize((size_t)width * height * 3);
        const int kScale = 16, kHalf = 1 << 15;
        const int f_1_40200 = (int)(1.40200 * 65536 + 0.5), f_1_77200 = (int)(1.77200 * 65536 + 0.5);
        const int f_0_71414 = (int)(0.71414 * 65536 + 0.5), f_0_34414 = (int)(0.34414 * 65536 + 0.5);
        for (size_t i = 0; i < (size_t)width * height; ++i) {
            const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
            const int r = y + ((f_1_40200 * cr + kHalf) >> kScale);
            const int g = y + (((-f_0_34414) * cb + kHalf + (-f_0_71414) * cr) >> kScale);
            const int b = y + ((f_1_77200 * cb + kHalf) >> kScale);
            pixels[3 * i + 0] = clamp_u8(b);
            pixels[3 * i + 1] = clamp_u8(g);
            pixels[3 * i + 2] = clamp_u8(r);
        }
        return true;
    };
    size_t pos = 2;
    while (pos + 4 <= size) {
        if (data[pos] != 0xFF) {
            ++pos;
            continue;
        }
        const int marker = data[pos + 1];
        pos += 2;
        if (marker == 0xD8 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0xFF) {
            if (marker == 0xFF) {
                --pos;
            }
            continue;
        }
        if (marker == 0xD9) {
            saw_eoi = true;
            break;
        }
        if (pos + 2 > size) {
            return false;
        }
        const int len = be16(data + pos);
        if (len < 2 || pos + len > size) {
            return false;
        }
        const uint8_t *seg = data + pos + 2;
        const int seglen = len - 2;
        if (marker == 0xDB) {  // DQT
            int i = 0;
            while (i < seglen) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > seglen) {
                    return false;
                }
                for (int k = 0; k < 64; ++k) {
                    qt[tq][kZigZag[k]] = pq ? (uint16_t)be16(seg + i + 2 * k) : seg[i + k];
                }
                i += pq ? 128 : 64;
                qt_present[tq] = true;
            }
        } else if (marker == 0xC4) {  // DHT
            int i = 0;
            while (i + 17 <= seglen) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                int n = 0;
                for (int k = 0; k < 16; ++k) {
                    n += seg[i + 1 + k];
                }
                if (th > 3 || tc > 1 || i + 17 + n > seglen) {
                    return false;
                }
                if (!build_huff(tc ? ac[th] : dc[th], seg + i + 1, seg + i + 17, n)) {
                    return false;
                }
                i += 17 + n;
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {  // SOF0 / SOF1 (sequential), SOF2 (progressive)
            if (have_sof) {
                return false;  // a second frame header would resize planes and coefficient arrays under the scans (libjpeg: JERR_SOF_DUPLICATE)
            }
            progressive = marker == 0xC2;
            if (seglen < 6 || seg[0] != 8) {
                return false;
            }
            height = be16(seg + 1);
            width = be16(seg + 3);
            ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || seglen < 6 + 3 * ncomp || width <= 0 || height <= 0 ||
                (long long)width * height > (1LL << 28)) {  // 268 Mpix: ten times the largest ETH3D image; keeps a corrupt header from asking for 4 Gpix
                return false;
            }
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = seg[6 + 3 * c];
                comp[c].h = seg[7 + 3 * c] >> 4;
                comp[c].v = seg[7 + 3 * c] & 15;
                comp[c].tq = seg[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].v < 1 || comp[c].h > 4 || comp[c].v > 4 || comp[c].tq > 3) {  // T.81 B.2.2
                    return false;
                }
                hmax = comp[c].h > hmax ? comp[c].h : hmax;
                vmax = comp[c].v > vmax ? comp[c].v : vmax;
            }
            have_sof = true;
        } else if (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
            return false;  // lossless / hierarchical / arithmetic coding: not built
        } else if (marker == 0xDD) {  // DRI
            if (seglen < 2) {
                return false;
            }
            restart_interval = be16(seg);
        } else if (marker == 0xDA) {  // SOS: baseline has a single interleaved scan (or one per component)
            if (!have_sof || seglen < 1) {
                return false;
            }
            const int ns = seg[0];
            if (ns < 1 || ns > ncomp || seglen < 1 + 2 * ns + 3) {
                return false;
            }
            int scan_comp[4];
            for (int s = 0; s < ns; ++s) {
                int ci = -1;
                for (int c = 0; c < ncomp; ++c) {
                    if (comp[c].id == seg[1 + 2 * s]) {
                        ci = c;
                    }
                }
                if (ci < 0) {
                    return false;
                }
                comp[ci].td = seg[2 + 2 * s] >> 4;
                comp[ci].ta = seg[2 + 2 * s] & 15;
                if (comp[ci].td > 3 || comp[ci].ta > 3) {
                    return false;  // baseline allows table ids 0..3 (T.81 B.2.3); anything else would index past dc[] / ac[]
                }
                scan_comp[s] = ci;
            }
            pos += len;
            // component planes padded to whole MCUs (grey output decodes the luma plane only)
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
            for (int c = 0; c < ncomp; ++c) {
                plane_w[c] = mcus_x * comp[c].h * 8;
                plane_h[c] = mcus_y * comp[c].v * 8;
                if ((c == yc || want_colour) && planes[c].size() != (size_t)plane_w[c] * plane_h[c]) {
                    planes[c].assign((size_t)plane_w[c] * plane_h[c], 0);
                }
            }
            BitReader br{data + pos, data + size};
            for (int c = 0; c < ncomp; ++c) {
                comp[c].dc_pred = 0;
            }
            if (progressive) {
                // T.81 Annex G: one scan = one spectral band [Ss, Se] at one successive-approximation step (Ah -> Al) of the
                // listed components; DC scans may interleave components, AC scans carry exactly one.
                const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
                if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1)) {
                    return false;
                }
                for (int c = 0; c < ncomp; ++c) {
                    blocks_w[c] = mcus_x * comp[c].h;
                    blocks_h[c] = mcus_y * comp[c].v;
                    if (coefs[c].empty()) {
                        coefs[c].assign((size_t)blocks_w[c] * blocks_h[c] * 64, 0);
                        memset(coef_al[c], 0xFF, sizeof(coef_al[c]));
                    }
                    if (coefs[c].size() != (size_t)blocks_w[c] * blocks_h[c] * 64) {
                        return false;  // cannot happen with a single SOF; guards every blk pointer below
                    }
                }
                for (int s2 = 0; s2 < ns; ++s2) {  // successive-approximation bookkeeping: which precision each coefficient has reached
                    for (int k = Ss; k <= Se; ++k) {
                        coef_al[scan_comp[s2]][k] = (int8_t)Al;
                    }
                }
                for (int s2 = 0; s2 < ns; ++s2) {
                    const int c = scan_comp[s2];
                    if (!qt_latched[c]) {
                        if (!qt_present[comp[c].tq]) {
                            return false;
                        }
                        memcpy(latched_qt[c], qt[comp[c].tq], sizeof(latched_qt[c]));
                        qt_latched[c] = true;
                    }
                    if ((Ss == 0 && Ah == 0 && !dc[comp[c].td].present) || (Ss > 0 && !ac[comp[c].ta].present)) {
                        return false;
                    }
                }
                const bool interleaved_scan = ns > 1;
                int units_x, units_y;
                if (interleaved_scan) {
                    units_x = mcus_x;
                    units_y = mcus_y;
                } else {  // a one-component scan walks the component's own blocks: only those that hold image samples
                    const Component &cc = comp[scan_comp[0]];
                    units_x = ((width * cc.h + hmax - 1) / hmax + 7) / 8;
                    units_y = ((height * cc.v + vmax - 1) / vmax + 7) / 8;
                }
                int eobrun = 0;
                const int p1 = 1 << Al, m1 = -(1 << Al);
                const long total_units = (long)units_x * units_y;
                for (long u = 0; u < total_units; ++u) {
                    if (restart_interval && u > 0 && (u % restart_interval) == 0) {
                        br.reset();
                        const uint8_t *q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                            ++q;
                        }
                        if (q + 1 >= br.end) {
                            return false;
                        }
                        br.p = q + 2;
                        for (int c = 0; c < ncomp; ++c) {
                            comp[c].dc_pred = 0;
                        }
                        eobrun = 0;
                    }
                    for (int s2 = 0; s2 < ns; ++s2) {
                        Component &cc = comp[scan_comp[s2]];
                        const int c = scan_comp[s2];
                        const int nbh = interleaved_scan ? cc.h : 1, nbv = interleaved_scan ? cc.v : 1;
                        for (int by = 0; by < nbv; ++by) {
                            for (int bx = 0; bx < nbh; ++bx) {
                                const long col = interleaved_scan ? (u % units_x) * cc.h + bx : (u % units_x);
                                const long row = interleaved_scan ? (u / units_x) * cc.v + by : (u / units_x);
                                if (col >= blocks_w[c] || row >= blocks_h[c]) {
                                    return false;
                                }
                                int16_t *blk = &coefs[c][((size_t)row * blocks_w[c] + col) * 64];
                                if (Ss == 0) {
                                    if (Ah == 0) {  // DC first scan (G.1.2.1)
                                        const int t = decode_symbol(br, dc[cc.td]);
                                        if (t < 0 || t > 15) {
                                            return false;
                                        }
                                        cc.dc_pred = (int)((unsigned)cc.dc_pred + (unsigned)extend(br.get_bits(t), t));  // wraps instead of overflowing on hostile files
                                        blk[0] = (int16_t)((unsigned)cc.dc_pred << Al);
                                    } else if (br.get_bit()) {  // DC refinement: one more bit
                                        blk[0] = (int16_t)(blk[0] | p1);
                                    }
                                    continue;
                                }
                                const HuffTable &tab = ac[cc.ta];
                                if (Ah == 0) {  // AC first scan (G.1.2.2)
                                    if (eobrun > 0) {
                                        --eobrun;
                                        continue;
                                    }
                                    for (int k = Ss; k <= Se; ++k) {
                                        const int rs = decode_symbol(br, tab);
                                        if (rs < 0) {
                                            return false;
                                        }
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) {
                                            if (r == 15) {
                                                k += 15;
                                                continue;
                                            }
                                            eobrun = (1 << r) - 1;
                                            if (r) {
                                                eobrun += br.get_bits(r);
                                            }
                                            break;
                                        }
                                        k += r;
                                        if (k > Se) {
                                            return false;
                                        }
                                        blk[kZigZag[k]] = (int16_t)(extend(br.get_bits(sz), sz) * (1 << Al));
                                    }
                                    continue;
                                }
                                // AC refinement scan (G.1.2.3), the control flow of libjpeg's decode_mcu_AC_refine
                                int k = Ss;
                                if (eobrun == 0) {
                                    for (; k <= Se; ++k) {
                                        const int rs = decode_symbol(br, tab);
                                        if (rs < 0) {
                                            return false;
                                        }
                                        int r = rs >> 4, sval = rs & 15;
                                        if (sval) {
                                            if (sval != 1) {
                                                return false;
                                            }
                                            sval = br.get_bit() ? p1 : m1;
                                        } else if (r != 15) {
                                            eobrun = 1 << r;
                                            if (r) {
                                                eobrun += br.get_bits(r);
                                            }
                                            break;
                                        }
                                        do {  // pass the already non-zero coefficients (one correction bit each) and r zero ones
                                            int16_t &cf = blk[kZigZag[k]];
                                            if (cf != 0) {
                                                if (br.get_bit() && (cf & p1) == 0) {
                                                    cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                                                }
                                            } else if (--r < 0) {
                                                break;
                                            }
                                            ++k;
                                        } while (k <= Se);
                                        if (sval && k <= Se) {
                                            blk[kZigZag[k]] = (int16_t)sval;
                                        }
                                    }
                                }
                                if (eobrun > 0) {
                                    for (; k <= Se; ++k) {
                                        int16_t &cf = blk[kZigZag[k]];
                                        if (cf != 0 && br.get_bit() && (cf & p1) == 0) {
                                            cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                                        }
                                    }
                                    --eobrun;
                                }
                            }
                        }
                    }
                }
                const uint8_t *q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) {
                    ++q;
                }
                pos = (size_t)(q - data);
                for (int s2 = 0; s2 < ns; ++s2) {
                    scanned[scan_comp[s2]] = true;
                }
                continue;  // more scans follow; the planes are produced after the last one
            }
            int restart_count = 0;
            const bool interleaved = ns > 1;
            int blocks_x = 0, blocks_y = 0;
            if (!interleaved) {  // non-interleaved scan of one component: its own block raster
                const Component &cc = comp[scan_comp[0]];
                blocks_x = ((width * cc.h + hmax - 1) / hmax + 7) / 8;
                blocks_y = ((height * cc.v + vmax - 1) / vmax + 7) / 8;
            }
            const int units = interleaved ? mcus_x * mcus_y : blocks_x * blocks_y;
            int coef[64];
            for (int u = 0; u < units; ++u) {
                if (restart_interval && u > 0 && (u % restart_interval) == 0) {
                    // align to the RSTn marker
                    br.reset();
                    const uint8_t *q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                        ++q;
                    }
                    if (q + 1 >= br.end) {
                        return false;
                    }
                    br.p = q + 2;
                    for (int c = 0; c < ncomp; ++c) {
                        comp[c].dc_pred = 0;
                    }
                    ++restart_count;
                }
                for (int s = 0; s < ns; ++s) {
                    Component &cc = comp[scan_comp[s]];
                    const int nbh = interleaved ? cc.h : 1, nbv = interleaved ? cc.v : 1;
                    if (!dc[cc.td].present || !ac[cc.ta].present || !qt_present[cc.tq]) {
                        return false;
                    }
                    for (int by = 0; by < nbv; ++by) {
                        for (int bx = 0; bx < nbh; ++bx) {
                            memset(coef, 0, sizeof(coef));
                            int t = decode_symbol(br, dc[cc.td]);
                            if (t < 0 || t > 11) {
                                return false;
                            }
                            cc.dc_pred = (int)((unsigned)cc.dc_pred + (unsigned)extend(br.get_bits(t), t));  // wraps instead of overflowing on hostile files
                            coef[0] = (int)((unsigned)cc.dc_pred * (unsigned)qt[cc.tq][0]);
                            for (int k = 1; k < 64;) {
                                const int rs = decode_symbol(br, ac[cc.ta]);
                                if (rs < 0) {
                                    return false;
                                }
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) {
                                    if (r == 15) {
                                        k += 16;
                                        continue;
                                    }
                                    break;  // EOB
                                }
                                k += r;
                                if (k > 63) {
                                    return false;
                                }
                                coef[kZigZag[k]] = extend(br.get_bits(sz), sz) * qt[cc.tq][kZigZag[k]];
                                ++k;
                            }
                            const int pc = scan_comp[s];
                            if (pc == yc || want_colour) {
                                int px, py;
                                if (interleaved) {
                                    px = ((u % mcus_x) * cc.h + bx) * 8;
                                    py = ((u / mcus_x) * cc.v + by) * 8;
                                } else {
                                    px = (u % blocks_x) * 8;
                                    py = (u / blocks_x) * 8;
                                }
                                if (px + 8 <= plane_w[pc] && py + 8 <= plane_h[pc]) {
                                    idct_islow(coef, &planes[pc][(size_t)py * plane_w[pc] + px], plane_w[pc]);
                                }
                            }
                        }
                    }
                }
            }
            (void)restart_count;
            // continue parsing after the entropy-coded segment (further scans of a non-interleaved file)
            const uint8_t *q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) {
                ++q;
            }
            pos = (size_t)(q - data);
            for (int s = 0; s < ns; ++s) {
                scanned[scan_comp[s]] = true;
            }
            bool done = scanned[yc];
            if (want_colour) {
                for (int c = 0; c < ncomp; ++c) {
                    done = done && scanned[c];
                }
            }
            if (!done) {
                continue;
            }
            return finish();
        }
        pos += len;
    }
    if (progressive && have_sof && scanned[yc]) {
        // All scans read: dequantise + IDCT.  Only files libjpeg decodes WITHOUT inter-block smoothing are accepted: EOI
        // reached, and the DC and the first five AC coefficients in zig-zag order (natural 1, 8, 16, 9, 2 -- the ones jdcoefct.c's
        // smoothing_ok() inspects) of every needed component refined down to bit 0.  For anything else libjpeg
        // interpolates the missing AC precision from the neighbouring blocks' DC values, which is not built here: such a
        // file is refused instead of being decoded to different pixels.
        for (size_t q = pos; !saw_eoi && q + 2 <= size; ++q) {  // the marker loop stops four bytes before the end: EOI is usually in them
            saw_eoi = data[q] == 0xFF && data[q + 1] == 0xD9;
        }
        if (!saw_eoi) {
            return false;
        }
        for (int c = 0; c < ncomp; ++c) {
            if (!(c == yc || want_colour)) {
                continue;
            }
            if (!scanned[c] || !qt_latched[c] || coefs[c].size() != (size_t)blocks_w[c] * blocks_h[c] * 64 ||
                planes[c].size() != (size_t)plane_w[c] * plane_h[c]) {
                return false;
            }
            for (int k = 0; k < 6; ++k) {  // coef_al is indexed by zig-zag position, like Ss / Se
                if (coef_al[c][k] != 0) {
                    return false;
                }
            }
            int block[64];
            for (int by = 0; by < blocks_h[c]; ++by) {
                for (int bx = 0; bx < blocks_w[c]; ++bx) {
                    const int16_t *blk = &coefs[c][((size_t)by * blocks_w[c] + bx) * 64];
                    for (int k = 0; k < 64; ++k) {
                        block[k] = blk[k] * latched_qt[c][k];
                    }
                    idct_islow(block, &planes[c][(size_t)by * 8 * plane_w[c] + (size_t)bx * 8], plane_w[c]);
                }
            }
        }
        return finish();
    }
    return false;
}

}  // namespace

// Decodes `data` (a whole .jpg file) to grey; returns false on unsupported / corrupt input.
bool DecodeJpegGray(const uint8_t *data, size_t size, std::vector<uint8_t> &gray, int &width, int &height)
{
    return decode_jpeg(data, size, false, gray, width, height);
}

// Decodes to interleaved 8-bit BGR, what cv::imread(IMREAD_COLOR) returns for the file (the reference's fusion colours,
// APD.cpp:859): libjpeg's islow IDCT, fancy chroma upsampling and fixed-point YCbCr -> RGB.
bool DecodeJpegBGR(const uint8_t *data, size_t size, std::vector<uint8_t> &bgr, int &width, int &height)
{
    return decode_jpeg(data, size, true, bgr, width, height);
}
