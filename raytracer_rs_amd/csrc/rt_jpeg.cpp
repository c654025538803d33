// Canvas::to_rgb_image + Canvas::to_jpeg (rendering/canvas.rs:92-114 of the reference): the canvas quantised to 8 bits
// per channel and written as a baseline JPEG at quality 100 through the `image` crate's encoder (0.24.6, pinned by the
// reference's Cargo.lock:234-235; `ImageOutputFormat::Jpeg(100)`).  That crate's source is not part of the reference
// checkout, so this file restates its published algorithm, host side:
//
//   * file layout: SOI, APP0 "JFIF" 1.02 (density 1:1, no units), SOF0 (8 bit, 3 components, all 1x1: no chroma
//     subsampling), two DQT segments, four DHT segments (the standard tables of ITU-T T.81 annex K.3), SOS, entropy
//     coded data, EOI;
//   * quantisation tables at quality 100: the libjpeg scaling (200 - 2q = 0) clamps every entry to 1;
//   * colour: RGB -> YCbCr in f32 with the JFIF coefficients scaled for a maximum of 255, truncated to u8;
//   * blocks of 8x8 pixels, the image's edge pixels repeated beyond its borders; forward DCT = the integer "slow" DCT
//     of libjpeg 9 (13-bit constants, 2 pass-1 bits) with the level shift folded into the DC term, its results scaled
//     by 8 and brought back by a truncating division;
//   * Huffman coding with the standard tables, an EOB only when a block's last coefficient is zero, bytes 0xFF
//     stuffed, the last byte padded with ones.
//
// What pins it: the reference ships both the PPM and the JPEG of five of its renderings (outputs/chapter_5_circle,
// chapter_6_sphere, chapter_10_patterns, chapter_11_reflections, chapter_13_rocket); fed the PPM's pixels this writer
// reproduces each of those JPEG files byte for byte (tests/test_jpeg.py).  One detail only that comparison could
// settle: in the second (column) pass the rounding term of the odd part is 1 << (CONST_BITS - PASS1_BITS - 1) as in
// the first pass, not libjpeg's 1 << (CONST_BITS + PASS1_BITS - 1).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_rs_amd.h"
#include "rt_capi_internal.hpp"

namespace rt {
namespace {

// ITU-T T.81 annex K.3: code lengths (number of codes of length 1..16) and values, in code order
const uint8_t LUMA_DC_LEN[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t CHROMA_DC_LEN[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t DC_VAL[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t LUMA_AC_LEN[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t LUMA_AC_VAL[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t CHROMA_AC_LEN[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t CHROMA_AC_VAL[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
// position in the 8x8 block of the k-th coefficient in zigzag order
const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    uint16_t code[256];
    uint8_t len[256];
    Huffman(const uint8_t *lengths, const uint8_t *values) {
        std::memset(code, 0, sizeof code);
        std::memset(len, 0, sizeof len);
        unsigned c = 0, k = 0;
        for (unsigned l = 1; l <= 16; ++l) {
            for (unsigned n = 0; n < lengths[l - 1]; ++n, ++k, ++c) {
                code[values[k]] = (uint16_t)c;
                len[values[k]] = (uint8_t)l;
            }
            c <<= 1;
        }
    }
};

struct BitWriter {
    std::string &out;
    uint32_t acc = 0;
    int bits = 0;
    explicit BitWriter(std::string &o) : out(o) {}
    void put(uint32_t value, int n) {  // n <= 16
        if (n == 0) return;
        acc = (acc << n) | (value & ((1u << n) - 1u));
        bits += n;
        while (bits >= 8) {
            const uint8_t byte = (uint8_t)(acc >> (bits - 8));
            out.push_back((char)byte);
            if (byte == 0xff) out.push_back('\0');
            bits -= 8;
        }
        acc &= (1u << bits) - 1u;
    }
    void pad() {
        if (bits) put(0x7f, 7);  // ones up to the byte boundary (more than enough: put() keeps what is left over below 8 bits)
        acc = 0;
        bits = 0;
    }
};

// Forward DCT on 8x8 samples 0..255, results scaled by 8 (the `image` crate's transform::fdct).
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int32_t FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
                  FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069, FIX_2_053119869 = 16819,
                  FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;

void fdct(const uint8_t *s, int32_t *c) {
    for (int y = 0; y < 8; ++y) {  // rows: scaled by sqrt(8) and by 2^PASS1_BITS
        const uint8_t *r = s + y * 8;
        int32_t *o = c + y * 8;
        int32_t t0 = r[0] + r[7], t1 = r[1] + r[6], t2 = r[2] + r[5], t3 = r[3] + r[4];
        const int32_t t10 = t0 + t3, t11 = t1 + t2;
        int32_t t12 = t0 - t3, t13 = t1 - t2;
        t0 = r[0] - r[7], t1 = r[1] - r[6], t2 = r[2] - r[5], t3 = r[3] - r[4];
        o[0] = (t10 + t11 - 8 * 128) * (1 << PASS1_BITS);  // the level shift of all eight samples
        o[4] = (t10 - t11) * (1 << PASS1_BITS);
        int32_t z1 = (t12 + t13) * FIX_0_541196100 + (1 << (CONST_BITS - PASS1_BITS - 1));
        o[2] = (z1 + t12 * FIX_0_765366865) >> (CONST_BITS - PASS1_BITS);
        o[6] = (z1 - t13 * FIX_1_847759065) >> (CONST_BITS - PASS1_BITS);
        t12 = t0 + t2;
        t13 = t1 + t3;
        z1 = (t12 + t13) * FIX_1_175875602 + (1 << (CONST_BITS - PASS1_BITS - 1));
        t12 = t12 * (-FIX_0_390180644) + z1;
        t13 = t13 * (-FIX_1_961570560) + z1;
        z1 = (t0 + t3) * (-FIX_0_899976223);
        const int32_t a0 = t0 * FIX_1_501321110 + z1 + t12, a3 = t3 * FIX_0_298631336 + z1 + t13;
        z1 = (t1 + t2) * (-FIX_2_562915447);
        const int32_t a1 = t1 * FIX_3_072711026 + z1 + t13, a2 = t2 * FIX_2_053119869 + z1 + t12;
        o[1] = a0 >> (CONST_BITS - PASS1_BITS);
        o[3] = a1 >> (CONST_BITS - PASS1_BITS);
        o[5] = a2 >> (CONST_BITS - PASS1_BITS);
        o[7] = a3 >> (CONST_BITS - PASS1_BITS);
    }
    for (int x = 7; x >= 0; --x) {  // columns: the PASS1_BITS come off, an overall factor of 8 stays
        int32_t *o = c + x;
        int32_t t0 = o[0] + o[56], t1 = o[8] + o[48], t2 = o[16] + o[40], t3 = o[24] + o[32];
        const int32_t t10 = t0 + t3 + (1 << (PASS1_BITS - 1)), t11 = t1 + t2;
        int32_t t12 = t0 - t3, t13 = t1 - t2;
        t0 = o[0] - o[56], t1 = o[8] - o[48], t2 = o[16] - o[40], t3 = o[24] - o[32];
        o[0] = (t10 + t11) >> PASS1_BITS;
        o[32] = (t10 - t11) >> PASS1_BITS;
        int32_t z1 = (t12 + t13) * FIX_0_541196100 + (1 << (CONST_BITS + PASS1_BITS - 1));
        o[16] = (z1 + t12 * FIX_0_765366865) >> (CONST_BITS + PASS1_BITS);
        o[48] = (z1 - t13 * FIX_1_847759065) >> (CONST_BITS + PASS1_BITS);
        t12 = t0 + t2;
        t13 = t1 + t3;
        z1 = (t12 + t13) * FIX_1_175875602 + (1 << (CONST_BITS - PASS1_BITS - 1));  // (sic: the first pass's rounding term)
        t12 = t12 * (-FIX_0_390180644) + z1;
        t13 = t13 * (-FIX_1_961570560) + z1;
        z1 = (t0 + t3) * (-FIX_0_899976223);
        const int32_t a0 = t0 * FIX_1_501321110 + z1 + t12, a3 = t3 * FIX_0_298631336 + z1 + t13;
        z1 = (t1 + t2) * (-FIX_2_562915447);
        const int32_t a1 = t1 * FIX_3_072711026 + z1 + t13, a2 = t2 * FIX_2_053119869 + z1 + t12;
        o[8] = a0 >> (CONST_BITS + PASS1_BITS);
        o[24] = a1 >> (CONST_BITS + PASS1_BITS);
        o[40] = a2 >> (CONST_BITS + PASS1_BITS);
        o[56] = a3 >> (CONST_BITS + PASS1_BITS);
    }
}

// f32 -> u8 the way Rust's `as u8` does it: towards zero, saturating, NaN -> 0
inline uint8_t to_u8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

// the encoder's rgb_to_ycbcr: JFIF coefficients for a maximum of 255, every product and sum in f32, left to right
inline void rgb_to_ycbcr(uint8_t r8, uint8_t g8, uint8_t b8, uint8_t &y, uint8_t &cb, uint8_t &cr) {
    const float max = 255.0f, r = (float)r8, g = (float)g8, b = (float)b8;
    const float fy = 76.245f / max * r + 149.685f / max * g + 29.07f / max * b;
    const float fcb = -43.0185f / max * r - 84.4815f / max * g + 127.5f / max * b + 128.0f;
    const float fcr = 127.5f / max * r - 106.7685f / max * g - 20.7315f / max * b + 128.0f;
    y = to_u8(fy);
    cb = to_u8(fcb);
    cr = to_u8(fcr);
}

inline void coefficient_bits(int32_t v, int &size, uint32_t &bits) {
    uint32_t a = (uint32_t)(v < 0 ? -v : v);
    size = 0;
    while (a) {
        ++size;
        a >>= 1;
    }
    bits = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
}

int32_t write_block(BitWriter &w, const int32_t *block, int32_t prev_dc, const Huffman &dc, const Huffman &ac) {
    int size;
    uint32_t bits;
    coefficient_bits(block[0] - prev_dc, size, bits);
    w.put(dc.code[size], dc.len[size]);
    w.put(bits, size);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        const int32_t v = block[ZIGZAG[k]];
        if (v == 0) {
            ++run;
            continue;
        }
        while (run > 15) {
            w.put(ac.code[0xf0], ac.len[0xf0]);
            run -= 16;
        }
        coefficient_bits(v, size, bits);
        const int symbol = (run << 4) | size;
        w.put(ac.code[symbol], ac.len[symbol]);
        w.put(bits, size);
        run = 0;
    }
    if (block[63] == 0) w.put(ac.code[0], ac.len[0]);  // end of block
    return block[0];
}

void segment(std::string &out, uint8_t marker, const std::string &body) {
    out.push_back((char)0xff);
    out.push_back((char)marker);
    const size_t n = body.size() + 2;
    out.push_back((char)(n >> 8));
    out.push_back((char)(n & 0xff));
    out += body;
}

void huffman_segment(std::string &out, uint8_t klass, uint8_t dest, const uint8_t *lengths, const uint8_t *values) {
    std::string body;
    body.push_back((char)((klass << 4) | dest));
    size_t n = 0;
    for (int i = 0; i < 16; ++i) {
        body.push_back((char)lengths[i]);
        n += lengths[i];
    }
    body.append(reinterpret_cast<const char *>(values), n);
    segment(out, 0xc4, body);
}

}  // namespace

// The canvas (f64, 3 values per pixel, row-major) as the bytes of the JPEG file Canvas::to_jpeg writes.
void canvas_to_jpeg(const double *rgb, uint32_t width, uint32_t height, std::string &out) {
    out.clear();
    out.reserve((size_t)width * height / 2 + 1024);
    out += "\xff\xd8";
    {  // JFIF 1.02, no units, density 1 x 1, no thumbnail
        const char app0[] = {'J', 'F', 'I', 'F', 0, 1, 2, 0, 0, 1, 0, 1, 0, 0};
        segment(out, 0xe0, std::string(app0, sizeof app0));
    }
    {  // baseline frame: 8 bits, three components, each 1 x 1; luma uses table 0, the chroma components table 1
        std::string sof;
        sof.push_back(8);
        sof.push_back((char)(height >> 8));
        sof.push_back((char)(height & 0xff));
        sof.push_back((char)(width >> 8));
        sof.push_back((char)(width & 0xff));
        sof.push_back(3);
        const char comps[] = {1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1};
        sof.append(comps, sizeof comps);
        segment(out, 0xc0, sof);
    }
    for (int t = 0; t < 2; ++t) {  // quality 100: every quantiser is 1
        std::string dqt(65, '\1');
        dqt[0] = (char)t;
        segment(out, 0xdb, dqt);
    }
    huffman_segment(out, 0, 0, LUMA_DC_LEN, DC_VAL);
    huffman_segment(out, 1, 0, LUMA_AC_LEN, LUMA_AC_VAL);
    huffman_segment(out, 0, 1, CHROMA_DC_LEN, DC_VAL);
    huffman_segment(out, 1, 1, CHROMA_AC_LEN, CHROMA_AC_VAL);
    {
        const char sos[] = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 0x3f, 0};
        segment(out, 0xda, std::string(sos, sizeof sos));
    }
    static const Huffman luma_dc(LUMA_DC_LEN, DC_VAL), luma_ac(LUMA_AC_LEN, LUMA_AC_VAL), chroma_dc(CHROMA_DC_LEN, DC_VAL),
        chroma_ac(CHROMA_AC_LEN, CHROMA_AC_VAL);
    // Transform first, code afterwards.  The strips of eight rows are independent up to the entropy coder -- canvas
    // quantiser (Canvas::to_rgb_image, canvas.rs:123-130), YCbCr, the blocks' DCTs -- and go to up to eight threads;
    // the coder itself runs over the finished coefficients in one pass (a block's DC is coded against the block before).
    const uint32_t blocks_x = (width + 7) / 8, strips = (height + 7) / 8;
    std::vector<int16_t> coded((size_t)blocks_x * strips * 3 * 64);  // |coefficient / 8| <= 1024
    auto quantize = [](double value) -> uint8_t {  // (value * 255.0).round() saturated: see rt_canvas.cpp
        const double x = value * 255.0;
        if (!(x >= 0.5)) return 0;
        if (x >= 254.5) return 255;
        const uint32_t q = (uint32_t)x;
        return (uint8_t)(q + (x - (double)q >= 0.5 ? 1u : 0u));
    };
    auto transform = [&](uint32_t s0, uint32_t s1) {
        std::vector<uint8_t> strip((size_t)8 * width * 3);
        uint8_t samples[3][64];
        int32_t coeffs[64];
        for (uint32_t sy = s0; sy < s1; ++sy) {
            const uint32_t y0 = sy * 8;
            for (uint32_t dy = 0; dy < 8; ++dy) {
                const uint32_t y = y0 + dy < height ? y0 + dy : height - 1;  // edge pixels repeat beyond the border
                const double *row = rgb + (size_t)y * width * 3;
                uint8_t *dst = strip.data() + (size_t)dy * width * 3;
                for (uint32_t x = 0; x < width; ++x)
                    rgb_to_ycbcr(quantize(row[x * 3]), quantize(row[x * 3 + 1]), quantize(row[x * 3 + 2]), dst[x * 3], dst[x * 3 + 1], dst[x * 3 + 2]);
            }
            for (uint32_t bx = 0; bx < blocks_x; ++bx) {
                const uint32_t x0 = bx * 8;
                for (uint32_t dy = 0; dy < 8; ++dy)
                    for (uint32_t dx = 0; dx < 8; ++dx) {
                        const uint32_t x = x0 + dx < width ? x0 + dx : width - 1;
                        const uint8_t *p = strip.data() + ((size_t)dy * width + x) * 3;
                        samples[0][dy * 8 + dx] = p[0];
                        samples[1][dy * 8 + dx] = p[1];
                        samples[2][dy * 8 + dx] = p[2];
                    }
                int16_t *dst = coded.data() + ((size_t)sy * blocks_x + bx) * 3 * 64;
                for (int c = 0; c < 3; ++c) {
                    fdct(samples[c], coeffs);
                    for (int i = 0; i < 64; ++i) dst[c * 64 + i] = (int16_t)(coeffs[i] / 8);  // towards zero; then divided by the quantiser, which is 1
                }
            }
        }
    };
    unsigned n_threads = (size_t)width * height < (1u << 18) ? 1u : std::thread::hardware_concurrency();
    n_threads = n_threads == 0 ? 1 : n_threads > 8 ? 8 : n_threads;
    if (n_threads == 1) {
        transform(0, strips);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back(transform, (uint32_t)((uint64_t)strips * t / n_threads), (uint32_t)((uint64_t)strips * (t + 1) / n_threads));
        for (std::thread &t : pool) t.join();
    }
    BitWriter w(out);
    int32_t prev[3] = {0, 0, 0};
    int32_t block[64];
    for (size_t b = 0; b < (size_t)blocks_x * strips; ++b)
        for (int c = 0; c < 3; ++c) {
            const int16_t *src = coded.data() + (b * 3 + c) * 64;
            for (int i = 0; i < 64; ++i) block[i] = src[i];
            prev[c] = write_block(w, block, prev[c], c == 0 ? luma_dc : chroma_dc, c == 0 ? luma_ac : chroma_ac);
        }
    w.pad();
    out += "\xff\xd9";
}

}  // namespace rt

extern "C" {

size_t rt_canvas_to_jpeg(const double *rgb, uint32_t width, uint32_t height, unsigned char *out, size_t cap) {
    if (!rgb || width == 0 || height == 0 || width > 65535 || height > 65535) return 0;
    std::string bytes;
    rt::canvas_to_jpeg(rgb, width, height, bytes);
    if (out && cap >= bytes.size()) std::memcpy(out, bytes.data(), bytes.size());
    return bytes.size();
}

int rt_canvas_write_jpeg(const double *rgb, uint32_t width, uint32_t height, const char *path) {
    if (!rgb || !path) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "rgb and path must not be NULL");
    if (width == 0 || height == 0 || width > 65535 || height > 65535)
        return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "a JPEG frame is 1 .. 65535 pixels wide and high");
    std::string bytes;
    rt::canvas_to_jpeg(rgb, width, height, bytes);
    FILE *f = std::fopen(path, "wb");
    if (!f) return rt::capi_set_error(RT_ERR_IO, std::string("cannot create ") + path);
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (std::fclose(f) != 0 || !ok) return rt::capi_set_error(RT_ERR_IO, std::string("cannot write ") + path);
    return RT_OK;
}

}  // extern "C"
