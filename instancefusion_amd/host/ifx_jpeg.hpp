// ifx_jpeg.hpp -- baseline JPEG decoder for the colour frames of .klg logs (IF/utilities/RawLogReader.cpp:96-106 decodes them with
// libjpeg through cvDecodeImage; the image of this build has no libjpeg headers).  Self-contained, header-only.
//
// Scope: baseline / extended sequential DCT (SOF0, SOF1), 8-bit samples, Huffman coding, 1 or 3 components (grey, YCbCr), chroma sampled
// 1x1, 2x1 (4:2:2) or 2x2 (4:2:0) relative to luma, restart intervals.  Progressive / arithmetic / 12-bit / CMYK files are refused.
// The arithmetic is libjpeg's default decompression path, so that a decoded frame equals what the reference (and PIL, which the Python
// reader uses) delivers: the accurate integer inverse DCT (jidctint.c, "islow": 13-bit constants, 2 extra bits after pass 1), "fancy"
// triangle-filter chroma upsampling (jdsample.c h2v1 / h2v2, context rows replicated at the image edges) and the 16-bit fixed-point
// YCbCr -> RGB tables of jdcolor.c.  tests/test_host_cpp.py compares it with PIL pixel for pixel.
#ifndef IFX_JPEG_HPP_
#define IFX_JPEG_HPP_

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace ifx_jpeg {

struct Huff {
    // canonical code tables (ITU T.81 Annex C / F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    int16_t fast[512];   // 9-bit lookup: (length << 8) | value, 0 = not a short code
    bool present = false;
    void build(const uint8_t* counts, const uint8_t* symbols, int nsym)
    {
        std::memcpy(vals, symbols, (size_t)nsym);
        int code = 0, k = 0;
        for (int i = 0; i < 512; i++) fast[i] = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            // the lengths must form a prefix code: after the codes of this length at most 2^len values are used (otherwise `first` below
            // would index past fast[] and the decoder could be steered out of vals[])
            if (code + counts[len - 1] > (1 << len)) throw std::runtime_error("JPEG: Huffman code lengths do not form a prefix code");
            for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
                if (len <= 9) {
                    const int first = code << (9 - len);
                    for (int j = 0; j < (1 << (9 - len)); j++) fast[first + j] = (int16_t)((len << 8) | symbols[k]);
                }
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t buf = 0;
    int bits = 0;
    bool hit_marker = false;
    void fill()
    {
        while (bits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    const int b2 = p < end ? *p : 0xD9;
                    if (b2 == 0) p++;                     // stuffed zero
                    else { hit_marker = true; p--; b = 0; }   // a marker: feed zeros (libjpeg does the same and warns)
                }
            }
            buf |= (uint32_t)b << (24 - bits);
            bits += 8;
        }
    }
    int peek(int n) { if (bits < n) fill(); return (int)(buf >> (32 - n)); }
    void skip(int n) { buf <<= n; bits -= n; }
    int get(int n)
    {
        if (n == 0) return 0;
        if (n < 0 || n > 16) throw std::runtime_error("JPEG: bad magnitude category");   // a corrupted table can name any symbol value
        const int v = peek(n);
        skip(n);
        return v;
    }
    void reset() { buf = 0; bits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h)
{
    const int look = br.peek(9);
    const int f = h.fast[look];
    if (f) { br.skip(f >> 8); return f & 0xFF; }
    int code = br.peek(16), len = 10;
    for (; len <= 16; len++) {
        const int c = code >> (16 - len);
        if (h.maxcode[len] >= 0 && c <= h.maxcode[len] && c >= h.mincode[len]) {
            br.skip(len);
            return h.vals[h.valptr[len] + c - h.mincode[len]];
        }
    }
    throw std::runtime_error("JPEG: bad Huffman code");
}
inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

// jidctint.c (jpeg_idct_islow): dequantised coefficients in natural order -> 64 samples (0..255)
inline void idct_islow(const int* in, uint8_t* out, int stride)
{
    const int CONST_BITS = 13, PASS1_BITS = 2;
    const long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373, F_1_175875602 = 9633,
               F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
    long ws[64];
    for (int c = 0; c < 8; c++) {
        const int* p = in + c;
        long z2 = p[16], z3 = p[48];
        long z1 = (z2 + z3) * F_0_541196100;
        long tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
        z2 = p[0]; z3 = p[32];
        long tmp0 = (z2 + z3) * (1L << CONST_BITS), tmp1 = (z2 - z3) * (1L << CONST_BITS);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int sh = CONST_BITS - PASS1_BITS;
        const long rnd = 1L << (sh - 1);
        long* w = ws + c;
        w[0] = (tmp10 + tmp3 + rnd) >> sh; w[56] = (tmp10 - tmp3 + rnd) >> sh;
        w[8] = (tmp11 + tmp2 + rnd) >> sh; w[48] = (tmp11 - tmp2 + rnd) >> sh;
        w[16] = (tmp12 + tmp1 + rnd) >> sh; w[40] = (tmp12 - tmp1 + rnd) >> sh;
        w[24] = (tmp13 + tmp0 + rnd) >> sh; w[32] = (tmp13 - tmp0 + rnd) >> sh;
    }
    for (int r = 0; r < 8; r++) {
        const long* p = ws + r * 8;
        long z2 = p[2], z3 = p[6];
        long z1 = (z2 + z3) * F_0_541196100;
        long tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
        long tmp0 = (p[0] + p[4]) * (1L << CONST_BITS), tmp1 = (p[0] - p[4]) * (1L << CONST_BITS);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = p[7]; tmp1 = p[5]; tmp2 = p[3]; tmp3 = p[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int sh = CONST_BITS + PASS1_BITS + 3;
        const long rnd = 1L << (sh - 1);
        const long v[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
        uint8_t* o = out + r * stride;
        for (int k = 0; k < 8; k++) {
            long s = ((v[k] + rnd) >> sh) + 128;
            o[k] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w = 0, hgt = 0;            // real (downsampled) size: ceil(image * samp / max_samp)
    int stride = 0, rows = 0;      // padded to whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
};

// Decodes a JPEG stream into interleaved RGB (grey is replicated).  Throws std::runtime_error on anything outside the scope above.
inline void decode(const uint8_t* data, size_t size, std::vector<uint8_t>& rgb, int& width, int& height)
{
    static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw std::runtime_error("JPEG: no SOI marker");
    int quant[4][64];
    bool have_q[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Component comp[3];
    int ncomp = 0, restart = 0, hmax = 1, vmax = 1;
    width = height = 0;
    size_t p = 2;
    bool decoded = false;
    while (p + 4 <= size && !decoded) {
        if (data[p] != 0xFF) { p++; continue; }
        const int m = data[p + 1];
        if (m == 0xFF) { p++; continue; }
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
        if (m == 0xD9) break;
        const size_t len = ((size_t)data[p + 2] << 8) | data[p + 3];
        if (len < 2 || p + 2 + len > size) throw std::runtime_error("JPEG: truncated segment");
        const uint8_t* s = data + p + 4;
        const uint8_t* se = data + p + 2 + len;
        if (m == 0xDB) {   // DQT
            while (s < se) {
                const int pq = s[0] >> 4, tq = s[0] & 15;
                if (tq > 3 || pq > 1) throw std::runtime_error("JPEG: bad quantisation table id / precision");
                s++;
                if (s + (pq ? 128 : 64) > se) throw std::runtime_error("JPEG: truncated quantisation table");
                for (int k = 0; k < 64; k++) {
                    quant[tq][zigzag[k]] = pq ? ((s[0] << 8) | s[1]) : s[0];
                    s += pq ? 2 : 1;
                }
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {   // DHT
            while (s < se) {
                if (s + 17 > se) throw std::runtime_error("JPEG: truncated Huffman table");
                const int tc = s[0] >> 4, th = s[0] & 15;
                if (th > 3 || tc > 1) throw std::runtime_error("JPEG: bad Huffman table id");
                int n = 0;
                for (int k = 0; k < 16; k++) n += s[1 + k];
                if (n > 256 || s + 17 + n > se) throw std::runtime_error("JPEG: bad Huffman table");
                (tc ? ac[th] : dc[th]).build(s + 1, s + 17, n);
                s += 17 + n;
            }
        } else if (m == 0xC0 || m == 0xC1) {   // SOF0 / SOF1
            if (s + 6 > se) throw std::runtime_error("JPEG: truncated frame header");
            if (s[0] != 8) throw std::runtime_error("JPEG: only 8-bit samples are supported");
            height = (s[1] << 8) | s[2];
            width = (s[3] << 8) | s[4];
            ncomp = s[5];
            if (ncomp != 1 && ncomp != 3) throw std::runtime_error("JPEG: only grey and YCbCr images are supported");
            if (s + 6 + 3 * ncomp > se) throw std::runtime_error("JPEG: truncated frame header");
            if (width <= 0 || height <= 0 || (size_t)width * height > ((size_t)1 << 28)) throw std::runtime_error("JPEG: bad image size");
            hmax = vmax = 1;
            for (int c = 0; c < ncomp; c++) {
                comp[c].id = s[6 + c * 3];
                comp[c].h = s[7 + c * 3] >> 4;
                comp[c].v = s[7 + c * 3] & 15;
                comp[c].tq = s[8 + c * 3];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2) throw std::runtime_error("JPEG: sampling factors other than 1 and 2 are not supported");
                if (comp[c].tq > 3) throw std::runtime_error("JPEG: bad quantisation table id");
                if (comp[c].h > hmax) hmax = comp[c].h;
                if (comp[c].v > vmax) vmax = comp[c].v;
            }
            if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }   // a single component is never interleaved: one block per MCU
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            throw std::runtime_error("JPEG: progressive / lossless / arithmetic-coded files are not supported");
        } else if (m == 0xDD) {
            if (s + 2 > se) throw std::runtime_error("JPEG: truncated restart interval");
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {   // SOS: the entropy-coded data follows
            if (!width || !height) throw std::runtime_error("JPEG: SOS before SOF");
            if (s + 1 > se) throw std::runtime_error("JPEG: truncated scan header");
            const int ns = s[0];
            if (ns != ncomp) throw std::runtime_error("JPEG: non-interleaved multi-scan files are not supported");
            if (s + 1 + 2 * ns > se) throw std::runtime_error("JPEG: truncated scan header");
            for (int c = 0; c < ncomp; c++) { comp[c].td = -1; comp[c].ta = -1; }
            for (int k = 0; k < ns; k++) {
                for (int c = 0; c < ncomp; c++)
                    if (comp[c].id == s[1 + k * 2]) { comp[c].td = s[2 + k * 2] >> 4; comp[c].ta = s[2 + k * 2] & 15; }
            }
            for (int c = 0; c < ncomp; c++)
                if (comp[c].td < 0 || comp[c].td > 3 || comp[c].ta < 0 || comp[c].ta > 3) throw std::runtime_error("JPEG: bad entropy table selector");
            if (hmax > 2 || vmax > 2) throw std::runtime_error("JPEG: sampling factors above 2 are not supported");
            for (int c = 1; c < ncomp; c++)
                if (comp[c].h != 1 || comp[c].v != 1) throw std::runtime_error("JPEG: subsampled luma / oversampled chroma is not supported");
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax, mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
            for (int c = 0; c < ncomp; c++) {
                Component& C = comp[c];
                if (!have_q[C.tq] || !dc[C.td].present || !ac[C.ta].present) throw std::runtime_error("JPEG: missing table");
                C.w = (width * C.h + hmax - 1) / hmax;
                C.hgt = (height * C.v + vmax - 1) / vmax;
                C.stride = mcus_x * C.h * 8;
                C.rows = mcus_y * C.v * 8;
                C.plane.assign((size_t)C.stride * C.rows, 0);
                C.dc_pred = 0;
            }
            BitReader br;
            br.p = data + p + 2 + len;
            br.end = data + size;
            int coef[64], until_restart = restart;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart && until_restart == 0) {   // RSTn: byte-align, skip the marker, reset the predictors
                        const uint8_t* q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                        br.p = q + 2 <= br.end ? q + 2 : br.end;
                        br.reset();
                        for (int c = 0; c < ncomp; c++) comp[c].dc_pred = 0;
                        until_restart = restart;
                    }
                    for (int c = 0; c < ncomp; c++) {
                        Component& C = comp[c];
                        for (int by = 0; by < C.v; by++)
                            for (int bx = 0; bx < C.h; bx++) {
                                std::memset(coef, 0, sizeof(coef));
                                int t = decode_symbol(br, dc[C.td]);
                                const int diff = t ? extend(br.get(t), t) : 0;
                                C.dc_pred += diff;
                                coef[0] = C.dc_pred * quant[C.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(br, ac[C.ta]), r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r != 15) break;   // EOB
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) throw std::runtime_error("JPEG: coefficient index out of range");
                                    coef[zigzag[k]] = extend(br.get(sz), sz) * quant[C.tq][zigzag[k]];
                                    k++;
                                }
                                idct_islow(coef, &C.plane[(size_t)((my * C.v + by) * 8) * C.stride + (size_t)(mx * C.h + bx) * 8], C.stride);
                            }
                    }
                    if (restart) until_restart--;
                }
            decoded = true;
        }
        p += 2 + len;
    }
    if (!decoded) throw std::runtime_error("JPEG: no scan found");

    rgb.assign((size_t)width * height * 3, 0);
    if (ncomp == 1) {
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                const uint8_t v = comp[0].plane[(size_t)y * comp[0].stride + x];
                uint8_t* o = &rgb[((size_t)y * width + x) * 3];
                o[0] = o[1] = o[2] = v;
            }
        return;
    }
    // ---- chroma upsampling to full resolution (jdsample.c: fullsize copy, h2v1_fancy_upsample, h2v2_fancy_upsample)
    std::vector<uint8_t> up[2];
    for (int c = 1; c < 3; c++) {
        const Component& C = comp[c];
        std::vector<uint8_t>& U = up[c - 1];
        const int ow = C.w * hmax, oh = C.hgt * vmax;   // >= image size
        U.assign((size_t)ow * oh, 0);
        auto row = [&](int r) { return &C.plane[(size_t)(r < 0 ? 0 : (r >= C.hgt ? C.hgt - 1 : r)) * C.stride]; };   // context rows replicate the edge rows
        if (hmax == 1 && vmax == 1) {
            for (int y = 0; y < oh; y++) std::memcpy(&U[(size_t)y * ow], row(y), (size_t)ow);
        } else if (hmax == 2 && vmax == 1) {
            for (int y = 0; y < oh; y++) {
                const uint8_t* in = row(y);
                uint8_t* o = &U[(size_t)y * ow];
                const int n = C.w;
                if (n == 1) { o[0] = o[1] = in[0]; continue; }
                o[0] = in[0];
                o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < n - 1; i++) {
                    const int v = in[i] * 3;
                    o[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
                    o[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
                }
                o[2 * (n - 1)] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
                o[2 * (n - 1) + 1] = in[n - 1];
            }
        } else if (hmax == 2 && vmax == 2) {
            for (int r = 0; r < C.hgt; r++)
                for (int v = 0; v < 2; v++) {
                    const uint8_t* in0 = row(r);                       // nearer input row
                    const uint8_t* in1 = row(v == 0 ? r - 1 : r + 1);  // further input row
                    uint8_t* o = &U[(size_t)(2 * r + v) * ow];
                    const int n = C.w;
                    if (n == 1) {
                        const int t = in0[0] * 3 + in1[0];
                        o[0] = (uint8_t)((t * 4 + 8) >> 4);
                        o[1] = (uint8_t)((t * 4 + 7) >> 4);
                        continue;
                    }
                    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                    o[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
                    o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                    for (int i = 1; i < n - 1; i++) {
                        nextcol = in0[i + 1] * 3 + in1[i + 1];
                        o[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                        o[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                        lastcol = thiscol; thiscol = nextcol;
                    }
                    o[2 * (n - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                    o[2 * (n - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
                }
        } else
            throw std::runtime_error("JPEG: 1x2 chroma sampling is not supported");
    }
    // ---- YCbCr -> RGB (jdcolor.c build_ycc_rgb_table / ycc_rgb_convert)
    struct YccTables {   // built once; a function-local static object is initialised thread-safely (the log reader decodes on several threads)
        int cr_r[256], cb_b[256];
        long cr_g[256], cb_g[256];
        YccTables()
        {
            const long ONE_HALF = 1L << 15;
            for (int i = 0; i < 256; i++) {
                const long x = i - 128;
                cr_r[i] = (int)((91881L * x + ONE_HALF) >> 16);      // FIX(1.40200)
                cb_b[i] = (int)((116130L * x + ONE_HALF) >> 16);     // FIX(1.77200)
                cr_g[i] = -46802L * x;                               // FIX(0.71414)
                cb_g[i] = -22554L * x + ONE_HALF;                    // FIX(0.34414)
            }
        }
    };
    static const YccTables T;
    const int* cr_r = T.cr_r;
    const int* cb_b = T.cb_b;
    const long* cr_g = T.cr_g;
    const long* cb_g = T.cb_g;
    const int ow = comp[1].w * hmax;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int Y = comp[0].plane[(size_t)y * comp[0].stride + x], cb = up[0][(size_t)y * ow + x], cr = up[1][(size_t)y * ow + x];
            const int r = Y + cr_r[cr], g = Y + (int)((cb_g[cb] + cr_g[cr]) >> 16), b = Y + cb_b[cb];
            uint8_t* o = &rgb[((size_t)y * width + x) * 3];
            o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
            o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
        }
}

}   // namespace ifx_jpeg
#endif   // IFX_JPEG_HPP_
