// Minimal PNG reader for image textures (8-bit, non-interlaced; grey, grey + alpha, RGB, RGBA, palette).
// The reference reads its textures through OpenCV (taichi-version/hittable.py:165: cv2.imread); there is no
// image library on the target, so this is a from-scratch inflate (RFC 1951) + PNG unfilter (RFC 2083).
// Alpha is dropped (cv2.imread(..., 1) returns three channels too).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rtmi {
namespace png {

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool ok = true;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    uint32_t bits(int n) {  // n <= 16, LSB first
        while (nbits < n) {
            if (p >= end) {
                ok = false;
                return 0;
            }
            acc |= (uint32_t)*p++ << nbits;
            nbits += 8;
        }
        const uint32_t v = acc & ((1u << n) - 1u);
        acc >>= n;
        nbits -= n;
        return v;
    }
    void align() { acc = 0, nbits = 0; }
};

struct Huffman {  // canonical code, decoded bit by bit (textures are small)
    uint16_t count[16] = {0}, symbol[320] = {0};
    bool build(const uint8_t *lengths, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// max_out: the most the stream may expand to (the caller knows the image size): a crafted stream of a few KB would otherwise
// grow to gigabytes (deflate reaches about 1000 : 1) before any size check
inline bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err, size_t max_out) {
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (n < 6) {
        err = "zlib stream too short";
        return false;
    }
    if ((src[0] & 15) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 32)) {
        err = "not a zlib stream";
        return false;
    }
    BitReader br(src + 2, src + n - 4);
    for (;;) {
        const int final = (int)br.bits(1), type = (int)br.bits(2);
        if (!br.ok) break;
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) break;
            const unsigned len = br.p[0] | br.p[1] << 8, nlen = br.p[2] | br.p[3] << 8;
            br.p += 4;
            if ((len ^ 0xffffu) != nlen || (size_t)(br.end - br.p) < len) break;
            if (out.size() + len > max_out) {
                err = "zlib stream expands beyond the image it belongs to";
                return false;
            }
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) break;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman clh;
                clh.build(cl, 19);
                int i = 0;
                bool bad = false;
                while (i < nlen + ndist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) {
                        bad = true;
                        break;
                    }
                    if (sym < 16) lengths[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) {
                                bad = true;
                                break;
                            }
                            val = lengths[i - 1], rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (i + rep > nlen + ndist) {
                            bad = true;
                            break;
                        }
                        while (rep--) lengths[i++] = (uint8_t)val;
                    }
                }
                if (bad || !br.ok) break;
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            bool done = false;
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) break;
                if (sym < 256 && out.size() >= max_out) {
                    err = "zlib stream expands beyond the image it belongs to";
                    return false;
                }
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) {
                    done = true;
                    break;
                } else {
                    const int li = sym - 257;
                    if (li >= 29) break;
                    const int len = len_base[li] + (int)br.bits(len_extra[li]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) break;
                    const size_t d = dist_base[ds] + br.bits(dist_extra[ds]);
                    if (!br.ok || d > out.size()) break;
                    if (out.size() + (size_t)len > max_out) {
                        err = "zlib stream expands beyond the image it belongs to";
                        return false;
                    }
                    const size_t from = out.size() - d;
                    for (int k = 0; k < len; ++k) out.push_back(out[from + k]);
                }
            }
            if (!done) break;
        } else break;
        if (final) {
            uint32_t a = 1, b = 0;  // adler32 of the output against the trailer
            for (uint8_t c : out) a = (a + c) % 65521u, b = (b + a) % 65521u;
            const uint8_t *t = src + n - 4;
            if (((b << 16) | a) != ((uint32_t)t[0] << 24 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 8 | t[3])) {
                err = "zlib checksum mismatch";
                return false;
            }
            return true;
        }
    }
    err = "corrupt deflate stream";
    return false;
}

// -> rows x cols x 3 bytes (R, G, B)
inline bool read(const std::vector<uint8_t> &file, int &rows, int &cols, std::vector<uint8_t> &rgb, std::string &err) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) {
        err = "not a PNG file";
        return false;
    }
    auto be32 = [&](size_t o) { return (uint32_t)file[o] << 24 | (uint32_t)file[o + 1] << 16 | (uint32_t)file[o + 2] << 8 | file[o + 3]; };
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(pos);
        const char *type = (const char *)&file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) break;
        const uint8_t *data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(pos + 8), h = be32(pos + 12);
            depth = data[8], ctype = data[9], interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    int ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: err = "PNG without a usable IHDR"; return false;
    }
    if (depth != 8 || interlace != 0 || w < 1 || h < 1 || w > 16384 || h > 16384) {
        err = "unsupported PNG (need 8 bits per channel, no interlace, at most 16384 x 16384)";
        return false;
    }
    std::vector<uint8_t> raw;
    const size_t stride = (size_t)w * ch;
    if (!inflate(idat.data(), idat.size(), raw, err, (stride + 1) * h)) return false;
    if (raw.size() < (stride + 1) * h) {
        err = "PNG image data too short";
        return false;
    }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {  // unfilter, RFC 2083 section 6
        const uint8_t *in = &raw[(stride + 1) * y];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        const int f = in[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int pred = 0;
            if (f == 1) pred = a;
            else if (f == 2) pred = b;
            else if (f == 3) pred = (a + b) >> 1;
            else if (f == 4) {
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (f != 0) {
                err = "PNG with an unknown filter type";
                return false;
            }
            cur[i] = (uint8_t)(in[1 + i] + pred);
        }
    }
    rows = (int)h, cols = (int)w;
    rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t *px = &img[i * ch];
        uint8_t *o = &rgb[i * 3];
        if (ctype == 3) {
            const size_t k = (size_t)px[0] * 3;
            if (k + 2 >= plte.size()) {
                err = "PNG palette index out of range";
                return false;
            }
            o[0] = plte[k], o[1] = plte[k + 1], o[2] = plte[k + 2];
        } else if (ch <= 2) o[0] = o[1] = o[2] = px[0];
        else o[0] = px[0], o[1] = px[1], o[2] = px[2];
    }
    return true;
}

}  // namespace png
}  // namespace rtmi
