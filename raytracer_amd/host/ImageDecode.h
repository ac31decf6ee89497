// Decoders for the two texture formats the reference's scene file names (assets/cerealBar.xml:169-180:
// textures/bgPlate.png, textures/alps_field_1k.hdr), so that SceneLoader reads them without outside help. The
// reference decodes them with its vendored lodepng and the framework's HDR reader (neither is on the hot path).
//   PNG:  8/16-bit grey, grey+alpha, RGB, RGBA and palette images, non-interlaced; zlib inflate with stored, fixed and
//         dynamic Huffman blocks; the five scanline filters. Alpha is dropped, 16-bit samples keep their high byte.
//   HDR:  Radiance RGBE ("#?RADIANCE" / "#?RGBE", -Y h +X w), flat and new-style run-length encoded scanlines,
//         decoded to float RGB = mantissa * 2^(e-136).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "scene.h"

namespace rtamd {
namespace imgdec {

inline bool ReadFile(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

// ---- inflate (RFC 1951) ----
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    bool ok = true;
    uint32_t bits(int k) {
        while (cnt < k) {
            if (pos >= n) {
                ok = false;
                return 0;
            }
            buf |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        const uint32_t v = buf & ((k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        buf >>= k;
        cnt -= k;
        return v;
    }
    void align() {
        buf = 0;
        cnt = 0;
    }
};
struct Huffman {
    uint16_t count[16] = {0};
    std::vector<uint16_t> symbol;
    bool build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        symbol.assign((size_t)n, 0);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[(size_t)(index + (code - first))];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};
inline bool Inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitReader br{src, n};
    for (;;) {
        const uint32_t final = br.bits(1), type = br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > n) return false;
            const uint32_t len = src[br.pos] | (src[br.pos + 1] << 8), nlen = src[br.pos + 2] | (src[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > n) return false;
            out.insert(out.end(), src + br.pos, src + br.pos + len);
            br.pos += len;
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
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman lencode;
                lencode.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lengths[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return false;
                            val = lengths[idx - 1];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (!br.ok || idx + rep > nlen + ndist) return false;
                        while (rep--) lengths[idx++] = (uint8_t)val;
                    }
                }
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    if (sym - 257 >= 29) return false;
                    const int len = lbase[sym - 257] + (int)br.bits(lext[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (!br.ok || d > out.size()) return false;
                    const size_t start = out.size() - d;
                    for (int k = 0; k < len; ++k) out.push_back(out[start + (size_t)k]);
                }
            }
        } else
            return false;
        if (final) break;
    }
    return true;
}

// ---- PNG ----
inline uint32_t Be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline std::shared_ptr<TextureMap> LoadPng(const std::string& path) {
    std::vector<uint8_t> file;
    if (!ReadFile(path, file) || file.size() < 33) return nullptr;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (std::memcmp(file.data(), sig, 8) != 0) return nullptr;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        const uint32_t len = Be32(&file[pos]);
        const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
        if (pos + 12 + (size_t)len > file.size()) return nullptr;
        const uint8_t* data = &file[pos + 8];
        if (!std::strncmp(type, "IHDR", 4) && len >= 13) {
            w = Be32(data);
            h = Be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!std::strncmp(type, "PLTE", 4)) palette.assign(data, data + len);
        else if (!std::strncmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::strncmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!w || !h || interlace || idat.size() < 6) return nullptr;
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return nullptr;
    }
    if (!(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3)))) return nullptr;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * channels * 2 + 1));
    if (!Inflate(idat.data() + 2, idat.size() - 2, raw)) return nullptr;   // 2-byte zlib header; the Adler-32 trailer is not read
    const size_t bpp = std::max<size_t>(1, (size_t)channels * depth / 8);
    const size_t stride = ((size_t)w * channels * depth + 7) / 8;
    if (raw.size() < (size_t)h * (stride + 1)) return nullptr;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    auto tex = std::make_shared<TextureMap>();
    tex->width = (int)w;
    tex->height = (int)h;
    tex->rgb8.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(size_t)y * (stride + 1)];
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return nullptr;
            }
            cur[i] = (uint8_t)(line[1 + i] + pred);
        }
        uint8_t* dst = &tex->rgb8[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t s[4] = {0, 0, 0, 255};
            for (int ch = 0; ch < channels; ++ch) {
                if (depth == 8) s[ch] = cur[(size_t)x * channels + ch];
                else if (depth == 16) s[ch] = cur[((size_t)x * channels + ch) * 2];
                else {   // 1, 2 or 4 bits: grey levels are stretched to 0..255, palette indices are kept
                    const size_t bit = (size_t)x * depth;
                    const int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                    s[ch] = ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
                }
            }
            if (ctype == 3) {
                const size_t k = (size_t)s[0] * 3;
                if (k + 2 >= palette.size()) return nullptr;
                dst[3 * x] = palette[k]; dst[3 * x + 1] = palette[k + 1]; dst[3 * x + 2] = palette[k + 2];
            } else if (channels <= 2) {
                dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = s[0];
            } else {
                dst[3 * x] = s[0]; dst[3 * x + 1] = s[1]; dst[3 * x + 2] = s[2];
            }
        }
        prev.swap(cur);
    }
    return tex;
}

// ---- Radiance HDR (RGBE) ----
inline std::shared_ptr<TextureMap> LoadHdr(const std::string& path) {
    std::vector<uint8_t> file;
    if (!ReadFile(path, file) || file.size() < 16) return nullptr;
    if (std::memcmp(file.data(), "#?", 2) != 0) return nullptr;
    size_t pos = 0;
    auto line = [&](std::string& s) {
        s.clear();
        while (pos < file.size() && file[pos] != '\n') s.push_back((char)file[pos++]);
        if (pos < file.size()) ++pos;
    };
    std::string s;
    bool fmtOk = false;
    for (;;) {   // header lines up to the empty one
        if (pos >= file.size()) return nullptr;
        line(s);
        if (s.empty()) break;
        if (s.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) fmtOk = true;
    }
    line(s);
    int w = 0, h = 0;
    if (std::sscanf(s.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || !fmtOk) return nullptr;
    auto tex = std::make_shared<TextureMap>();
    tex->width = w;
    tex->height = h;
    tex->isFloat = true;
    tex->rgb32f.resize((size_t)w * h * 3);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; ++y) {
        if (pos + 4 > file.size()) return nullptr;
        if (w >= 8 && w < 32768 && file[pos] == 2 && file[pos + 1] == 2 && ((file[pos + 2] << 8) | file[pos + 3]) == w) {
            pos += 4;   // new-style RLE: the four components of the scanline one after the other
            for (int ch = 0; ch < 4; ++ch) {
                int x = 0;
                while (x < w) {
                    if (pos >= file.size()) return nullptr;
                    int c = file[pos++];
                    if (c > 128) {
                        c -= 128;
                        if (pos >= file.size() || x + c > w) return nullptr;
                        const uint8_t v = file[pos++];
                        while (c--) scan[(size_t)(x++) * 4 + ch] = v;
                    } else {
                        if (c == 0 || pos + (size_t)c > file.size() || x + c > w) return nullptr;
                        while (c--) scan[(size_t)(x++) * 4 + ch] = file[pos++];
                    }
                }
            }
        } else {   // flat scanline
            if (pos + (size_t)w * 4 > file.size()) return nullptr;
            std::memcpy(scan.data(), &file[pos], (size_t)w * 4);
            pos += (size_t)w * 4;
        }
        float* dst = &tex->rgb32f[(size_t)y * w * 3];
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = &scan[(size_t)x * 4];
            if (p[3] == 0) dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = 0.0f;
            else {
                const float f = std::ldexp(1.0f, (int)p[3] - 136);
                dst[3 * x] = p[0] * f; dst[3 * x + 1] = p[1] * f; dst[3 * x + 2] = p[2] * f;
            }
        }
    }
    return tex;
}

}  // namespace imgdec
}  // namespace rtamd
