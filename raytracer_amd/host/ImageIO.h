// Image outputs of the render loop (reference: RenderImage::SaveImage / SaveZImage / SaveSampleCountImage called from
// RayTracer/Main.cpp:370-373 and RayCaster.cpp:1342-1346; encoder there: the vendored lodepng).
// A self-contained PNG writer: 8-bit RGB or grey, zlib "stored" blocks (no compression), CRC-32 and Adler-32 as the
// PNG / zlib specifications define them. Any PNG reader decodes the result; files are larger than lodepng's.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace rtamd {
namespace png {

inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}
inline void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, crc32(body.data(), body.size()));
}

// channels = 3 (RGB) or 1 (grey); pixels row-major, tightly packed
inline std::vector<uint8_t> encode(const uint8_t* pixels, int width, int height, int channels) {
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width);
    put32(ihdr, (uint32_t)height);
    ihdr.push_back(8);
    ihdr.push_back(channels == 3 ? 2 : 0);
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    // raw scanlines, filter type 0
    const size_t row = (size_t)width * channels;
    std::vector<uint8_t> raw;
    raw.reserve((row + 1) * height);
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), pixels + row * y, pixels + row * (y + 1));
    }
    // zlib stream of stored blocks (<= 65535 bytes each)
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    put32(z, (b << 16) | a);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return out;
}

inline bool save(const char* filename, const uint8_t* pixels, int width, int height, int channels) {
    if (!filename || !pixels || width <= 0 || height <= 0) return false;
    const std::vector<uint8_t> bytes = encode(pixels, width, height, channels);
    FILE* f = std::fopen(filename, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    return ok;
}

}  // namespace png
}  // namespace rtamd
