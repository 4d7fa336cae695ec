#include "ImageIO.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace acgpt {

static std::vector<uint8_t> flip_rgb(const uint8_t* rgba, int w, int h, bool with_filter_byte)
{
    const size_t stride = (size_t)w * 3 + (with_filter_byte ? 1 : 0);
    std::vector<uint8_t> out(stride * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* src = rgba + (size_t)(h - 1 - y) * w * 4;      // file row y = buffer row h-1-y
        uint8_t* dst = out.data() + stride * y;
        if (with_filter_byte) *dst++ = 0;
        for (int x = 0; x < w; x++) { dst[3 * x] = src[4 * x]; dst[3 * x + 1] = src[4 * x + 1]; dst[3 * x + 2] = src[4 * x + 2]; }
    }
    return out;
}

bool savePPM(const std::string& filename, const uint8_t* rgba, int w, int h)
{
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<uint8_t> rgb = flip_rgb(rgba, w, h, false);
    const bool ok = fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    fclose(f);
    return ok;
}

static uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
    return ~crc;
}

static void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

static void chunk(FILE* f, const char* tag, const std::vector<uint8_t>& data)
{
    std::vector<uint8_t> buf;
    put32(buf, (uint32_t)data.size());
    buf.insert(buf.end(), tag, tag + 4);
    buf.insert(buf.end(), data.begin(), data.end());
    const uint32_t c = crc32(buf.data() + 4, buf.size() - 4);
    put32(buf, c);
    fwrite(buf.data(), 1, buf.size(), f);
}

// PNG with stored (uncompressed) deflate blocks: no zlib dependency.
bool savePNG(const std::string& filename, const uint8_t* rgba, int w, int h)
{
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
    chunk(f, "IHDR", ihdr);
    const std::vector<uint8_t> raw = flip_rgb(rgba, w, h, true);
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        const size_t n = raw.size() - pos > 65535 ? 65535 : raw.size() - pos;
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 255); z.push_back(n >> 8); z.push_back(~n & 255); z.push_back((~n >> 8) & 255);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", std::vector<uint8_t>());
    fclose(f);
    return true;
}

bool saveImage(const std::string& filename, const uint8_t* rgba, int w, int h)
{
    const size_t n = filename.size();
    if (n >= 4 && (filename.compare(n - 4, 4, ".ppm") == 0 || filename.compare(n - 4, 4, ".PPM") == 0)) return savePPM(filename, rgba, w, h);
    if (n >= 4 && (filename.compare(n - 4, 4, ".png") == 0 || filename.compare(n - 4, 4, ".PNG") == 0)) return savePNG(filename, rgba, w, h);
    return false;
}

}  // namespace acgpt
