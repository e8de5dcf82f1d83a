/*
 * image_io.cpp -- image output in place of the reference's present blit
 * (/root/reference/src/renderer/main.cpp:140-145): 8-bit PPM and PNG of the
 * accumulator, beside the float PFM of renderer.cpp.  The reference shows the
 * accumulator through an RGBA texture, i.e. clamped to [0,1] with no tone
 * curve; to_srgb8 with gamma 1 is that mapping, gamma 2.2 is offered for viewing.
 */
#include <cmath>
#include <cstdio>
#include <cstring>

#include "bdpt_host.h"

namespace bdpt_host {

std::vector<uint8_t> to_rgb8(const float *rgba, int width, int height, float scale, float gamma)
{
    std::vector<uint8_t> out((size_t)width * height * 3);
    const float inv_gamma = 1.0f / gamma;
    for (size_t i = 0; i < (size_t)width * height; i++)
        for (int k = 0; k < 3; k++) {
            float v = rgba[i * 4 + k] * scale;
            v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);              /* NaN -> 0 */
            if (gamma != 1.0f) v = std::pow(v, inv_gamma);
            out[i * 3 + k] = (uint8_t)std::lround(v * 255.0f);
        }
    return out;
}

bool write_ppm(const std::string &path, const float *rgba, int width, int height, float scale, float gamma)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const std::vector<uint8_t> px = to_rgb8(rgba, width, height, scale, gamma);
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    const bool ok = std::fwrite(px.data(), 1, px.size(), f) == px.size();
    return (std::fclose(f) == 0) && ok;
}

namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

bool write_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    uint32_t crc = crc32_update(0xFFFFFFFFu, (const uint8_t *)type, 4);
    if (!data.empty()) crc = crc32_update(crc, data.data(), data.size());
    std::vector<uint8_t> tail;
    put_be32(tail, crc ^ 0xFFFFFFFFu);
    return std::fwrite(head.data(), 1, 4, f) == 4 && std::fwrite(type, 1, 4, f) == 4 &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) && std::fwrite(tail.data(), 1, 4, f) == 4;
}

}  // namespace

/* 8-bit RGB PNG; the zlib stream uses stored (uncompressed) deflate blocks, which every decoder reads */
bool write_png(const std::string &path, const float *rgba, int width, int height, float scale, float gamma)
{
    if (width <= 0 || height <= 0) return false;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const std::vector<uint8_t> px = to_rgb8(rgba, width, height, scale, gamma);
    /* scanlines, each behind filter byte 0 */
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), px.begin() + (size_t)y * width * 3, px.begin() + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);                                 /* zlib header: deflate, 32 K window, no preset */
    uint32_t a = 1, b = 0;                                                /* Adler-32 of raw */
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);                       /* BFINAL, BTYPE = 00 */
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   /* 8-bit, RGB */
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    const bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) &&
                    write_chunk(f, "IEND", std::vector<uint8_t>());
    return (std::fclose(f) == 0) && ok;
}

bool write_image(const std::string &path, const float *rgba, int width, int height, float scale, float gamma)
{
    const size_t dot = path.rfind('.');
    const std::string ext = dot == std::string::npos ? std::string() : path.substr(dot + 1);
    if (ext == "pfm") return write_pfm(path, rgba, width, height);
    if (ext == "ppm") return write_ppm(path, rgba, width, height, scale, gamma);
    if (ext == "png") return write_png(path, rgba, width, height, scale, gamma);
    return false;
}

}  // namespace bdpt_host
