// png_io.h -- dependency-free PNG read/write for the CLI host (the reference decodes with
// ffmpeg and always encodes PNG, src/imagefileio.rs:84-183,:217-271; codecs are out of the
// hot path, this is just enough to move real images through `reforge -i in.png -o out.png`).
//
// Reader: 8-bit greyscale / greyscale+alpha / RGB / RGBA, non-interlaced, any zlib stream
// (stored, fixed and dynamic Huffman blocks), all five scanline filters.  Output is RGBA8.
// Writer: RGBA8, filter 0, stored deflate blocks.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace pngio {

inline uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n)
{
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
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}

// ---- inflate (RFC 1951) ------------------------------------------------------------
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    bool ok = true;
    uint32_t bits(int k)
    {
        while (cnt < k) {
            if (pos >= n) { ok = false; return 0; }
            buf |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        uint32_t v = k ? (buf & ((1u << k) - 1)) : 0;
        buf >>= k;
        cnt -= k;
        return v;
    }
    void align() { buf = 0; cnt = 0; }
};

struct Huffman {
    uint16_t count[16] = {};
    uint16_t symbol[320] = {};
    void build(const uint8_t* lengths, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

inline bool inflate(const uint8_t* data, size_t n, std::vector<uint8_t>& out)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n < 2) return false;
    BitReader br{data + 2, n - 2};   // skip the zlib header (CMF, FLG)
    int last;
    do {
        last = (int)br.bits(1);
        int type = (int)br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
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
                int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman lencode;
                lencode.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lengths[idx++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return false;
                            val = lengths[idx - 1];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)br.bits(3);
                        } else {
                            rep = 11 + (int)br.bits(7);
                        }
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lengths[idx++] = (uint8_t)val;
                    }
                }
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0 || !br.ok) return false;
                if (sym < 256) {
                    out.push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    int len = lbase[sym] + (int)br.bits(lext[sym]);
                    int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (d > out.size()) return false;
                    size_t from = out.size() - d;
                    for (int i = 0; i < len; ++i) out.push_back(out[from + (size_t)i]);
                }
            }
        } else {
            return false;
        }
    } while (!last);
    return true;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// Decodes a PNG file into RGBA8.  err explains a refusal.
inline bool read_png(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open file"; return false; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    w = h = 0;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        if (pos + 12 + len > file.size()) { err = "truncated chunk"; return false; }
        const uint8_t* type = &file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        // chunk CRC-32 over type + data: a corrupt file fails here instead of decoding silently
        if ((crc32_update(0xFFFFFFFFu, type, 4 + (size_t)len) ^ 0xFFFFFFFFu) != be32(data + len)) { err = "chunk CRC mismatch"; return false; }
        if (std::memcmp(type, "IHDR", 4) == 0 && len >= 13) {
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + len;
    }
    if (w < 1 || h < 1) { err = "missing IHDR"; return false; }
    // a crafted IHDR must not drive the allocations below: 2^28 pixels (1 GiB of RGBA8) is far
    // beyond any frame this host uploads
    if ((uint64_t)w * (uint64_t)h > (1ull << 28)) { err = "image dimensions too large"; return false; }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (depth != 8 || ch == 0 || interlace != 0) { err = "only 8-bit grey/RGB/RGBA non-interlaced PNGs are decoded"; return false; }
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * ch + 1));
    if (!inflate(idat.data(), idat.size(), raw)) { err = "corrupt zlib stream"; return false; }
    const size_t stride = (size_t)w * ch;
    if (raw.size() < (size_t)h * (stride + 1)) { err = "image data too short"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(size_t)y * (stride + 1)];
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: err = "bad scanline filter"; return false;
            }
            cur[i] = (uint8_t)(line[1 + i] + pred);
        }
        for (int x = 0; x < w; ++x) {
            uint8_t* o = &rgba[((size_t)y * w + x) * 4];
            const uint8_t* s = &cur[(size_t)x * ch];
            if (ch == 1) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
            else if (ch == 2) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
            else if (ch == 3) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; }
            else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }
        }
        prev.swap(cur);
    }
    return true;
}

inline void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

inline void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data)
{
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}

// RGBA8 PNG with stored (uncompressed) deflate blocks: valid for every decoder
inline bool write_png(const std::string& path, int w, int h, const uint8_t* rgba)
{
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);   // filter: none
        raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    size_t pos = 0;
    do {
        size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)out.data(), (std::streamsize)out.size());
    return (bool)f;
}

}  // namespace pngio
