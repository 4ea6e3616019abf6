// sl3d_shim_io.h -- the file readers of the drop-in shim (sl3d_shim.cpp), on their own: plain C++17, no HIP, no shim state, so that
// the test suite can compile them with -fsanitize=address,undefined and feed them truncated / oversized / garbage files
// (tests/native/shim_io_check.cpp, tests/test_shim_io.py).
//   read_bmp_gray / read_bmp_bgr : what cvLoadImage yields for the reference's BMP files (3/wrapped_phase.cpp:44,
//                                  4/phase_unwrap.cpp:78,84, 8/save_point_cloud.cpp:46)
//   parse_xml_matrix             : the numbers of an OpenCV XML matrix (cvReadByName, 7/triangulation.cpp:152-168,1069-1083)
//   read_ply                     : the vertex list of Point_cloud/point_cloud_<i>.ply (9/register_point_clouds.cpp:83-128)
// Every size a file CLAIMS is checked against the bytes the file HAS before anything is allocated or read with it: a malformed file
// is a `false`, never an allocation of its header's vertex count or a read through its header's offsets.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sl3d_io {

// OpenCV's fixed-point gray weights (B 1868, G 9617, R 4899, >> 14)
inline uint8_t bgr2gray(int b, int gch, int r) { return (uint8_t)((b * 1868 + gch * 9617 + r * 4899 + (1 << 13)) >> 14); }

struct File {
    FILE *f = nullptr;
    long size = 0;
    explicit File(const std::string &path)
    {
        f = fopen(path.c_str(), "rb");
        if (f && (fseek(f, 0, SEEK_END) != 0 || (size = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0)) {
            fclose(f);
            f = nullptr;
        }
    }
    ~File() { if (f) fclose(f); }
    File(const File &) = delete;
    File &operator=(const File &) = delete;
};

// the 54 header bytes both BMP readers share, checked against the file: W x H (bottom-up, or top-down with a negative height),
// uncompressed, 8 or 24 bits, palette and pixel array inside the file
struct BmpHeader {
    int bpp = 0;
    bool top_down = false;
    uint32_t data_off = 0, palette_off = 0, ncolors = 0;
    size_t rowbytes = 0;
};
inline bool read_bmp_header(File &in, int W, int H, BmpHeader &h)
{
    uint8_t hdr[54];
    if (!in.f || W < 1 || H < 1 || fread(hdr, 1, 54, in.f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') return false;
    auto u32 = [&](int o) { return (uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8) | ((uint32_t)hdr[o + 2] << 16) | ((uint32_t)hdr[o + 3] << 24); };
    const uint32_t dib = u32(14);
    const int32_t w = (int32_t)u32(18), hgt = (int32_t)u32(22);
    h.bpp = hdr[28] | (hdr[29] << 8);
    h.data_off = u32(10);
    h.ncolors = u32(46);
    if (w != W || (hgt != H && hgt != -H) || u32(30) != 0 || (h.bpp != 8 && h.bpp != 24) || dib < 40 || dib > 124) return false;
    h.top_down = hgt < 0;
    h.rowbytes = (((size_t)W * (size_t)h.bpp + 31) / 32) * 4;
    h.palette_off = 14 + dib;
    const uint64_t fsize = (uint64_t)in.size;
    if (h.bpp == 8) {
        if (h.ncolors == 0) h.ncolors = 256;
        if (h.ncolors > 256 || (uint64_t)h.palette_off + 4ull * h.ncolors > fsize) return false;
    }
    if (h.data_off < 54 || (uint64_t)h.data_off + (uint64_t)h.rowbytes * (uint64_t)H > fsize) return false;
    return true;
}

// out: W * H bytes, top-down rows of W bytes.  scratch: the file's pixel array is read into it in one piece; a caller that decodes
// the same number of files scan after scan hands in the same vectors again, so their pages are touched once, not once per scan
inline bool read_bmp_gray(const std::string &path, int W, int H, uint8_t *out, std::vector<uint8_t> *scratch = nullptr)
{
    File in(path);
    BmpHeader h;
    if (!read_bmp_header(in, W, H, h)) return false;
    uint8_t pal[256];
    for (int i = 0; i < 256; i++) pal[i] = (uint8_t)i;
    if (h.bpp == 8) {
        if (fseek(in.f, (long)h.palette_off, SEEK_SET) != 0) return false;
        for (uint32_t i = 0; i < h.ncolors; i++) {
            uint8_t q[4];
            if (fread(q, 1, 4, in.f) != 4) return false;
            pal[i] = bgr2gray(q[0], q[1], q[2]);
        }
    }
    bool identity = h.bpp == 8;  // the grey ramp cvSaveImage writes for a 1-channel image: rows are copied, not looked up
    for (int i = 0; i < 256 && identity; i++) identity = pal[i] == (uint8_t)i;
    std::vector<uint8_t> own;
    std::vector<uint8_t> &file = scratch ? *scratch : own;
    file.resize(h.rowbytes * (size_t)H);  // one read for the whole pixel array (its size was checked against the file's)
    if (fseek(in.f, (long)h.data_off, SEEK_SET) != 0 || fread(file.data(), 1, file.size(), in.f) != file.size()) return false;
    for (int i = 0; i < H; i++) {
        const uint8_t *row = file.data() + (size_t)i * h.rowbytes;
        uint8_t *dst = out + (size_t)(h.top_down ? i : H - 1 - i) * W;  // bottom-up unless the height is negative
        if (identity) memcpy(dst, row, (size_t)W);
        else if (h.bpp == 8)
            for (int x = 0; x < W; x++) dst[x] = pal[row[x]];
        else
            for (int x = 0; x < W; x++) dst[x] = bgr2gray(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
    }
    return true;
}

// B,G,R interleaved, top-down: what cvLoadImage(path) (colour) yields; 8-bit files go through their palette, 24-bit files are copied
inline bool read_bmp_bgr(const std::string &path, int W, int H, std::vector<uint8_t> &out)
{
    File in(path);
    BmpHeader h;
    if (!read_bmp_header(in, W, H, h)) return false;
    uint8_t pal[256][3];
    for (int i = 0; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = (uint8_t)i;
    if (h.bpp == 8) {
        if (fseek(in.f, (long)h.palette_off, SEEK_SET) != 0) return false;
        for (uint32_t i = 0; i < h.ncolors; i++) {
            uint8_t q[4];
            if (fread(q, 1, 4, in.f) != 4) return false;
            pal[i][0] = q[0]; pal[i][1] = q[1]; pal[i][2] = q[2];
        }
    }
    std::vector<uint8_t> row(h.rowbytes);
    out.assign((size_t)W * H * 3, 0);
    if (fseek(in.f, (long)h.data_off, SEEK_SET) != 0) return false;
    for (int i = 0; i < H; i++) {
        if (fread(row.data(), 1, h.rowbytes, in.f) != h.rowbytes) return false;
        uint8_t *dst = out.data() + (size_t)(h.top_down ? i : H - 1 - i) * W * 3;
        if (h.bpp == 24) memcpy(dst, row.data(), (size_t)W * 3);
        else
            for (int x = 0; x < W; x++) memcpy(dst + 3 * x, pal[row[x]], 3);
    }
    return true;
}

// the first `count` numbers between <data> and </data> of an OpenCV XML matrix; false if the text holds fewer
inline bool parse_xml_matrix(const std::string &text, int count, double *out)
{
    const size_t a = text.find("<data>"), b = text.find("</data>");
    if (count < 0 || a == std::string::npos || b == std::string::npos || b < a + 6) return false;
    const std::string body = text.substr(a + 6, b - (a + 6));  // (a copy: strtod must not run past </data> into whatever follows)
    const char *p = body.c_str();
    for (int i = 0; i < count; i++) {
        char *q = nullptr;
        out[i] = strtod(p, &q);
        if (q == p) return false;
        p = q;
    }
    return true;
}
inline bool read_text_file(const std::string &path, std::string &s, size_t limit = (size_t)16 << 20)
{
    File in(path);
    if (!in.f || (size_t)in.size > limit) return false;
    s.resize((size_t)in.size);
    return in.size == 0 || fread(&s[0], 1, s.size(), in.f) == s.size();
}

struct PlyCloud {
    std::vector<float> xyz;
    std::vector<uint8_t> rgb;
};
// ASCII or binary_little_endian PLY with scalar vertex properties (what save_point_cloud() writes in either format).  The vertex
// count of the header is believed only as far as the file can hold it: a binary vertex takes its record size, an ASCII vertex at
// least two bytes per property.
inline bool read_ply(const std::string &path, PlyCloud &c)
{
    File in(path);
    if (!in.f) return false;
    char line[512];
    long long nv = -1;
    std::vector<std::string> props, types;
    bool ascii = false, binary = false, in_vertex = false, header_ok = false;
    int lines = 0;
    while (fgets(line, sizeof line, in.f)) {
        if (++lines > 4096) return false;  // (no header of ours is anywhere near that long)
        char a[64] = "", b[64] = "", d[64] = "";
        const int k = sscanf(line, "%63s %63s %63s", a, b, d);
        if (lines == 1 && (k < 1 || strcmp(a, "ply"))) return false;
        if (k >= 1 && !strcmp(a, "end_header")) { header_ok = true; break; }
        if (k >= 2 && !strcmp(a, "format")) { ascii = !strcmp(b, "ascii"); binary = !strcmp(b, "binary_little_endian"); }
        if (k >= 3 && !strcmp(a, "element")) {
            in_vertex = !strcmp(b, "vertex");
            if (in_vertex) {
                char *e = nullptr;
                nv = strtoll(d, &e, 10);
                if (e == d || *e != '\0') return false;
            }
        }
        if (k >= 3 && !strcmp(a, "property") && in_vertex && strcmp(b, "list")) {
            if (props.size() >= 64) return false;
            props.push_back(d);
            types.push_back(b);
        }
    }
    if (!header_ok || (!ascii && !binary) || nv < 0 || props.empty()) return false;
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1;
    for (int i = 0; i < (int)props.size(); i++) {
        if (props[i] == "x") ix = i; else if (props[i] == "y") iy = i; else if (props[i] == "z") iz = i;
        else if (props[i] == "red" || props[i] == "r") ir = i; else if (props[i] == "green" || props[i] == "g") ig = i;
        else if (props[i] == "blue" || props[i] == "b") ib = i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return false;
    // byte size of a scalar PLY type (0 = unknown)
    auto tsize = [](const std::string &t) -> int {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        return 0;
    };
    size_t rec = 0;
    if (binary)
        for (auto &t : types) { if (!tsize(t)) return false; rec += (size_t)tsize(t); }
    const long here = ftell(in.f);
    if (here < 0 || here > in.size) return false;
    const unsigned long long left = (unsigned long long)(in.size - here), per_vertex = binary ? rec : 2ull * props.size();
    if ((unsigned long long)nv > left / per_vertex) return false;  // the header claims more vertices than the file can hold
    c.xyz.resize((size_t)nv * 3);
    c.rgb.assign((size_t)nv * 3, 0);
    std::vector<double> v(props.size());
    std::vector<uint8_t> buf(rec);
    auto u8 = [](double x) { return (uint8_t)(x >= 255.0 ? 255 : x > 0.0 ? (int)x : 0); };  // (NaN and negatives: 0)
    for (long long p = 0; p < nv; p++) {
        if (binary) {
            if (fread(buf.data(), 1, rec, in.f) != rec) return false;
            size_t o = 0;
            for (size_t i = 0; i < props.size(); i++) {
                const std::string &t = types[i];
                const int sz = tsize(t);
                if (t == "float" || t == "float32") { float q; memcpy(&q, &buf[o], 4); v[i] = q; }
                else if (t == "double" || t == "float64") { double q; memcpy(&q, &buf[o], 8); v[i] = q; }
                else if (sz == 1) v[i] = (t == "char" || t == "int8") ? (double)(int8_t)buf[o] : (double)buf[o];
                else if (sz == 2) { uint16_t q; memcpy(&q, &buf[o], 2); v[i] = (t == "short" || t == "int16") ? (double)(int16_t)q : (double)q; }
                else { uint32_t q; memcpy(&q, &buf[o], 4); v[i] = (t == "int" || t == "int32") ? (double)(int32_t)q : (double)q; }
                o += (size_t)sz;
            }
        } else {
            for (size_t i = 0; i < props.size(); i++)
                if (fscanf(in.f, "%lf", &v[i]) != 1) return false;
        }
        // (a double beyond the float range becomes +-inf by saturation, not by an out-of-range conversion)
        auto f32 = [](double x) { return std::fabs(x) > 3.4028234663852886e38 ? (x > 0 ? HUGE_VALF : -HUGE_VALF) : (float)x; };
        c.xyz[3 * p] = f32(v[ix]); c.xyz[3 * p + 1] = f32(v[iy]); c.xyz[3 * p + 2] = f32(v[iz]);
        if (ir >= 0 && ig >= 0 && ib >= 0) { c.rgb[3 * p] = u8(v[ir]); c.rgb[3 * p + 1] = u8(v[ig]); c.rgb[3 * p + 2] = u8(v[ib]); }
    }
    return true;
}

}  // namespace sl3d_io
