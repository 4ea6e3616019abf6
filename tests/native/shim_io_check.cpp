// Malformed-input check of the shim's file readers (3dscan_amd/csrc/sl3d_shim_io.h), built with -fsanitize=address,undefined by
// tests/test_shim_io.py: well-formed files must parse to the right values; truncated, oversized, garbage and lying files must be
// refused (or parsed) WITHOUT touching memory they do not own, allocating what their headers claim, or running into undefined
// behaviour.  Test infrastructure only.   usage: shim_io_check <scratch directory>   -> exit code 0, "ok <cases>" on stdout
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "sl3d_shim_io.h"

using namespace sl3d_io;
static int failures = 0, cases = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        cases++;                                                                \
        if (!(cond)) { failures++; fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); } \
    } while (0)

static void put(const std::string &path, const std::vector<uint8_t> &b)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (b.size()) fwrite(b.data(), 1, b.size(), f);
    fclose(f);
}
static void put_text(const std::string &path, const std::string &s) { put(path, std::vector<uint8_t>(s.begin(), s.end())); }
static void le32(std::vector<uint8_t> &b, size_t o, uint32_t v) { for (int k = 0; k < 4; k++) b[o + k] = (uint8_t)(v >> (8 * k)); }

// a well-formed BMP of w x h: 8 bits with a grey ramp palette (value = (x + 2 y) & 255) or 24 bits (B, G, R = x, y, x ^ y)
static std::vector<uint8_t> bmp(int w, int h, int bpp, bool top_down = false, uint32_t ncolors = 0)
{
    const size_t rowbytes = (((size_t)w * bpp + 31) / 32) * 4, pal = bpp == 8 ? 4u * (ncolors ? ncolors : 256u) : 0u, off = 54 + pal;
    std::vector<uint8_t> b(off + rowbytes * h, 0);
    b[0] = 'B'; b[1] = 'M';
    le32(b, 2, (uint32_t)b.size()); le32(b, 10, (uint32_t)off); le32(b, 14, 40); le32(b, 18, (uint32_t)w); le32(b, 22, (uint32_t)(top_down ? -h : h));
    b[26] = 1; b[28] = (uint8_t)bpp; le32(b, 46, ncolors);
    if (bpp == 8)
        for (uint32_t i = 0; i < (ncolors ? ncolors : 256u); i++) b[54 + 4 * i] = b[54 + 4 * i + 1] = b[54 + 4 * i + 2] = (uint8_t)i;
    for (int i = 0; i < h; i++) {
        const int y = top_down ? i : h - 1 - i;
        uint8_t *row = &b[off + (size_t)i * rowbytes];
        for (int x = 0; x < w; x++) {
            if (bpp == 8) row[x] = (uint8_t)((x + 2 * y) & (ncolors ? (int)ncolors - 1 : 255));
            else { row[3 * x] = (uint8_t)x; row[3 * x + 1] = (uint8_t)y; row[3 * x + 2] = (uint8_t)(x ^ y); }
        }
    }
    return b;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const std::string p = dir + "/case.bin";
    const int W = 37, H = 11;
    std::vector<uint8_t> gray((size_t)W * H), bgr, scratch;
    std::mt19937 rng(7);

    // ---- BMP: well-formed files
    for (int bpp : {8, 24})
        for (bool td : {false, true}) {
            put(p, bmp(W, H, bpp, td));
            CHECK(read_bmp_gray(p, W, H, gray.data(), &scratch));
            bool same = true;
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) same &= gray[(size_t)y * W + x] == (bpp == 8 ? (uint8_t)((x + 2 * y) & 255) : bgr2gray(x, y, x ^ y));
            CHECK(same);
            CHECK(read_bmp_bgr(p, W, H, bgr) && bgr.size() == (size_t)W * H * 3);
            CHECK(bpp == 8 ? bgr[3 * (size_t)(5 * W + 3)] == (uint8_t)(3 + 10) : (bgr[3 * (size_t)(5 * W + 3)] == 3 && bgr[3 * (size_t)(5 * W + 3) + 1] == 5));
        }
    put(p, bmp(W, H, 8, false, 16));  // a short palette
    CHECK(read_bmp_gray(p, W, H, gray.data()) && gray[0] == 0 && gray[(size_t)W + 1] == 3);
    // ---- BMP: every truncation of a well-formed file, wrong sizes, lying headers
    for (int bpp : {8, 24}) {
        const std::vector<uint8_t> good = bmp(W, H, bpp);
        for (size_t n = 0; n < good.size(); n += (n < 1200 ? 1 : 97)) {
            put(p, std::vector<uint8_t>(good.begin(), good.begin() + (long)n));
            CHECK(!read_bmp_gray(p, W, H, gray.data(), &scratch));
            CHECK(!read_bmp_bgr(p, W, H, bgr));
        }
        CHECK(!read_bmp_gray(p, W + 1, H, gray.data()) && !read_bmp_gray(p, W, H - 1, gray.data()));
        std::vector<uint8_t> b = good;
        le32(b, 10, 0xfffffff0u);  // pixel array far behind the file
        put(p, b); CHECK(!read_bmp_gray(p, W, H, gray.data()) && !read_bmp_bgr(p, W, H, bgr));
        b = good; le32(b, 10, 10);  // ... inside the header
        put(p, b); CHECK(!read_bmp_gray(p, W, H, gray.data()) && !read_bmp_bgr(p, W, H, bgr));
        b = good; le32(b, 14, 0x7fffffffu);  // a DIB header that puts the palette beyond the file
        put(p, b); CHECK(!read_bmp_gray(p, W, H, gray.data()) && !read_bmp_bgr(p, W, H, bgr));
        b = good; le32(b, 46, 0x40000000u);  // a billion palette entries
        put(p, b); CHECK(bpp == 24 || (!read_bmp_gray(p, W, H, gray.data()) && !read_bmp_bgr(p, W, H, bgr)));
        b = good; le32(b, 30, 1);  // compressed
        put(p, b); CHECK(!read_bmp_gray(p, W, H, gray.data()));
        b = good; b[28] = 4;  // 4 bits per pixel
        put(p, b); CHECK(!read_bmp_gray(p, W, H, gray.data()) && !read_bmp_bgr(p, W, H, bgr));
    }
    // ---- BMP: garbage of every length up to a few KB, and a good header over garbage
    for (int t = 0; t < 300; t++) {
        std::vector<uint8_t> b(rng() % 3000);
        for (auto &x : b) x = (uint8_t)rng();
        if (t % 3 == 0 && b.size() > 60) { const std::vector<uint8_t> g = bmp(W, H, t % 2 ? 8 : 24); std::copy(g.begin(), g.begin() + 30, b.begin()); }
        put(p, b);
        (void)read_bmp_gray(p, W, H, gray.data(), &scratch);  // any answer; no crash, no sanitizer report
        (void)read_bmp_bgr(p, W, H, bgr);
        cases++;
    }
    CHECK(!read_bmp_gray(dir + "/does_not_exist.bmp", W, H, gray.data()));

    // ---- OpenCV XML matrices
    double m[9];
    CHECK(parse_xml_matrix("<?xml version=\"1.0\"?><opencv_storage><M><rows>3</rows><data>\n 1. 2.5e+00 -3 4 5 6 7 8 9e-1</data></M></opencv_storage>", 9, m) && m[1] == 2.5 && m[8] == 0.9);
    CHECK(!parse_xml_matrix("<data>1 2 3</data>", 4, m));                 // too few numbers
    CHECK(!parse_xml_matrix("<data>1 2 3", 3, m));                        // no closing tag
    CHECK(!parse_xml_matrix("</data> 1 2 3 <data>", 3, m));               // tags in the wrong order
    CHECK(!parse_xml_matrix("<data>1 2</data> 3 4 5", 3, m));             // the numbers behind </data> do not count
    CHECK(!parse_xml_matrix("<data>abc</data>", 1, m) && !parse_xml_matrix("", 1, m) && !parse_xml_matrix("<data></data>", 1, m));
    CHECK(parse_xml_matrix("<data>1e999 nan -inf</data>", 3, m));         // whatever strtod makes of it, but no failure of ours
    for (int t = 0; t < 200; t++) {
        std::string s(rng() % 400, ' ');
        for (auto &ch : s) ch = "<>/data0123456789.e-+ \n"[rng() % 23];
        (void)parse_xml_matrix(s, (int)(rng() % 10), m);
        cases++;
    }
    std::string txt;
    put_text(p, "<data>1 2 3</data>");
    CHECK(read_text_file(p, txt) && txt.size() == 18 && !read_text_file(p, txt, 5) && !read_text_file(dir + "/none.xml", txt));

    // ---- PLY
    PlyCloud c;
    const std::string hdr_a = "ply\nformat ascii 1.0\ncomment x\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    put_text(p, hdr_a + "1 2 3 10 20 30\n4 5 6 40 50 60\n-7.5 8 9e1 255 0 128\n");
    CHECK(read_ply(p, c) && c.xyz.size() == 9 && c.xyz[6] == -7.5f && c.xyz[8] == 90.f && c.rgb[3] == 40 && c.rgb[8] == 128);
    put_text(p, hdr_a + "1 2 3 10 20 30\n4 5 6 40 50\n");                  // rows missing
    CHECK(!read_ply(p, c));
    put_text(p, hdr_a + "1 2 3 1e9 -5 nan\n4 5 6 40 50 60\n1e60 -1e60 9 255 0 128\n");   // colours / coordinates out of range: saturate, no UB
    CHECK(read_ply(p, c) && c.rgb[0] == 255 && c.rgb[1] == 0 && c.rgb[2] == 0 && std::isinf(c.xyz[6]) && c.xyz[6] > 0 && c.xyz[7] < 0);
    for (const char *nv : {"4000000000000", "-3", "99999999999999999999999", "12abc", "1000000"}) {   // vertex counts the file cannot hold
        put_text(p, std::string("ply\nformat ascii 1.0\nelement vertex ") + nv + "\nproperty float x\nproperty float y\nproperty float z\nend_header\n1 2 3\n");
        CHECK(!read_ply(p, c));
        put_text(p, std::string("ply\nformat binary_little_endian 1.0\nelement vertex ") + nv + "\nproperty float x\nproperty float y\nproperty float z\nend_header\n123456789012");
        CHECK(!read_ply(p, c));
    }
    {   // binary, mixed types, and its truncations
        std::string b = "ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty double x\nproperty float y\nproperty short z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty int extra\nend_header\n";
        const size_t h = b.size();
        for (int v = 0; v < 2; v++) {
            double x = 1.5 + v; float y = -2.f * (v + 1); int16_t z = (int16_t)(-7 - v); uint8_t col[3] = {1, 2, (uint8_t)(3 + v)}; int32_t e = 77;
            b.append((const char *)&x, 8); b.append((const char *)&y, 4); b.append((const char *)&z, 2); b.append((const char *)col, 3); b.append((const char *)&e, 4);
        }
        put_text(p, b);
        CHECK(read_ply(p, c) && c.xyz.size() == 6 && c.xyz[3] == 2.5f && c.xyz[4] == -4.f && c.xyz[5] == -8.f && c.rgb[5] == 4);
        for (size_t n = 0; n < b.size(); n += (n < h ? 3 : 1)) {
            put_text(p, b.substr(0, n));
            CHECK(!read_ply(p, c));
        }
    }
    put_text(p, "ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty quaternion x\nproperty float y\nproperty float z\nend_header\nxxxxxxxxxxxx");
    CHECK(!read_ply(p, c));                                                  // unknown scalar type
    put_text(p, "ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nend_header\n1 2\n");
    CHECK(!read_ply(p, c));                                                  // no z
    {
        std::string many = "ply\nformat ascii 1.0\nelement vertex 1\n";
        for (int i = 0; i < 100; i++) many += "property float p" + std::to_string(i) + "\n";
        put_text(p, many + "end_header\n");
        CHECK(!read_ply(p, c));                                              // more properties than anything of ours writes
    }
    for (int t = 0; t < 300; t++) {
        std::string s = t % 2 ? hdr_a : std::string("ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nend_header\n");
        const size_t n = rng() % 200;
        for (size_t i = 0; i < n; i++) s.push_back((char)rng());
        if (t % 5 == 0) s = s.substr(rng() % (s.size() + 1));
        put_text(p, s);
        (void)read_ply(p, c);
        cases++;
    }
    CHECK(!read_ply(dir + "/none.ply", c));
    remove(p.c_str());
    if (failures) { fprintf(stderr, "%d of %d checks failed\n", failures, cases); return 1; }
    printf("ok %d\n", cases);
    return 0;
}
