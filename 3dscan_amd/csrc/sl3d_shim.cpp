// sl3d_shim.cpp -- drop-in for the reference's stage objects 3/4/5/7: same four C++ entry points, same
// global arrays, same input files, but the arithmetic runs in the HIP kernels behind the C ABI.
// See include/sl3d_shim.h.  Plain C++ (no HIP here); links against libsl3d.so.
#include "../../include/sl3d_shim.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../../include/sl3d.h"

namespace {

constexpr int W = Camera_imagewidth, H = Camera_imageheight;
const char *kReferenceRoot = "/home/pranav/Desktop/M_tech_project_console";  // 3/wrapped_phase.cpp:39, 7/triangulation.cpp:152

// One scan runs on one GPU (a single context) or, with SL3D_DEVICES=0,1,..., as row stripes on several (sl3d_group_*: one
// context per listed device, stripe order = row order).  Every stage function below walks the parts; a part's results are
// rows [row0, row0 + rows) of the row-major planes the reference's [col][row] globals are filled from.
struct Part {
    sl3d_ctx *ctx;
    int row0, rows;
};

struct Shim {
    sl3d_ctx *ctx = nullptr;    // the first part's context (library-wide calls, error texts)
    sl3d_group *group = nullptr;
    std::vector<Part> parts;
    std::string root;
    bool root_set = false;
    bool write_debug = false;
    int status = SL3D_OK;
    std::string err;
    // the configuration the context was created with (the scalar globals may change between scans)
    int F = 0, Nv = 0, Nh = 0, fwv = 0, fwh = 0, ncv = 0, nch = 0;
} g;

std::string data_root()
{
    if (g.root_set) return g.root;
    const char *e = getenv("SL3D_DATA_ROOT");
    return e ? std::string(e) : std::string(kReferenceRoot);
}

bool fail(int code, const std::string &msg)
{
    g.status = code;
    g.err = msg;
    fprintf(stderr, "\nsl3d shim: %s", msg.c_str());  // the reference reports with printf and carries on
    return false;
}

bool ok(int rc, const char *what)
{
    if (rc == SL3D_OK) return true;
    return fail(rc, std::string(what) + ": " + sl3d_strerror(rc) + ": " + sl3d_last_error(g.ctx));
}

// ---- 8-bit gray planes from BMP files: what cvLoadImage(..., CV_LOAD_IMAGE_GRAYSCALE) yields ----
// (3/wrapped_phase.cpp:44, 4/phase_unwrap.cpp:78,84).  8-bit palettised and 24-bit BMPs; colour is converted
// with OpenCV's fixed-point weights (B 1868, G 9617, R 4899, >> 14).
inline uint8_t bgr2gray(int b, int gch, int r) { return (uint8_t)((b * 1868 + gch * 9617 + r * 4899 + (1 << 13)) >> 14); }

bool read_bmp_gray(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t hdr[54];
    if (fread(hdr, 1, 54, f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') { fclose(f); return false; }
    auto u32 = [&](int o) { return (uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8) | ((uint32_t)hdr[o + 2] << 16) | ((uint32_t)hdr[o + 3] << 24); };
    const uint32_t data_off = u32(10), dib = u32(14);
    const int32_t w = (int32_t)u32(18), hgt = (int32_t)u32(22);
    const int bpp = hdr[28] | (hdr[29] << 8);
    const uint32_t compression = u32(30);
    uint32_t ncolors = u32(46);
    if (w != W || (hgt != H && hgt != -H) || compression != 0 || (bpp != 8 && bpp != 24)) { fclose(f); return false; }
    uint8_t pal[256];
    for (int i = 0; i < 256; i++) pal[i] = (uint8_t)i;
    if (bpp == 8) {
        if (ncolors == 0) ncolors = 256;
        fseek(f, 14 + dib, SEEK_SET);
        for (uint32_t i = 0; i < ncolors && i < 256; i++) {
            uint8_t q[4];
            if (fread(q, 1, 4, f) != 4) { fclose(f); return false; }
            pal[i] = bgr2gray(q[0], q[1], q[2]);
        }
    }
    const size_t rowbytes = (((size_t)w * bpp + 31) / 32) * 4;
    std::vector<uint8_t> row(rowbytes);
    out.assign((size_t)W * H, 0);
    fseek(f, data_off, SEEK_SET);
    for (int i = 0; i < H; i++) {
        if (fread(row.data(), 1, rowbytes, f) != rowbytes) { fclose(f); return false; }
        const int y = hgt > 0 ? H - 1 - i : i;  // bottom-up unless the height is negative
        uint8_t *dst = out.data() + (size_t)y * W;
        if (bpp == 8)
            for (int x = 0; x < W; x++) dst[x] = pal[row[x]];
        else
            for (int x = 0; x < W; x++) dst[x] = bgr2gray(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
    }
    fclose(f);
    return true;
}

// B,G,R interleaved, top-down: what cvLoadImage(path) (colour) yields (8/save_point_cloud.cpp:46); 8-bit files go through
// their palette, 24-bit files are copied
bool read_bmp_bgr(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t hdr[54];
    if (fread(hdr, 1, 54, f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') { fclose(f); return false; }
    auto u32 = [&](int o) { return (uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8) | ((uint32_t)hdr[o + 2] << 16) | ((uint32_t)hdr[o + 3] << 24); };
    const uint32_t data_off = u32(10), dib = u32(14);
    const int32_t w = (int32_t)u32(18), hgt = (int32_t)u32(22);
    const int bpp = hdr[28] | (hdr[29] << 8);
    uint32_t ncolors = u32(46);
    if (w != W || (hgt != H && hgt != -H) || u32(30) != 0 || (bpp != 8 && bpp != 24)) { fclose(f); return false; }
    uint8_t pal[256][3];
    for (int i = 0; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = (uint8_t)i;
    if (bpp == 8) {
        if (ncolors == 0) ncolors = 256;
        fseek(f, 14 + dib, SEEK_SET);
        for (uint32_t i = 0; i < ncolors && i < 256; i++) {
            uint8_t q[4];
            if (fread(q, 1, 4, f) != 4) { fclose(f); return false; }
            pal[i][0] = q[0]; pal[i][1] = q[1]; pal[i][2] = q[2];
        }
    }
    const size_t rowbytes = (((size_t)w * bpp + 31) / 32) * 4;
    std::vector<uint8_t> row(rowbytes);
    out.assign((size_t)W * H * 3, 0);
    fseek(f, data_off, SEEK_SET);
    for (int i = 0; i < H; i++) {
        if (fread(row.data(), 1, rowbytes, f) != rowbytes) { fclose(f); return false; }
        uint8_t *dst = out.data() + (size_t)(hgt > 0 ? H - 1 - i : i) * W * 3;
        if (bpp == 24) memcpy(dst, row.data(), (size_t)W * 3);
        else
            for (int x = 0; x < W; x++) memcpy(dst + 3 * x, pal[row[x]], 3);
    }
    fclose(f);
    return true;
}

// 8-bit palettised BMP exactly as the reference's cvSaveImage (OpenCV 2.4 BMP encoder) writes a 1-channel image:
// 14 + 40 byte headers with biSizeImage = biClrUsed = 0, 256 grey palette entries, bottom-up rows padded to 4 bytes.
// (tests/test_gpu_shim.py compares whole files with the SHA-256 of the reference's own pattern images.)
bool write_bmp_gray(const std::string &path, const uint8_t *img, int w = W, int h = H)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint32_t rowbytes = ((uint32_t)w + 3) & ~3u, off = 54 + 1024, size = off + rowbytes * h;
    uint8_t hdr[54] = {0};
    auto put32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = v >> 24; };
    hdr[0] = 'B'; hdr[1] = 'M';
    put32(2, size); put32(10, off); put32(14, 40); put32(18, w); put32(22, h);
    hdr[26] = 1; hdr[28] = 8;
    fwrite(hdr, 1, 54, f);
    for (int i = 0; i < 256; i++) { uint8_t q[4] = {(uint8_t)i, (uint8_t)i, (uint8_t)i, 0}; fwrite(q, 1, 4, f); }
    std::vector<uint8_t> row(rowbytes, 0);
    for (int y = h - 1; y >= 0; y--) { memcpy(row.data(), img + (size_t)y * w, w); fwrite(row.data(), 1, rowbytes, f); }
    fclose(f);
    return true;
}

// first readable of the given names below the data root
bool load_frame(const std::vector<std::string> &names, std::vector<uint8_t> &out)
{
    for (const auto &n : names)
        if (read_bmp_gray(data_root() + "/" + n, out)) return true;
    return fail(SL3D_E_INVALID_ARG, "cannot read " + data_root() + "/" + names[0] + " (8/24-bit BMP of " + std::to_string(W) + "x" + std::to_string(H) + ")");
}

// the numbers inside <data>...</data> of an OpenCV XML matrix (cvReadByName of 7/triangulation.cpp:152-168,1069-1083)
bool read_xml_matrix(const std::string &rel, int count, double *out)
{
    const std::string path = data_root() + "/" + rel;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(SL3D_E_INVALID_ARG, "cannot open " + path);
    std::string s;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    const size_t a = s.find("<data>"), b = s.find("</data>");
    if (a == std::string::npos || b == std::string::npos) return fail(SL3D_E_INVALID_ARG, "no <data> in " + path);
    const char *p = s.c_str() + a + 6;
    const char *end = s.c_str() + b;
    for (int i = 0; i < count; i++) {
        char *q = nullptr;
        out[i] = strtod(p, &q);
        if (q == p || q > end) return fail(SL3D_E_INVALID_ARG, "too few numbers in " + path);
        p = q;
    }
    return true;
}

void drop_ctx()
{
    if (g.group) sl3d_group_destroy(g.group);
    else if (g.ctx) sl3d_destroy(g.ctx);
    g.group = nullptr;
    g.ctx = nullptr;
    g.parts.clear();
}

// every part in turn; stops at the first failure (reported with that part's error text)
template <typename Fn>
bool each_part(const char *what, Fn fn)
{
    for (const Part &p : g.parts) {
        const int rc = fn(p);
        if (rc != SL3D_OK) return fail(rc, std::string(what) + ": " + sl3d_strerror(rc) + ": " + sl3d_last_error(p.ctx));
    }
    return true;
}

bool ensure_ctx()
{
    const bool same = g.ctx && g.F == number_of_patterns_fringe && g.Nv == number_of_patterns_binary_vertical &&
                      g.Nh == number_of_patterns_binary_horizontal && g.fwv == fringe_width_pixels_vertical &&
                      g.fwh == fringe_width_pixels_horizontal && g.ncv == number_of_codes_vertical && g.nch == number_of_codes_horizontal;
    if (same) return true;
    drop_ctx();
    sl3d_config c;
    memset(&c, 0, sizeof c);
    c.width = W; c.height = H; c.proj_width = Projector_imagewidth; c.proj_height = Projector_imageheight;
    c.n_fringe = g.F = number_of_patterns_fringe;
    c.n_gray_v = g.Nv = number_of_patterns_binary_vertical;
    c.n_gray_h = g.Nh = number_of_patterns_binary_horizontal;
    c.fringe_width_v = g.fwv = fringe_width_pixels_vertical;
    c.fringe_width_h = g.fwh = fringe_width_pixels_horizontal;
    c.n_codes_v = g.ncv = number_of_codes_vertical;
    c.n_codes_h = g.nch = number_of_codes_horizontal;
    c.max_views = 1;
    c.device = getenv("SL3D_DEVICE") ? atoi(getenv("SL3D_DEVICE")) : 0;
    c.flags = SL3D_FLAG_KEEP_STAGES;
    std::vector<int> devs;
    if (const char *e = getenv("SL3D_DEVICES")) {  // "0,1,2,3": one row stripe per listed device (a device may repeat)
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const long d = strtol(q, &end, 10);
            if (end == q) break;
            devs.push_back((int)d);
            q = *end == ',' ? end + 1 : end;
        }
    }
    if (devs.size() > 1) {
        const int rc = sl3d_group_create(&c, devs.data(), (int)devs.size(), &g.group);
        if (rc != SL3D_OK) return fail(rc, std::string("sl3d_group_create: ") + sl3d_strerror(rc) + ": " + sl3d_group_last_error(nullptr));
        for (int i = 0; i < sl3d_group_size(g.group); i++) {
            Part p{nullptr, 0, 0};
            sl3d_group_stripe(g.group, i, &p.row0, &p.rows, nullptr, &p.ctx);
            g.parts.push_back(p);
        }
        g.ctx = g.parts[0].ctx;
        return true;
    }
    if (devs.size() == 1) c.device = devs[0];
    const int rc = sl3d_create(&c, &g.ctx);
    if (rc != SL3D_OK) return fail(rc, std::string("sl3d_create: ") + sl3d_strerror(rc) + ": " + sl3d_last_error(nullptr));
    g.parts.push_back(Part{g.ctx, 0, H});
    return true;
}

template <typename T, typename U>
void to_col_row(const std::vector<T> &rowmajor, U (*dst)[Camera_imageheight])
{
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) dst[c][r] = (U)rowmajor[(size_t)r * W + c];
}

const char *axis_dir(int pattern_type) { return pattern_type == 0 ? "Vertical" : "Horizontal"; }

// the frames of one axis to every part: a part takes its own rows of every plane (a contiguous byte range)
bool upload_axis(const std::vector<std::vector<uint8_t>> &img, int pattern_type)
{
    return each_part("sl3d_set_frames", [&](const Part &q) {
        std::vector<const uint8_t *> planes;
        for (auto &v : img) planes.push_back(v.data() + (size_t)q.row0 * W);
        return sl3d_set_frames(q.ctx, 0, pattern_type, planes.data(), (int)planes.size(), W);
    });
}

}  // namespace

extern "C" void sl3d_shim_set_data_root(const char *dir)
{
    g.root = dir ? dir : "";
    g.root_set = dir != nullptr;
}
extern "C" void sl3d_shim_write_debug_images(int enable) { g.write_debug = enable != 0; }
extern "C" int sl3d_shim_last_status(void) { return g.status; }
extern "C" const char *sl3d_shim_last_error(void) { return g.err.c_str(); }
extern "C" void sl3d_shim_reset(void) { drop_ctx(); }

// ---- stage 1: generate_pattern() ----------------------------------------------------------------------
// 1/pattern_generator.cpp:513-544.  The reference's allocate_memory() asks for the number of fringe patterns and the two
// fringe widths with scanf (:204-222); the shim takes them from the globals number_of_patterns_fringe and
// fringe_width_pixels_{vertical,horizontal}, derives number_of_codes_* / number_of_patterns_binary_* exactly as
// :224-229 does (and stores them in the globals, as the reference does), generates every pattern on the device and
// saves the same files save_pattern_images() writes (:414-470) below <data root>/Generated_patterns/.
void generate_pattern()
{
    g.status = SL3D_OK;
    if (number_of_patterns_fringe < 3 || number_of_patterns_fringe > 5) { fail(SL3D_E_INVALID_ARG, "generate_pattern: 3, 4 or 5 fringe patterns"); return; }
    if (!ok(sl3d_pattern_counts(Projector_imagewidth, fringe_width_pixels_vertical, &number_of_codes_vertical, &number_of_patterns_binary_vertical), "sl3d_pattern_counts")) return;
    if (!ok(sl3d_pattern_counts(Projector_imageheight, fringe_width_pixels_horizontal, &number_of_codes_horizontal, &number_of_patterns_binary_horizontal), "sl3d_pattern_counts")) return;
    if (!ensure_ctx()) return;
    const int PWs = Projector_imagewidth, PHs = Projector_imageheight;
    std::vector<uint8_t> img((size_t)PWs * PHs);
    const std::string root = data_root() + "/Generated_patterns";
    auto emit = [&](int kind, int axis, int index, const std::string &rel) {
        if (!ok(sl3d_generate_pattern(g.ctx, kind, axis, index, img.data(), (size_t)PWs, nullptr, nullptr), "sl3d_generate_pattern")) return false;
        const std::string path = root + "/" + rel;
        const std::string dir = path.substr(0, path.rfind('/'));
        for (size_t i = 1; i <= dir.size(); i++)
            if (i == dir.size() || dir[i] == '/') mkdir(dir.substr(0, i).c_str(), 0777);
        if (!write_bmp_gray(path, img.data(), PWs, PHs)) return fail(SL3D_E_INVALID_ARG, "cannot write " + path);
        return true;
    };
    for (int axis = 0; axis < 2; axis++) {
        const std::string ax = axis_dir(axis);
        const int N = axis == 0 ? number_of_patterns_binary_vertical : number_of_patterns_binary_horizontal;
        for (int i = 0; i < number_of_patterns_fringe; i++)  // :419-431
            if (!emit(SL3D_PATTERN_FRINGE, axis, i, "Fringe_patterns/" + ax + "/Pattern_" + std::to_string(i) + ".bmp")) return;
        for (int j = 0; j < N + 1; j++) {  // :433-465: one image more than there are bit planes
            if (!emit(SL3D_PATTERN_BINARY, axis, j, "Coded_patterns/Binary_coded/" + ax + "/Pattern_" + std::to_string(j) + ".bmp")) return;
            if (!emit(SL3D_PATTERN_GRAY, axis, j, "Coded_patterns/Gray_coded/" + ax + "/Pattern_" + std::to_string(j) + ".bmp")) return;
            if (!emit(SL3D_PATTERN_INVERSE_GRAY, axis, j, "Coded_patterns/Gray_coded/" + ax + "/inverse_Pattern_" + std::to_string(j) + ".bmp")) return;
        }
    }
}

// ---- stage 3 ----------------------------------------------------------------------------------------
void compute_wrapped_phase(int pattern_type)
{
    g.status = SL3D_OK;
    if (pattern_type != 0 && pattern_type != 1) return;
    if (!ensure_ctx()) return;
    // the reference allocates these with new[] on every call and never frees them (3/wrapped_phase.cpp:410-424)
    int (*&vm)[Camera_imageheight] = pattern_type == 0 ? valid_map_vertical : valid_map_horizontal;
    float (*&wp)[Camera_imageheight] = pattern_type == 0 ? wrapped_phi_vertical : wrapped_phi_horizontal;
    if (!vm) vm = new int[Camera_imagewidth][Camera_imageheight];
    if (!wp) wp = new float[Camera_imagewidth][Camera_imageheight];

    // selection mask from image_scissor (m_tech_project_console.cpp:146-238); without one: 1 inside the border
    std::vector<uint8_t> mask((size_t)W * H, 0);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            mask[(size_t)r * W + c] = selected_region ? (selected_region[c][r] == 1) : (r > 0 && r < H - 1 && c > 0 && c < W - 1);
    if (!each_part("sl3d_set_mask", [&](const Part &p) { return sl3d_set_mask(p.ctx, 0, mask.data(), W); })) return;

    // read_image: F fringe frames (3/wrapped_phase.cpp:29-58); the Gray/inverse planes are supplied by stage 4
    const int F = number_of_patterns_fringe;
    const int N = pattern_type == 0 ? number_of_patterns_binary_vertical : number_of_patterns_binary_horizontal;
    std::vector<std::vector<uint8_t>> img(F + 2 * N, std::vector<uint8_t>((size_t)W * H, 0));
    char name[256], alt[256];
    for (int i = 0; i < F; i++) {
        snprintf(name, sizeof name, "Captured_patterns/Fringe_patterns/%s/Undistorted/Captured_image_%d.bmp", axis_dir(pattern_type), i);
        snprintf(alt, sizeof alt, "Captured_patterns/Fringe_patterns/%s/Undistorted/Gray_captured_image_%d.bmp", axis_dir(pattern_type), i);
        if (!load_frame({name, alt}, img[i])) return;
    }
    // Gray planes may already be on disk: load them now so one upload covers the axis (stage 4 reloads them anyway)
    for (int i = 0; i < N; i++) {
        snprintf(name, sizeof name, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/Captured_image_%d.bmp", axis_dir(pattern_type), i);
        snprintf(alt, sizeof alt, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/Gray_captured_image_%d.bmp", axis_dir(pattern_type), i);
        read_bmp_gray(data_root() + "/" + name, img[F + i]) || read_bmp_gray(data_root() + "/" + alt, img[F + i]);
        snprintf(name, sizeof name, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/inverse_Captured_image_%d.bmp", axis_dir(pattern_type), i);
        snprintf(alt, sizeof alt, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/inverse_Gray_captured_image_%d.bmp", axis_dir(pattern_type), i);
        read_bmp_gray(data_root() + "/" + name, img[F + N + i]) || read_bmp_gray(data_root() + "/" + alt, img[F + N + i]);
    }
    if (!upload_axis(img, pattern_type)) return;
    if (!each_part("sl3d_compute_wrapped_phase", [&](const Part &p) { return sl3d_compute_wrapped_phase(p.ctx, 0, pattern_type); })) return;

    std::vector<uint8_t> v((size_t)W * H);
    std::vector<float> p((size_t)W * H);
    if (!each_part("sl3d_get_valid_map", [&](const Part &q) { return sl3d_get_valid_map(q.ctx, 0, pattern_type, v.data() + (size_t)q.row0 * W, W); })) return;
    if (!each_part("sl3d_get_wrapped_phase", [&](const Part &q) { return sl3d_get_wrapped_phase(q.ctx, 0, pattern_type, p.data() + (size_t)q.row0 * W, W); })) return;
    to_col_row(v, vm);
    to_col_row(p, wp);
    if (g.write_debug) {  // save_wrapped_image :346
        std::vector<uint8_t> d((size_t)W * H);
        if (each_part("sl3d_get_debug_image", [&](const Part &q) { return sl3d_get_debug_image(q.ctx, 0, 3, pattern_type, d.data() + (size_t)q.row0 * W, W); }))
            write_bmp_gray(data_root() + "/Wrapped_phase_images/" + axis_dir(pattern_type) + "/Wrapped_phase_image.bmp", d.data());
    }
}

// ---- stage 4 ----------------------------------------------------------------------------------------
void unwrap_phase(int pattern_type)
{
    g.status = SL3D_OK;
    if (pattern_type != 0 && pattern_type != 1) return;
    if (!g.ctx) { fail(SL3D_E_STATE, "unwrap_phase before compute_wrapped_phase"); return; }
    int (*&code)[Camera_imageheight] = pattern_type == 0 ? code_vertical : code_horizontal;
    float (*&uw)[Camera_imageheight] = pattern_type == 0 ? unwrapped_phi_vertical : unwrapped_phi_horizontal;
    float (*&wp)[Camera_imageheight] = pattern_type == 0 ? wrapped_phi_vertical : wrapped_phi_horizontal;
    if (!code) code = new int[Camera_imagewidth][Camera_imageheight];     // 4/phase_unwrap.cpp:373-376
    if (!uw) uw = new float[Camera_imagewidth][Camera_imageheight];       // :282 / :300

    // read_captured_images :51-131: N Gray + N inverse-Gray frames (frame index N is loaded there but never used)
    const int F = number_of_patterns_fringe;
    const int N = pattern_type == 0 ? number_of_patterns_binary_vertical : number_of_patterns_binary_horizontal;
    std::vector<std::vector<uint8_t>> img(F + 2 * N, std::vector<uint8_t>((size_t)W * H, 0));
    char name[256], alt[256];
    for (int i = 0; i < F; i++) {  // the axis is uploaded as a whole: fringe frames again
        snprintf(name, sizeof name, "Captured_patterns/Fringe_patterns/%s/Undistorted/Captured_image_%d.bmp", axis_dir(pattern_type), i);
        snprintf(alt, sizeof alt, "Captured_patterns/Fringe_patterns/%s/Undistorted/Gray_captured_image_%d.bmp", axis_dir(pattern_type), i);
        if (!load_frame({name, alt}, img[i])) return;
    }
    for (int i = 0; i < N; i++) {
        snprintf(name, sizeof name, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/Captured_image_%d.bmp", axis_dir(pattern_type), i);
        snprintf(alt, sizeof alt, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/Gray_captured_image_%d.bmp", axis_dir(pattern_type), i);
        if (!load_frame({name, alt}, img[F + i])) return;
        snprintf(name, sizeof name, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/inverse_Captured_image_%d.bmp", axis_dir(pattern_type), i);
        snprintf(alt, sizeof alt, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/inverse_Gray_captured_image_%d.bmp", axis_dir(pattern_type), i);
        if (!load_frame({name, alt}, img[F + N + i])) return;
    }
    if (!upload_axis(img, pattern_type)) return;
    if (!each_part("sl3d_unwrap_phase", [&](const Part &q) { return sl3d_unwrap_phase(q.ctx, 0, pattern_type); })) return;

    std::vector<int32_t> cd((size_t)W * H);
    std::vector<float> u((size_t)W * H), p((size_t)W * H);
    if (!each_part("sl3d_get_code", [&](const Part &q) { return sl3d_get_code(q.ctx, 0, pattern_type, cd.data() + (size_t)q.row0 * W, W); })) return;
    if (!each_part("sl3d_get_unwrapped_phase", [&](const Part &q) { return sl3d_get_unwrapped_phase(q.ctx, 0, pattern_type, u.data() + (size_t)q.row0 * W, W); })) return;
    if (!each_part("sl3d_get_wrapped_phase", [&](const Part &q) { return sl3d_get_wrapped_phase(q.ctx, 0, pattern_type, p.data() + (size_t)q.row0 * W, W); })) return;
    to_col_row(cd, code);
    to_col_row(u, uw);
    if (wp) to_col_row(p, wp);  // stage 4 shifts wrapped_phi in place by +Pi (:290, :308)
    if (g.write_debug) {       // save_unwrap_phase_image :321-364
        std::vector<uint8_t> d((size_t)W * H);
        if (each_part("sl3d_get_debug_image", [&](const Part &q) { return sl3d_get_debug_image(q.ctx, 0, 4, pattern_type, d.data() + (size_t)q.row0 * W, W); }))
            write_bmp_gray(data_root() + (pattern_type == 0 ? "/Unwrapped_phase_images/Gray_coded/Vertical/Unwrapped_phase_vertical.bmp"
                                                            : "/Unwrapped_phase_images/Gray_coded/Horizontal/Unwrapped_phase_horizontal.bmp"),
                           d.data());
    }
}

// ---- stage 5 ----------------------------------------------------------------------------------------
void compute_c_p_map()
{
    g.status = SL3D_OK;
    if (!g.ctx) { fail(SL3D_E_STATE, "compute_c_p_map before the phase stages"); return; }
    if (!valid_map) valid_map = new int[Camera_imagewidth][Camera_imageheight];  // 5/compute_correspondance.cpp:635
    if (!c_p_map) c_p_map = new long int[total_camera_pixels][2];                 // :640
    if (!each_part("sl3d_compute_c_p_map", [&](const Part &q) { return sl3d_compute_c_p_map(q.ctx, 0); })) return;
    std::vector<uint8_t> v((size_t)W * H);
    if (!each_part("sl3d_get_valid_map", [&](const Part &q) { return sl3d_get_valid_map(q.ctx, 0, SL3D_VALID_MERGED, v.data() + (size_t)q.row0 * W, W); })) return;
    to_col_row(v, valid_map);
    static_assert(sizeof(long int) == sizeof(int64_t), "c_p_map is long[ ][2] on LP64");
    each_part("sl3d_get_c_p_map", [&](const Part &q) { return sl3d_get_c_p_map(q.ctx, 0, (int64_t *)c_p_map + 2 * (size_t)q.row0 * W); });
}

// ---- stage 7 ----------------------------------------------------------------------------------------
void triangulate()
{
    g.status = SL3D_OK;
    if (!g.ctx) { fail(SL3D_E_STATE, "triangulate before compute_c_p_map"); return; }
    double Kc[9], dc[5], rc[3], tc[3], Kp[9], dp[5], rp[3], tp[3];
    if (!read_xml_matrix("Camera_calibration/Matrices/cam_intrinsic_mat.xml", 9, Kc) ||            // 7/triangulation.cpp:152
        !read_xml_matrix("Camera_calibration/Matrices/cam_distortion_vect.xml", 5, dc) ||          // :157
        !read_xml_matrix("Projector_calibration/Matrices/proj_intrinsic_mat.xml", 9, Kp) ||        // :162
        !read_xml_matrix("Projector_calibration/Matrices/proj_distortion_vect.xml", 5, dp) ||      // :167
        !read_xml_matrix("Triangulation/Camera_extrinsic_parametrs/world_to_cam_rot_vect.xml", 3, rc) ||      // :1069
        !read_xml_matrix("Triangulation/Camera_extrinsic_parametrs/world_to_cam_trans_vect.xml", 3, tc) ||    // :1074
        !read_xml_matrix("Triangulation/Projector_extrinsic_parametrs/world_to_proj_rot_vect.xml", 3, rp) ||  // :1077
        !read_xml_matrix("Triangulation/Projector_extrinsic_parametrs/world_to_proj_trans_vect.xml", 3, tp))  // :1082
        return;
    if (!each_part("sl3d_set_calibration", [&](const Part &q) { return sl3d_set_calibration(q.ctx, Kc, dc, rc, tc, Kp, dp, rp, tp); })) return;
    if (!intersection_points) intersection_points = new double[Camera_imagewidth][Camera_imageheight][3];  // :1513
    if (!each_part("sl3d_triangulate", [&](const Part &q) { return sl3d_triangulate(q.ctx, 0); })) return;
    std::vector<double> pts((size_t)W * H * 3);
    if (!each_part("sl3d_get_intersection_points", [&](const Part &q) { return sl3d_get_intersection_points(q.ctx, 0, pts.data() + 3 * (size_t)q.row0 * W); })) return;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) memcpy(intersection_points[c][r], &pts[3 * ((size_t)r * W + c)], 3 * sizeof(double));
}

// ---- stage 8: save_point_cloud() ------------------------------------------------------------------------
// 8/save_point_cloud.cpp:19-217: the valid pixels in row-major scan order (:85-104) as float xyz with the r,g,b of
// Point_cloud/texture.bmp (:46-52,70-72), saved as Point_cloud/point_cloud_<i>.pcd (ASCII) and .ply.  Compaction and
// colour gather run on the device on the result of the last triangulate().  The reference writes the two files with
// PCL 1.6 (pcl::io::savePCDFileASCII / savePLYFile); PCL is not available here, so the files are standard PCD v0.7 ASCII
// (fields x y z rgb, rgb as the packed 0x00RRGGBB integer) and PLY ASCII (x y z red green blue) that PCL, MeshLab and
// CloudCompare read -- the same points, colours and order, not PCL's exact text (unpinned).
void save_point_cloud(unsigned cloud_index)
{
    g.status = SL3D_OK;
    if (!g.ctx) { fail(SL3D_E_STATE, "save_point_cloud before triangulate"); return; }
    std::vector<uint8_t> tex;
    if (!read_bmp_bgr(data_root() + "/Point_cloud/texture.bmp", tex)) {
        fail(SL3D_E_INVALID_ARG, "cannot read " + data_root() + "/Point_cloud/texture.bmp (8/24-bit BMP of the camera size)");
        return;
    }
    if (!each_part("sl3d_set_texture", [&](const Part &q) { return sl3d_set_texture(q.ctx, 0, tex.data() + 3 * (size_t)q.row0 * W, (size_t)W * 3); })) return;
    // the parts' clouds one after the other: stripe order = row order = the scan order of :85-104
    int64_t n = 0;
    std::vector<int64_t> cnt(g.parts.size(), 0);
    size_t k = 0;
    if (!each_part("sl3d_get_cloud_rgb", [&](const Part &q) { const int rc = sl3d_get_cloud_rgb(q.ctx, 0, nullptr, nullptr, 0, &cnt[k]); n += cnt[k++]; return rc; })) return;
    std::vector<float> xyz((size_t)n * 3);
    std::vector<uint8_t> rgb((size_t)n * 3);
    int64_t off = 0;
    k = 0;
    if (!each_part("sl3d_get_cloud_rgb", [&](const Part &q) {
            int64_t m = 0;
            const int rc = sl3d_get_cloud_rgb(q.ctx, 0, xyz.data() + 3 * off, rgb.data() + 3 * off, cnt[k], &m);
            off += cnt[k++];
            return rc;
        }))
        return;
    mkdir((data_root() + "/Point_cloud").c_str(), 0777);
    const std::string base = data_root() + "/Point_cloud/point_cloud_" + std::to_string(cloud_index);
    FILE *f = fopen((base + ".pcd").c_str(), "w");
    if (!f) { fail(SL3D_E_INVALID_ARG, "cannot write " + base + ".pcd"); return; }
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
               "WIDTH %lld\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %lld\nDATA ascii\n", (long long)n, (long long)n);
    for (int64_t i = 0; i < n; i++)
        fprintf(f, "%.9g %.9g %.9g %u\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2],
                ((unsigned)rgb[3 * i] << 16) | ((unsigned)rgb[3 * i + 1] << 8) | (unsigned)rgb[3 * i + 2]);
    fclose(f);
    f = fopen((base + ".ply").c_str(), "w");
    if (!f) { fail(SL3D_E_INVALID_ARG, "cannot write " + base + ".ply"); return; }
    fprintf(f, "ply\nformat ascii 1.0\ncomment generated by sl3d (3dscan_amd)\nelement vertex %lld\nproperty float x\nproperty float y\n"
               "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", (long long)n);
    for (int64_t i = 0; i < n; i++)
        fprintf(f, "%.9g %.9g %.9g %u %u %u\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    fclose(f);
    fprintf(stderr, "Saved %lld data points to %s.pcd / .ply\n", (long long)n, base.c_str());
}


// ---- stage 9: register_point_clouds() -------------------------------------------------------------------
// 9/register_point_clouds.cpp:23-155: Point_cloud/point_cloud_<i>.ply, i = 0..n-1, each rotated about the Y axis through
// (tx,ty,tz) by theta_i (theta_0 = 0, theta_{i+1} = theta_i + rot_step in float, degrees with Pi = 22/7), colours kept,
// concatenated into Point_cloud/registered_point_cloud.ply.  Reads the ASCII PLY files save_point_cloud() writes (vertex
// properties x y z [red green blue] in any order, other properties ignored); the rotation runs on the device.
namespace {
struct PlyCloud {
    std::vector<float> xyz;
    std::vector<uint8_t> rgb;
};
bool read_ply_ascii(const std::string &path, PlyCloud &c)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    char line[512];
    long nv = -1;
    std::vector<std::string> props;
    bool ascii = false, in_vertex = false, header_ok = false;
    while (fgets(line, sizeof line, f)) {
        char a[64] = "", b[64] = "", d[64] = "";
        const int k = sscanf(line, "%63s %63s %63s", a, b, d);
        if (k >= 1 && !strcmp(a, "end_header")) { header_ok = true; break; }
        if (k >= 2 && !strcmp(a, "format")) ascii = !strcmp(b, "ascii");
        if (k >= 3 && !strcmp(a, "element")) { in_vertex = !strcmp(b, "vertex"); if (in_vertex) nv = atol(d); }
        if (k >= 3 && !strcmp(a, "property") && in_vertex && strcmp(b, "list")) props.push_back(d);
    }
    if (!header_ok || !ascii || nv < 0) { fclose(f); return false; }
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1;
    for (int i = 0; i < (int)props.size(); i++) {
        if (props[i] == "x") ix = i; else if (props[i] == "y") iy = i; else if (props[i] == "z") iz = i;
        else if (props[i] == "red" || props[i] == "r") ir = i; else if (props[i] == "green" || props[i] == "g") ig = i;
        else if (props[i] == "blue" || props[i] == "b") ib = i;
    }
    if (ix < 0 || iy < 0 || iz < 0) { fclose(f); return false; }
    c.xyz.resize((size_t)nv * 3);
    c.rgb.assign((size_t)nv * 3, 0);
    std::vector<double> v(props.size());
    for (long p = 0; p < nv; p++) {
        for (size_t i = 0; i < props.size(); i++)
            if (fscanf(f, "%lf", &v[i]) != 1) { fclose(f); return false; }
        c.xyz[3 * p] = (float)v[ix]; c.xyz[3 * p + 1] = (float)v[iy]; c.xyz[3 * p + 2] = (float)v[iz];
        if (ir >= 0 && ig >= 0 && ib >= 0) { c.rgb[3 * p] = (uint8_t)v[ir]; c.rgb[3 * p + 1] = (uint8_t)v[ig]; c.rgb[3 * p + 2] = (uint8_t)v[ib]; }
    }
    fclose(f);
    return true;
}
}  // namespace

void register_point_clouds(unsigned num_point_clouds, float tx, float ty, float tz, float rot_step)
{
    g.status = SL3D_OK;
    if (!ensure_ctx()) return;
    std::vector<float> all_xyz;
    std::vector<uint8_t> all_rgb;
    float theta = 0.0;  // :79
    for (unsigned i = 0; i < num_point_clouds; i++) {
        PlyCloud c;
        const std::string path = data_root() + "/Point_cloud/point_cloud_" + std::to_string(i) + ".ply";
        if (!read_ply_ascii(path, c)) { fail(SL3D_E_INVALID_ARG, "cannot read " + path + " (ASCII PLY with x y z vertex properties)"); return; }
        const int64_t n = (int64_t)c.xyz.size() / 3;
        std::vector<float> out((size_t)n * 3);
        if (!ok(sl3d_transform_cloud(g.ctx, c.xyz.data(), n, theta, tx, ty, tz, out.data()), "sl3d_transform_cloud")) return;
        all_xyz.insert(all_xyz.end(), out.begin(), out.end());
        all_rgb.insert(all_rgb.end(), c.rgb.begin(), c.rgb.end());
        theta += rot_step;  // :145
    }
    const std::string outp = data_root() + "/Point_cloud/registered_point_cloud.ply";
    FILE *f = fopen(outp.c_str(), "w");
    if (!f) { fail(SL3D_E_INVALID_ARG, "cannot write " + outp); return; }
    const long long n = (long long)all_xyz.size() / 3;
    fprintf(f, "ply\nformat ascii 1.0\ncomment generated by sl3d (3dscan_amd)\nelement vertex %lld\nproperty float x\nproperty float y\n"
               "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", n);
    for (long long i = 0; i < n; i++)
        fprintf(f, "%.9g %.9g %.9g %u %u %u\n", all_xyz[3 * i], all_xyz[3 * i + 1], all_xyz[3 * i + 2], all_rgb[3 * i], all_rgb[3 * i + 1], all_rgb[3 * i + 2]);
    fclose(f);
}
