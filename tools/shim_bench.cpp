// Times the Level-1 drop-in path -- the reference's own six stage calls + save_point_cloud() through the shim
// (include/sl3d_shim.h) -- at the reference's compile-time size (1600x1200 camera, 1280x720 projector, N_v = 6, N_h = 5, fringe
// width 32: PROJECT_GLOBAL/global_cv.h:49-53, common_variables.h:6-10,23-24), as main() runs it once per scan
// (m_tech_project_console.cpp:366-395).  Not a product path and not `value`: a side figure of bench.py (tools/shim_timing.py
// builds and runs it) that says what the host side around the kernels costs.
//
//   shim_bench <cal.bin: 40 doubles Kc dc rc tc Kp dp rp tp> <scratch dir> [scans]
//
// Inputs: one synthetic capture generated on the device (sl3d_synth_view on a helper context), handed to the shim
//   files   : as the reference's BMP / XML files (8-bit palettised BMPs as cvSaveImage writes them; decoded by the shim)
//   memory  : the same planes in pageable host memory (an IplImage's imageData) through sl3d_shim_provide_image
//   pinned  : the same in pinned host memory (sl3d_host_alloc)
// each with the globals produced in the reference's [col][row] layout on the DEVICE (default) and with the pre-round-3 route
// (row-major download + transposes by the host: sl3d_shim_host_transpose(1)).  Prints one JSON object: medians in ms.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "sl3d.h"
#include "sl3d_shim.h"

namespace {
constexpr int W = Camera_imagewidth, H = Camera_imageheight, PW = Projector_imagewidth, PH = Projector_imageheight;
constexpr int NV = 6, NH = 5, FW = 32, F = 3;

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void mkdirs(const std::string &path)
{
    for (size_t i = 1; i <= path.size(); i++)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}

void write_bmp8(const std::string &path, const uint8_t *img)
{
    mkdirs(path.substr(0, path.rfind('/')));
    FILE *f = fopen(path.c_str(), "wb");
    const uint32_t rowbytes = ((uint32_t)W + 3) & ~3u, off = 54 + 1024, size = off + rowbytes * H;
    uint8_t hdr[54] = {0};
    auto put32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = v >> 24; };
    hdr[0] = 'B'; hdr[1] = 'M';
    put32(2, size); put32(10, off); put32(14, 40); put32(18, W); put32(22, H);
    hdr[26] = 1; hdr[28] = 8;
    fwrite(hdr, 1, 54, f);
    for (int i = 0; i < 256; i++) { uint8_t q[4] = {(uint8_t)i, (uint8_t)i, (uint8_t)i, 0}; fwrite(q, 1, 4, f); }
    std::vector<uint8_t> row(rowbytes, 0);
    for (int y = H - 1; y >= 0; y--) { memcpy(row.data(), img + (size_t)y * W, W); fwrite(row.data(), 1, rowbytes, f); }
    fclose(f);
}

void write_bmp24(const std::string &path, const uint8_t *bgr)
{
    mkdirs(path.substr(0, path.rfind('/')));
    FILE *f = fopen(path.c_str(), "wb");
    const uint32_t rowbytes = ((uint32_t)W * 3 + 3) & ~3u, off = 54, size = off + rowbytes * H;
    uint8_t hdr[54] = {0};
    auto put32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = v >> 24; };
    hdr[0] = 'B'; hdr[1] = 'M';
    put32(2, size); put32(10, off); put32(14, 40); put32(18, W); put32(22, H);
    hdr[26] = 1; hdr[28] = 24;
    fwrite(hdr, 1, 54, f);
    std::vector<uint8_t> row(rowbytes, 0);
    for (int y = H - 1; y >= 0; y--) { memcpy(row.data(), bgr + (size_t)y * W * 3, (size_t)W * 3); fwrite(row.data(), 1, rowbytes, f); }
    fclose(f);
}

void write_xml(const std::string &path, const char *name, int rows, int cols, const double *v)
{
    mkdirs(path.substr(0, path.rfind('/')));
    FILE *f = fopen(path.c_str(), "w");
    fprintf(f, "<?xml version=\"1.0\"?>\n<opencv_storage>\n<%s type_id=\"opencv-matrix\">\n  <rows>%d</rows>\n  <cols>%d</cols>\n  <dt>d</dt>\n  <data>\n   ", name, rows, cols);
    for (int i = 0; i < rows * cols; i++) fprintf(f, " %.17e", v[i]);
    fprintf(f, "</data></%s>\n</opencv_storage>\n", name);
    fclose(f);
}

std::string frame_name(int axis, int kind, int i)  // kind 0 fringe, 1 gray, 2 inverse
{
    char b[256];
    const char *ax = axis == 0 ? "Vertical" : "Horizontal";
    if (kind == 0) snprintf(b, sizeof b, "Captured_patterns/Fringe_patterns/%s/Undistorted/Captured_image_%d.bmp", ax, i);
    else snprintf(b, sizeof b, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/%sCaptured_image_%d.bmp", ax, kind == 2 ? "inverse_" : "", i);
    return b;
}

const char *kMats[8] = {"Camera_calibration/Matrices/cam_intrinsic_mat.xml", "Camera_calibration/Matrices/cam_distortion_vect.xml",
                        "Triangulation/Camera_extrinsic_parametrs/world_to_cam_rot_vect.xml", "Triangulation/Camera_extrinsic_parametrs/world_to_cam_trans_vect.xml",
                        "Projector_calibration/Matrices/proj_intrinsic_mat.xml", "Projector_calibration/Matrices/proj_distortion_vect.xml",
                        "Triangulation/Projector_extrinsic_parametrs/world_to_proj_rot_vect.xml", "Triangulation/Projector_extrinsic_parametrs/world_to_proj_trans_vect.xml"};
const char *kMatNames[8] = {"cam_intrinsic_mat", "cam_distortion_vect", "world_to_cam_rot_vect", "world_to_cam_trans_vect",
                            "proj_intrinsic_mat", "proj_distortion_vect", "world_to_proj_rot_vect", "world_to_proj_trans_vect"};
const int kMatRows[8] = {3, 5, 3, 3, 3, 5, 3, 3}, kMatCols[8] = {3, 1, 1, 1, 3, 1, 1, 1};

double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    double cal[40];
    {
        FILE *f = fopen(argv[1], "rb");
        if (!f || fread(cal, sizeof(double), 40, f) != 40) return 3;
        fclose(f);
    }
    const std::string root = argv[2];
    const int scans = argc > 3 ? atoi(argv[3]) : 5;
    const size_t px = (size_t)W * H;
    const int ppa[2] = {F + 2 * NV, F + 2 * NH};
    // ---- one synthetic capture, generated on the device --------------------------------------------------------------
    std::vector<uint8_t> pageable((size_t)(ppa[0] + ppa[1]) * px);
    uint8_t *pinned = (uint8_t *)sl3d_host_alloc(pageable.size());
    if (!pinned) return 4;
    {
        sl3d_config c;
        memset(&c, 0, sizeof c);
        c.width = W; c.height = H; c.proj_width = PW; c.proj_height = PH; c.n_fringe = F; c.n_gray_v = NV; c.n_gray_h = NH;
        c.fringe_width_v = c.fringe_width_h = FW; c.max_views = 1;
        sl3d_ctx *h = nullptr;
        if (sl3d_create(&c, &h) != SL3D_OK) { fprintf(stderr, "%s\n", sl3d_last_error(nullptr)); return 5; }
        const double plane[3] = {0.0, 0.05, 0.05};
        int rc = sl3d_set_calibration(h, cal, cal + 9, cal + 14, cal + 17, cal + 20, cal + 29, cal + 34, cal + 37);
        if (!rc) rc = sl3d_synth_view(h, 0, plane, 0x3D5CA11ull, 0, 2, 0.8f, 10.0f);
        for (int a = 0, o = 0; a < 2 && !rc; o += ppa[a], a++) {
            std::vector<uint8_t *> pl((size_t)ppa[a]);
            for (int p = 0; p < ppa[a]; p++) pl[(size_t)p] = pageable.data() + (size_t)(o + p) * px;
            rc = sl3d_get_frames(h, 0, a, pl.data(), ppa[a], (size_t)W);
        }
        if (rc) { fprintf(stderr, "%s\n", sl3d_last_error(h)); return 6; }
        sl3d_destroy(h);
    }
    memcpy(pinned, pageable.data(), pageable.size());
    std::vector<uint8_t> texture(px * 3);
    for (size_t i = 0; i < texture.size(); i++) texture[i] = (uint8_t)((i * 2654435761u) >> 24);
    // the reference's files
    for (int a = 0, o = 0; a < 2; o += ppa[a], a++) {
        const int N = a == 0 ? NV : NH;
        for (int i = 0; i < F; i++) write_bmp8(root + "/" + frame_name(a, 0, i), pageable.data() + (size_t)(o + i) * px);
        for (int i = 0; i < N; i++) {
            write_bmp8(root + "/" + frame_name(a, 1, i), pageable.data() + (size_t)(o + F + i) * px);
            write_bmp8(root + "/" + frame_name(a, 2, i), pageable.data() + (size_t)(o + F + N + i) * px);
        }
    }
    write_bmp24(root + "/Point_cloud/texture.bmp", texture.data());
    for (int k = 0, o = 0; k < 8; o += kMatRows[k] * kMatCols[k], k++) write_xml(root + "/" + kMats[k], kMatNames[k], kMatRows[k], kMatCols[k], cal + o);

    sl3d_shim_set_data_root(root.c_str());
    number_of_patterns_fringe = F;
    number_of_patterns_binary_vertical = NV;
    number_of_patterns_binary_horizontal = NH;
    fringe_width_pixels_vertical = fringe_width_pixels_horizontal = FW;
    number_of_codes_vertical = (PW + FW - 1) / FW;
    number_of_codes_horizontal = (PH + FW - 1) / FW;
    // image_scissor's result as main() holds it: int [col][row], 1 inside the border
    selected_region = new int[Camera_imagewidth][Camera_imageheight];
    for (int c = 0; c < W; c++)
        for (int r = 0; r < H; r++) selected_region[c][r] = (c > 0 && c < W - 1 && r > 0 && r < H - 1) ? 1 : 0;

    auto provide = [&](const uint8_t *base) {
        for (int a = 0, o = 0; a < 2; o += ppa[a], a++) {
            const int N = a == 0 ? NV : NH;
            for (int i = 0; i < F; i++) sl3d_shim_provide_image(frame_name(a, 0, i).c_str(), base ? base + (size_t)(o + i) * px : nullptr, W, H, 1, W);
            for (int i = 0; i < N; i++) {
                sl3d_shim_provide_image(frame_name(a, 1, i).c_str(), base ? base + (size_t)(o + F + i) * px : nullptr, W, H, 1, W);
                sl3d_shim_provide_image(frame_name(a, 2, i).c_str(), base ? base + (size_t)(o + F + N + i) * px : nullptr, W, H, 1, W);
            }
        }
        sl3d_shim_provide_image("Point_cloud/texture.bmp", base ? texture.data() : nullptr, W, H, 3, (size_t)W * 3);
        for (int k = 0, o = 0; k < 8; o += kMatRows[k] * kMatCols[k], k++) sl3d_shim_provide_matrix(kMats[k], base ? cal + o : nullptr, kMatRows[k] * kMatCols[k]);
    };

    printf("{\"camera\": \"%dx%d\", \"projector\": \"%dx%d\", \"n_gray\": [%d, %d], \"scans\": %d, \"unit\": \"ms (median per scan)\"", W, H, PW, PH, NV, NH, scans);
    long long npoints = 0;
    const char *inputs[3] = {"files", "memory", "pinned"};
    for (int in = 0; in < 3; in++) {
        provide(in == 0 ? nullptr : in == 1 ? pageable.data() : pinned);
        // host: 0 = globals transposed on the device, 1 = by the host (both stage by stage, every global after every stage);
        // 2, 3 = DEFERRED (sl3d_shim_globals): one launch of the timed fused kernel inside triangulate(), then valid_map +
        // intersection_points (FINAL: what 8/save_point_cloud.cpp reads) or no global at all (NONE: the shim's own save_point_cloud()
        // follows); the triangulate lap of NONE includes the wait for the launch (sl3d_shim_materialize(0))
        for (int host = 0; host < 4; host++) {
            sl3d_shim_host_transpose(host == 1);
            sl3d_shim_globals(host == 2 ? (unsigned)SL3D_SHIM_G_FINAL : host == 3 ? (unsigned)SL3D_SHIM_G_NONE : (unsigned)SL3D_SHIM_G_ALL);
            std::vector<double> t[9];
            for (int s = 0; s < scans + 1; s++) {  // scan 0 warms up (context creation, first-touch of the globals)
                double t0 = now_ms(), t1;
                auto lap = [&](int k) { t1 = now_ms(); if (s) t[k].push_back(t1 - t0); t0 = t1; };
                compute_wrapped_phase(0); lap(0);
                compute_wrapped_phase(1); lap(1);
                unwrap_phase(0); lap(2);
                unwrap_phase(1); lap(3);
                compute_c_p_map(); lap(4);
                triangulate();
                if (host == 3) sl3d_shim_materialize(0);
                lap(5);
                if (sl3d_shim_last_status()) { fprintf(stderr, "%s\n", sl3d_shim_last_error()); return 10; }
            }
            // the cloud files are written from the device-resident result whatever the inputs were: timed in the first
            // configuration only, in three writes of their own AFTER the stage timings (round 3 wrote them between the scans: the
            // ~160 MB of fresh page-cache pages and freed heap of a save made the first stage call of the NEXT scan ~10 ms slower,
            // and the median of the five scans reported that artefact as the cost of compute_wrapped_phase(0))
            if (in == 0 && host == 0) {
                for (int w = 0; w < 3; w++) {
                    double t0 = now_ms();
                    sl3d_shim_cloud_format(0);
                    save_point_cloud(0);
                    double t1 = now_ms();
                    sl3d_shim_cloud_format(1);
                    save_point_cloud(1);
                    double t2 = now_ms();
                    if (sl3d_shim_last_status()) { fprintf(stderr, "%s\n", sl3d_shim_last_error()); return 11; }
                    if (w) { t[6].push_back(t1 - t0); t[7].push_back(t2 - t1); }
                }
            }
            if (host == 3) sl3d_shim_materialize(SL3D_SHIM_G_VALID);
            long long n = 0;
            for (int c = 0; c < W; c++)
                for (int r = 0; r < H; r++) n += valid_map[c][r];
            if (npoints && n != npoints) { fprintf(stderr, "valid points differ between the routes: %lld vs %lld\n", n, npoints); return 12; }
            npoints = n;
            double six = 0;
            for (int k = 0; k < 6; k++) six += median(t[k]);
            printf(", \"%s/%s\": {\"wrapped_v\": %.3f, \"wrapped_h\": %.3f, \"unwrap_v\": %.3f, \"unwrap_h\": %.3f, \"c_p_map\": %.3f, \"triangulate\": %.3f, "
                   "\"six_stages\": %.3f",
                   inputs[in], host == 0 ? "device_colrow" : host == 1 ? "host_transposes" : host == 2 ? "deferred_final" : "deferred_none", median(t[0]), median(t[1]), median(t[2]), median(t[3]), median(t[4]), median(t[5]), six);
            if (!t[6].empty()) printf(", \"save_point_cloud_ascii\": %.3f, \"save_point_cloud_binary\": %.3f", median(t[6]), median(t[7]));
            printf("}");
        }
    }
    printf(", \"valid_points\": %lld}\n", npoints);
    return 0;
}
