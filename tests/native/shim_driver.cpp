// Headless stand-in for the reference's main() (m_tech_project_console.cpp:366-395): sets the scalar globals
// and selected_region, calls the four stage functions in main()'s order through the drop-in shim, and dumps
// the reference-layout global arrays to a binary file for the Python test to compare with the oracle.
//   shim_driver <data_root> <out.bin> Nv Nh fwv fwh ncv nch
//   shim_driver register <data_root> n tx ty tz rot_step : register_point_clouds() only (9/register_point_clouds.cpp:23)
//   shim_driver patterns <data_root> F fwv fwh      : generate_pattern() only (1/pattern_generator.cpp:513); prints the counts
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sl3d_shim.h"

int main(int argc, char **argv)
{
    if (getenv("SL3D_SHIM_HOST_TRANSPOSE")) sl3d_shim_host_transpose(atoi(getenv("SL3D_SHIM_HOST_TRANSPOSE")));
    if (getenv("SL3D_SHIM_BINARY")) sl3d_shim_cloud_format(atoi(getenv("SL3D_SHIM_BINARY")));
    if (argc >= 8 && std::string(argv[1]) == "register") {
        sl3d_shim_set_data_root(argv[2]);
        register_point_clouds((unsigned)atoi(argv[3]), (float)atof(argv[4]), (float)atof(argv[5]), (float)atof(argv[6]), (float)atof(argv[7]));
        if (sl3d_shim_last_status()) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 30; }
        return 0;
    }
    if (argc >= 6 && std::string(argv[1]) == "patterns") {
        sl3d_shim_set_data_root(argv[2]);
        number_of_patterns_fringe = atoi(argv[3]);
        fringe_width_pixels_vertical = atoi(argv[4]);
        fringe_width_pixels_horizontal = atoi(argv[5]);
        generate_pattern();
        if (sl3d_shim_last_status()) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 20; }
        printf("%d %d %d %d\n", number_of_codes_vertical, number_of_patterns_binary_vertical, number_of_codes_horizontal, number_of_patterns_binary_horizontal);
        return 0;
    }
    if (argc < 9) return 2;
    sl3d_shim_set_data_root(argv[1]);
    sl3d_shim_write_debug_images(getenv("SL3D_SHIM_NO_DEBUG") ? 0 : 1);
    // switches of the test: SL3D_SHIM_HOST_TRANSPOSE=1 (row-major download + host transposes), SL3D_SHIM_BINARY=1 (binary PCD / PLY),
    // SL3D_SHIM_MEMORY=1: the frames, the texture and the calibration are handed over in memory (frames.raw: per axis F fringe,
    // N gray, N inverse planes of W*H bytes; texture.raw: H*W*3 B,G,R; cal.raw: 40 doubles) -- no BMP / XML file exists then
    static std::vector<unsigned char> mem_frames, mem_texture;
    static double mem_cal[40];
    if (getenv("SL3D_SHIM_MEMORY") && atoi(getenv("SL3D_SHIM_MEMORY"))) {
        const size_t px = (size_t)Camera_imagewidth * Camera_imageheight;
        const int Nv = atoi(argv[3]), Nh = atoi(argv[4]), F = 3;
        auto slurp = [&](const char *name, std::vector<unsigned char> &v, size_t n) {
            v.resize(n);
            FILE *f = fopen((std::string(argv[1]) + "/" + name).c_str(), "rb");
            const bool ok = f && fread(v.data(), 1, n, f) == n;
            if (f) fclose(f);
            return ok;
        };
        if (!slurp("frames.raw", mem_frames, px * (size_t)(2 * F + 2 * Nv + 2 * Nh)) || !slurp("texture.raw", mem_texture, px * 3)) return 4;
        FILE *f = fopen((std::string(argv[1]) + "/cal.raw").c_str(), "rb");
        if (!f || fread(mem_cal, sizeof(double), 40, f) != 40) return 5;
        fclose(f);
        const unsigned char *p = mem_frames.data();
        char name[256];
        for (int a = 0; a < 2; a++) {
            const char *ax = a == 0 ? "Vertical" : "Horizontal";
            const int N = a == 0 ? Nv : Nh;
            for (int i = 0; i < F; i++, p += px) {
                snprintf(name, sizeof name, "Captured_patterns/Fringe_patterns/%s/Undistorted/Captured_image_%d.bmp", ax, i);
                sl3d_shim_provide_image(name, p, Camera_imagewidth, Camera_imageheight, 1, Camera_imagewidth);
            }
            for (int i = 0; i < N; i++, p += px) {
                snprintf(name, sizeof name, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/Captured_image_%d.bmp", ax, i);
                sl3d_shim_provide_image(name, p, Camera_imagewidth, Camera_imageheight, 1, Camera_imagewidth);
            }
            for (int i = 0; i < N; i++, p += px) {
                snprintf(name, sizeof name, "Captured_patterns/Coded_patterns/Gray_coded/%s/Undistorted/inverse_Captured_image_%d.bmp", ax, i);
                sl3d_shim_provide_image(name, p, Camera_imagewidth, Camera_imageheight, 1, Camera_imagewidth);
            }
        }
        sl3d_shim_provide_image("Point_cloud/texture.bmp", mem_texture.data(), Camera_imagewidth, Camera_imageheight, 3, (size_t)Camera_imagewidth * 3);
        const char *mats[8] = {"Camera_calibration/Matrices/cam_intrinsic_mat.xml", "Camera_calibration/Matrices/cam_distortion_vect.xml",
                               "Triangulation/Camera_extrinsic_parametrs/world_to_cam_rot_vect.xml", "Triangulation/Camera_extrinsic_parametrs/world_to_cam_trans_vect.xml",
                               "Projector_calibration/Matrices/proj_intrinsic_mat.xml", "Projector_calibration/Matrices/proj_distortion_vect.xml",
                               "Triangulation/Projector_extrinsic_parametrs/world_to_proj_rot_vect.xml", "Triangulation/Projector_extrinsic_parametrs/world_to_proj_trans_vect.xml"};
        const int cnt[8] = {9, 5, 3, 3, 9, 5, 3, 3};
        for (int k = 0, o = 0; k < 8; o += cnt[k], k++) sl3d_shim_provide_matrix(mats[k], mem_cal + o, cnt[k]);
    }
    number_of_patterns_binary_vertical = atoi(argv[3]);
    number_of_patterns_binary_horizontal = atoi(argv[4]);
    fringe_width_pixels_vertical = atoi(argv[5]);
    fringe_width_pixels_horizontal = atoi(argv[6]);
    number_of_codes_vertical = atoi(argv[7]);
    number_of_codes_horizontal = atoi(argv[8]);
    // image_scissor's result: selected_region[col][row] from mask.raw (row-major bytes)
    selected_region = new int[Camera_imagewidth][Camera_imageheight];
    {
        std::vector<unsigned char> m((size_t)Camera_imagewidth * Camera_imageheight);
        FILE *f = fopen((std::string(argv[1]) + "/mask.raw").c_str(), "rb");
        if (!f || fread(m.data(), 1, m.size(), f) != m.size()) return 3;
        fclose(f);
        for (int r = 0; r < Camera_imageheight; r++)
            for (int c = 0; c < Camera_imagewidth; c++) selected_region[c][r] = m[(size_t)r * Camera_imagewidth + c];
    }
    // SL3D_SHIM_GLOBALS=<hex mask>: the deferred mode (sl3d_shim_globals); SL3D_SHIM_SCANS=n: main()'s scan loop n times -- every scan
    // but the last one with ANOTHER selection (a block cleared), so that a stale mask or stale frames would show in the dump
    // (all | final | none | a hex mask -- the shim reads the same variable by itself when sl3d_shim_globals is never called:
    // SL3D_SHIM_DRIVER_NO_CALL=1 leaves the choice to it, the way a relinked main() without any source change would)
    unsigned gmask = (unsigned)SL3D_SHIM_G_ALL;
    if (const char *ge = getenv("SL3D_SHIM_GLOBALS")) {
        const std::string v = ge;
        gmask = v == "all" ? (unsigned)SL3D_SHIM_G_ALL : v == "final" ? (unsigned)SL3D_SHIM_G_FINAL : v == "none" ? (unsigned)SL3D_SHIM_G_NONE
                                                                                                                : (unsigned)strtoul(ge, nullptr, 16);
    }
    if (!getenv("SL3D_SHIM_DRIVER_NO_CALL")) sl3d_shim_globals(gmask);
    const int scans = getenv("SL3D_SHIM_SCANS") ? atoi(getenv("SL3D_SHIM_SCANS")) : 1;
    for (int scan = 0; scan < scans; scan++) {
    if (scans > 1) {
        static std::vector<int> orig;
        int *flat = &selected_region[0][0];
        const size_t npx = (size_t)Camera_imagewidth * Camera_imageheight;
        if (orig.empty()) orig.assign(flat, flat + npx);
        std::copy(orig.begin(), orig.end(), flat);
        if (scan != scans - 1)
            for (int c = Camera_imagewidth / 4; c < Camera_imagewidth / 2; c++)
                for (int r = Camera_imageheight / 4; r < Camera_imageheight / 2; r++) selected_region[c][r] = 0;
    }
    compute_wrapped_phase(0);
    if (sl3d_shim_last_status()) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 10; }
    compute_wrapped_phase(1);
    if (sl3d_shim_last_status()) return 11;
    unwrap_phase(0);
    if (sl3d_shim_last_status()) return 12;
    unwrap_phase(1);
    if (sl3d_shim_last_status()) return 13;
    compute_c_p_map();
    if (sl3d_shim_last_status()) return 14;
    triangulate();
    if (sl3d_shim_last_status()) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 15; }
    FILE *t = fopen((std::string(argv[1]) + "/Point_cloud/texture.bmp").c_str(), "rb");
    if (t || !mem_texture.empty()) {  // main() calls it after triangulate()
        if (t) fclose(t);
        save_point_cloud(3);
        if (sl3d_shim_last_status()) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 16; }
    }
    }  // scan loop
    // a deferred scan filled only the globals its mask names: the dump below wants them all, so the rest is asked for now
    // (intersection_points stays what the mask made it: the f32 result widened with SL3D_SHIM_G_FINAL, the doubles otherwise)
    if (gmask != (unsigned)SL3D_SHIM_G_ALL) {
        unsigned rest = (unsigned)SL3D_SHIM_G_EVERY & ~gmask & ~(unsigned)SL3D_SHIM_G_INTERSECTION_POINTS_F32;
        if (gmask & SL3D_SHIM_G_INTERSECTION_POINTS_F32) rest &= ~(unsigned)SL3D_SHIM_G_INTERSECTION_POINTS;
        if (rest && sl3d_shim_materialize(rest)) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 17; }
    }

    FILE *o = fopen(argv[2], "wb");
    const size_t n = (size_t)Camera_imagewidth * Camera_imageheight;
    fwrite(valid_map_vertical, sizeof(int), n, o);
    fwrite(valid_map_horizontal, sizeof(int), n, o);
    fwrite(valid_map, sizeof(int), n, o);
    fwrite(wrapped_phi_vertical, sizeof(float), n, o);
    fwrite(wrapped_phi_horizontal, sizeof(float), n, o);
    fwrite(unwrapped_phi_vertical, sizeof(float), n, o);
    fwrite(unwrapped_phi_horizontal, sizeof(float), n, o);
    fwrite(code_vertical, sizeof(int), n, o);
    fwrite(code_horizontal, sizeof(int), n, o);
    fwrite(c_p_map, sizeof(long), 2 * n, o);
    fwrite(intersection_points, sizeof(double), 3 * n, o);
    fclose(o);
    return 0;
}
