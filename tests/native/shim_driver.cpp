// Headless stand-in for the reference's main() (m_tech_project_console.cpp:366-395): sets the scalar globals
// and selected_region, calls the four stage functions in main()'s order through the drop-in shim, and dumps
// the reference-layout global arrays to a binary file for the Python test to compare with the oracle.
//   shim_driver <data_root> <out.bin> Nv Nh fwv fwh ncv nch
//   shim_driver register <data_root> n tx ty tz rot_step : register_point_clouds() only (9/register_point_clouds.cpp:23)
//   shim_driver patterns <data_root> F fwv fwh      : generate_pattern() only (1/pattern_generator.cpp:513); prints the counts
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sl3d_shim.h"

int main(int argc, char **argv)
{
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
    sl3d_shim_write_debug_images(1);
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
    if (FILE *t = fopen((std::string(argv[1]) + "/Point_cloud/texture.bmp").c_str(), "rb")) {  // main() calls it after triangulate()
        fclose(t);
        save_point_cloud(3);
        if (sl3d_shim_last_status()) { fprintf(stderr, "\n%s\n", sl3d_shim_last_error()); return 16; }
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
