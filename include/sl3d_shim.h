/*
 * sl3d_shim.h -- the reference's own stage interface, re-exported on top of the C ABI (sl3d.h).
 *
 * Link libsl3d_shim.so (+ libsl3d.so) INSTEAD OF the reference's stage objects
 *     3/wrapped_phase.o  4/phase_unwrap.o  5/compute_correspondance.o  7/triangulation.o
 * (unit list: M_tech_project_console/M_tech_project_console.cbp:71-85) and main() keeps calling
 *     compute_wrapped_phase(0); compute_wrapped_phase(1); unwrap_phase(0); unwrap_phase(1);
 *     compute_c_p_map(); triangulate();                 [m_tech_project_console.cpp:372-395]
 * The four functions have the reference's C++ signatures (PROJECT_GLOBAL/intermodule_dependencies.h:10-22)
 * and fill the reference's global arrays, which stay DEFINED where the reference defines them
 * (PROJECT_GLOBAL/common_variables.h, pulled in by 1/pattern_generator.cpp); for a build without the
 * reference's objects, sl3d_shim_globals.cpp provides the same definitions.
 *
 * Compile-time dimensions follow PROJECT_GLOBAL/global_cv.h:49-53 and can be overridden with -D.
 */
#ifndef SL3D_SHIM_H
#define SL3D_SHIM_H

#include <stddef.h>

#ifndef Camera_imagewidth
#define Camera_imagewidth 1600
#endif
#ifndef Camera_imageheight
#define Camera_imageheight 1200
#endif
#ifndef Projector_imagewidth
#define Projector_imagewidth 1280
#endif
#ifndef Projector_imageheight
#define Projector_imageheight 720
#endif
#define total_camera_pixels (Camera_imagewidth * Camera_imageheight)

/* ---- the reference's globals on the path (common_variables.h:6-24,56-62), all [col][row] ---- */
extern int number_of_codes_vertical, number_of_codes_horizontal;
extern int number_of_patterns_binary_vertical, number_of_patterns_binary_horizontal;
extern int number_of_patterns_fringe;
extern int fringe_width_pixels_vertical, fringe_width_pixels_horizontal;
extern int (*code_vertical)[Camera_imageheight];
extern int (*code_horizontal)[Camera_imageheight];
extern long int (*c_p_map)[2];
extern int (*selected_region)[Camera_imageheight];
extern int (*valid_map_vertical)[Camera_imageheight];
extern int (*valid_map_horizontal)[Camera_imageheight];
extern int (*valid_map)[Camera_imageheight];
extern float (*wrapped_phi_vertical)[Camera_imageheight];
extern float (*wrapped_phi_horizontal)[Camera_imageheight];
extern float (*unwrapped_phi_vertical)[Camera_imageheight];
extern float (*unwrapped_phi_horizontal)[Camera_imageheight];
extern double (*intersection_points)[Camera_imageheight][3];

/* ---- the entry points (C++ linkage, as in the reference) ---- */
void generate_pattern();                      /* 1/pattern_generator.cpp:513; the scanf answers come from number_of_patterns_fringe
                                                 and fringe_width_pixels_*; files go to <data root>/Generated_patterns/... */
void compute_wrapped_phase(int pattern_type); /* 3/wrapped_phase.cpp:402 */
void unwrap_phase(int pattern_type);          /* 4/phase_unwrap.cpp:367 (declared int at intermodule_dependencies.h:13, defined void) */
void compute_c_p_map();                       /* 5/compute_correspondance.cpp:630 */
void triangulate();                           /* 7/triangulation.cpp:1444 */
void save_point_cloud(unsigned cloud_index);  /* 8/save_point_cloud.cpp:19: Point_cloud/texture.bmp -> point_cloud_<i>.pcd / .ply
                                                 (device compaction + colour gather of the last triangulate(); standard PCD / PLY
                                                 ASCII text, not PCL 1.6's exact bytes; sl3d_shim_cloud_format(1): binary) */

void register_point_clouds(unsigned num_point_clouds, float tx, float ty, float tz, float rot_step); /* 9/register_point_clouds.cpp:23:
                                                 Point_cloud/point_cloud_<i>.ply (the files save_point_cloud() writes, ASCII or binary) ->
                                                 Point_cloud/registered_point_cloud.ply, rotation on the device */

/* ---- shim configuration (not in the reference) ----
 * The reference reads its inputs from hard-coded paths: absolute /home/pranav/Desktop/M_tech_project_console/...
 * in stages 3 and 7 (3/wrapped_phase.cpp:39-50, 7/triangulation.cpp:152-167,1069-1082), relative paths in
 * stage 4 (4/phase_unwrap.cpp:68-86).  The shim reads the same files below one root directory:
 * sl3d_shim_set_data_root(), else $SL3D_DATA_ROOT, else the reference's absolute path. */
extern "C" {
void sl3d_shim_set_data_root(const char *dir);
/* write the stage-3/4 debug images (Wrapped_phase_image.bmp, Unwrapped_phase_*.bmp) like the reference does */
void sl3d_shim_write_debug_images(int enable);
/* status of the last shim call (an sl3d_status); the reference's functions return nothing */
int sl3d_shim_last_status(void);
const char *sl3d_shim_last_error(void);
void sl3d_shim_reset(void); /* drop the context (e.g. before changing the scalar globals) */
/* Inputs handed over in MEMORY instead of through files: whenever a stage would load <data root>/<relative_path> (the names the
 * reference hard-codes, e.g. "Captured_patterns/Fringe_patterns/Vertical/Undistorted/Captured_image_0.bmp",
 * "Point_cloud/texture.bmp" (channels = 3, B,G,R), "Camera_calibration/Matrices/cam_intrinsic_mat.xml"), it takes the provided
 * buffer instead -- an IplImage's (imageData, widthStep) / a CvMat's doubles.  The image memory is not copied: it must stay
 * valid until the stage that reads it has returned (pinned memory makes the upload an asynchronous DMA).  data / values == NULL
 * withdraws an entry. */
void sl3d_shim_provide_image(const char *relative_path, const unsigned char *data, int width, int height, int channels, size_t stride);
void sl3d_shim_provide_matrix(const char *relative_path, const double *values, int count);
/* save_point_cloud() / register_point_clouds(): 0 = ASCII PCD / PLY like the reference (8/save_point_cloud.cpp:211-217), 1 = the
 * binary flavours of the same formats (PCD "DATA binary", PLY "binary_little_endian") */
void sl3d_shim_cloud_format(int binary);
/* Which of the reference's globals the stage functions fill.
 *   SL3D_SHIM_G_ALL (default): every stage call runs its own kernel and fills its globals before it returns, exactly the hand-over
 *     the reference's stages have among themselves (the contexts keep every stage plane; ~146 MB of [col][row] arrays come down per scan).
 *   anything else = DEFERRED: main()'s own calls (m_tech_project_console.cpp:372-395) stay as they are, but compute_wrapped_phase /
 *     unwrap_phase only bring the mask and their frames to the GPU (from pinned memory or files: asynchronously), compute_c_p_map
 *     does nothing, and triangulate() runs the whole scan as ONE launch of the timed fused kernel -- the kernel behind sl3d_run, the
 *     one bench.py measures -- and then fills the globals the mask names:
 *       SL3D_SHIM_G_FINAL = valid_map + intersection_points: all that the rest of the reference reads (main() reads none of the
 *         globals; 8/save_point_cloud.cpp:85-104 reads these two).  intersection_points then holds the kernel's f32 result widened to
 *         double -- the values save_point_cloud.cpp:94-96 casts them to anyway (the average point spacing it prints, :160-191, then
 *         carries the f32 rounding: ~1e-4 relative); SL3D_SHIM_G_INTERSECTION_POINTS asks for the doubles
 *       SL3D_SHIM_G_NONE: nothing (the shim's own save_point_cloud() follows: it reads the device-resident result)
 *       any other bit: that global too, from the per-stage kernels run on a second context that keeps the stage planes (created on
 *         first use; frames and mask are copied device to device) -- also available after the scan through sl3d_shim_materialize(which).
 *     A deferred scan assumes what main() does: selected_region and the scalar globals do not change between the first stage call of
 *     a scan and its triangulate().  Changing the mode drops the contexts. */
enum {
    SL3D_SHIM_G_NONE = 0u,
    SL3D_SHIM_G_VALID_V = 1u << 0, SL3D_SHIM_G_VALID_H = 1u << 1, SL3D_SHIM_G_VALID = 1u << 2,
    SL3D_SHIM_G_WRAPPED_V = 1u << 3, SL3D_SHIM_G_WRAPPED_H = 1u << 4, SL3D_SHIM_G_UNWRAPPED_V = 1u << 5, SL3D_SHIM_G_UNWRAPPED_H = 1u << 6,
    SL3D_SHIM_G_CODE_V = 1u << 7, SL3D_SHIM_G_CODE_H = 1u << 8, SL3D_SHIM_G_C_P_MAP = 1u << 9,
    SL3D_SHIM_G_INTERSECTION_POINTS = 1u << 10,      /* the fp64 solve's doubles (parity launch) */
    SL3D_SHIM_G_INTERSECTION_POINTS_F32 = 1u << 11,  /* the timed kernel's f32 result widened to double */
    SL3D_SHIM_G_FINAL = (1u << 2) | (1u << 11),
    SL3D_SHIM_G_EVERY = (1u << 12) - 1u,             /* deferred, every global (intersection_points as doubles) */
    SL3D_SHIM_G_ALL = 0xffffffffu                    /* stage by stage (default) */
};
void sl3d_shim_globals(unsigned mask);  /* (without a call: $SL3D_SHIM_GLOBALS = all | final | none | <hex mask>; default all) */
/* after a deferred scan: fill the named globals now (returns an sl3d_status, also in sl3d_shim_last_status); which = 0 fills
 * nothing and only waits until the scan's launch has finished (triangulate() with SL3D_SHIM_G_NONE returns right after launching) */
int sl3d_shim_materialize(unsigned which);
/* measurement switch: 1 = fetch row-major planes and transpose them on the host (the shim's behaviour before the globals were
 * transposed on the device); the results are identical */
void sl3d_shim_host_transpose(int enable);
}

#endif /* SL3D_SHIM_H */
