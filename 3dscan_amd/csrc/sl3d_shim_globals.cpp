// sl3d_shim_globals.cpp -- definitions of the reference's path globals for builds that do NOT link the
// reference's own objects (there they are defined by PROJECT_GLOBAL/common_variables.h:6-24,56-62 through
// 1/pattern_generator.cpp).  Initial values are the reference's.
#include "../../include/sl3d_shim.h"

int number_of_codes_vertical = 40;
int number_of_codes_horizontal = 23;
int number_of_patterns_binary_vertical = 6;
int number_of_patterns_binary_horizontal = 5;
int number_of_patterns_fringe = 3;
int fringe_width_pixels_vertical = 32;
int fringe_width_pixels_horizontal = 32;

int (*code_vertical)[Camera_imageheight];
int (*code_horizontal)[Camera_imageheight];
long int (*c_p_map)[2];
int (*selected_region)[Camera_imageheight];
int (*valid_map_vertical)[Camera_imageheight];
int (*valid_map_horizontal)[Camera_imageheight];
int (*valid_map)[Camera_imageheight];
float (*wrapped_phi_vertical)[Camera_imageheight];
float (*wrapped_phi_horizontal)[Camera_imageheight];
float (*unwrapped_phi_vertical)[Camera_imageheight];
float (*unwrapped_phi_horizontal)[Camera_imageheight];
double (*intersection_points)[Camera_imageheight][3];
