// followgap_ref_shim.cpp — builds the REFERENCE's own FollowGap class into oracle/_ref/.
// Test infrastructure only.  This file contains no reference code: it #includes
// /root/reference/followgap/followgap.hpp where it lies (see oracle/Makefile) and exposes one
// C function so tests can pin orc_followgap_eval and the product's rl_followgap_eval against the
// reference's compiled behaviour (followgap/followgap.hpp:104-129; wrapper followgap/followgap.pyx:30-31).
#include "followgap.hpp"

extern "C" float ref_followgap_eval(float *lidar, int size, int window_size, float max_distance,
                                    float max_angle, float angle_inc)
{
    FollowGap fg(window_size, max_distance, max_angle, angle_inc);
    return fg.eval(lidar, size);
}
