// racecar_ref_shim.cpp — builds the REFERENCE's own Car class into oracle/_ref/.
// Test infrastructure only.  This file contains no reference code: it #includes
// /root/reference/racecar/src/racecar.cpp where it lies (see oracle/Makefile) and
// exposes a C ABI so tests can pin orc_edge_distances / orc_is_crashed and the
// product's fused crash test against the reference's compiled behaviour
// (racecar/src/racecar.cpp:239-292, :305-328; wrapper racecar/pywrapper/racecar.pyx:75-114).
#include "src/racecar.cpp"

extern "C" {

void *ref_car_create(const double *p /* 17 ctor args, racecar.hpp:32-36 order */)
{
    return new Car(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11],
                   p[12], p[13], p[14], p[15], p[16]);
}
void ref_car_destroy(void *c) { delete static_cast<Car *>(c); }
void ref_car_set_edge_distances(void *c, int num_rays, double ang_min, double inc, double d)
{
    static_cast<Car *>(c)->setCarEdgeDistances(num_rays, ang_min, inc, d);
}
int ref_car_is_crashed(void *c, float *rays, int num_rays, int poses)
{
    return static_cast<Car *>(c)->isCrashed(rays, num_rays, poses);
}
void ref_car_control(void *c, double speed, double steer) { static_cast<Car *>(c)->control(speed, steer); }
void ref_car_update_position(void *c, double dt) { static_cast<Car *>(c)->updatePosition(dt); }
void ref_car_get_state(void *c, double *s) { static_cast<Car *>(c)->getState(s); }
void ref_car_set_state(void *c, double *s) { static_cast<Car *>(c)->setState(s); }
void ref_car_get_scan_pose(void *c, double d, double *pose) { static_cast<Car *>(c)->getScanPose(d, pose); }

}  // extern "C"
