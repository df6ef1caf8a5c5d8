// car_kernels.h — the step in front of the scan path: the MCTS roll-out
// pose generator.  The reference produces the 200 poses of a roll-out with 200 serial
// Python -> Cython -> C++ steps (scripts/mcts.py:214-231 -> Car::control / Car::updatePosition,
// racecar/src/racecar.cpp:53-98,118-237) and ships them to the GPU; here one lane integrates one
// roll-out in float64 and writes the float32 poses straight into the device buffer the march
// kernel reads, so a batch of roll-outs never touches the host.
//
// Behaviour restated from racecar.cpp (kept quirks: acceleration MAX_DECEL when starting from
// rest :140-143, the st_dyn hysteresis thresholds 0.5 / 0.53 racecar.hpp:112-114, division by the
// velocity in the single-track branch :212-213).  float64 throughout; libm vs OCML trig differ in
// the last ulp, so parity with the reference's compiled Car is <= 1e-9 relative, not bitwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scan {

struct CarParams {            // constructor order of Car (racecar/include/racecar.hpp:32-36)
    double WB, FC, H_CG, L_F, L_R, CS_F, CS_R, MASS, I_Z, CRASH_THRESH, WIDTH, LENGTH,
        MAX_STEER_VEL, MAX_STEER_ANG, MAX_SPEED, MAX_ACCEL, MAX_DECEL;
};

struct CarState {             // Car::getState / setState layout (racecar.cpp:330-376)
    double x, y, theta, velocity, steer_angle, angular_velocity, slip_angle;
    bool st_dyn;
    double travel_dist, total_velo;
    int update_count;
};

__host__ __device__ inline double clampd(double v, double lo, double hi)
{
    return v < lo ? lo : (v > hi ? hi : v);      // std::min(std::max(v, lo), hi)
}

// Car::control + Car::updatePosition(dt)
__device__ inline void car_step(const CarParams &P, CarState &cs, double input_speed,
                                double input_steer, double dt)
{
    const double G = 9.81, K_THRESH = 0.5, ST_THRESH = 0.53;
    const double KP = 2.0 * P.MAX_ACCEL / P.MAX_SPEED;
    // computeFromInput (:118-169)
    double accel, steer_ang_vel;
    const double dif_speed = input_speed - cs.velocity;
    if (cs.velocity > 0) accel = dif_speed > 0 ? clampd(KP * dif_speed, -P.MAX_ACCEL, P.MAX_ACCEL) : -P.MAX_DECEL;
    else                 accel = dif_speed > 0 ? P.MAX_DECEL : clampd(KP * dif_speed, -P.MAX_ACCEL, P.MAX_ACCEL);
    const double dif_steer = input_steer - cs.steer_angle;
    if (fabs(dif_steer) > 0.0001) steer_ang_vel = dif_steer > 0 ? P.MAX_STEER_VEL : -P.MAX_STEER_VEL;
    else steer_ang_vel = 0;

    const double x_before = cs.x, y_before = cs.y;
    const double switch_at = cs.st_dyn ? ST_THRESH : K_THRESH;
    if (cs.velocity < switch_at) {
        // updateNormal (:171-194): kinematic single track
        double sin_h, cos_h;                       // one argument reduction for both (the chain of 200
        sincos(cs.theta, &sin_h, &cos_h);          //  dependent steps of a lone roll-out is latency, all of it)
        const double vx = cs.velocity * cos_h;
        const double vy = cs.velocity * sin_h;
        const double yaw_rate = cs.velocity / P.WB * tan(cs.steer_angle);
        cs.x += vx * dt;
        cs.y += vy * dt;
        cs.theta += yaw_rate * dt;
        cs.velocity += accel * dt;
        cs.steer_angle += steer_ang_vel * dt;
        cs.angular_velocity = 0;
        cs.slip_angle = 0;
        cs.st_dyn = false;
    } else {
        // updateSingle (:196-237): dynamic single track.  The operation order of every expression is
        // the reference's (parity with its compiled Car is <= 1e-9, tests/golden/car_rollouts_ref.npz).
        double sin_h, cos_h;
        sincos(cs.theta + cs.slip_angle, &sin_h, &cos_h);
        const double vx = cs.velocity * cos_h;
        const double vy = cs.velocity * sin_h;
        const double yaw_rate = cs.angular_velocity;
        const double load_rear = G * P.L_R - accel * P.H_CG;        // axle loads under longitudinal acceleration
        const double load_front = G * P.L_F + accel * P.H_CG;
        const double yaw_per_speed = cs.angular_velocity / cs.velocity;
        const double slip_gain = P.FC / (cs.velocity * (P.L_R + P.L_F));
        const double yaw_accel =
            (P.FC * P.MASS / (P.I_Z * P.WB)) *
            (P.L_F * P.CS_F * cs.steer_angle * load_rear +
             cs.slip_angle * (P.L_R * P.CS_R * load_front - P.L_F * P.CS_F * load_rear) -
             yaw_per_speed * ((P.L_F * P.L_F) * P.CS_F * load_rear + (P.L_R * P.L_R) * P.CS_R * load_front));
        const double slip_rate =
            slip_gain * (P.CS_F * cs.steer_angle * (load_rear) -
                         cs.slip_angle * (P.CS_R * load_front + P.CS_F * load_rear) +
                         yaw_per_speed * (P.CS_R * P.L_R * load_front - P.CS_F * P.L_F * load_rear)) -
            cs.angular_velocity;
        cs.x += vx * dt;
        cs.y += vy * dt;
        cs.theta += yaw_rate * dt;
        cs.velocity += accel * dt;
        cs.steer_angle += steer_ang_vel * dt;
        cs.angular_velocity += yaw_accel * dt;
        cs.slip_angle += slip_rate * dt;
        cs.st_dyn = true;
    }
    const double moved_x = x_before - cs.x, moved_y = y_before - cs.y;
    cs.travel_dist += sqrt(moved_x * moved_x + moved_y * moved_y);
    cs.total_velo += cs.velocity;
    cs.update_count++;
    cs.velocity = clampd(cs.velocity, -P.MAX_SPEED, P.MAX_SPEED);
    cs.steer_angle = clampd(cs.steer_angle, -P.MAX_STEER_ANG, P.MAX_STEER_ANG);
}

// One lane per roll-out.  states: 11 doubles per roll-out (getState layout), updated in place when
// states_out != nullptr.  actions: (speed, steer) pairs, one per `action_every` steps
// (scripts/mcts.py:216-222 draws a new pair every 10th step).  poses_out: float32 (x, y, theta) per
// step — the car pose, NOT the lidar pose, exactly what mcts.py:228-231 stores.  velocities_out
// (optional): state[3] after every step = the roll-out's per-step reward (mcts.py:235).
__global__ __launch_bounds__(64) void rollout_kernel(CarParams P, const double *__restrict__ states_in,
                                                     const double *__restrict__ actions,
                                                     int n_rollouts, int n_steps, int action_every,
                                                     double dt, float *__restrict__ poses_out,
                                                     double *__restrict__ states_out,
                                                     double *__restrict__ velocities_out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rollouts) return;
    const double *s = states_in + (size_t)r * 11;
    CarState cs;
    cs.x = s[0]; cs.y = s[1]; cs.theta = s[2]; cs.velocity = s[3]; cs.steer_angle = s[4];
    cs.angular_velocity = s[5]; cs.slip_angle = s[6]; cs.st_dyn = s[7] > 0.0;
    cs.travel_dist = s[8]; cs.total_velo = s[9]; cs.update_count = (int)s[10];
    const int n_act = (n_steps + action_every - 1) / action_every;
    double speed = 0.0, steer = 0.0;
    for (int i = 0; i < n_steps; ++i) {
        if (i % action_every == 0) {
            const double *a = actions + ((size_t)r * n_act + i / action_every) * 2;
            speed = a[0];
            steer = a[1];
        }
        car_step(P, cs, speed, steer, dt);
        float *p = poses_out + ((size_t)r * n_steps + i) * 3;
        p[0] = (float)cs.x;
        p[1] = (float)cs.y;
        p[2] = (float)cs.theta;
        if (velocities_out) velocities_out[(size_t)r * n_steps + i] = cs.velocity;
    }
    if (states_out) {
        double *o = states_out + (size_t)r * 11;
        o[0] = cs.x; o[1] = cs.y; o[2] = cs.theta; o[3] = cs.velocity; o[4] = cs.steer_angle;
        o[5] = cs.angular_velocity; o[6] = cs.slip_angle; o[7] = cs.st_dyn ? 1.0 : 0.0;
        o[8] = cs.travel_dist; o[9] = cs.total_velo; o[10] = (double)cs.update_count;
    }
}

}  // namespace scan
