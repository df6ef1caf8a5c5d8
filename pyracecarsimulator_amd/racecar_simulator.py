"""``RacecarSimulator`` — the reference's simulator façade over the MI355X scan path.

Same constructor config and methods as /root/reference/scripts/racecar_simulator_v2.py
(``__init__`` :7-66, ``setState``/``getState`` :68-83, ``getMeanVelocity`` :85-90,
``getTravelDistance`` :92-97, ``getScan`` :99-105, ``runScan`` :108-116, ``drive`` :118-124,
``updatePose`` :126-132, ``checkCollision`` :134-144, ``checkCollisionMany`` :146-167, ``stop``
:169-188, ``setMap`` :190-197, ``setRaytracingMethod`` :199-204), so ``scripts/mcts.py`` can drive
it unchanged.  The two native pieces behind it are on the GPU:

* the vehicle (``racecar.PyCar``) -> ``racecar.CarBatch`` with one roll-out (float64 dynamics);
* ``car.isCrashed(scanMany(poses))`` -> the crash test fused into the scan kernel, so
  ``checkCollisionMany`` moves 12 B per pose down and one int back.

``rolloutMany`` is the batched form of ``MCTS.rollout`` + ``checkCollisionMany``
(scripts/mcts.py:202-245) for any number of roll-outs per call.
"""
from __future__ import annotations

import math

import numpy as np

from . import racecar as RC
from .scan_simulator import ScanSimulator2D


class RacecarSimulator:

    def __init__(self, config, verbose=False, device=0):
        self.verbose = verbose
        self.map_frame = "map"
        self.base_frame = "base_link"
        self.scan_frame = "laser"
        self.config = config

        self.scan_dist_to_base = config["scan_dist_to_base"]
        self.max_speed = config["max_speed"]
        self.max_accel = config["max_accel"]
        self.max_steer_ang = config["max_steer_ang"]
        self.max_steer_vel = config["max_steer_vel"]
        self.max_decel = config["max_decel"]
        self.width = config["width"]
        self.length = config["length"]
        self.batch_size = config["batch_size"]
        self.num_rays = config["scan_beams"]
        self.scan_fov = config["scan_fov"]
        self.scan_std = config["scan_std"]
        self.scan_max_range = config["scan_max_range"]
        self.free_thresh = config["free_thresh"]
        self.ttc_thresh = config["ttc_thresh"]

        # car object: racecar.PyCar(...) in the reference (:37-44)
        self.car_params = {k: config[k] for k in RC.CAR_PARAM_ORDER}
        self.car = RC.CarBatch(self.car_params, device=device)
        self._state = np.zeros(11, dtype=np.float64)
        # where the lidar beams leave the car body: setCarEdgeDistances (:47-50)
        self.edge_distances = RC.edge_distances(self.num_rays, -self.scan_fov / 2.0,
                                                self.scan_fov / self.num_rays,
                                                self.scan_dist_to_base, config["width"], config["wb"])
        self.scan_simulator = ScanSimulator2D(self.num_rays, self.scan_fov, self.scan_std,
                                              self.batch_size)
        self.scan = np.zeros(self.num_rays, dtype=np.float32)
        self.desired_speed = 0.0
        self.desired_steer_ang = 0.0
        if self.verbose:
            print("Simulator constructed")

    # -- state ---------------------------------------------------------------------
    def setState(self, state):
        self._state = np.array(state, dtype=np.float64)
        self._state[7] = 1.0 if self._state[7] > 0.0 else 0.0      # racecar.cpp:345-352

    def getState(self):
        return self._state.copy()

    def getMeanVelocity(self):
        return self._state[9] / self._state[10]                    # racecar.cpp:100-107

    def getTravelDistance(self):
        return self._state[8]

    def getScan(self):
        return self.scan

    def laserScanFields(self, stamp=None, frame_id="laser"):
        """The fields RunSimulationViz.lidarPub puts into sensor_msgs/LaserScan
        (scripts/ros_interface.py:332-348) as a plain dict (there is no ROS on the GPU box): the fan
        convention every consumer of the scan relies on.  ``ranges`` and ``intensities`` alias the
        current scan, as the reference's message does."""
        return {
            "header": {"stamp": stamp, "frame_id": frame_id},
            "angle_min": -self.scan_fov / 2.0,
            "angle_max": self.scan_fov / 2.0,
            "angle_increment": self.scan_fov / self.num_rays,
            "range_max": self.config["scan_max_range"],
            "ranges": self.scan,
            "intensities": self.scan,
        }

    # -- one tick -------------------------------------------------------------------
    def getScanPose(self):
        """Car::getScanPose (racecar.cpp:378-387): the lidar sits scan_dist_to_base ahead."""
        x, y, th = self._state[0], self._state[1], self._state[2]
        return (x + self.scan_dist_to_base * math.cos(th), y + self.scan_dist_to_base * math.sin(th), th)

    def runScan(self):
        """One scan from the lidar pose (:108-116); the ranges land in the simulator's cached output vector
        (same alias semantics as ``scan()``)."""
        self.scan = self.scan_simulator.scan(*self.getScanPose())

    def drive(self, desired_speed, desired_steer_ang):
        self.desired_speed = desired_speed
        self.desired_steer_ang = desired_steer_ang

    def updatePose(self, dt=0.01):
        """car.control(...) + car.updatePosition(dt) (:126-132)."""
        _, out, _ = self.car.rollout(self._state[None, :],
                                     np.array([[[self.desired_speed, self.desired_steer_ang]]]),
                                     n_steps=1, action_every=1, dt=dt)
        self._state = out[0]

    def checkCollision(self):
        """isCrashed(scan, num_rays, 1) on the scan array's CURRENT contents, as the reference evaluates it
        (:134-144): 0 when the scan touches the car outline, else -2.  ``self.scan`` aliases the simulator's
        cached vector — a later ``scan()`` or an in-place edit changes what is tested, exactly as in the
        reference — so nothing is cached here: the native host test is one pass over num_rays floats."""
        return RC.is_crashed(self.scan, self.num_rays, 1, self.edge_distances, self.ttc_thresh)

    def checkCollisionMany(self, poses):
        """scanMany + isCrashed fused on the device: index of the first crashed pose of the first
        ``batch_size`` poses, else -(batch_size+1)."""
        b = self.batch_size
        p = np.ascontiguousarray(np.asarray(poses, dtype=np.float32)[:b, :3])
        if p.shape[0] < b:
            raise IndexError("checkCollisionMany needs batch_size poses")     # scan_simulator.py:119
        return self.scan_simulator.scan_method.check_collision_many(
            p, self.scan_fov, self.num_rays, self.edge_distances, self.ttc_thresh)

    def rolloutMany(self, states, actions, n_steps=None, action_every=10, dt=0.01):
        """R roll-outs at once: (first crashed pose per roll-out or -(n_steps+1), final states,
        per-step velocities) — poses and ranges stay on the GPU."""
        n_steps = self.batch_size if n_steps is None else n_steps
        return self.car.rollout_check(self.scan_simulator.scan_method, states, actions, self.scan_fov,
                                      self.num_rays, self.edge_distances, self.ttc_thresh,
                                      n_steps=n_steps, action_every=action_every, dt=dt)

    def stop(self):
        state = self.getState()
        state[:11] = 0.0
        self.setState(state)
        self.desired_speed = 0.0
        self.desired_steer_ang = 0.0

    # -- map / method -----------------------------------------------------------------
    def setMap(self, ros_map, resolution, origin):
        max_range_px = int(self.scan_max_range / resolution)        # :196
        self.scan_simulator.setMap(ros_map, max_range_px, resolution, origin)

    def setRaytracingMethod(self, method="RMGPU"):
        self.scan_simulator.setRaytracingMethod(method)
