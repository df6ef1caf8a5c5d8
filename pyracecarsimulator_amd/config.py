"""Config dict of the simulator from the reference's flat rosparam YAML (params.yaml /
simple_params.yaml), without ROS.

``scripts/ros_interface.py:44-71`` and ``scripts/mcts_driver.py:65-92`` build ``car_config`` with
``rospy.get_param("~<name>")`` after ``launch/simulate.launch:12-14`` loaded the YAML; this is the same
key mapping applied to the file directly.  Defaults are the values of the reference's params.yaml
for the keys the scan path reads (params.yaml:3-47,126-130).
"""
from __future__ import annotations

#: config key -> rosparam name (scripts/ros_interface.py:44-71)
ROSPARAM_OF = {
    "scan_beams": "scan_beams", "scan_fov": "scan_fov", "scan_std": "scan_std",
    "free_thresh": "free_thresh", "scan_dist_to_base": "scan_dist_to_base",
    "max_speed": "max_speed", "max_accel": "max_accel", "max_decel": "max_decel",
    "max_steer_ang": "max_steer_ang", "max_steer_vel": "max_steer_vel", "ttc_thresh": "ttc_thresh",
    "width": "width", "length": "length", "scan_max_range": "scan_max_range",
    "update_pose_rate": "update_pose_rate", "wb": "wheelbase", "fc": "friction_coeff",
    "h_cg": "height_cg", "l_r": "l_cg2rear", "l_f": "l_cg2front", "cs_f": "C_S_front",
    "cs_r": "C_S_rear", "I_z": "moment_inertia", "mass": "mass", "batch_size": "batch_size",
}

DEFAULTS = {
    "scan_beams": 1080, "scan_fov": 4.71, "scan_std": 0.01, "free_thresh": 0.8,
    "scan_dist_to_base": 0.275, "max_speed": 7.0, "max_accel": 3.0, "max_decel": 20.0,
    "max_steer_ang": 0.4189, "max_steer_vel": 5.0, "ttc_thresh": 0.001, "width": 0.2032,
    "length": 0.4064, "scan_max_range": 15.0, "update_pose_rate": 0.05, "wb": 0.3302, "fc": 1.0,
    "h_cg": 0.08255, "l_r": 0.17145, "l_f": 0.15875, "cs_f": 2.3, "cs_r": 2.3, "I_z": 0.0398378,
    "mass": 3.17, "batch_size": 200,
}


def load_params(path=None, **overrides):
    """Read a params.yaml-style file into the ``config`` dict ``RacecarSimulator`` takes.  Keys the
    file lacks fall back to ``DEFAULTS``; ``scan_method`` (params.yaml:130) and ``budget`` ride along
    when present."""
    cfg = dict(DEFAULTS)
    if path is not None:
        import yaml
        with open(path) as f:
            raw = yaml.safe_load(f) or {}
        for key, ros_name in ROSPARAM_OF.items():
            if ros_name in raw:
                cfg[key] = raw[ros_name]
        for extra in ("scan_method", "budget", "update_action_rate"):
            if extra in raw:
                cfg[extra] = raw[extra]
    cfg.update(overrides)
    return cfg
