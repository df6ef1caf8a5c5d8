"""Where the host-visible latency of the reference's own calls goes: scan() (one pose) and scanMany(200) (one MCTS
roll-out), colombia + cfg2 map: the full ScanSimulator2D path, the shim's calc_range_fan, the raw C call on prepared
pointers, and the device time of the launch (library events)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyracecarsimulator_amd import ScanSimulator2D, _lib, maps, range_libc, workloads

if os.environ.get("SPIN") == "1":                       # hipSetDeviceFlags(hipDeviceScheduleSpin) before anything else
    import torch  # noqa: F401  (the process must share torch's HIP runtime)
    hip = C.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(C.c_uint(1)))


def med(f, n=300, warm=30):
    for _ in range(warm): f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e6

for name, w in (("colombia", workloads.cfg4()), ("cfg2 maze", workloads.cfg2())):
    g = w.gmap
    omap = range_libc.PyOMap(g)
    B = 1081
    sim = ScanSimulator2D(B, 4.71, 0.01, batch_size=200)
    sim.setMap(omap, 300, g.resolution, g.origin)
    sim.setRaytracingMethod("RMGPU")
    poses = maps.sample_free_poses(g, 200, 3)
    m = sim.scan_method
    out = sim.output_vector_many
    p32 = np.ascontiguousarray(poses, np.float32)
    raw = _lib.raw("rl_calc_range_fan")
    a_p, a_o = p32.ctypes.data, out.ctypes.data
    t_scanmany = med(lambda: sim.scanMany(poses))
    t_shim = med(lambda: m.calc_range_fan(p32, out, 4.71, B))
    t_raw = med(lambda: raw(m._h, a_p, 200, 4.71, B, a_o, None, None))
    t_scan = med(lambda: sim.scan(float(poses[0, 0]), float(poses[0, 1]), float(poses[0, 2])))
    a_o1 = sim.output_vector.ctypes.data
    t_raw1 = med(lambda: raw(m._h, a_p, 1, 4.71, B, a_o1, None, None))
    m.set_option("timing", 1)
    ks = []
    for _ in range(50):
        raw(m._h, a_p, 200, 4.71, B, a_o, None, None); ks.append(m.last_kernel_ms() * 1e3)
    k1 = []
    for _ in range(50):
        raw(m._h, a_p, 1, 4.71, B, a_o1, None, None); k1.append(m.last_kernel_ms() * 1e3)
    m.set_option("timing", 0)
    print("%-10s scanMany(200): ScanSimulator2D %.1f us | shim calc_range_fan %.1f | raw C call %.1f | device %.1f      scan(): ScanSimulator2D %.1f us | raw C call %.1f | device %.1f"
          % (name, t_scanmany, t_shim, t_raw, float(np.median(ks)), t_scan, t_raw1, float(np.median(k1))))
