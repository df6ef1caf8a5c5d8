// abi_map.hip — errors, pinned host blocks and the map handle of libscan_amd.so (C ABI: include/scanlib.h).
//
// Replaces, for the scan path only, range_libc's PyOMap (scripts/scan_simulator.py:72, scripts/ros_interface.py:210,
// scripts/two_player/scan.py:45): occupancy grid -> exact EDT, bit map, edge-cell list, all on the device.
// There is no CPU fallback in this library.
#include "abi_internal.h"
#include "edt_kernels.h"

// ------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------
static thread_local std::string g_err = "";

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

const std::string &last_error() { return g_err; }
void set_last_error(const std::string &msg) { g_err = msg; }

extern "C" const char *rl_last_error(void) { return g_err.c_str(); }
extern "C" const char *rl_version(void) { return "scanlib-amd 0.6 (gfx950)"; }

extern "C" int rl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------
// pinned host blocks handed to callers (rl_host_alloc): a scan whose output buffer lies in one of
// them is written by the kernel directly — no staging copy on the way back
// ------------------------------------------------------------------------------
static std::mutex g_host_mu;
struct HostBlock {
    char *p;
    size_t bytes;
    uint64_t devices;      // devices that may have work in flight on the block: where it was allocated and
};                         // every device a scan was launched from with its output inside the block
static std::vector<HostBlock> g_host_blocks;

bool in_host_block(const void *p, size_t bytes, int device)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (auto &b : g_host_blocks)
        if ((const char *)p >= b.p && (const char *)p + bytes <= b.p + b.bytes) {
            if (device >= 0 && device < 64) b.devices |= 1ull << device;
            return true;
        }
    return false;
}

extern "C" int rl_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return fail(RL_ERR_INVALID, "rl_host_alloc: bad arguments");
    if (rl_device_count() <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available");
    void *p = nullptr;
    // (portable + mapped: every device of a multi-device handle writes its pose block's ranges straight into it)
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
        return fail(RL_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    memset(p, 0, bytes);
    int cur = 0;
    (void)hipGetDevice(&cur);
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        g_host_blocks.push_back(HostBlock{(char *)p, bytes, (cur >= 0 && cur < 64) ? 1ull << cur : 0ull});
    }
    *out = p;
    return RL_OK;
}

extern "C" int rl_host_free(void *p)
{
    if (!p) return RL_OK;
    uint64_t devices = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = std::find_if(g_host_blocks.begin(), g_host_blocks.end(),
                               [&](const HostBlock &b) { return b.p == (char *)p; });
        if (it == g_host_blocks.end()) return fail(RL_ERR_INVALID, "rl_host_free: not a block of rl_host_alloc");
        devices = it->devices;
        g_host_blocks.erase(it);
    }
    // a kernel may still be writing into it: wait for the devices that were handed the block — not for every
    // visible device (a rank of an N-GPU job would create contexts on, and stall, its neighbours' GPUs)
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && hipGetDevice(&cur) == hipSuccess) {
        for (int d = 0; d < ndev && d < 64; ++d)
            if (((devices >> d) & 1ull) && hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
        (void)hipSetDevice(cur);
    }
    HIPCHK(hipHostFree(p));
    return RL_OK;
}

int set_device(const rl_map *m)
{
    HIPCHK(hipSetDevice(m->device));
    return RL_OK;
}

// ------------------------------------------------------------------------------
// map
// ------------------------------------------------------------------------------
void host_sincosf(float x, float &s, float &c)
{
    // host twin of scan::det_sincosf (same operations; this TU is built with
    // -ffp-contract=off and fmaf is a single rounding on the host too)
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float P1 = 0x1.921fb6p+0f, P2 = -0x1.777a5cp-25f, P3 = -0x1.ee59dap-50f;
    float k = rintf(x * TWO_OVER_PI);
    float r = fmaf(-k, P1, x);
    r = fmaf(-k, P2, r);
    r = fmaf(-k, P3, r);
    float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float ss = (q & 1) ? cr : sr;
    float cc = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    s = ss;
    c = cc;
}

int map_build_tables(rl_map *m)
{
    // K0: exact EDT + bit-packed occupancy, all on the device
    const int rows = m->rows, cols = m->cols;
    hipLaunchKernelGGL(edt_cols_kernel, dim3((cols + 63) / 64), dim3(1024), 0, m->stream,
                       m->d_occ, rows, cols, m->d_g);
    hipLaunchKernelGGL(edt_rows_kernel, dim3(rows), dim3(256), (size_t)cols * sizeof(int),
                       m->stream, m->d_g, rows, cols, m->d_dt);
    hipLaunchKernelGGL(pack_bits_kernel, dim3((m->bits_stride + 255) / 256, rows), dim3(256), 0,
                       m->stream, m->d_occ, rows, cols, m->bits_stride, m->d_bits);
    if (m->want_edges) {
        if (!m->d_edges) {
            HIPCHK(hipMalloc((void **)&m->d_edges, (size_t)rows * cols * sizeof(uint32_t)));
            HIPCHK(hipMalloc((void **)&m->d_n_edges, 256));
            HIPCHK(hipHostMalloc((void **)&m->pin_n_edges, 64, hipHostMallocDefault));
        }
        HIPCHK(hipMemsetAsync(m->d_n_edges, 0, 4, m->stream));
        hipLaunchKernelGGL(cddt_edges_kernel, dim3((cols + 255) / 256, (rows + EDGE_ROWS_PER_WG - 1) / EDGE_ROWS_PER_WG),
                           dim3(256), 0, m->stream,
                           m->d_occ, rows, cols, m->d_n_edges, m->d_edges);
        HIPCHK(hipMemcpyAsync(m->pin_n_edges, m->d_n_edges, 4, hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->want_edges) m->n_edges = *m->pin_n_edges;
    return RL_OK;
}

extern "C" int rl_map_create(const uint8_t *occ, int rows, int cols, float res, float ox,
                             float oy, float oyaw, int device, rl_map **out)
{
    if (!occ || !out) return fail(RL_ERR_INVALID, "rl_map_create: null pointer");
    if (rows <= 0 || cols <= 0 || rows > 16384 || cols > 16384)
        return fail(RL_ERR_INVALID, "rl_map_create: rows/cols must be in [1,16384] (got %dx%d)",
                    rows, cols);
    if (!(res > 0.0f)) return fail(RL_ERR_INVALID, "rl_map_create: resolution must be > 0");
    int ndev = rl_device_count();
    if (ndev <= 0)
        return fail(RL_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    rl_map *m = new (std::nothrow) rl_map();
    if (!m) return fail(RL_ERR_NOMEM, "out of host memory");
    m->device = device;
    m->rows = rows;
    m->cols = cols;
    m->res = res;
    m->ox = ox;
    m->oy = oy;
    m->oyaw = oyaw;
    m->bits_stride = (cols + 31) / 32;
    auto bail = [&](int code) {
        rl_map_destroy(m);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(RL_ERR_HIP, "hipSetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        m->n_cu = prop.multiProcessorCount;
        m->clock_khz = prop.clockRate;
    }
    const size_t n = (size_t)rows * cols;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&m->d_occ, n) != hipSuccess ||
        hipMalloc((void **)&m->d_g, n * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&m->d_dt, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&m->d_bits, (size_t)rows * m->bits_stride * sizeof(uint32_t)) !=
            hipSuccess)
        return bail(fail(RL_ERR_NOMEM, "device allocation for a %dx%d map failed", rows, cols));
    if (hipMemcpyAsync(m->d_occ, occ, n, hipMemcpyHostToDevice, m->stream) != hipSuccess)
        return bail(fail(RL_ERR_HIP, "map upload failed"));
    int rc = map_build_tables(m);
    if (rc != RL_OK) return bail(rc);

    MapParams &p = m->mp;
    p.dt = m->d_dt;
    p.bits = m->d_bits;
    p.bits_stride = m->bits_stride;
    p.rows = rows;
    p.cols = cols;
    p.frows = (float)rows;
    p.fcols = (float)cols;
    p.res = res;
    p.inv_res = (float)(1.0 / (double)res);
    p.ox = ox;
    p.oy = oy;
    p.wa = -oyaw;                                   // PyOMap: world_angle = -yaw
    host_sincosf(p.wa, p.wa_sin, p.wa_cos);
    if (hipMalloc((void **)&m->d_mp, sizeof(MapParams)) != hipSuccess ||
        hipMemcpy(m->d_mp, &m->mp, sizeof(MapParams), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(RL_ERR_NOMEM, "map parameter upload failed"));
    *out = m;
    return RL_OK;
}

extern "C" int rl_map_create_multi(const uint8_t *occ, int rows, int cols, float res, float ox, float oy,
                                   float oyaw, const int *devices, int n_devices, rl_map **out)
{
    if (!occ || !out || !devices) return fail(RL_ERR_INVALID, "rl_map_create_multi: null pointer");
    if (n_devices < 1 || n_devices > 64) return fail(RL_ERR_INVALID, "rl_map_create_multi: 1..64 devices (got %d)", n_devices);
    rl_map *m = new (std::nothrow) rl_map();
    if (!m) return fail(RL_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n_devices; ++i) {
        rl_map *r = nullptr;
        const int rc = rl_map_create(occ, rows, cols, res, ox, oy, oyaw, devices[i], &r);
        if (rc) {
            const std::string keep = g_err;
            rl_map_destroy(m);
            g_err = keep;
            return rc;
        }
        m->reps.push_back(r);
    }
    const rl_map *r0 = m->reps[0];
    m->device = r0->device;
    m->rows = rows;
    m->cols = cols;
    m->res = res;
    m->ox = ox;
    m->oy = oy;
    m->oyaw = oyaw;
    m->n_cu = r0->n_cu;
    m->clock_khz = r0->clock_khz;
    m->mp = r0->mp;
    *out = m;
    return RL_OK;
}

extern "C" int rl_map_n_devices(const rl_map *m) { return m ? (m->reps.empty() ? 1 : (int)m->reps.size()) : 0; }

extern "C" rl_map *rl_map_replica(rl_map *m, int i)
{
    if (!m) return nullptr;
    if (m->reps.empty()) return i == 0 ? m : nullptr;
    return (i >= 0 && i < (int)m->reps.size()) ? m->reps[i] : nullptr;
}

extern "C" int rl_map_update(rl_map *m, const uint8_t *occ)
{
    if (!m || !occ) return fail(RL_ERR_INVALID, "rl_map_update: null pointer");
    if (!m->reps.empty()) {
        std::lock_guard<std::mutex> lk(m->mu);
        // exclusive against every multi_* call in progress: a batch sees ONE occupancy on all of its devices
        std::unique_lock<std::shared_mutex> wl(m->multi_mu);
        if (m->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
        for (size_t i = 0; i < m->reps.size(); ++i) {
            const int rc = rl_map_update(m->reps[i], occ);
            if (rc) {
                // replicas [0, i) hold the new cells, the others the old ones: no roll-back (the old cells are
                // gone from the host) — the handle is marked and refuses further scans
                if (i > 0) {
                    m->broken.store(true);
                    const std::string keep = g_err;
                    return fail(rc, "rl_map_update failed on replica %zu of %zu after %zu replica(s) had been updated — "
                                    "the multi-device map is now invalid: %s", i, m->reps.size(), i, keep.c_str());
                }
                return rc;
            }
        }
        m->epoch++;
        return RL_OK;
    }
    std::lock_guard<std::mutex> lk(m->mu);
    // exclusive: no host-pointer call of any method of this map is in progress; the device
    // synchronisation covers launches the asynchronous *_device entry points left in flight
    std::unique_lock<std::shared_mutex> wl(m->tables_mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyAsync(m->d_occ, occ, (size_t)m->rows * m->cols, hipMemcpyHostToDevice,
                          m->stream));
    if (m->d_occ_base) {                              // (a new base for later stamps; no outline is in place any more)
        HIPCHK(hipMemcpyAsync(m->d_occ_base, m->d_occ, (size_t)m->rows * m->cols, hipMemcpyDeviceToDevice, m->stream));
        m->n_stamped = 0;
    }
    rc = map_build_tables(m);
    if (rc) return rc;
    m->epoch++;
    return RL_OK;
}

// The two-player tick (scripts/two_player/rcs_two_player.py:105-124): the map as it was created (or last rl_map_update'd)
// with the other car's outline laid over it, tables rebuilt — without the grid crossing PCIe again: the caller sends
// the n outline cells (4 B each) instead of rows x cols bytes.  Every stamp starts from the BASE occupancy: the previous
// outline is gone, like the reference's `ego_map[:] = org_map`.
extern "C" int rl_map_stamp_cells(rl_map *m, const int32_t *flat_idx, int n, uint8_t value)
{
    if (!m || (n > 0 && !flat_idx)) return fail(RL_ERR_INVALID, "rl_map_stamp_cells: null pointer");
    if (n < 0) return fail(RL_ERR_INVALID, "rl_map_stamp_cells: n must be >= 0");
    if (!m->reps.empty()) {
        std::lock_guard<std::mutex> lk(m->mu);
        std::unique_lock<std::shared_mutex> wl(m->multi_mu);
        if (m->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
        for (size_t i = 0; i < m->reps.size(); ++i) {
            const int rc = rl_map_stamp_cells(m->reps[i], flat_idx, n, value);
            if (rc) {
                if (i > 0) m->broken.store(true);
                return rc;
            }
        }
        m->epoch++;
        return RL_OK;
    }
    std::lock_guard<std::mutex> lk(m->mu);
    std::unique_lock<std::shared_mutex> wl(m->tables_mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());                  // launches the *_device entry points left in flight still read the tables
    const size_t cells = (size_t)m->rows * m->cols;
    if (!m->d_occ_base) {                            // (first stamp: the map as it stands is the base)
        HIPCHK(hipMalloc((void **)&m->d_occ_base, cells));
        HIPCHK(hipMemcpyAsync(m->d_occ_base, m->d_occ, cells, hipMemcpyDeviceToDevice, m->stream));
        m->n_stamped = 0;
    }
    if (n > m->stamp_cap) {
        // (the list in place is lost with its buffer: restore the whole grid once)
        HIPCHK(hipMemcpyAsync(m->d_occ, m->d_occ_base, cells, hipMemcpyDeviceToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        if (m->d_stamp) (void)hipFree(m->d_stamp);
        if (m->pin_stamp) (void)hipHostFree(m->pin_stamp);
        m->d_stamp = m->pin_stamp = nullptr;
        m->stamp_cap = m->n_stamped = 0;
        const int cap = std::max(n + 256, 1024);
        HIPCHK(hipMalloc((void **)&m->d_stamp, (size_t)cap * sizeof(int32_t)));
        HIPCHK(hipHostMalloc((void **)&m->pin_stamp, (size_t)cap * sizeof(int32_t), hipHostMallocDefault));
        m->stamp_cap = cap;
    }
    // the indices land in pinned memory the kernel reads directly; ONE launch puts the previous outline's cells back
    // and sets the new ones (the stream was synchronised above: the pinned buffer is not in use)
    if (n > 0) memcpy(m->pin_stamp, flat_idx, (size_t)n * sizeof(int32_t));
    if (n > 0 || m->n_stamped > 0)
        hipLaunchKernelGGL(stamp_swap_kernel, dim3(1), dim3(1024), 0, m->stream, m->d_occ, (const uint8_t *)m->d_occ_base, cells,
                           m->d_stamp, m->n_stamped, (const int32_t *)m->pin_stamp, n, value);
    m->n_stamped = n;
    rc = map_build_tables(m);
    if (rc) return rc;
    m->epoch++;
    return RL_OK;
}

extern "C" void rl_map_destroy(rl_map *m)
{
    if (!m) return;
    if (!m->reps.empty()) {
        for (rl_map *r : m->reps) rl_map_destroy(r);
        delete m;
        return;
    }
    (void)hipSetDevice(m->device);
    if (m->d_occ) (void)hipFree(m->d_occ);
    if (m->d_occ_base) (void)hipFree(m->d_occ_base);
    if (m->d_stamp) (void)hipFree(m->d_stamp);
    if (m->pin_stamp) (void)hipHostFree(m->pin_stamp);
    if (m->d_g) (void)hipFree(m->d_g);
    if (m->d_dt) (void)hipFree(m->d_dt);
    if (m->d_bits) (void)hipFree(m->d_bits);
    if (m->d_mp) (void)hipFree(m->d_mp);
    if (m->d_edges) (void)hipFree(m->d_edges);
    if (m->d_n_edges) (void)hipFree(m->d_n_edges);
    if (m->pin_n_edges) (void)hipHostFree(m->pin_n_edges);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int rl_map_rows(const rl_map *m) { return m ? m->rows : 0; }
extern "C" int rl_map_cols(const rl_map *m) { return m ? m->cols : 0; }
extern "C" int rl_map_device(const rl_map *m) { return m ? m->device : -1; }

extern "C" int rl_map_get_dt(rl_map *m, float *dt_out)
{
    if (!m || !dt_out) return fail(RL_ERR_INVALID, "rl_map_get_dt: null pointer");
    if (!m->reps.empty()) return rl_map_get_dt(m->reps[0], dt_out);
    std::lock_guard<std::mutex> lk(m->mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipMemcpy(dt_out, m->d_dt, (size_t)m->rows * m->cols * sizeof(float),
                     hipMemcpyDeviceToHost));
    return RL_OK;
}

extern "C" int rl_map_get_occ(rl_map *m, uint8_t *occ_out)
{
    if (!m || !occ_out) return fail(RL_ERR_INVALID, "rl_map_get_occ: null pointer");
    if (!m->reps.empty()) return rl_map_get_occ(m->reps[0], occ_out);
    std::lock_guard<std::mutex> lk(m->mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipMemcpy(occ_out, m->d_occ, (size_t)m->rows * m->cols, hipMemcpyDeviceToHost));
    return RL_OK;
}

