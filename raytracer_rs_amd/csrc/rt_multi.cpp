// One frame over several GPUs of a node, from one process: the row loop of Camera::render (rendering/camera.rs:58-64)
// cut into bands that the GPUs take from a shared cursor as they become free (a band costs anything between one ray
// per pixel in the sky and dozens on glass, so who renders what is decided at run time, not dealt in advance).  The
// world is small and read-only: every GPU holds its own copy (its own rt_scene).  Built on the public C ABI plus the
// HIP runtime; RCCL (librccl.so.1, loaded when a second distinct GPU is asked for) moves finished bands into the
// frame on the first GPU: grouped ncclSend/ncclRecv pairs, each peer's bands over that peer's own xGMI link.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_rs_amd.h"
#include "rt_capi_internal.hpp"

namespace {

// The handful of RCCL entry points used, resolved at run time so that the library itself does not depend on RCCL.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;

    bool load(std::string &err) {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            err = std::string("RCCL is needed to collect bands from a second GPU and could not be loaded: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(handle, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            err = "librccl lacks one of ncclCommInitAll / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

struct Worker {
    int device = 0;
    int rank = -1;              // index of `device` among the distinct devices (= its rank in the communicator)
    bool on_first = false;      // renders straight into the frame: same GPU as devices[0]
    rt_scene *scene = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ring[2] = {nullptr, nullptr};
    char *d_buf = nullptr;      // this GPU's bands, at the offsets they have in the frame (workers not on the first GPU,
    size_t d_cap = 0;           // and every worker of rt_multi_render)
    std::vector<uint32_t> mine; // bands taken during the current call
    rt_stats stats{};
    int rc = RT_OK;
    std::string err;
};

}  // namespace

struct rt_multi {
    std::vector<Worker> workers;
    std::vector<int> distinct;        // distinct devices, in order of first appearance
    std::vector<ncclComm_t> comms;    // one per distinct device (empty when there is only one)
    char *h_stage = nullptr;          // pinned staging memory of rt_multi_render, the size of the frame
    size_t h_cap = 0;
    std::vector<uint32_t> last_taken; // bands per worker in the last call
};

namespace {

struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int device) {
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int fail_hip(hipError_t e, const char *what) { return rt::capi_set_error(RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

hipError_t grow_device(Worker &w, size_t bytes) {
    if (bytes <= w.d_cap) return hipSuccess;
    if (w.d_buf) (void)hipFree(w.d_buf);
    w.d_buf = nullptr;
    w.d_cap = 0;
    hipError_t e = hipMalloc((void **)&w.d_buf, bytes);
    if (e == hipSuccess) w.d_cap = bytes;
    return e;
}

void add_stats(rt_stats &into, const rt_stats &s) {
    into.rays += s.rays; into.primary += s.primary; into.shadow += s.shadow; into.reflect += s.reflect; into.refract += s.refract;
    for (int i = 0; i < 5; ++i) { into.prim_tests[i] += s.prim_tests[i]; into.prim_hits[i] += s.prim_hits[i]; }
    into.aabb_tests += s.aabb_tests; into.ray_transforms += s.ray_transforms; into.shades += s.shades; into.lit += s.lit;
    into.specular += s.specular; into.patterns += s.patterns; into.refract_setups += s.refract_setups; into.schlick += s.schlick;
}

struct Plan {
    uint32_t band_rows = 0, n_bands = 0;
    size_t row_bytes = 0;
    uint32_t flags = 0;
    int32_t max_depth = 0;
    bool stats = false;
};

// Validates a multi-GPU render call and cuts the frame into bands.
int make_plan(const rt_multi *m, const rt_camera_desc *camera, const rt_render_opts *opts, bool device_frame, Plan &plan) {
    if (!m || !camera) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "multi and camera must not be NULL");
    if (camera->hsize == 0 || camera->vsize == 0) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "camera has zero size");
    plan = Plan{};
    if (opts) {
        if (opts->row_begin || opts->row_end || opts->band_first || opts->band_step || opts->stream)
            return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "rt_multi_render* renders whole frames on its own streams: row range, band selection and stream must be zero");
        const uint32_t allowed = RT_RENDER_EXACT_DEPTH | (device_frame ? RT_RENDER_QUANTIZED : 0u);
        if (opts->flags & ~allowed) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "flags not available for this call");
        plan.flags = opts->flags;
        plan.max_depth = opts->max_depth;
        plan.stats = opts->collect_stats != 0;
        plan.band_rows = opts->band_rows;
    }
    if (plan.band_rows == 0) {
        // about four bands per worker, so that a slow band can be made up for; never less than 64 rows (a launch
        // needs thousands of 8x8 tiles to fill a GPU), always whole tiles
        uint32_t rows = camera->vsize / (4u * (uint32_t)m->workers.size());
        rows = std::max(64u, (rows + 7u) / 8u * 8u);
        plan.band_rows = std::min(rows, (camera->vsize + 7u) / 8u * 8u);
    }
    plan.n_bands = (camera->vsize + plan.band_rows - 1) / plan.band_rows;
    plan.row_bytes = (size_t)camera->hsize * 3 * ((plan.flags & RT_RENDER_QUANTIZED) ? 1 : sizeof(double));
    return RT_OK;
}

// One worker's part of a call: takes bands from the cursor while its GPU has fewer than two in flight.
// `frame_on_first`: where workers on the first GPU render to (nullptr: every worker renders into its own buffer);
// `stage` / `rgb`: when set, every band is copied to pinned staging memory behind its kernel and from there into the
// caller's frame by this thread.
void run_worker(Worker &w, const rt_camera_desc &camera, const Plan &plan, std::atomic<uint32_t> &cursor, char *frame_on_first,
                char *stage, char *rgb) {
    w.mine.clear();
    std::memset(&w.stats, 0, sizeof w.stats);
    w.rc = RT_OK;
    if (hipSetDevice(w.device) != hipSuccess) {
        w.rc = RT_ERR_HIP;
        w.err = "hipSetDevice failed";
        return;
    }
    char *base = (frame_on_first && w.on_first) ? frame_on_first : w.d_buf;
    uint32_t in_flight[2] = {UINT32_MAX, UINT32_MAX};
    auto band_bytes = [&](uint32_t b) {
        const uint32_t y0 = b * plan.band_rows, y1 = std::min(camera.vsize, y0 + plan.band_rows);
        return (size_t)(y1 - y0) * plan.row_bytes;
    };
    auto retire = [&](int slot) {  // waits for the band issued into `slot` and delivers it
        if (in_flight[slot] == UINT32_MAX) return true;
        const hipError_t e = hipEventSynchronize(w.ring[slot]);
        if (e != hipSuccess) {
            w.rc = RT_ERR_HIP;
            w.err = std::string("hipEventSynchronize: ") + hipGetErrorString(e);
            return false;
        }
        if (rgb) {
            const size_t off = (size_t)in_flight[slot] * plan.band_rows * plan.row_bytes;
            std::memcpy(rgb + off, stage + off, band_bytes(in_flight[slot]));
        }
        in_flight[slot] = UINT32_MAX;
        return true;
    };
    for (int slot = 0;; slot ^= 1) {
        if (!retire(slot)) break;
        const uint32_t b = cursor.fetch_add(1);
        if (b >= plan.n_bands) break;
        rt_render_opts o;
        std::memset(&o, 0, sizeof o);
        o.row_begin = b * plan.band_rows;
        o.row_end = std::min(camera.vsize, o.row_begin + plan.band_rows);
        o.max_depth = plan.max_depth;
        o.flags = plan.flags;
        o.collect_stats = plan.stats ? 1 : 0;
        o.stream = w.stream;
        const size_t off = (size_t)o.row_begin * plan.row_bytes;
        rt_stats st;
        int rc = rt_render_device(w.scene, &camera, &o, reinterpret_cast<double *>(base + off), plan.stats ? &st : nullptr);
        if (rc != RT_OK) {
            w.rc = rc;
            w.err = rt_last_error();
            break;
        }
        if (plan.stats) add_stats(w.stats, st);
        hipError_t e = hipSuccess;
        if (stage) e = hipMemcpyAsync(stage + off, base + off, band_bytes(b), hipMemcpyDeviceToHost, w.stream);
        if (e == hipSuccess) e = hipEventRecord(w.ring[slot], w.stream);
        if (e != hipSuccess) {
            w.rc = RT_ERR_HIP;
            w.err = std::string("queueing a band: ") + hipGetErrorString(e);
            break;
        }
        in_flight[slot] = b;
        w.mine.push_back(b);
    }
    (void)retire(0);
    (void)retire(1);
    if (hipStreamSynchronize(w.stream) != hipSuccess && w.rc == RT_OK) {
        w.rc = RT_ERR_HIP;
        w.err = "hipStreamSynchronize failed";
    }
}

int run_all(rt_multi *m, const rt_camera_desc &camera, const Plan &plan, char *frame_on_first, char *stage, char *rgb, rt_stats *stats,
            double *wall_ms) {
    std::atomic<uint32_t> cursor{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    for (size_t k = 1; k < m->workers.size(); ++k)
        threads.emplace_back([&, k] { run_worker(m->workers[k], camera, plan, cursor, frame_on_first, stage, rgb); });
    {
        int prev = -1;
        (void)hipGetDevice(&prev);
        run_worker(m->workers[0], camera, plan, cursor, frame_on_first, stage, rgb);  // the caller's thread drives the first GPU
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    for (std::thread &t : threads) t.join();
    *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->last_taken.assign(m->workers.size(), 0);
    for (size_t k = 0; k < m->workers.size(); ++k) {
        const Worker &w = m->workers[k];
        if (w.rc != RT_OK) return rt::capi_set_error(w.rc, "GPU " + std::to_string(w.device) + ": " + w.err);
        m->last_taken[k] = (uint32_t)w.mine.size();
        if (stats && plan.stats) add_stats(*stats, w.stats);
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_multi_create(const rt_world_desc *world, const int *devices, uint32_t n_devices, rt_multi **out) {
    if (!world || !devices || !out || n_devices == 0) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "world, devices and out must not be NULL, n_devices must be positive");
    if (n_devices > 64) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "at most 64 workers");
    *out = nullptr;
    int visible = 0;
    hipError_t e = hipGetDeviceCount(&visible);
    if (e != hipSuccess || visible == 0) return rt::capi_set_error(RT_ERR_NO_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    for (uint32_t k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= visible) return rt::capi_set_error(RT_ERR_NO_DEVICE, "device ordinal " + std::to_string(devices[k]) + " out of range");
    std::unique_ptr<rt_multi, void (*)(rt_multi *)> m(new rt_multi, rt_multi_destroy);
    m->workers.resize(n_devices);
    for (uint32_t k = 0; k < n_devices; ++k) {
        Worker &w = m->workers[k];
        w.device = devices[k];
        auto it = std::find(m->distinct.begin(), m->distinct.end(), w.device);
        if (it == m->distinct.end()) {
            m->distinct.push_back(w.device);
            it = m->distinct.end() - 1;
        }
        w.rank = (int)(it - m->distinct.begin());
        w.on_first = w.device == devices[0];
        int rc = rt_scene_create(world, w.device, &w.scene);  // a handle of its own per worker: handles serve one call at a time
        if (rc != RT_OK) return rc;
        DeviceScope scope(w.device);
        e = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking);
        for (int s = 0; e == hipSuccess && s < 2; ++s) e = hipEventCreateWithFlags(&w.ring[s], hipEventDisableTiming);
        if (e != hipSuccess) return fail_hip(e, "creating a worker's stream and events");
    }
    if (m->distinct.size() > 1) {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        std::string err;
        if (!g_rccl.load(err)) return rt::capi_set_error(RT_ERR_UNSUPPORTED, err);
        m->comms.assign(m->distinct.size(), nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), (int)m->distinct.size(), m->distinct.data());
        if (r != ncclSuccess) {
            m->comms.clear();
            return rt::capi_set_error(RT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
        }
    }
    *out = m.release();
    return RT_OK;
}

void rt_multi_destroy(rt_multi *m) {
    if (!m) return;
    for (ncclComm_t c : m->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    for (Worker &w : m->workers) {
        if (!w.scene && !w.stream && !w.d_buf) continue;  // never set up
        DeviceScope scope(w.device);
        if (w.d_buf) (void)hipFree(w.d_buf);
        for (hipEvent_t ev : w.ring)
            if (ev) (void)hipEventDestroy(ev);
        if (w.stream) (void)hipStreamDestroy(w.stream);
        if (w.scene) rt_scene_destroy(w.scene);
    }
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    delete m;
}

int rt_multi_render_device(rt_multi *m, const rt_camera_desc *camera, const rt_render_opts *opts, void *frame_device, rt_stats *stats) {
    Plan plan;
    int rc = make_plan(m, camera, opts, true, plan);
    if (rc != RT_OK) return rc;
    if (!frame_device) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "frame_device must not be NULL");
    if (plan.stats && !stats) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "collect_stats needs a stats pointer");
    if (stats) std::memset(stats, 0, sizeof *stats);
    const size_t frame_bytes = (size_t)camera->vsize * plan.row_bytes;
    for (Worker &w : m->workers)
        if (!w.on_first) {
            DeviceScope scope(w.device);
            hipError_t e = grow_device(w, frame_bytes);
            if (e != hipSuccess) return fail_hip(e, "allocating a GPU's band buffer");
        }
    double wall_ms = 0.0;
    rc = run_all(m, *camera, plan, static_cast<char *>(frame_device), nullptr, nullptr, stats, &wall_ms);
    if (rc != RT_OK) return rc;
    // The bands rendered on other GPUs travel to the frame: runs of consecutive bands of one worker as one message,
    // all messages of the call in one group so that every peer's link is busy at the same time.
    if (!m->comms.empty()) {
        Worker &first = m->workers[0];
        bool any = false;
        ncclResult_t r = g_rccl.GroupStart();
        for (Worker &w : m->workers) {
            if (w.on_first || r != ncclSuccess) continue;
            std::sort(w.mine.begin(), w.mine.end());
            for (size_t k = 0; k < w.mine.size() && r == ncclSuccess;) {
                size_t j = k + 1;
                while (j < w.mine.size() && w.mine[j] == w.mine[j - 1] + 1) ++j;
                const size_t off = (size_t)w.mine[k] * plan.band_rows * plan.row_bytes;
                const size_t end = std::min<size_t>(frame_bytes, (size_t)(w.mine[j - 1] + 1) * plan.band_rows * plan.row_bytes);
                r = g_rccl.Send(w.d_buf + off, end - off, ncclUint8, first.rank, m->comms[(size_t)w.rank], w.stream);
                if (r == ncclSuccess) r = g_rccl.Recv(static_cast<char *>(frame_device) + off, end - off, ncclUint8, w.rank, m->comms[(size_t)first.rank], first.stream);
                any = true;
                k = j;
            }
        }
        const ncclResult_t r_end = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r_end;
        if (r != ncclSuccess) return rt::capi_set_error(RT_ERR_HIP, std::string("collecting the bands with RCCL: ") + g_rccl.GetErrorString(r));
        if (any)
            for (Worker &w : m->workers) {
                DeviceScope scope(w.device);
                hipError_t e = hipStreamSynchronize(w.stream);
                if (e != hipSuccess) return fail_hip(e, "waiting for the collected bands");
            }
    }
    if (stats) stats->kernel_ms = wall_ms;
    return RT_OK;
}

int rt_multi_render(rt_multi *m, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb, rt_stats *stats) {
    Plan plan;
    int rc = make_plan(m, camera, opts, false, plan);
    if (rc != RT_OK) return rc;
    if (!rgb) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "rgb must not be NULL");
    if (plan.stats && !stats) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "collect_stats needs a stats pointer");
    if (stats) std::memset(stats, 0, sizeof *stats);
    const size_t frame_bytes = (size_t)camera->vsize * plan.row_bytes;
    // Every GPU sends its bands to the host over its own PCIe link: nothing travels between GPUs for a host frame.
    for (Worker &w : m->workers) {
        DeviceScope scope(w.device);
        hipError_t e = grow_device(w, frame_bytes);
        if (e != hipSuccess) return fail_hip(e, "allocating a GPU's band buffer");
    }
    if (frame_bytes > m->h_cap) {
        if (m->h_stage) (void)hipHostFree(m->h_stage);
        m->h_stage = nullptr;
        m->h_cap = 0;
        hipError_t e = hipHostMalloc((void **)&m->h_stage, frame_bytes, hipHostMallocPortable);
        if (e != hipSuccess) return fail_hip(e, "allocating pinned staging memory");
        m->h_cap = frame_bytes;
    }
    double wall_ms = 0.0;
    rc = run_all(m, *camera, plan, nullptr, m->h_stage, reinterpret_cast<char *>(rgb), stats, &wall_ms);
    if (rc != RT_OK) return rc;
    if (stats) stats->kernel_ms = wall_ms;
    return RT_OK;
}

int rt_multi_bands_taken(const rt_multi *m, uint32_t *per_worker, uint32_t n) {
    if (!m || (!per_worker && n)) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t k = 0; k < n; ++k) per_worker[k] = k < m->last_taken.size() ? m->last_taken[k] : 0u;
    return RT_OK;
}

}  // extern "C"
