// One frame over several GPUs of a node, from one process: the row loop of Camera::render (rendering/camera.rs:58-64)
// cut into units of work that the GPUs take from a shared cursor as they become free, so who renders what is decided
// at run time, not dealt in advance.  The world is small and read-only: every GPU holds its own copies (rt_scene
// handles).  Built on the public C ABI plus the HIP runtime.
//
// Units.  The frame is cut into stripes of `band_rows` rows (16 unless the caller says otherwise) and unit u of U is
// the set of stripes u, u + U, u + 2U, ...: every unit holds the same mix of sky, floor and glass, so units cost about
// the same and one launch per unit sees the whole frame's spread of tile costs (contiguous bands of rows differ by an
// order of magnitude in cost and each ends in a tail of its own slowest tiles: measured on one GPU's share of a
// 3840x2160 chapter_11 frame, interleaved units are 4-10 % faster than contiguous bands of the same size).  Two units
// per worker by default: one per stream of a worker, the next one taken by whichever GPU is free first.
//
// Workers.  One per entry of `devices`: a host thread that lives as long as the handle (the caller's thread drives
// worker 0), with two slots -- a stream, a scene handle and two events each -- so that two units are in flight on its
// GPU at any time: the under-occupied end of one launch overlaps the start of the next.
//
// Collection of the frame on devices[0].  Workers on that GPU render their stripes straight into the frame
// (RT_RENDER_IN_PLACE).  The others render a unit's stripes packed one after the other; the packed unit travels as ONE
// message into staging memory on devices[0] and a small kernel there puts its stripes in place.  Two ways to travel,
// both overlapped with the rendering that is still going on elsewhere:
//   RT_GATHER_RCCL   a worker that has issued its last unit sends its units with RCCL -- one group of ncclSend /
//                    ncclRecv pairs on a communicator from ncclCommInitAll over the distinct GPUs -- over its own
//                    xGMI link; the receives and the placement run on a stream of devices[0] that belongs to that
//                    worker.  RCCL is looked for among the libraries already in the process first (a host program
//                    such as PyTorch brings its own copy; two copies in one process would each set up their own
//                    transports) and loaded by name otherwise, so the library itself depends on the HIP runtime only.
//   RT_GATHER_COPY   every unit is copied by the GPU that rendered it (hipMemcpyPeerAsync on the unit's stream, behind
//                    its kernel), and the placement on devices[0] waits for that copy's event: no communicator, no
//                    library; used when RCCL is not available, cannot be initialised, or RT_MULTI_GATHER=copy is set
//                    in the environment.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_rs_amd.h"
#include "rt_capi_internal.hpp"
#include "rt_device.hpp"

namespace {

// The handful of RCCL entry points used, declared here (their C signatures are stable across RCCL 2.x) and resolved at
// run time, so that neither building nor loading this library needs RCCL.
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
enum { kNcclSuccess = 0, kNcclUint8 = 1 /* ncclUint8 == ncclChar + 1 in nccl.h's ncclDataType_t */ };

struct Rccl {
    void *handle = nullptr;
    bool preloaded = false;  // found in the process already (not loaded by us)
    std::string path;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;

    // A copy of RCCL that the process has mapped already: its path from /proc/self/maps.
    static std::string mapped_copy() {
        std::string found;
        if (FILE *f = std::fopen("/proc/self/maps", "r")) {
            char line[4096];
            while (std::fgets(line, sizeof line, f)) {
                const char *p = std::strstr(line, "librccl.so");
                if (!p) continue;
                const char *start = std::strchr(line, '/');
                if (!start) continue;
                found.assign(start);
                while (!found.empty() && (found.back() == '\n' || found.back() == ' ')) found.pop_back();
                break;
            }
            std::fclose(f);
        }
        return found;
    }

    bool load(std::string &err) {
        if (handle) return true;
        const std::string mapped = mapped_copy();
        if (!mapped.empty()) {
            handle = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (handle) {
                preloaded = true;
                path = mapped;
            }
        }
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            if (handle) break;
            handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (handle) {
                preloaded = true;
                path = name;
            }
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (handle) break;
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) path = name;
        }
        if (!handle) {
            const char *why = dlerror();
            err = std::string("RCCL could not be loaded: ") + (why ? why : "librccl.so.1 not found");
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
            handle = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
// One communicator is never used by two threads at a time: every RCCL call of this file is made under this mutex.
std::mutex g_rccl_mutex;

using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Slot {
    rt_scene *scene = nullptr;      // a handle of its own: handles serve one launch at a time
    hipStream_t stream = nullptr;
    hipEvent_t rendered = nullptr;  // behind the band's kernel
    hipEvent_t done = nullptr;      // behind the band's transfer, if it has one (else behind the kernel)
    uint32_t band = UINT32_MAX;     // the unit in flight
};

struct Worker {
    int device = 0;
    int rank = -1;              // index of `device` among the distinct devices (= its rank in the communicator)
    bool on_first = false;      // renders straight into the frame: same GPU as devices[0]
    Slot slot[2];
    hipStream_t recv = nullptr;  // on devices[0]: where this worker's units are received and put in place
    char *d_buf = nullptr;       // this GPU's units, packed (workers not on the first GPU, and every worker of
    size_t d_cap = 0;            // rt_multi_render)
    std::vector<uint32_t> mine;  // units taken during the current call
    rt_stats stats{};
    double rendered_ms = 0.0, finished_ms = 0.0;  // since the start of the call: last kernel done; last band delivered
    int rc = RT_OK;
    std::string err;
    std::thread thread;
};

struct Plan {
    uint32_t band_rows = 0;  // height of a stripe
    uint32_t n_units = 0;    // unit u = stripes u, u + n_units, ...
    uint32_t n_stripes = 0;
    size_t row_bytes = 0;
    uint32_t flags = 0;
    int32_t max_depth = 0;
    bool stats = false;
    std::vector<uint32_t> unit_rows;  // rows of unit u
    std::vector<size_t> unit_off;     // where unit u starts in a buffer that holds all units packed, in bytes (n_units + 1)
};

struct Job {
    rt_camera_desc camera{};
    Plan plan;
    char *frame_device = nullptr;    // the frame on devices[0] (workers there render into it, other GPUs' units are collected into it), or nullptr
    char *stage = nullptr, *rgb = nullptr;  // host frame: pinned staging memory and the caller's buffer
    Clock::time_point t0;
};

}  // namespace

struct rt_multi {
    std::vector<Worker> workers;
    std::vector<int> distinct;        // distinct devices, in order of first appearance
    std::vector<ncclComm_t> comms;    // one per distinct device (empty unless gather == RT_GATHER_RCCL)
    int gather = RT_GATHER_NONE;
    std::string gather_note;          // why RCCL is not in use, when it was wanted
    char *h_stage = nullptr;          // pinned staging memory of rt_multi_render, the size of the frame
    size_t h_cap = 0;
    char *stage0 = nullptr;           // on devices[0]: where packed units of other GPUs arrive, the size of the frame
    size_t stage0_cap = 0;
    bool test_peers = false;          // RT_MULTI_TEST_PEERS=1: workers other than the first are treated as if on another GPU
    std::vector<uint32_t> last_taken; // units per worker in the last call
    std::vector<double> last_ms;      // per worker: when its last kernel had finished, since the start of the call
    Plan last_plan;
    // the job the worker threads wait for
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::atomic<uint64_t> generation{0};
    std::atomic<uint32_t> pending{0};
    std::atomic<bool> quit{false};
    Job job;
    std::atomic<uint32_t> cursor{0};
};

namespace {

struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int device) {
        (void)hipGetDevice(&prev);
        if (prev != device) (void)hipSetDevice(device);
        else prev = -1;
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

// Into how many units the frame is cut when the caller names no number: two per worker (one for each of its
// streams), fewer when a unit would hold less than 2 048 of the 8x8-pixel tiles a launch hands to its waves (a GPU
// runs 4 096 waves at a time); one worker alone takes the frame as one unit (exactly what rt_render_device does).
uint32_t default_units(const rt_camera_desc &camera, size_t n_workers, uint32_t n_stripes) {
    if (n_workers <= 1) return 1;
    const uint64_t tiles = (uint64_t)((camera.hsize + 7u) / 8u) * ((camera.vsize + 7u) / 8u);
    uint64_t units = 2u * n_workers;
    while (units > n_workers && tiles / units < 2048u) --units;
    return (uint32_t)std::max<uint64_t>(1u, std::min<uint64_t>(units, n_stripes));
}

// Validates a multi-GPU render call and cuts the frame into units.
int make_plan(const rt_multi *m, const rt_camera_desc *camera, const rt_render_opts *opts, bool device_frame, Plan &plan) {
    if (!m || !camera) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "multi and camera must not be NULL");
    if (camera->hsize == 0 || camera->vsize == 0) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "camera has zero size");
    plan = Plan{};
    if (opts) {
        if (opts->row_begin || opts->row_end || opts->band_first || opts->stream)
            return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "rt_multi_render* renders whole frames on its own streams: row range, band_first and stream must be zero");
        const uint32_t allowed = RT_RENDER_EXACT_DEPTH | (device_frame ? RT_RENDER_QUANTIZED : 0u);
        if (opts->flags & ~allowed) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "flags not available for this call");
        if (opts->band_rows % 8u) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "band_rows must be a multiple of 8, the height of a tile");
        plan.flags = opts->flags;
        plan.max_depth = opts->max_depth;
        plan.stats = opts->collect_stats != 0;
        plan.band_rows = opts->band_rows;
        plan.n_units = opts->band_step;
    }
    if (m->workers.size() == 1 && plan.n_units == 0 && plan.band_rows == 0) plan.band_rows = (camera->vsize + 7u) / 8u * 8u;  // the frame as one launch
    if (plan.band_rows == 0) plan.band_rows = 16;
    plan.n_stripes = (camera->vsize + plan.band_rows - 1) / plan.band_rows;
    if (plan.n_units == 0) plan.n_units = default_units(*camera, m->workers.size(), plan.n_stripes);
    plan.n_units = std::min(plan.n_units, plan.n_stripes);
    plan.row_bytes = (size_t)camera->hsize * 3 * ((plan.flags & RT_RENDER_QUANTIZED) ? 1 : sizeof(double));
    plan.unit_rows.assign(plan.n_units, 0);
    plan.unit_off.assign(plan.n_units + 1, 0);
    for (uint32_t st = 0; st < plan.n_stripes; ++st)
        plan.unit_rows[st % plan.n_units] += std::min(plan.band_rows, camera->vsize - st * plan.band_rows);
    for (uint32_t u = 0; u < plan.n_units; ++u) {
        // (packed units start on 256-byte boundaries: the placement kernel and the transports like aligned messages)
        const size_t bytes = ((size_t)plan.unit_rows[u] * plan.row_bytes + 255u) / 256u * 256u;
        plan.unit_off[u + 1] = plan.unit_off[u] + bytes;
    }
    return RT_OK;
}

void worker_failed(Worker &w, int rc, const std::string &what) {
    if (w.rc != RT_OK) return;
    w.rc = rc;
    w.err = what;
}

// A packed unit that has arrived in the staging memory on devices[0] is put in place: its stripes go to their rows of
// the frame.  Queued on `stream`, a stream of devices[0].
hipError_t place_unit(const rt_multi *m, const Job &job, uint32_t u, hipStream_t stream) {
    const Plan &plan = job.plan;
    if (plan.unit_rows[u] == 0) return hipSuccess;
    DeviceScope first(m->workers[0].device);
    const size_t stripe_bytes = (size_t)plan.band_rows * plan.row_bytes;
    return rt::launch_place_stripes(m->stage0 + plan.unit_off[u], job.frame_device + (size_t)u * stripe_bytes, stripe_bytes,
                                    (size_t)plan.n_units * stripe_bytes, (size_t)plan.unit_rows[u] * plan.row_bytes, stream);
}

// RCCL collection of one worker's units into the frame on devices[0]: every unit one message, all of them in one group;
// the sends wait for both of the worker's streams, the receives run on the worker's stream on devices[0], and so does
// the placement of the stripes behind them.  Called by the worker's thread once it has issued its last unit; returns
// when its part of the frame is complete.
void collect_with_rccl(rt_multi *m, Worker &w, const Job &job) {
    if (w.mine.empty()) return;
    hipError_t e = hipStreamWaitEvent(w.slot[0].stream, w.slot[1].rendered, 0);  // (an event never recorded counts as complete)
    if (e != hipSuccess) return worker_failed(w, RT_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
    const Plan &plan = job.plan;
    const int first_rank = m->workers[0].rank;
    ncclResult_t r;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        r = g_rccl.GroupStart();
        for (size_t k = 0; k < w.mine.size() && r == kNcclSuccess; ++k) {
            const uint32_t u = w.mine[k];
            const size_t off = plan.unit_off[u], bytes = (size_t)plan.unit_rows[u] * plan.row_bytes;
            if (!bytes) continue;
            r = g_rccl.Send(w.d_buf + off, bytes, kNcclUint8, first_rank, m->comms[(size_t)w.rank], w.slot[0].stream);
            if (r == kNcclSuccess) r = g_rccl.Recv(m->stage0 + off, bytes, kNcclUint8, w.rank, m->comms[(size_t)first_rank], w.recv);
        }
        const ncclResult_t r_end = g_rccl.GroupEnd();
        if (r == kNcclSuccess) r = r_end;
    }
    if (r != kNcclSuccess) return worker_failed(w, RT_ERR_HIP, std::string("collecting the units with RCCL: ") + g_rccl.GetErrorString(r));
    for (size_t k = 0; k < w.mine.size() && e == hipSuccess; ++k) e = place_unit(m, job, w.mine[k], w.recv);
    if (e == hipSuccess) e = hipStreamSynchronize(w.slot[0].stream);
    if (e == hipSuccess) {
        DeviceScope first(m->workers[0].device);
        e = hipStreamSynchronize(w.recv);
    }
    if (e != hipSuccess) worker_failed(w, RT_ERR_HIP, std::string("waiting for the collected units: ") + hipGetErrorString(e));
}

// One worker's part of a call: takes units from the cursor while its GPU has fewer than two in flight.
void run_worker(rt_multi *m, Worker &w) {
    const Job &job = m->job;
    const Plan &plan = job.plan;
    const rt_camera_desc &camera = job.camera;
    w.mine.clear();
    std::memset(&w.stats, 0, sizeof w.stats);
    w.rc = RT_OK;
    w.err.clear();
    w.rendered_ms = w.finished_ms = 0.0;
    w.slot[0].band = w.slot[1].band = UINT32_MAX;
    if (hipSetDevice(w.device) != hipSuccess) return worker_failed(w, RT_ERR_HIP, "hipSetDevice failed");
    const bool straight = job.frame_device && w.on_first;   // renders its stripes into the frame itself
    const bool collect = job.frame_device && !w.on_first;   // its packed units travel to devices[0]
    if (plan.stats)
        for (Slot &s : w.slot)
            if (rt::capi_counters_reset(s.scene, s.stream) != RT_OK) return worker_failed(w, RT_ERR_HIP, rt_last_error());
    auto retire = [&](Slot &s) {  // waits for the unit in flight in `s` and delivers it
        if (s.band == UINT32_MAX) return true;
        const hipError_t e = hipEventSynchronize(s.done);
        if (e != hipSuccess) {
            worker_failed(w, RT_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
            return false;
        }
        if (job.rgb) {  // staged packed unit -> its stripes of the caller's frame
            const char *from = job.stage + plan.unit_off[s.band];
            for (uint32_t st = s.band; st < plan.n_stripes; st += plan.n_units) {
                const uint32_t y0 = st * plan.band_rows, y1 = std::min(camera.vsize, y0 + plan.band_rows);
                std::memcpy(job.rgb + (size_t)y0 * plan.row_bytes, from, (size_t)(y1 - y0) * plan.row_bytes);
                from += (size_t)(y1 - y0) * plan.row_bytes;
            }
        }
        s.band = UINT32_MAX;
        return true;
    };
    for (int k = 0;; k ^= 1) {
        Slot &s = w.slot[k];
        if (!retire(s)) break;
        const uint32_t u = m->cursor.fetch_add(1);
        if (u >= plan.n_units) break;
        rt_render_opts o;
        std::memset(&o, 0, sizeof o);
        o.band_rows = plan.band_rows;
        o.band_first = u;
        o.band_step = plan.n_units;
        o.max_depth = plan.max_depth;
        o.flags = plan.flags | (straight ? RT_RENDER_IN_PLACE : 0u);
        o.stream = s.stream;
        const size_t off = plan.unit_off[u], bytes = (size_t)plan.unit_rows[u] * plan.row_bytes;
        double *target = reinterpret_cast<double *>(straight ? job.frame_device : w.d_buf + off);
        const int rc = plan.stats ? rt::capi_render_device_counting(s.scene, &camera, &o, target) : rt_render_device(s.scene, &camera, &o, target, nullptr);
        if (rc != RT_OK) {
            worker_failed(w, rc, rt_last_error());
            break;
        }
        hipError_t e = hipEventRecord(s.rendered, s.stream);
        if (e == hipSuccess && job.stage && bytes) e = hipMemcpyAsync(job.stage + off, w.d_buf + off, bytes, hipMemcpyDeviceToHost, s.stream);
        else if (e == hipSuccess && collect && m->gather == RT_GATHER_COPY && bytes)
            e = hipMemcpyPeerAsync(m->stage0 + off, m->workers[0].device, w.d_buf + off, w.device, bytes, s.stream);
        if (e == hipSuccess) e = hipEventRecord(s.done, s.stream);
        if (e == hipSuccess && collect && m->gather == RT_GATHER_COPY) {
            // on devices[0], behind the copy: the unit's stripes to their rows of the frame
            DeviceScope first(m->workers[0].device);
            e = hipStreamWaitEvent(w.recv, s.done, 0);
            if (e == hipSuccess) e = place_unit(m, job, u, w.recv);
        }
        if (e != hipSuccess) {
            worker_failed(w, RT_ERR_HIP, std::string("queueing a unit: ") + hipGetErrorString(e));
            break;
        }
        s.band = u;
        w.mine.push_back(u);
    }
    // the kernels of this GPU have finished when both slots' `rendered` events have
    for (Slot &s : w.slot)
        if (s.band != UINT32_MAX && hipEventSynchronize(s.rendered) != hipSuccess) worker_failed(w, RT_ERR_HIP, "hipEventSynchronize failed");
    w.rendered_ms = ms_since(job.t0);
    if (w.rc == RT_OK && collect && m->gather == RT_GATHER_RCCL) collect_with_rccl(m, w, job);
    (void)retire(w.slot[0]);
    (void)retire(w.slot[1]);
    for (Slot &s : w.slot)
        if (hipStreamSynchronize(s.stream) != hipSuccess) worker_failed(w, RT_ERR_HIP, "hipStreamSynchronize failed");
    if (collect && m->gather == RT_GATHER_COPY) {
        DeviceScope first(m->workers[0].device);
        if (hipStreamSynchronize(w.recv) != hipSuccess) worker_failed(w, RT_ERR_HIP, "hipStreamSynchronize failed (placement)");
    }
    if (plan.stats && w.rc == RT_OK)
        for (Slot &s : w.slot) {
            rt_stats st;
            if (rt::capi_counters_read(s.scene, s.stream, &st) != RT_OK) worker_failed(w, RT_ERR_HIP, rt_last_error());
            else add_stats(w.stats, st);
        }
    w.finished_ms = ms_since(job.t0);
}

// A worker thread: waits for the next job (spinning for a short while first: in a frame loop the next job is a few
// microseconds away, and waking from a condition variable costs more than a small frame's band), runs its part, reports.
void worker_main(rt_multi *m, size_t k) {
    uint64_t seen = 0;
    (void)hipSetDevice(m->workers[k].device);
    for (;;) {
        const Clock::time_point idle_since = Clock::now();
        while (m->generation.load(std::memory_order_acquire) == seen && !m->quit.load(std::memory_order_acquire)) {
            if (ms_since(idle_since) < 0.3) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                continue;
            }
            std::unique_lock<std::mutex> lock(m->mu);
            m->cv_go.wait(lock, [&] { return m->generation.load(std::memory_order_acquire) != seen || m->quit.load(std::memory_order_acquire); });
        }
        if (m->quit.load(std::memory_order_acquire)) return;
        seen = m->generation.load(std::memory_order_acquire);
        run_worker(m, m->workers[k]);
        if (m->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lock(m->mu);
            m->cv_done.notify_all();
        }
    }
}

int run_all(rt_multi *m, rt_stats *stats) {
    m->cursor.store(0);
    m->job.t0 = Clock::now();
    const uint32_t others = (uint32_t)m->workers.size() - 1;
    if (others) {
        std::lock_guard<std::mutex> lock(m->mu);
        m->pending.store(others, std::memory_order_release);
        m->generation.fetch_add(1, std::memory_order_acq_rel);
        m->cv_go.notify_all();
    }
    {
        int prev = -1;
        (void)hipGetDevice(&prev);
        run_worker(m, m->workers[0]);  // the caller's thread drives the first worker
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    if (others) {
        const Clock::time_point waiting_since = Clock::now();
        while (m->pending.load(std::memory_order_acquire) != 0) {
            if (ms_since(waiting_since) < 0.3) continue;
            std::unique_lock<std::mutex> lock(m->mu);
            m->cv_done.wait(lock, [&] { return m->pending.load(std::memory_order_acquire) == 0; });
        }
    }
    m->last_plan = m->job.plan;
    m->last_taken.assign(m->workers.size(), 0);
    m->last_ms.assign(m->workers.size(), 0.0);
    double rendered = 0.0, finished = 0.0;
    for (size_t k = 0; k < m->workers.size(); ++k) {
        const Worker &w = m->workers[k];
        if (w.rc != RT_OK) return rt::capi_set_error(w.rc, "GPU " + std::to_string(w.device) + ": " + w.err);
        m->last_taken[k] = (uint32_t)w.mine.size();
        m->last_ms[k] = w.rendered_ms;
        rendered = std::max(rendered, w.rendered_ms);
        finished = std::max(finished, w.finished_ms);
        if (stats && m->job.plan.stats) add_stats(*stats, w.stats);
    }
    if (stats) {
        stats->kernel_ms = rendered;
        stats->gather_ms = std::max(0.0, finished - rendered);
    }
    return RT_OK;
}

// Lets every other GPU of the handle reach memory of devices[0] (and the reverse, which RCCL's transports want):
// peer copies then go over xGMI directly instead of through host memory.  Failing here costs speed, not correctness.
void enable_peer_access(const std::vector<int> &distinct) {
    for (size_t a = 0; a < distinct.size(); ++a)
        for (size_t b = 0; b < distinct.size(); ++b) {
            if (a == b || (a != 0 && b != 0)) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, distinct[a], distinct[b]) != hipSuccess || !can) continue;
            DeviceScope scope(distinct[a]);
            const hipError_t e = hipDeviceEnablePeerAccess(distinct[b], 0);
            if (e != hipSuccess) (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled included: clear the sticky error
        }
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
    // Test aid: treat every worker but the first as if it sat on another GPU (own buffer, peer copy into the staging
    // memory, placement), so that the whole collection by copies can be exercised on a box with a single GPU.
    if (const char *t = getenv("RT_MULTI_TEST_PEERS")) m->test_peers = t[0] == '1';
    for (uint32_t k = 0; k < n_devices; ++k) {
        Worker &w = m->workers[k];
        w.device = devices[k];
        auto it = std::find(m->distinct.begin(), m->distinct.end(), w.device);
        if (it == m->distinct.end()) {
            m->distinct.push_back(w.device);
            it = m->distinct.end() - 1;
        }
        w.rank = (int)(it - m->distinct.begin());
        w.on_first = w.device == devices[0] && !(m->test_peers && k > 0);
        for (Slot &s : w.slot) {
            int rc = rt_scene_create(world, w.device, &s.scene);
            if (rc != RT_OK) return rc;
            DeviceScope scope(w.device);
            e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
            if (e != hipSuccess) return fail_hip(e, "creating a worker's streams and events");
        }
        if (!w.on_first) {
            DeviceScope scope(devices[0]);
            e = hipStreamCreateWithFlags(&w.recv, hipStreamNonBlocking);
            if (e != hipSuccess) return fail_hip(e, "creating a worker's receiving stream");
        }
    }
    if (m->distinct.size() == 1 && m->test_peers && n_devices > 1) m->gather = RT_GATHER_COPY;
    if (m->distinct.size() > 1) {
        enable_peer_access(m->distinct);
        m->gather = RT_GATHER_COPY;
        const char *want = getenv("RT_MULTI_GATHER");
        if (want && std::strcmp(want, "copy") == 0) {
            m->gather_note = "RT_MULTI_GATHER=copy";
        } else {
            std::lock_guard<std::mutex> lock(g_rccl_mutex);
            std::string err;
            if (!g_rccl.load(err)) {
                m->gather_note = err;
            } else {
                m->comms.assign(m->distinct.size(), nullptr);
                const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), (int)m->distinct.size(), m->distinct.data());
                if (r != kNcclSuccess) {
                    m->comms.clear();
                    m->gather_note = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
                } else {
                    m->gather = RT_GATHER_RCCL;
                }
            }
            if (m->gather != RT_GATHER_RCCL && want && std::strcmp(want, "rccl") == 0)  // asked for by name: no silent fallback
                return rt::capi_set_error(RT_ERR_UNSUPPORTED, "RT_MULTI_GATHER=rccl: " + m->gather_note);
        }
    }
    for (size_t k = 1; k < m->workers.size(); ++k) m->workers[k].thread = std::thread(worker_main, m.get(), k);
    *out = m.release();
    return RT_OK;
}

void rt_multi_destroy(rt_multi *m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lock(m->mu);
        m->quit.store(true, std::memory_order_release);
        m->cv_go.notify_all();
    }
    for (Worker &w : m->workers)
        if (w.thread.joinable()) w.thread.join();
    if (!m->comms.empty()) {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        for (size_t r = 0; r < m->comms.size(); ++r)
            if (m->comms[r]) {
                DeviceScope scope(m->distinct[r]);
                (void)g_rccl.CommDestroy(m->comms[r]);
            }
    }
    for (Worker &w : m->workers) {
        if (w.recv) {
            DeviceScope scope(m->workers[0].device);
            (void)hipStreamDestroy(w.recv);
        }
        DeviceScope scope(w.device);
        if (w.d_buf) (void)hipFree(w.d_buf);
        for (Slot &s : w.slot) {
            if (s.rendered) (void)hipEventDestroy(s.rendered);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            if (s.scene) rt_scene_destroy(s.scene);
        }
    }
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->stage0 && !m->workers.empty()) {
        DeviceScope scope(m->workers[0].device);
        (void)hipFree(m->stage0);
    }
    delete m;
}

int rt_multi_get_info(const rt_multi *m, rt_multi_info *out) {
    if (!m || !out) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(out, 0, sizeof *out);
    out->n_workers = (uint32_t)m->workers.size();
    out->n_devices = (uint32_t)m->distinct.size();
    out->gather = m->gather;
    out->rccl_ranks = (uint32_t)m->comms.size();
    out->band_rows = m->last_plan.band_rows;
    out->n_units = m->last_plan.n_units;
    std::snprintf(out->note, sizeof out->note, "%s", m->gather == RT_GATHER_RCCL ? g_rccl.path.c_str() : m->gather_note.c_str());
    return RT_OK;
}

int rt_multi_render_device(rt_multi *m, const rt_camera_desc *camera, const rt_render_opts *opts, void *frame_device, rt_stats *stats) {
    Plan plan;
    int rc = make_plan(m, camera, opts, true, plan);
    if (rc != RT_OK) return rc;
    if (!frame_device) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "frame_device must not be NULL");
    if (plan.stats && !stats) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "collect_stats needs a stats pointer");
    if (stats) std::memset(stats, 0, sizeof *stats);
    const size_t packed_bytes = plan.unit_off[plan.n_units];
    bool peers = false;
    for (Worker &w : m->workers)
        if (!w.on_first) {
            DeviceScope scope(w.device);
            hipError_t e = grow_device(w, packed_bytes);
            if (e != hipSuccess) return fail_hip(e, "allocating a GPU's buffer of packed units");
            peers = true;
        }
    if (peers && packed_bytes > m->stage0_cap) {
        DeviceScope scope(m->workers[0].device);
        if (m->stage0) (void)hipFree(m->stage0);
        m->stage0 = nullptr;
        m->stage0_cap = 0;
        hipError_t e = hipMalloc((void **)&m->stage0, packed_bytes);
        if (e != hipSuccess) return fail_hip(e, "allocating the staging memory for other GPUs' units");
        m->stage0_cap = packed_bytes;
    }
    m->job = Job{};
    m->job.camera = *camera;
    m->job.plan = plan;
    m->job.frame_device = static_cast<char *>(frame_device);
    return run_all(m, stats);
}

int rt_multi_render(rt_multi *m, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb, rt_stats *stats) {
    Plan plan;
    int rc = make_plan(m, camera, opts, false, plan);
    if (rc != RT_OK) return rc;
    if (!rgb) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "rgb must not be NULL");
    if (plan.stats && !stats) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "collect_stats needs a stats pointer");
    if (stats) std::memset(stats, 0, sizeof *stats);
    const size_t packed_bytes = plan.unit_off[plan.n_units];
    // Every GPU sends its units to the host over its own PCIe link: nothing travels between GPUs for a host frame.
    for (Worker &w : m->workers) {
        DeviceScope scope(w.device);
        hipError_t e = grow_device(w, packed_bytes);
        if (e != hipSuccess) return fail_hip(e, "allocating a GPU's buffer of packed units");
    }
    if (packed_bytes > m->h_cap) {
        if (m->h_stage) (void)hipHostFree(m->h_stage);
        m->h_stage = nullptr;
        m->h_cap = 0;
        hipError_t e = hipHostMalloc((void **)&m->h_stage, packed_bytes, hipHostMallocPortable);
        if (e != hipSuccess) return fail_hip(e, "allocating pinned staging memory");
        m->h_cap = packed_bytes;
    }
    m->job = Job{};
    m->job.camera = *camera;
    m->job.plan = plan;
    m->job.stage = m->h_stage;
    m->job.rgb = reinterpret_cast<char *>(rgb);
    return run_all(m, stats);
}

int rt_multi_units_taken(const rt_multi *m, uint32_t *per_worker, uint32_t n) {
    if (!m || (!per_worker && n)) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t k = 0; k < n; ++k) per_worker[k] = k < m->last_taken.size() ? m->last_taken[k] : 0u;
    return RT_OK;
}

int rt_multi_worker_ms(const rt_multi *m, double *per_worker, uint32_t n) {
    if (!m || (!per_worker && n)) return rt::capi_set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t k = 0; k < n; ++k) per_worker[k] = k < m->last_ms.size() ? m->last_ms[k] : 0.0;
    return RT_OK;
}

}  // extern "C"
