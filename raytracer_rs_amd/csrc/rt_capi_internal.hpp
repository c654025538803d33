// Shared by the translation units that implement the C ABI.
#pragma once

#include <string>

#include "../../include/raytracer_rs_amd.h"

namespace rt {

// Records `msg` as the calling thread's last error (rt_last_error) and returns `code`.
int capi_set_error(int code, const std::string &msg);

// rt_render_device with the counting kernel ADDING to the handle's counters: nothing is cleared before and nothing is
// read back after, so the call does not synchronise with the host.  A caller that renders a frame in several launches
// brackets them with capi_counters_reset and capi_counters_read (the latter waits for `stream`).
int capi_render_device_counting(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb_device);
int capi_counters_reset(const rt_scene *scene, void *stream);
int capi_counters_read(const rt_scene *scene, void *stream, rt_stats *stats);

}  // namespace rt
