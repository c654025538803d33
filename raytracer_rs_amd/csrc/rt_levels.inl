// Camera::render (rendering/camera.rs:56-67) level by level, for scenes with groups.  Included by rt_device.hip inside
// namespace rt, after the traversal, shading and tile-store code it shares with render_kernel.
//
// render_kernel keeps a pixel's pending reflection and refraction rays on a per-lane stack and walks them in the
// pixel's own wave, round after round; deep in the ray trees the 64 lanes of a wave hold rays of different
// generations and kinds, the wave walks the union of their paths through the node tables, and the state it carries
// across a traversal no longer fits the registers of five waves per SIMD.  Here a launch renders ONE level of every
// ray tree:
//
//   level 0      one wave per 8x8 pixel tile (tiles taken in 8x8 blocks, so that consecutive waves are neighbours in
//                both directions): ray through the pixel, closest hit, shading, shadow ray;
//   level k > 0  one wave per 64 consecutive rays of the queue the level before wrote.
//
// A wave writes the reflection and refraction rays it spawns into slots of its own (64 per kind) and counts them per
// kind; an exclusive scan of the counts numbers the next level's rays kind after kind, each kind in the order of the
// parent waves, and a stable sort then groups them by (kind, octant of the direction, leaf the ray leaves): a wave's 64
// rays are of one kind, start on one object, travel in one octant (the node tables are ordered front to back per
// octant) and, within that, come from neighbouring pixels.  What each step bought, 8K frame of the generated 10 001-leaf
// scene / 4K frame of chapter_14: rays appended in the order in which waves happen to finish (the textbook wavefront
// queue) 3-4x slower at the deeper levels; parent-wave order 63 ms / 4.5 ms; octant bins 52.7 ms / 5.4 ms (chapter_14's
// small mirror balls scatter rays, octants alone cost it screen locality); sorted by octant and leaf 47.9 ms / 3.6 ms.
// Keys with a few bits of direction below the leaf: chapter_14 -1.5 %, generated scene +0.5 %; direction above the leaf:
// +10 % / +5 %.  The numbering is a property of the view (scene, camera, rows, budget), not of the frame: a view
// rendered twice in a row reuses it, and its frames launch nothing but the level kernels and the sums.
//
// Sums.  Each ray stores its term of its pixel's colour (path weight x surface colour, as in radiance()) under its own
// number and remembers which of its wave's children are its own; when the last level is done, one pass per level,
// deepest first, adds to every ray the sums of its two children -- reflection, then refraction: the order of
// world.rs:107-110.  A pixel's colour is therefore a function of its own ray tree alone: it does not depend on the
// launch, the rows rendered with it, or the order in which waves run.
//
// Launch order.  A view's first frame also records what every wave of every level cost (shader clocks); later frames
// of the view launch a level's waves by the quarter-octave of that cost, most expensive first and otherwise in their own
// order, as render_kernel launches its tiles by estimated cost: the under-occupied end of every launch shrinks (4K
// chapter_14 3.63 -> 3.41 ms, generated scene at 1080p 6.2 -> 4.7 ms, at 8K 48.2 -> 46.9 ms).
// Five waves per SIMD (96 VGPRs, ten to twenty of the traversal's values spilled).  The generated scene at 8K: 55.9 ms
// at five, 57.2 at six, 67.4 at seven (and, before rays were binned by octant, 63.8 / 62.7 at five / six, 68.9 at four).
// Shadow traversals stop as soon as every ray of the wave has its occluder.  In render_kernel that test costs more than
// it saves (a wave's hits lie on many objects); here a wave's rays left one object in one octant, their hits lie
// together, and so do their shadow rays: chapter_14 at 4K -1 %, the generated scene at 8K -3 %.
#ifndef RT_LEVEL_SHADOW_EXIT
#define RT_LEVEL_SHADOW_EXIT true
#endif
#ifndef RT_LEVEL_WAVES
#define RT_LEVEL_WAVES 5
#endif
namespace {

constexpr unsigned NO_CHILD = 0xffu;

struct LevelIO {
    // the rays of this level (unused at level 0)
    const double *in_f = nullptr;       // eight doubles per slot, one 64-byte line: origin, direction, weight, remaining budget
    const uint32_t *in_index = nullptr; // ray number -> slot
    uint32_t n_rays = 0;
    // what it writes
    double *out_f = nullptr;            // 128 slots per wave: kind * 64 * waves + wave * 64 + k
    uint32_t *out_counts = nullptr;     // [2][waves + 1]: children per kind (reflection 0, refraction 1) and wave;
                                        // nullptr: not wanted (the numbering of this view's rays is in place already)
    uint32_t *out_keys = nullptr;       // per slot of out_f: what the next level's rays are sorted by (nullptr: not wanted)
    uint32_t leaf_bits = 0;             // of a key: (kind * 8 + direction octant) << leaf_bits | leaf
    uint32_t *out_cost = nullptr;       // per wave: the shader clocks it took (nullptr: not wanted)
    const uint32_t *wave_order = nullptr; // which wave's rays the k-th workgroup takes (nullptr: its own)
    uint32_t waves = 0;
    double *terms = nullptr;            // three doubles per ray
    uint8_t *links = nullptr;           // two arrays of 64 * waves bytes: k of the ray's reflection / refraction, or NO_CHILD
    uint32_t blocks_x = 0;              // level 0: blocks of 8x8 tiles per row of blocks
};

// Level 0: which tile wave `wave` renders.  Tiles go in blocks of 8x8; a block on the right or lower edge may hold
// tiles outside the frame.
__device__ __forceinline__ bool level0_tile(unsigned wave, unsigned blocks_x, unsigned tiles_x, unsigned tiles_y, unsigned &tile) {
    const unsigned block = wave >> 6, within = wave & 63u;
    const unsigned tx = (block % blocks_x) * 8 + (within & 7u), ty = (block / blocks_x) * 8 + (within >> 3);
    tile = ty * tiles_x + tx;
    return tx < tiles_x && ty < tiles_y;
}

template <int NEST, bool TIES, bool FIRST>
__global__ __launch_bounds__(64, RT_LEVEL_WAVES) void level_kernel(DeviceScene scene, RtCameraDev cam, RowMap rows, int max_depth, LevelIO io) {
    const unsigned lane = threadIdx.x & 63;
    const long long began = io.out_cost ? clock64() : 0ll;
    const unsigned wave = io.wave_order ? (unsigned)__builtin_amdgcn_readfirstlane((int)io.wave_order[blockIdx.x]) : blockIdx.x;
    SceneK S = scene_k<NEST, false>(scene, max_depth);
    Ctr<false> ctr;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    double w = 1.0;
    int remaining = max_depth;
    bool have;
    if constexpr (FIRST) {
        const unsigned tiles_x = (cam.hsize + 7) >> 3, tiles_y = (rows.n_rows + 7) >> 3;
        unsigned tile;
        const bool tile_ok = level0_tile(wave, io.blocks_x, tiles_x, tiles_y, tile);
        const unsigned x = (tile % tiles_x) * 8 + (lane & 7);
        const unsigned r = (tile / tiles_x) * 8 + (lane >> 3);
        const unsigned y = image_row(rows, r);
        have = tile_ok && x < cam.hsize && r < rows.n_rows && y < rows.row_end;
        if (have) ray_for_pixel(cam, x, y, ro, rd);
    } else {
        const unsigned i = wave * 64 + lane;
        have = i < io.n_rays;
        if (have) {
            const unsigned slot = io.in_index[i];
            const double4 *f = reinterpret_cast<const double4 *>(io.in_f + (size_t)slot * 8);
            const double4 a = f[0], b = f[1];
            ro = mk(a.x, a.y, a.z);
            rd = mk(a.w, b.x, b.y);
            w = b.z;
            remaining = __double2loint(b.w);
        }
    }
    if (FIRST && __ballot(have) == 0ull) {  // a tile outside the frame (the launch is padded to whole blocks of tiles)
        const size_t n = (size_t)io.waves * 64, i = (size_t)wave * 64 + lane;
        if (io.out_counts && lane < 2u) io.out_counts[lane * (io.waves + 1u) + wave] = 0u;
        if (io.out_cost && lane == 0u) io.out_cost[wave] = 0u;
        io.terms[3 * i] = 0.0, io.terms[3 * i + 1] = 0.0, io.terms[3 * i + 2] = 0.0;
        io.links[i] = (uint8_t)NO_CHILD, io.links[n + i] = (uint8_t)NO_CHILD;
        return;
    }
    // The wave's children of one kind, packed into its 64 slots for that kind; returns the lane's k (NO_CHILD if none).
    // `leaf`: the object the ray leaves.
    auto append = [&](bool want, unsigned kind, V3 o, V3 d, double pw, int prem, int leaf) -> unsigned {
        const unsigned long long m = __ballot(want);
        if (io.out_counts && lane == 0u) io.out_counts[kind * (io.waves + 1u) + wave] = (unsigned)__popcll(m);
        if (!want) return NO_CHILD;
        const unsigned k = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const size_t slot = (size_t)kind * 64 * io.waves + (size_t)wave * 64 + k;
        double4 *f = reinterpret_cast<double4 *>(io.out_f + slot * 8);
        f[0] = make_double4(o.x, o.y, o.z, d.x);
        f[1] = make_double4(d.y, d.z, pw, __hiloint2double(0, prem));
        if (io.out_keys) {
            // (bit a of the octant set: towards -a, as in DeviceScene::oct_tables)
            const unsigned oct = (d.x < 0.0 ? 1u : 0u) | (d.y < 0.0 ? 2u : 0u) | (d.z < 0.0 ? 4u : 0u);
            io.out_keys[slot] = ((kind * 8u + oct) << io.leaf_bits) | (unsigned)leaf;
        }
        return k;
    };

    const V3 light_pos = mk(S.light.position[0], S.light.position[1], S.light.position[2]);
    const Closest<TIES> best = trace<NEST, false, TIES>(S, ro, rd, have, (double)INFINITY, ctr);
    const bool shade = have && best.leaf >= 0 && S.light.has_light;  // else black: world.rs:124, world.rs:86-89
    V3 term_lit = mk(0.0, 0.0, 0.0), term = mk(0.0, 0.0, 0.0);
    HitShade h;
    h.reflect = h.refract = false;
    h.own_shadow = false;
    h.over = h.under = ro, h.normal = h.refract_dir = h.shadow_dir = rd;
    h.w_reflect = h.w_refract = h.light_distance = 0.0;
    if (shade) {
        shade_hit<false, false, RT_OWN_SHADOW>(S.light, light_pos, S.surfaces + best.leaf, ro, rd, best.t, w, remaining, h, ctr);
        term_lit = h.lit * w;
        term = h.shadowed * w;
    }
    const unsigned k_reflect = append(shade && h.reflect, 0u, h.over, reflect(rd, h.normal), h.w_reflect, remaining - 1, best.leaf);
    const unsigned k_refract = append(shade && h.refract, 1u, h.under, h.refract_dir, h.w_refract, remaining - 1, best.leaf);
    const bool walk = shade && !h.own_shadow;  // (a hit known to lie in its own object's shadow needs no shadow ray walked)
    if (__ballot(walk) != 0ull) {
        const bool in_shadow = occluded<NEST, false, RT_LEVEL_SHADOW_EXIT>(S, h.over, h.shadow_dir, walk, h.light_distance, ctr);  // world.rs:145-148
        if (walk && !in_shadow) term = term_lit;
    }
    const size_t n = (size_t)io.waves * 64, i = (size_t)wave * 64 + lane;
    io.terms[3 * i] = term.x;
    io.terms[3 * i + 1] = term.y;
    io.terms[3 * i + 2] = term.z;
    io.links[i] = (uint8_t)k_reflect;
    io.links[n + i] = (uint8_t)k_refract;
    if (io.out_cost && lane == 0u) {
        const long long took = clock64() - began;
        io.out_cost[wave] = took > 0x7fffffffll ? 0x7fffffffu : (uint32_t)took;
    }
}

// The launch order of a level's waves: by the quarter-octave of what they cost the last time, most expensive first,
// waves of one class in their own order.  Keys for a stable sort, and the identity to sort along.
__global__ __launch_bounds__(256) void level_cost_key_kernel(const uint32_t *__restrict__ cost, uint32_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ ids) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cost[i] | 4u;
    const uint32_t e = 31u - (uint32_t)__clz((int)c);
    const uint32_t cls = (e << 2) | ((c >> (e - 2u)) & 3u);  // < 128
    keys[i] = 127u - cls;
    ids[i] = (uint32_t)i;
}

// Numbers the children of one level: child k of kind `kind` of parent wave p lies in slot kind * 64 * waves + p * 64 + k,
// and its number in the next level is the scanned count of (kind, p) plus k.  Writes both directions of the mapping.
// One wave per (parent, kind), a lane per slot: the counts are wave-uniform reads (scalar loads) and the writes
// contiguous; four waves to a workgroup.
__global__ __launch_bounds__(256) void level_index_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets, uint32_t waves,
                                                          uint32_t *__restrict__ slot_of_ray, uint32_t *__restrict__ ray_of_slot) {
    // (readfirstlane: the unit is the same for a wave's 64 lanes, and saying so keeps the counts in scalar loads)
    const unsigned unit = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));  // (parent, kind)
    if (unit >= 2u * waves) return;
    const unsigned kind = unit >= waves ? 1u : 0u, p = unit - kind * waves, k = threadIdx.x & 63u;
    const size_t at = (size_t)kind * (waves + 1u) + p;
    if (k < counts[at]) {
        const size_t slot = (size_t)kind * 64 * waves + (size_t)p * 64 + k;
        const uint32_t ray = offsets[at] + k;
        slot_of_ray[ray] = (uint32_t)slot;
        ray_of_slot[slot] = ray;
    }
}

// Sorting a level's rays: the key of every ray in the order of the numbering, and after the (stable) sort both
// directions of the mapping again.
__global__ __launch_bounds__(256) void level_key_gather_kernel(const uint32_t *__restrict__ slot_of_ray, const uint32_t *__restrict__ key_of_slot, uint32_t n,
                                                               uint32_t *__restrict__ keys) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = key_of_slot[slot_of_ray[i]];
}
__global__ __launch_bounds__(256) void level_renumber_kernel(const uint32_t *__restrict__ sorted_slots, uint32_t n, uint32_t *__restrict__ slot_of_ray,
                                                             uint32_t *__restrict__ ray_of_slot) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t slot = sorted_slots[i];
        slot_of_ray[i] = slot;
        ray_of_slot[slot] = (uint32_t)i;
    }
}

// Children before parents: a ray's sum = its own term + the sum of its reflection + the sum of its refraction
// (the order of world.rs:107-110).  `child` holds the finished sums of the next level.
__global__ __launch_bounds__(256) void level_sum_kernel(double *__restrict__ terms, const uint8_t *__restrict__ links, const uint32_t *__restrict__ ray_of_slot,
                                                        uint32_t waves, uint32_t n_rays, const double *__restrict__ child) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    const size_t n = (size_t)waves * 64;
    const size_t block = (i >> 6) * 64;
    const unsigned k0 = links[i], k1 = links[n + i];
    if (k0 == NO_CHILD && k1 == NO_CHILD) return;
    V3 sum = mk(terms[3 * i], terms[3 * i + 1], terms[3 * i + 2]);
    if (k0 != NO_CHILD) {
        const size_t j = 3 * (size_t)ray_of_slot[block + k0];
        sum = sum + mk(child[j], child[j + 1], child[j + 2]);
    }
    if (k1 != NO_CHILD) {
        const size_t j = 3 * (size_t)ray_of_slot[n + block + k1];
        sum = sum + mk(child[j], child[j + 1], child[j + 2]);
    }
    terms[3 * i] = sum.x;
    terms[3 * i + 1] = sum.y;
    terms[3 * i + 2] = sum.z;
}

// The sums of level 0 are the pixels: a pixel's own term plus the finished sums of its two children (none when the
// frame has a single level), then Canvas::write_pixel per tile, as render_kernel does it.
__global__ __launch_bounds__(64) void level_store_kernel(RtCameraDev cam, RowMap rows, const double *__restrict__ terms, const uint8_t *__restrict__ links,
                                                         const uint32_t *__restrict__ ray_of_slot, const double *__restrict__ child, uint32_t waves,
                                                         uint32_t blocks_x, double *__restrict__ rgb) {
    __shared__ double staging[192];
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x;
    const unsigned tiles_x = (cam.hsize + 7) >> 3, tiles_y = (rows.n_rows + 7) >> 3;
    unsigned tile;
    if (!level0_tile(wave, blocks_x, tiles_x, tiles_y, tile)) return;
    const unsigned x = (tile % tiles_x) * 8 + (lane & 7);
    const unsigned r = (tile / tiles_x) * 8 + (lane >> 3);
    const unsigned y = image_row(rows, r);
    const bool valid = x < cam.hsize && r < rows.n_rows && y < rows.row_end;
    const size_t i = (size_t)wave * 64 + lane;
    V3 acc = mk(terms[3 * i], terms[3 * i + 1], terms[3 * i + 2]);
    if (child) {
        const size_t n = (size_t)waves * 64, block = (size_t)wave * 64;
        const unsigned k0 = links[i], k1 = links[n + i];
        if (k0 != NO_CHILD) {
            const size_t j = 3 * (size_t)ray_of_slot[block + k0];
            acc = acc + mk(child[j], child[j + 1], child[j + 2]);
        }
        if (k1 != NO_CHILD) {
            const size_t j = 3 * (size_t)ray_of_slot[n + block + k1];
            acc = acc + mk(child[j], child[j + 1], child[j + 2]);
        }
    }
    store_tile(cam, rows, tile, lane, valid, x, r, y, acc, staging, rgb);
}

hipError_t reserve(ArenaBuffer &b, size_t bytes, bool frozen) {
    if (b.bytes >= bytes) return hipSuccess;
    if (frozen) return hipErrorNotReady;  // (the caller must not allocate)
    if (b.p) {
        hipError_t e = hipFree(b.p);  // (waits for the device: nothing in flight reads it afterwards)
        b.p = nullptr;
        b.bytes = 0;
        if (e != hipSuccess) return e;
    }
    const size_t grown = bytes + bytes / 8;  // a little room, so that a slightly larger frame does not allocate again
    hipError_t e = hipMalloc(&b.p, grown);
    if (e != hipSuccess) return e;
    b.bytes = grown;
    return hipSuccess;
}

template <int NEST, bool TIES>
hipError_t render_levels(const DeviceScene &scene, const RtCameraDev &camera, const RowMap &rows, int max_depth, double *rgb, LevelArena &A,
                         hipStream_t stream) {
#define RT_TRY(expr)                       \
    do {                                   \
        hipError_t rt_try_e = (expr);      \
        if (rt_try_e != hipSuccess) return rt_try_e; \
    } while (0)
    if (max_depth < 0 || max_depth >= LEVELS_MAX) return hipErrorInvalidValue;
    const unsigned tiles_x = (camera.hsize + 7) / 8, tiles_y = (rows.n_rows + 7) / 8;
    const unsigned blocks_x = (tiles_x + 7) / 8, blocks_y = (tiles_y + 7) / 8;
    if ((unsigned long long)blocks_x * blocks_y * 64 > 0x00ffffffull) return hipErrorInvalidValue;  // 2^24 waves: 2^31 queue slots
    if (!A.host_totals) {
        if (A.frozen) return hipErrorNotReady;
        RT_TRY(hipHostMalloc((void **)&A.host_totals, 2 * sizeof(uint32_t), 0));
    }
    const char *sort_env = getenv("RT_LEVELS_SORT");  // =0: rays stay in the order of their parent waves (tests compare the two)
    const bool sorted = !(sort_env && sort_env[0] == '0');
    const char *order_env = getenv("RT_LEVELS_ORDER");  // =0: a level's waves are launched in the order of their rays
    const bool ordered = !(order_env && order_env[0] == '0');
    uint32_t leaf_bits = 1;
    while (leaf_bits < 28 && (1u << leaf_bits) < (uint32_t)scene.n_leaves) ++leaf_bits;
    if ((uint32_t)scene.n_leaves > (1u << 28)) return hipErrorInvalidValue;
    // Seen this view before?  Then every level's ray count is known and nothing below waits for the GPU.
    LevelArena::Known *known = nullptr;
    const uint32_t view[5] = {rows.row_begin, rows.row_end, rows.n_rows, rows.band_rows, rows.band_pitch};
    const char *remember_env = getenv("RT_LEVELS_REMEMBER");  // =0: always ask the GPU (tests compare the two)
    const bool remember = !(remember_env && remember_env[0] == '0');
    for (uint32_t k = 0; remember && k < A.n_known; ++k)
        if (A.known[k].max_depth == max_depth && std::memcmp(&A.known[k].cam, &camera, sizeof camera) == 0 &&
            std::memcmp(A.known[k].rows, view, sizeof view) == 0) {
            known = &A.known[k];
            known->used = ++A.clock;
        }
    if (A.frozen && !known) return hipErrorNotReady;  // (learning a view's ray counts means waiting for the GPU)
    // Is a numbering of this view's rays still in place?  It is the same in every frame.
    // (Not on a handle through which a call has been captured into a graph: the graph holds the numbering kernels -- it
    // must, or replaying it after another view went through the handle would read that view's numbering -- and a replay
    // rewrites the handle's numbering whenever the caller chooses, unseen from here.  Such a handle numbers every frame.)
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream) RT_TRY(hipStreamIsCapturing(stream, &capture));
    if (capture != hipStreamCaptureStatusNone) A.capture_seen = true;
    const int32_t known_at = known ? (int32_t)(known - A.known) : -1;
    LevelArena::Numbering *N = nullptr;
    bool numbered = false;
    if (known && !A.capture_seen)
        for (LevelArena::Numbering &n : A.numberings)
            if (n.view == known_at && n.sorted == sorted) N = &n, numbered = true;
    if (!N) {
        // Where the new numbering goes: over an old one of the same view, else into an unused set of buffers, else over
        // the numbering unused for longest.  A caller that must not allocate needs buffers that are large enough: the
        // ones this view was numbered in when it was first rendered still are (buffers never shrink).
        auto fits = [&](const LevelArena::Numbering &n) {
            if (!A.frozen) return true;
            for (uint32_t level = 0; level + 1 < known->levels; ++level) {
                const size_t waves_l = level == 0 ? (size_t)blocks_x * blocks_y * 64 : ((size_t)known->level_rays[level] + 63) / 64;
                if (n.index[level].bytes < (size_t)known->level_rays[level + 1] * sizeof(uint32_t)) return false;
                if (n.ray_of_slot[level].bytes < waves_l * 128 * sizeof(uint32_t)) return false;
            }
            return true;
        };
        LevelArena::Numbering *same = nullptr, *unused = nullptr, *oldest = nullptr;
        for (LevelArena::Numbering &n : A.numberings) {
            if (!fits(n)) continue;
            if (known && n.view == known_at) same = same ? same : &n;
            else if (n.view < 0) unused = unused ? unused : &n;
            else if (!oldest || n.used < oldest->used) oldest = &n;
        }
        N = same ? same : unused ? unused : oldest;
        if (!N) return hipErrorNotReady;
        N->view = -1;  // (about to be overwritten)
    }
    // (What a numbering's waves cost is recorded with it -- unless the caller must not allocate and the buffers for the
    // launch orders are not there: then this numbering goes without.)
    bool record_order = ordered && !numbered;
    if (record_order && A.frozen)
        for (uint32_t level = 0; level < known->levels; ++level) {
            const size_t waves_l = level == 0 ? (size_t)blocks_x * blocks_y * 64 : ((size_t)known->level_rays[level] + 63) / 64;
            const size_t need = waves_l * sizeof(uint32_t);
            if (N->wave_order[level].bytes < need || A.wave_cost.bytes < need || A.sort_buf[0].bytes < need || A.sort_buf[1].bytes < need ||
                A.sort_buf[2].bytes < need)
                record_order = false;
        }
    // (Buffers of a numbering are only ever resized while it is being made, never while it is in use.)
    auto reserve_numbering = [&](ArenaBuffer &b, size_t need, int kind, int level) {
        size_t &cap = A.numbering_cap[kind][level];
        if (need > cap) cap = need;
        return reserve(b, A.frozen ? need : cap, A.frozen);
    };
    const size_t lds = lds_bytes<NEST, false>(64);
    unsigned waves = blocks_x * blocks_y * 64, n_rays = waves * 64, levels = 0;
    uint32_t level_waves[LEVELS_MAX] = {};
    for (int level = 0; level <= max_depth; ++level) {
        const int out = level & 1, in = out ^ 1;
        const bool last = level == max_depth;  // its rays have no budget left: nothing is appended
        RT_TRY(reserve(A.terms[level], (size_t)waves * 64 * 3 * sizeof(double), A.frozen));
        RT_TRY(reserve(A.links[level], (size_t)waves * 64 * 2, A.frozen));
        const size_t table = ((size_t)waves + 1) * 2;  // per kind: one count per wave and a last entry that stays zero
        if (table > 0x7fffffffull) return hipErrorInvalidValue;  // (the scan counts its items in an int)
        RT_TRY(reserve(A.counts, table * sizeof(uint32_t), A.frozen));
        if (!last) {
            RT_TRY(reserve(A.queue_f[out], (size_t)waves * 128 * 8 * sizeof(double), A.frozen));
            RT_TRY(reserve(A.offsets, table * sizeof(uint32_t), A.frozen));
            if (!numbered) RT_TRY(reserve_numbering(N->ray_of_slot[level], (size_t)waves * 128 * sizeof(uint32_t), 1, level));
            if (sorted && !numbered) RT_TRY(reserve(A.key_of_slot, (size_t)waves * 128 * sizeof(uint32_t), A.frozen));
        }
        LevelIO io;
        io.in_f = (const double *)A.queue_f[in].p;
        io.in_index = level == 0 ? nullptr : (const uint32_t *)N->index[level - 1].p;
        io.n_rays = n_rays;
        io.out_f = (double *)A.queue_f[out].p;
        io.out_counts = numbered ? nullptr : (uint32_t *)A.counts.p;
        io.out_keys = sorted && !numbered && !last ? (uint32_t *)A.key_of_slot.p : nullptr;
        io.leaf_bits = leaf_bits;
        if (record_order) {
            RT_TRY(reserve(A.wave_cost, (size_t)waves * sizeof(uint32_t), A.frozen));
            RT_TRY(reserve_numbering(N->wave_order[level], (size_t)waves * sizeof(uint32_t), 2, level));
            io.out_cost = (uint32_t *)A.wave_cost.p;
        }
        if (ordered && numbered && N->ordered) io.wave_order = (const uint32_t *)N->wave_order[level].p;
        io.waves = waves;
        io.terms = (double *)A.terms[level].p;
        io.links = (uint8_t *)A.links[level].p;
        io.blocks_x = blocks_x;
        level_waves[level] = waves;
        A.level_rays[level] = n_rays;
        levels = level + 1;
        uint32_t *counts = (uint32_t *)A.counts.p, *offsets = (uint32_t *)A.offsets.p;
        if (!last && !numbered) RT_TRY(hipMemsetAsync(counts, 0, table * sizeof(uint32_t), stream));  // (the last entry of either kind must read as zero)
        if (level == 0) level_kernel<NEST, TIES, true><<<dim3(waves), 64, lds, stream>>>(scene, camera, rows, max_depth, io);
        else level_kernel<NEST, TIES, false><<<dim3(waves), 64, lds, stream>>>(scene, camera, rows, max_depth, io);
        RT_TRY(hipGetLastError());
        if (record_order) {
            for (int k = 0; k < 3; ++k) RT_TRY(reserve(A.sort_buf[k], (size_t)waves * sizeof(uint32_t), A.frozen));
            uint32_t *keys_in = (uint32_t *)A.sort_buf[0].p, *keys_out = (uint32_t *)A.sort_buf[1].p, *ids = (uint32_t *)A.sort_buf[2].p;
            level_cost_key_kernel<<<dim3((waves + 255) / 256), 256, 0, stream>>>((const uint32_t *)A.wave_cost.p, waves, keys_in, ids);
            size_t sort_need = 0;
            RT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_need, keys_in, keys_out, ids, (uint32_t *)N->wave_order[level].p, (int)waves, 0, 7, stream));
            RT_TRY(reserve(A.scan_temp, sort_need, A.frozen));
            size_t sort_bytes = A.scan_temp.bytes;
            RT_TRY(hipcub::DeviceRadixSort::SortPairs(A.scan_temp.p, sort_bytes, keys_in, keys_out, ids, (uint32_t *)N->wave_order[level].p, (int)waves, 0, 7, stream));
        }
        if (last) break;
        if (numbered) {
            if ((uint32_t)level + 1 >= known->levels) break;
            n_rays = known->level_rays[level + 1];
            waves = (n_rays + 63) / 64;
            continue;
        }
        // number the children: kind after kind, each kind in the order of the parent waves
        size_t need = 0;
        RT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, counts, offsets, (int)table, stream));
        RT_TRY(reserve(A.scan_temp, need, A.frozen));
        size_t bytes = A.scan_temp.bytes;
        RT_TRY(hipcub::DeviceScan::ExclusiveSum(A.scan_temp.p, bytes, counts, offsets, (int)table, stream));
        unsigned long long children;
        if (known) {
            children = (uint32_t)level + 1 < known->levels ? known->level_rays[level + 1] : 0;
        } else {
            RT_TRY(hipMemcpyAsync(A.host_totals, offsets + table - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            RT_TRY(hipStreamSynchronize(stream));
            children = A.host_totals[0];
        }
        if (children == 0) break;
        if (children > 0x3fffffffull) return hipErrorOutOfMemory;
        RT_TRY(reserve_numbering(N->index[level], (size_t)children * sizeof(uint32_t), 0, level));
        level_index_kernel<<<dim3((2 * waves + 3) / 4), 256, 0, stream>>>(counts, offsets, waves, (uint32_t *)N->index[level].p,
                                                                          (uint32_t *)N->ray_of_slot[level].p);
        if (sorted) {  // ... then group them by (kind, octant, leaf), keeping that order within a group
            const unsigned n = (unsigned)children;
            for (int k = 0; k < 3; ++k) RT_TRY(reserve(A.sort_buf[k], (size_t)n * sizeof(uint32_t), A.frozen));
            uint32_t *keys_in = (uint32_t *)A.sort_buf[0].p, *keys_out = (uint32_t *)A.sort_buf[1].p, *vals_out = (uint32_t *)A.sort_buf[2].p;
            uint32_t *slot_of_ray = (uint32_t *)N->index[level].p;
            level_key_gather_kernel<<<dim3((n + 255) / 256), 256, 0, stream>>>(slot_of_ray, (const uint32_t *)A.key_of_slot.p, n, keys_in);
            size_t sort_need = 0;
            RT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_need, keys_in, keys_out, slot_of_ray, vals_out, (int)n, 0, (int)leaf_bits + 4, stream));
            RT_TRY(reserve(A.scan_temp, sort_need, A.frozen));
            size_t sort_bytes = A.scan_temp.bytes;
            RT_TRY(hipcub::DeviceRadixSort::SortPairs(A.scan_temp.p, sort_bytes, keys_in, keys_out, slot_of_ray, vals_out, (int)n, 0, (int)leaf_bits + 4, stream));
            level_renumber_kernel<<<dim3((n + 255) / 256), 256, 0, stream>>>(vals_out, n, slot_of_ray, (uint32_t *)N->ray_of_slot[level].p);
        }
        n_rays = (unsigned)children;
        waves = (n_rays + 63) / 64;
    }
    for (int level = (int)levels - 2; level >= 1; --level) {
        const unsigned w = level_waves[level];
        level_sum_kernel<<<dim3((A.level_rays[level] + 255) / 256), 256, 0, stream>>>((double *)A.terms[level].p, (const uint8_t *)A.links[level].p,
                                                                                     (const uint32_t *)N->ray_of_slot[level].p, w, A.level_rays[level],
                                                                                     (const double *)A.terms[level + 1].p);
    }
    level_store_kernel<<<dim3(level_waves[0]), 64, 0, stream>>>(camera, rows, (const double *)A.terms[0].p, (const uint8_t *)A.links[0].p,
                                                                levels > 1 ? (const uint32_t *)N->ray_of_slot[0].p : nullptr,
                                                                levels > 1 ? (const double *)A.terms[1].p : nullptr, level_waves[0], blocks_x, rgb);
    A.levels = levels;
    if (!numbered) N->sorted = sorted, N->ordered = record_order;
    N->used = ++A.clock;
    if (known) N->view = known_at;
    if (!known && remember) {  // remember this view's counts (in place of the entry unused for longest when all are taken)
        uint32_t at = A.n_known;
        if (at == LevelArena::KNOWN_MAX) {
            at = 0;
            for (uint32_t k = 1; k < A.n_known; ++k)
                if (A.known[k].used < A.known[at].used) at = k;
        } else {
            A.n_known++;
        }
        for (LevelArena::Numbering &n : A.numberings)
            if (n.view == (int32_t)at) n.view = -1;  // (the numbering of the view that is being forgotten)
        LevelArena::Known &e = A.known[at];
        std::memcpy(&e.cam, &camera, sizeof camera);
        std::memcpy(e.rows, view, sizeof view);
        e.max_depth = max_depth;
        e.levels = levels;
        std::memcpy(e.level_rays, A.level_rays, sizeof e.level_rays);
        e.used = ++A.clock;
        N->view = (int32_t)at;
    }
    return hipGetLastError();
#undef RT_TRY
}

}  // namespace

size_t LevelArena::device_bytes() const {
    size_t total = queue_f[0].bytes + queue_f[1].bytes + counts.bytes + offsets.bytes + scan_temp.bytes + key_of_slot.bytes + sort_buf[0].bytes + sort_buf[1].bytes +
                   sort_buf[2].bytes + wave_cost.bytes;
    for (const Numbering &n : numberings)
        for (int k = 0; k < LEVELS_MAX; ++k) total += n.index[k].bytes + n.ray_of_slot[k].bytes + n.wave_order[k].bytes;
    for (int k = 0; k < LEVELS_MAX; ++k) total += terms[k].bytes + links[k].bytes;
    return total;
}

void level_arena_release(LevelArena &A) {
    auto drop = [](ArenaBuffer &b) {
        if (b.p) (void)hipFree(b.p);
        b = ArenaBuffer();
    };
    for (int k = 0; k < 2; ++k) drop(A.queue_f[k]);
    drop(A.counts), drop(A.offsets), drop(A.scan_temp), drop(A.key_of_slot), drop(A.wave_cost);
    for (LevelArena::Numbering &n : A.numberings) {
        for (int k = 0; k < LEVELS_MAX; ++k) drop(n.index[k]), drop(n.ray_of_slot[k]), drop(n.wave_order[k]);
        n.view = -1;
    }
    for (auto &caps : A.numbering_cap)
        for (size_t &c : caps) c = 0;
    for (int k = 0; k < 3; ++k) drop(A.sort_buf[k]);
    for (int k = 0; k < LEVELS_MAX; ++k) drop(A.terms[k]), drop(A.links[k]);
    if (A.host_totals) (void)hipHostFree(A.host_totals);
    A.host_totals = nullptr;
    A.levels = 0;
    A.n_known = 0;
}

// Frames of more pixels than this are rendered in chunks of rows, one after the other through the same arena: the
// queues of a frame grow with its pixels (about 0.5 KB per pixel for a scene whose ray trees are full), a pixel's
// colour does not depend on what is rendered with it, and a chunk of this size still fills the GPU hundreds of times
// over.  48 Mpixel: a 7680x4320 frame is one chunk.  RT_LEVELS_CHUNK_PIXELS in the environment overrides (tests).
constexpr unsigned long long LEVELS_CHUNK_PIXELS = 48ull << 20;

hipError_t launch_render_levels(const DeviceScene &scene, const RtCameraDev &camera, const RowMap &rows_in, int max_depth, double *rgb,
                                LevelArena &arena, hipStream_t stream, hipEvent_t *marks) {
    if (rows_in.n_rows == 0 || camera.hsize == 0 || rows_in.band_rows == 0) return hipSuccess;
    if (scene.max_nesting == 0) return hipErrorInvalidValue;
    if (marks) {
        hipError_t e = hipEventRecord(marks[0], stream);
        if (e == hipSuccess) e = hipEventRecord(marks[1], stream);
        if (e != hipSuccess) return e;
    }
    unsigned long long limit = LEVELS_CHUNK_PIXELS;
    if (const char *env = getenv("RT_LEVELS_CHUNK_PIXELS")) {
        const unsigned long long v = strtoull(env, nullptr, 10);
        if (v) limit = v;
    }
    // chunks of packed rows [r0, r1): whole bands when the rows are bands of a frame, multiples of 64 rows otherwise
    const bool banded = rows_in.band_pitch != 0;
    const unsigned unit = banded ? rows_in.band_rows : 64u;
    unsigned long long chunk = limit / camera.hsize / unit * unit;
    if (chunk < unit) chunk = unit;
    hipError_t e = hipSuccess;
    for (unsigned long long r0 = 0; r0 < rows_in.n_rows && e == hipSuccess; r0 += chunk) {
        const unsigned r1 = (unsigned)std::min<unsigned long long>(rows_in.n_rows, r0 + chunk);
        RowMap rows = rows_in;
        rows.n_rows = r1 - (unsigned)r0;
        if (banded) {
            rows.row_begin = rows_in.row_begin + (unsigned)(r0 / rows_in.band_rows) * rows_in.band_pitch;
        } else {
            rows.row_begin = rows_in.row_begin + (unsigned)r0;
            rows.row_end = std::min(rows_in.row_end, rows_in.row_begin + r1);
            rows.band_rows = rows.n_rows;
        }
        double *out = rgb;
        if (!rows.in_place) {  // packed output: the chunk's rows follow the rows before them
            const size_t values = (size_t)r0 * camera.hsize * 3;
            if (rows.quantized) rows.quantized += values;
            else out += values;
        }
        rows.n_tiles = ((camera.hsize + 7) / 8) * ((rows.n_rows + 7) / 8);
        rows.tile_list = nullptr;
        const int build = scene.any_order != 0 && scene.cone_nodes == 0 ? 2 : scene.any_order != 0 ? 1 : 0;
#define RT_LEVELS(N, T) render_levels<N, T>(scene, camera, rows, max_depth, out, arena, stream)
        e = build == 2 ? RT_LEVELS(2, true) : build == 1 ? RT_LEVELS(1, true) : RT_LEVELS(1, false);
#undef RT_LEVELS
    }
    if (e == hipSuccess && marks) e = hipEventRecord(marks[2], stream);
    return e;
}
