// Canvas quantiser and P3 writer (rendering/canvas.rs:44-90,116-130 of the reference), host side.
//
// The text is byte-identical to Canvas::to_ppm_string: header "P3\n{w} {h}\n255\n"; one text line per image row,
// values separated by single spaces; a row whose joined text is longer than 70 characters is re-flowed greedily (a
// value moves to a new line when the running count of value lengths plus one space each would exceed 70); every
// image row starts on a new line; the file ends with one extra "\n".  The reference re-flows with a quadratic
// `Vec::remove(0)` loop; this writer is linear in the number of values, and rows are written by up to eight threads.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_rs_amd.h"

namespace rt {

// canvas.rs:123-130: (value * 255.0).round() as u32, then min(255).  `round` is half-away-from-zero; the Rust cast
// saturates and maps NaN to 0.  Without the library call: for x > 0 the nearest integer, halves upwards, is trunc(x)
// plus one when the fraction -- exact, x being far below 2^52 -- is at least a half.
static inline uint32_t quantize(double value) {
    const double x = value * 255.0;
    if (!(x >= 0.5)) return 0;  // rounds to 0 or below: small values, negatives, -0.0 and NaN
    if (x >= 254.5) return 255;
    const uint32_t q = (uint32_t)x;
    return q + (x - (double)q >= 0.5 ? 1u : 0u);
}

namespace {

// The decimal text of 0..255 and its length.
struct Decimals {
    char text[256][4];
    uint8_t len[256];
    Decimals() {
        for (int q = 0; q < 256; ++q) len[q] = (uint8_t)std::snprintf(text[q], sizeof text[q], "%d", q);
    }
};
const Decimals &decimals() {
    static const Decimals d;
    return d;
}

// One image row as text (canvas.rs:57-84) from its 8-bit values; returns the number of characters written (dst == nullptr:
// only counted).  Values are separated by single spaces; a row whose joined text is longer than 70 characters is
// re-flowed greedily: a value moves to a new line when the running count of value lengths plus one space each would
// exceed 70.
size_t row_text(const uint8_t *q, size_t n, char *dst) {
    const Decimals &d = decimals();
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += d.len[q[i]] + 1u;
    if (!dst) return total + (n ? 0 : 1);  // values, separators between them, the newline (an empty row is a lone newline)
    if (total) total -= 1;
    const bool split = total > 70;  // MAX_LINE_LENGTH, canvas.rs:14,66
    char *p = dst;
    size_t line_length = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned wl = d.len[q[i]];
        if (i) {
            char sep = ' ';
            if (split) {
                line_length += wl + 1u;
                if (line_length > 70) {
                    sep = '\n';
                    line_length = wl + 1u;
                }
            }
            *p++ = sep;
        } else {
            line_length = wl + 1u;
        }
        // (four bytes at once where the row's own next token overwrites the excess; the row's last token exactly: what
        // follows it belongs to the next row, which another thread may be writing)
        std::memcpy(p, d.text[q[i]], i + 1 < n ? 4 : wl);
        p += wl;
    }
    *p++ = '\n';
    return (size_t)(p - dst);
}

unsigned writer_threads(size_t values) {
    if (values < (1u << 20)) return 1;
    unsigned n = std::thread::hardware_concurrency();
    return n == 0 ? 1 : n > 8 ? 8 : n;
}

}  // namespace

// Two passes, both over rows in parallel on large frames: quantise and measure every row's text (PpmPlan), then write
// each row at its place (rows have text of different lengths, so a row's place is the sum of the lengths before it).
namespace {

struct PpmPlan {
    char head[64];
    size_t head_len = 0, row_vals = 0;
    uint32_t height = 0;
    unsigned n_threads = 1;
    std::vector<uint8_t> q;   // the frame as 8-bit values
    std::vector<size_t> at;   // where row y's text starts; at[height]: where the closing newline goes
    size_t size() const { return at[height] + 1; }

    template <class Body>
    void in_parallel(Body &&body) const {
        if (n_threads == 1) return body(0u, height);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back(body, (uint32_t)((uint64_t)height * t / n_threads), (uint32_t)((uint64_t)height * (t + 1) / n_threads));
        for (std::thread &t : pool) t.join();
    }

    PpmPlan(const double *rgb, uint32_t width, uint32_t h) : height(h) {
        head_len = (size_t)std::snprintf(head, sizeof head, "P3\n%u %u\n255\n", width, h);
        row_vals = (size_t)width * 3;
        q.resize(row_vals * height);
        at.assign((size_t)height + 1, 0);
        n_threads = writer_threads(q.size());
        in_parallel([&](uint32_t y0, uint32_t y1) {
            for (uint32_t y = y0; y < y1; ++y) {
                const double *row = rgb + (size_t)y * row_vals;
                uint8_t *qr = q.data() + (size_t)y * row_vals;
                for (size_t i = 0; i < row_vals; ++i) qr[i] = (uint8_t)quantize(row[i]);
                at[y + 1] = row_text(qr, row_vals, nullptr);
            }
        });
        at[0] = head_len;
        for (uint32_t y = 0; y < height; ++y) at[y + 1] += at[y];
    }

    void write(char *dst) const {  // size() characters
        std::memcpy(dst, head, head_len);
        in_parallel([&](uint32_t y0, uint32_t y1) {
            for (uint32_t y = y0; y < y1; ++y) row_text(q.data() + (size_t)y * row_vals, row_vals, dst + at[y]);
        });
        dst[at[height]] = '\n';
    }
};

}  // namespace

void canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, std::string &out) {
    const PpmPlan plan(rgb, width, height);
    out.resize(plan.size());
    plan.write(&out[0]);
}

// The same into a caller's buffer of `cap` characters: returns the length of the whole text; writes it only when it
// fits (a NULL buffer or cap == 0 asks for the length alone, which costs the first pass only).
size_t canvas_to_ppm_buffer(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap) {
    const PpmPlan plan(rgb, width, height);
    if (out && cap >= plan.size()) {
        plan.write(out);
    } else if (out && cap) {  // a buffer too small gets the head of the text, as before
        std::string whole(plan.size(), '\0');
        plan.write(&whole[0]);
        std::memcpy(out, whole.data(), cap);
    }
    return plan.size();
}

}  // namespace rt

extern "C" {

size_t rt_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap) {
    if (!rgb && (size_t)width * height) return 0;
    return rt::canvas_to_ppm_buffer(rgb, width, height, out, cap);
}

int rt_canvas_write_ppm(const double *rgb, uint32_t width, uint32_t height, const char *path) {
    if (!path || (!rgb && (size_t)width * height)) return RT_ERR_INVALID_ARGUMENT;
    std::string s;
    rt::canvas_to_ppm(rgb, width, height, s);
    FILE *f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    size_t n = std::fwrite(s.data(), 1, s.size(), f);
    int rc = std::fclose(f);
    return (n == s.size() && rc == 0) ? RT_OK : RT_ERR_IO;
}

int rt_canvas_quantize(const double *rgb, size_t n, uint8_t *out) {
    if ((!rgb || !out) && n) return RT_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)rt::quantize(rgb[i]);
    return RT_OK;
}

}  // extern "C"
