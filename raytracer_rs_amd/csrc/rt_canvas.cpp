// Canvas quantiser and P3 writer (rendering/canvas.rs:44-90,116-130 of the reference), host side.
//
// The text is byte-identical to Canvas::to_ppm_string: header "P3\n{w} {h}\n255\n"; one text line per image row,
// values separated by single spaces; a row whose joined text is longer than 70 characters is re-flowed greedily (a
// value moves to a new line when the running count of value lengths plus one space each would exceed 70); every
// image row starts on a new line; the file ends with one extra "\n".  The reference re-flows with a quadratic
// `Vec::remove(0)` loop; this writer is linear in the number of values.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/raytracer_rs_amd.h"

namespace rt {

// canvas.rs:123-130: (value * 255.0).round() as u32, then min(255).  `round` is half-away-from-zero; the Rust cast
// saturates and maps NaN to 0.
static inline uint32_t quantize(double value) {
    double r = std::round(value * 255.0);
    if (!(r > 0.0)) return 0;  // negatives, -0.0 and NaN
    if (r >= 255.0) return 255;
    return (uint32_t)r;
}

static inline int put_u8(char *dst, uint32_t q) {
    if (q >= 100) {
        dst[0] = (char)('0' + q / 100);
        dst[1] = (char)('0' + (q / 10) % 10);
        dst[2] = (char)('0' + q % 10);
        return 3;
    }
    if (q >= 10) {
        dst[0] = (char)('0' + q / 10);
        dst[1] = (char)('0' + q % 10);
        return 2;
    }
    dst[0] = (char)('0' + q);
    return 1;
}

void canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, std::string &out) {
    char head[64];
    int hn = std::snprintf(head, sizeof head, "P3\n%u %u\n255\n", width, height);
    const size_t row_vals = (size_t)width * 3;
    out.clear();
    out.reserve((size_t)hn + (size_t)height * (row_vals * 4 + 1) + 1);
    out.append(head, (size_t)hn);
    std::string line;
    line.reserve(row_vals * 4 + 2);
    for (uint32_t y = 0; y < height; ++y) {
        const double *row = rgb + (size_t)y * row_vals;
        size_t total = 0;
        for (size_t i = 0; i < row_vals; ++i) {
            uint32_t q = quantize(row[i]);
            total += (q >= 100 ? 3 : q >= 10 ? 2 : 1) + 1;
        }
        if (total) total -= 1;
        const bool split = total > 70;  // MAX_LINE_LENGTH, canvas.rs:14,66
        line.clear();
        size_t line_length = 0;
        char tok[4];
        for (size_t i = 0; i < row_vals; ++i) {
            int wl = put_u8(tok, quantize(row[i]));
            if (i) {
                if (split) {
                    line_length += (size_t)wl + 1;
                    if (line_length > 70) {
                        line.push_back('\n');
                        line_length = (size_t)wl + 1;
                    } else {
                        line.push_back(' ');
                    }
                } else {
                    line.push_back(' ');
                }
            } else {
                line_length = (size_t)wl + 1;
            }
            line.append(tok, (size_t)wl);
        }
        line.push_back('\n');
        out.append(line);
    }
    out.push_back('\n');
}

}  // namespace rt

extern "C" {

size_t rt_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap) {
    if (!rgb && (size_t)width * height) return 0;
    std::string s;
    rt::canvas_to_ppm(rgb, width, height, s);
    if (out && cap) std::memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
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
