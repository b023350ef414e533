// Host-side text formatting of the resample's result table (no device code in this file).
//
// The reference ends process_csv_with_pixel_coordinates_nearest with `df_out.to_csv(output_csv, index=False)`
// (pixels_coords_testing.py:223-226): 32 768 rows x 5 fp64 columns per case as text.  Once the resample itself takes
// microseconds on the GPU, that pandas call is the whole preprocess time (~150 ms per case on the GPU box's host).  This
// writer produces the SAME BYTES: pandas turns each double into `repr(float)` -- the shortest decimal string that reads
// back to the same double, laid out by CPython's rule (float_repr_style 'short', format code 'r': exponent form when
// the decimal point position is <= -4 or > 16, at least two exponent digits, ".0" appended to integers) -- and NaN into
// the empty field.  std::to_chars yields the same shortest digit string; the layout below is CPython's.
#include <charconv>
#include <cmath>
#include <cstring>

#include "common.h"

namespace {

// writes repr(v) at p, returns the new end (at most 25 bytes)
inline char *py_repr(double v, char *p) {
  if (std::isnan(v)) return p;                       // pandas na_rep = ''
  if (std::signbit(v)) { *p++ = '-'; v = -v; }
  if (std::isinf(v)) { std::memcpy(p, "inf", 3); return p + 3; }
  char sci[40];
  auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
  char *e = sci;
  while (*e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (char *c = sci; c < e; ++c)
    if (*c != '.') digits[nd++] = *c;
  int ex = 0;
  for (char *c = e + 2; c < r.ptr; ++c) ex = ex * 10 + (*c - '0');
  if (e[1] == '-') ex = -ex;
  const int decpt = ex + 1;                           // position of the decimal point relative to the digit string
  if (decpt <= -4 || decpt > 16) {                    // exponent form: d[.ddd]e[+-]XX
    *p++ = digits[0];
    if (nd > 1) { *p++ = '.'; std::memcpy(p, digits + 1, nd - 1); p += nd - 1; }
    *p++ = 'e';
    *p++ = ex < 0 ? '-' : '+';
    int a = ex < 0 ? -ex : ex;
    if (a >= 100) { *p++ = char('0' + a / 100); a %= 100; }
    *p++ = char('0' + a / 10);
    *p++ = char('0' + a % 10);
  } else if (decpt <= 0) {                            // 0.000ddd
    *p++ = '0'; *p++ = '.';
    for (int i = 0; i < -decpt; ++i) *p++ = '0';
    std::memcpy(p, digits, nd); p += nd;
  } else if (decpt >= nd) {                           // ddd000.0
    std::memcpy(p, digits, nd); p += nd;
    for (int i = nd; i < decpt; ++i) *p++ = '0';
    *p++ = '.'; *p++ = '0';
  } else {                                            // dd.ddd
    std::memcpy(p, digits, decpt); p += decpt;
    *p++ = '.';
    std::memcpy(p, digits + decpt, nd - decpt); p += nd - decpt;
  }
  return p;
}

}  // namespace

extern "C" int64_t cfd_csv_format_rows(const double *rows, int64_t n_rows, int n_cols, char *out, int64_t cap) {
  CFD_REQUIRE(rows && out && n_rows >= 0 && n_cols > 0, "csv_format_rows: null pointer or bad shape (%lld x %d)",
              (long long)n_rows, n_cols);
  CFD_REQUIRE(cap >= n_rows * (int64_t)n_cols * 26, "csv_format_rows: output capacity %lld < %lld (26 bytes per value)",
              (long long)cap, (long long)(n_rows * (int64_t)n_cols * 26));
  char *p = out;
  for (int64_t i = 0; i < n_rows; ++i) {
    const double *r = rows + i * n_cols;
    for (int j = 0; j < n_cols; ++j) {
      p = py_repr(r[j], p);
      *p++ = (j + 1 < n_cols) ? ',' : '\n';
    }
  }
  return (int64_t)(p - out);
}
