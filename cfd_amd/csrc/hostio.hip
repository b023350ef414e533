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
#include <cstdlib>
#include <cstring>
#include <locale.h>

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


// ---------------------------------------------------------------------------------------------------------------------
// The reverse direction: two named columns of a numeric CSV as doubles, parsed as Python's float() parses them
// (the reference's ingest: utils.py:72-83, csv.DictReader + float(row[col])).  float() is correctly rounded, and so is
// glibc's strtod: the same double for every decimal spelling (CPU-tested on 200 000 values against a DictReader +
// float() restatement).  Why here: pandas' default parser is NOT correctly rounded, and its round-trip parser (the same
// strtod) holds the GIL -- a folder read by 16 threads took 0.8 s PER CASE on the GPU box, 86 s of a 100-case training
// script whose epoch is 0.6 s.  This call runs without the GIL (ctypes releases it), ~3 ms a case.
// Anything this plain parser does not read exactly as csv + float() would (quotes, '_' digit separators, hex floats,
// empty or malformed fields, a missing column) returns -2: the caller falls back to pandas' round-trip parser.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
locale_t c_locale() {
  static locale_t loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
  return loc;
}

// one field [b, e) -> double as float() would; false: not a spelling this parser vouches for
inline bool parse_field(const char *b, const char *e, double &v) {
  while (b < e && (*b == ' ' || *b == '\t')) ++b;
  while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
  if (b == e || e - b > 63) return false;
  char tmp[64];
  for (const char *c = b; c < e; ++c) {
    const char ch = *c;
    const bool ok = (ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E' ||
                    ch == 'n' || ch == 'a' || ch == 'i' || ch == 'f' || ch == 'N' || ch == 'A' || ch == 'I' || ch == 'F' ||
                    ch == 't' || ch == 'y' || ch == 'T' || ch == 'Y';
    if (!ok) return false;                              // quotes, '_', 'x', anything else: not ours
    tmp[c - b] = ch;
  }
  tmp[e - b] = 0;
  char *end = nullptr;
  v = strtod_l(tmp, &end, c_locale());
  return end == tmp + (e - b);
}
}  // namespace

extern "C" int64_t cfd_csv_parse_columns(const char *text, int64_t len, const char *name_a, const char *name_b,
                                         double *out_a, double *out_b, int64_t cap) {
  CFD_REQUIRE(text && name_a && name_b && out_a && out_b && len >= 0 && cap >= 0, "csv_parse_columns: bad arguments");
  const char *p = text, *end = text + len;
  // header: positions of the two columns
  const char *eol = (const char *)memchr(p, '\n', end - p);
  if (!eol) return -2;
  int col_a = -1, col_b = -1, ncol = 0;
  {
    const char *f = p;
    while (f <= eol) {
      const char *g = f;
      while (g < eol && *g != ',') ++g;
      const char *h = g;
      while (h > f && (h[-1] == '\r')) --h;
      const size_t n = h - f;
      if (memchr(f, '"', n)) return -2;
      if (n == strlen(name_a) && memcmp(f, name_a, n) == 0 && col_a < 0) col_a = ncol;
      if (n == strlen(name_b) && memcmp(f, name_b, n) == 0 && col_b < 0) col_b = ncol;
      ++ncol;
      f = g + 1;
      if (g == eol) break;
    }
  }
  if (col_a < 0 || col_b < 0) return -2;
  int64_t rows = 0;
  p = eol + 1;
  while (p < end) {
    eol = (const char *)memchr(p, '\n', end - p);
    if (!eol) eol = end;
    if (eol == p || (eol == p + 1 && *p == '\r')) { p = eol + 1; continue; }   // (csv.DictReader skips blank lines)
    if (rows >= cap) return -2;                          // more rows than the caller expects: let pandas report it
    const char *f = p;
    int c = 0;
    for (;;) {                                           // every field of the row
      const char *g = f;
      while (g < eol && *g != ',') {
        if (*g == '"') return -2;                        // quoting: the csv module's business
        ++g;
      }
      if (c == col_a || c == col_b) {
        double v;
        if (!parse_field(f, g, v)) return -2;
        if (c == col_a) out_a[rows] = v;
        if (c == col_b) out_b[rows] = v;                 // (name_a == name_b is allowed)
      }
      ++c;
      if (g >= eol) break;
      f = g + 1;
    }
    if (c != ncol) return -2;                            // short or long row: DictReader treats those differently
    ++rows;
    p = eol + 1;
  }
  return rows;
}
