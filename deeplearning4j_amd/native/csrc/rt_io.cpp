// Native record parsing for the DataVec bridge: numeric CSV and SVMLight/LibSVM files to dense
// float matrices, threaded by line ranges. (Reference role: datavec-api CSVRecordReader /
// SVMLightRecordReader feeding RecordReaderDataSetIterator; those parse on the JVM one Writable at
// a time. Here a whole file becomes one contiguous fp32 block ready for a pinned-host -> HBM copy.)
#include "rt_common.h"

#include <cstdio>
#include <cstdlib>
#include <cmath>

namespace {
struct Parsed {
  std::vector<float> data;
  LongType rows = 0, cols = 0;
};

bool read_all(const char* path, std::string& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize((size_t)n);
  size_t got = n ? std::fread(&out[0], 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}

void line_starts(const std::string& s, std::vector<size_t>& starts) {
  size_t p = 0;
  while (p < s.size()) {
    size_t e = s.find('\n', p);
    if (e == std::string::npos) e = s.size();
    size_t l = e;
    while (l > p && (s[l - 1] == '\r' || s[l - 1] == ' ')) --l;
    if (l > p) starts.push_back(p);
    p = e + 1;
  }
}
}  // namespace

// Returns an opaque handle (or null). Non-numeric fields become NaN. quote char '"' is stripped.
RT_API void* dv_csv_parse(const char* path, char delim, int skip_lines) {
  std::string txt;
  if (!read_all(path, txt)) {
    rt::set_error(40, std::string("cannot read ") + path);
    return nullptr;
  }
  std::vector<size_t> starts;
  line_starts(txt, starts);
  if ((int)starts.size() <= skip_lines) {
    auto* p = new Parsed();
    return p;
  }
  starts.erase(starts.begin(), starts.begin() + skip_lines);
  // columns from the first line
  size_t e0 = txt.find('\n', starts[0]);
  if (e0 == std::string::npos) e0 = txt.size();
  LongType cols = 1;
  bool inq = false;
  for (size_t i = starts[0]; i < e0; ++i) {
    if (txt[i] == '"') inq = !inq;
    else if (txt[i] == delim && !inq) ++cols;
  }
  auto* out = new Parsed();
  out->rows = (LongType)starts.size();
  out->cols = cols;
  out->data.assign((size_t)(out->rows * cols), NAN);
  std::vector<int> bad((size_t)out->rows, 0);
  rt::parallel_for(0, out->rows, 2048, [&](LongType b, LongType e) {
    for (LongType r = b; r < e; ++r) {
      size_t p = starts[(size_t)r];
      size_t end = txt.find('\n', p);
      if (end == std::string::npos) end = txt.size();
      LongType c = 0;
      while (p <= end && c < cols) {
        size_t q = p;
        bool quoted = false;
        if (q < end && txt[q] == '"') {
          quoted = true;
          ++q;
        }
        char* stop = nullptr;
        float v = std::strtof(txt.c_str() + q, &stop);
        size_t fe = p;
        bool iq = false;
        while (fe < end && (txt[fe] != delim || iq)) {
          if (txt[fe] == '"') iq = !iq;
          ++fe;
        }
        if (stop != txt.c_str() + q) out->data[(size_t)(r * cols + c)] = v;
        (void)quoted;
        ++c;
        p = fe + 1;
      }
      if (c != cols) bad[(size_t)r] = 1;
    }
  });
  for (int b : bad)
    if (b) {
      rt::set_error(41, "csv: ragged rows (column count differs from first line)");
      break;
    }
  return out;
}

// SVMLight: "<label> idx:val idx:val ... # comment". Features are 1-based unless zero_based.
// Output matrix = [rows, 1 + num_features] with the label in column 0.
RT_API void* dv_svmlight_parse(const char* path, int num_features, int zero_based) {
  std::string txt;
  if (!read_all(path, txt)) {
    rt::set_error(40, std::string("cannot read ") + path);
    return nullptr;
  }
  std::vector<size_t> starts;
  line_starts(txt, starts);
  auto* out = new Parsed();
  out->rows = (LongType)starts.size();
  out->cols = num_features + 1;
  out->data.assign((size_t)(out->rows * out->cols), 0.0f);
  rt::parallel_for(0, out->rows, 2048, [&](LongType b, LongType e) {
    for (LongType r = b; r < e; ++r) {
      const char* p = txt.c_str() + starts[(size_t)r];
      const char* end = txt.c_str() + txt.size();
      const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
      if (nl) end = nl;
      char* stop;
      out->data[(size_t)(r * out->cols)] = std::strtof(p, &stop);
      p = stop;
      while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t')) ++p;
        if (p >= end || *p == '#') break;
        long idx = std::strtol(p, &stop, 10);
        if (stop == p || *stop != ':') break;
        p = stop + 1;
        float v = std::strtof(p, &stop);
        p = stop;
        long col = zero_based ? idx : idx - 1;
        if (col >= 0 && col < num_features) out->data[(size_t)(r * out->cols + 1 + col)] = v;
      }
    }
  });
  return out;
}

RT_API LongType dv_rows(void* h) { return ((Parsed*)h)->rows; }
RT_API LongType dv_cols(void* h) { return ((Parsed*)h)->cols; }
RT_API const float* dv_data(void* h) { return ((Parsed*)h)->data.data(); }
RT_API void dv_copy(void* h, float* dst) {
  auto* p = (Parsed*)h;
  std::memcpy(dst, p->data.data(), p->data.size() * sizeof(float));
}
RT_API void dv_free(void* h) { delete (Parsed*)h; }
