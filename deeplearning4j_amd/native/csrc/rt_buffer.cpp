// DataBuffer host-side helpers: big-endian (de)serialisation of typed buffers and .npy IO.
//
// Reference: nd4j-api linalg/api/buffer/BaseDataBuffer.java:1852-1912,2060-2100 (Java
// DataOutputStream = big-endian elements; this is the coefficients.bin / updaterState.bin payload)
// and libnd4j cnpy (legacy/impl/cnpy.cpp — .npy/.npz). A ResNet-50 checkpoint is ~100 MB of floats
// (and models sized for 288 GB HBM are far larger), so the byte swap is threaded native code.
#include "rt_common.h"

#include <cstdio>
#include <cstdlib>

namespace {
template <typename T>
inline T bswap(T v);
template <>
inline uint16_t bswap(uint16_t v) { return __builtin_bswap16(v); }
template <>
inline uint32_t bswap(uint32_t v) { return __builtin_bswap32(v); }
template <>
inline uint64_t bswap(uint64_t v) { return __builtin_bswap64(v); }

template <typename T>
void swap_copy(const void* src, void* dst, LongType n) {
  const T* s = (const T*)src;
  T* d = (T*)dst;
  rt::parallel_for(0, n, 1 << 18, [&](LongType b, LongType e) {
    for (LongType i = b; i < e; ++i) d[i] = bswap<T>(s[i]);
  });
}
}  // namespace

// dst[i] = byteswap(src[i]); elem_size in {1,2,4,8}; src == dst allowed.
RT_API int nd4j_bswap_copy(const void* src, void* dst, LongType n, int elem_size) {
  switch (elem_size) {
    case 1:
      if (src != dst) std::memcpy(dst, src, (size_t)n);
      return 0;
    case 2: swap_copy<uint16_t>(src, dst, n); return 0;
    case 4: swap_copy<uint32_t>(src, dst, n); return 0;
    case 8: swap_copy<uint64_t>(src, dst, n); return 0;
  }
  rt::set_error(30, "unsupported element size");
  return 30;
}

// ---- .npy (format 1.0) --------------------------------------------------------------------------
// descr: e.g. "<f4", "<f8", "<i4", "<i8", "|u1", "|b1", "<f2"
RT_API int nd4j_npy_save(const char* path, const void* data, const LongType* shape, int rank, const char* descr,
                         int elem_size, int fortran_order) {
  FILE* f = std::fopen(path, "wb");
  if (!f) {
    rt::set_error(31, std::string("cannot open for writing: ") + path);
    return 31;
  }
  std::string dict = "{'descr': '" + std::string(descr) + "', 'fortran_order': " + (fortran_order ? "True" : "False") +
                     ", 'shape': (";
  LongType n = 1;
  for (int i = 0; i < rank; ++i) {
    dict += std::to_string(shape[i]);
    if (rank == 1 || i + 1 < rank) dict += ",";
    if (i + 1 < rank) dict += " ";
    n *= shape[i];
  }
  dict += "), }";
  size_t total = 10 + dict.size() + 1;
  size_t pad = (64 - total % 64) % 64;
  dict.append(pad, ' ');
  dict.push_back('\n');
  unsigned char hdr[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(dict.size() & 0xff),
                           (unsigned char)(dict.size() >> 8)};
  bool ok = std::fwrite(hdr, 1, 10, f) == 10 && std::fwrite(dict.data(), 1, dict.size(), f) == dict.size() &&
            std::fwrite(data, (size_t)elem_size, (size_t)n, f) == (size_t)n;
  std::fclose(f);
  if (!ok) {
    rt::set_error(32, "short write");
    return 32;
  }
  return 0;
}

// Parses the header. Returns data offset (>0) or a negative error. descr_out needs >= 8 bytes.
RT_API LongType nd4j_npy_header(const char* path, LongType* shape_out, int* rank_out, char* descr_out, int* fortran_out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) {
    rt::set_error(33, std::string("cannot open: ") + path);
    return -33;
  }
  unsigned char hdr[12];
  if (std::fread(hdr, 1, 10, f) != 10 || hdr[0] != 0x93 || std::memcmp(hdr + 1, "NUMPY", 5) != 0) {
    std::fclose(f);
    rt::set_error(34, "not an npy file");
    return -34;
  }
  size_t hlen = hdr[8] | (hdr[9] << 8);
  size_t pre = 10;
  if (hdr[6] >= 2) {
    if (std::fread(hdr + 10, 1, 2, f) != 2) {
      std::fclose(f);
      return -34;
    }
    hlen |= ((size_t)hdr[10] << 16) | ((size_t)hdr[11] << 24);
    pre = 12;
  }
  std::string dict(hlen, '\0');
  if (std::fread(&dict[0], 1, hlen, f) != hlen) {
    std::fclose(f);
    rt::set_error(34, "truncated npy header");
    return -34;
  }
  std::fclose(f);
  auto find_val = [&](const char* key) -> size_t {
    size_t p = dict.find(key);
    return p == std::string::npos ? p : dict.find(':', p) + 1;
  };
  size_t p = find_val("'descr'");
  size_t q1 = dict.find('\'', p), q2 = dict.find('\'', q1 + 1);
  std::string descr = dict.substr(q1 + 1, q2 - q1 - 1);
  std::strncpy(descr_out, descr.c_str(), 7);
  descr_out[7] = 0;
  p = find_val("'fortran_order'");
  *fortran_out = dict.compare(dict.find_first_not_of(' ', p), 4, "True") == 0;
  p = find_val("'shape'");
  size_t l = dict.find('(', p), r = dict.find(')', l);
  int rank = 0;
  size_t i = l + 1;
  while (i < r) {
    while (i < r && (dict[i] == ' ' || dict[i] == ',')) ++i;
    if (i >= r) break;
    shape_out[rank++] = std::strtoll(dict.c_str() + i, nullptr, 10);
    while (i < r && dict[i] != ',') ++i;
  }
  *rank_out = rank;
  return (LongType)(pre + hlen);
}

RT_API int nd4j_npy_read_data(const char* path, LongType offset, void* dst, LongType bytes) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return 33;
  std::fseek(f, (long)offset, SEEK_SET);
  size_t got = std::fread(dst, 1, (size_t)bytes, f);
  std::fclose(f);
  if ((LongType)got != bytes) {
    rt::set_error(35, "truncated npy payload");
    return 35;
  }
  return 0;
}
