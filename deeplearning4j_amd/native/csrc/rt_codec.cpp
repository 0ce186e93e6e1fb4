// Host gradient-compression codecs: threshold (sparse ±(idx+1) stream) and bitmap (2 bits/element).
//
// Wire format is the reference's (libnd4j/include/loops/cuda/type_conversions.cu:215-499,
// nd4j-cuda CudaThreshold.java:85-110): int32 message = [count, length, floatBits(tau), 0, payload..].
//  * threshold payload: one int per encoded element, +(idx+1) for x>=tau, -(idx+1) for x<=-tau; the
//    encoded amount (tau) is subtracted from the residual in place; the decoder adds ±tau.
//  * bitmap payload: one int per 16 elements; bit b = "|x|>=tau", bit b+16 = sign, and "bit b clear,
//    bit b+16 set" = negative half-threshold step. Residual updated in place.
// Own design: chunked two-pass (count -> exclusive scan -> emit) so the host path is multi-threaded
// and produces exactly the ordering of the device kernels (codec.hip).
#include "rt_common.h"

#include <algorithm>
#include <cmath>
#include <random>

namespace {
inline int float_bits(float f) {
  int i;
  std::memcpy(&i, &f, 4);
  return i;
}
inline float bits_float(int i) {
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}
constexpr LongType CHUNK = 1 << 16;
}  // namespace

RT_API LongType nd4j_threshold_count_f32(const float* x, LongType n, float tau) {
  LongType nchunks = (n + CHUNK - 1) / CHUNK;
  std::vector<LongType> cnt((size_t)nchunks, 0);
  rt::parallel_for(0, nchunks, 4, [&](LongType b, LongType e) {
    for (LongType c = b; c < e; ++c) {
      LongType k = 0, hi = std::min(n, (c + 1) * CHUNK);
      for (LongType i = c * CHUNK; i < hi; ++i) k += std::fabs(x[i]) >= tau;
      cnt[(size_t)c] = k;
    }
  });
  LongType t = 0;
  for (auto v : cnt) t += v;
  return t;
}

// Encodes at most `max_elems` elements (lowest indices first). out must hold 4 + max_elems ints.
RT_API LongType nd4j_threshold_encode_f32(float* x, LongType n, float tau, int* out, LongType max_elems) {
  if (n >= (LongType)INT32_MAX) {
    rt::set_error(20, "threshold codec indexes with int32: length too large");
    return -1;
  }
  LongType nchunks = (n + CHUNK - 1) / CHUNK;
  std::vector<LongType> cnt((size_t)nchunks + 1, 0);
  rt::parallel_for(0, nchunks, 4, [&](LongType b, LongType e) {
    for (LongType c = b; c < e; ++c) {
      LongType k = 0, hi = std::min(n, (c + 1) * CHUNK);
      for (LongType i = c * CHUNK; i < hi; ++i) k += std::fabs(x[i]) >= tau;
      cnt[(size_t)c + 1] = k;
    }
  });
  for (LongType c = 0; c < nchunks; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
  LongType total = std::min(cnt[(size_t)nchunks], max_elems);
  rt::parallel_for(0, nchunks, 4, [&](LongType b, LongType e) {
    for (LongType c = b; c < e; ++c) {
      LongType pos = cnt[(size_t)c], hi = std::min(n, (c + 1) * CHUNK);
      for (LongType i = c * CHUNK; i < hi && pos < max_elems; ++i) {
        float v = x[i];
        if (std::fabs(v) >= tau) {
          if (v > 0) {
            out[4 + pos] = (int)(i + 1);
            x[i] = v - tau;
          } else {
            out[4 + pos] = -(int)(i + 1);
            x[i] = v + tau;
          }
          ++pos;
        }
      }
    }
  });
  out[0] = (int)total;
  out[1] = (int)n;
  out[2] = float_bits(tau);
  out[3] = 0;
  return total;
}

RT_API int nd4j_threshold_decode_f32(const int* enc, float* z, LongType z_len) {
  LongType count = enc[0];
  float tau = bits_float(enc[2]);
  if (enc[1] > z_len) {
    rt::set_error(21, "threshold decode: target shorter than encoded length");
    return 21;
  }
  // indices are unique within one message, so chunks can be applied concurrently
  rt::parallel_for(0, count, 1 << 15, [&](LongType b, LongType e) {
    for (LongType k = b; k < e; ++k) {
      int el = enc[4 + k];
      LongType idx = (el < 0 ? -(LongType)el : (LongType)el) - 1;
      z[idx] += el < 0 ? -tau : tau;
    }
  });
  return 0;
}

// Bitmap: out must hold 4 + ceil(n/16) ints. Returns the number of elements that were stepped.
RT_API LongType nd4j_bitmap_encode_f32(float* x, LongType n, float tau, int* out) {
  LongType words = (n + 15) / 16;
  std::vector<LongType> cnt;
  LongType nchunks = (words + 4095) / 4096;
  cnt.assign((size_t)nchunks, 0);
  float half = tau / 2.0f;
  rt::parallel_for(0, nchunks, 1, [&](LongType b, LongType e) {
    for (LongType c = b; c < e; ++c) {
      LongType k = 0;
      for (LongType w = c * 4096; w < std::min(words, (c + 1) * 4096); ++w) {
        int word = 0;
        for (int bit = 0; bit < 16; ++bit) {
          LongType i = w * 16 + bit;
          if (i >= n) break;
          float v = x[i], a = std::fabs(v);
          if (a >= tau) {
            word |= 1 << bit;
            if (v < 0) {
              word |= 1 << (bit + 16);
              x[i] = v + tau;
            } else {
              x[i] = v - tau;
            }
            ++k;
          } else if (a >= half && v < 0) {
            word |= 1 << (bit + 16);
            x[i] = v + half;
            ++k;
          }
        }
        out[4 + w] = word;
      }
      cnt[(size_t)c] = k;
    }
  });
  LongType total = 0;
  for (auto v : cnt) total += v;
  out[0] = (int)total;
  out[1] = (int)n;
  out[2] = float_bits(tau);
  out[3] = 1;
  return total;
}

RT_API int nd4j_bitmap_decode_f32(const int* enc, float* z, LongType z_len) {
  LongType n = enc[1];
  if (n > z_len) {
    rt::set_error(22, "bitmap decode: target shorter than encoded length");
    return 22;
  }
  float tau = bits_float(enc[2]), half = tau / 2.0f;
  LongType words = (n + 15) / 16;
  rt::parallel_for(0, words, 1 << 12, [&](LongType b, LongType e) {
    for (LongType w = b; w < e; ++w) {
      int word = enc[4 + w];
      if (!word) continue;
      for (int bit = 0; bit < 16; ++bit) {
        LongType i = w * 16 + bit;
        if (i >= n) break;
        bool has = word & (1 << bit), sign = word & (1 << (bit + 16));
        if (has) z[i] += sign ? -tau : tau;
        else if (sign) z[i] -= half;
      }
    }
  });
  return 0;
}

// Pick tau so that roughly `target_count` elements satisfy |x| >= tau (sampled quantile).
RT_API float nd4j_estimate_threshold_f32(const float* x, LongType n, LongType target_count, uint64_t seed) {
  if (n <= 0 || target_count <= 0) return INFINITY;
  if (target_count >= n) return 0.0f;
  LongType m = std::min<LongType>(n, 1 << 18);
  std::vector<float> s((size_t)m);
  if (m == n) {
    for (LongType i = 0; i < n; ++i) s[(size_t)i] = std::fabs(x[i]);
  } else {
    std::mt19937_64 g(seed);
    for (LongType i = 0; i < m; ++i) s[(size_t)i] = std::fabs(x[g() % (uint64_t)n]);
  }
  double frac = (double)target_count / (double)n;
  LongType k = std::min<LongType>(m - 1, std::max<LongType>(0, (LongType)((1.0 - frac) * (double)m)));
  std::nth_element(s.begin(), s.begin() + k, s.end());
  return s[(size_t)k];
}
