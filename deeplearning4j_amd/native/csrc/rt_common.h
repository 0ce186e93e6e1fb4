// Host runtime (libnd4jrocm_rt.so) shared declarations.
// Error channel mirrors the reference's lastErrorCode/lastErrorMessage contract
// (libnd4j/include/legacy/NativeOps.h:150-156): every entry point catches, records, returns a code.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <functional>

#define RT_API extern "C" __attribute__((visibility("default")))

typedef int64_t LongType;

namespace rt {

constexpr int MAX_RANK = 32;   // reference: system/common.h:195-200

void set_error(int code, const std::string& msg);

// shapeInfo = [rank, shape.., strides.., extras, ews, order]  (helpers/shape.h)
inline int si_rank(const LongType* si) { return (int)si[0]; }
inline const LongType* si_shape(const LongType* si) { return si + 1; }
inline const LongType* si_strides(const LongType* si) { return si + 1 + si[0]; }
inline LongType si_extras(const LongType* si) { return si[1 + 2 * si[0]]; }
inline LongType si_ews(const LongType* si) { return si[2 + 2 * si[0]]; }
inline char si_order(const LongType* si) { return (char)si[3 + 2 * si[0]]; }
inline int si_len(int rank) { return 2 * rank + 4; }
inline LongType si_length(const LongType* si) {
  LongType n = 1;
  for (int i = 0; i < si_rank(si); ++i) n *= si_shape(si)[i];
  return n;
}

// simple fork-join helper (host analogue of samediff::Threads::parallel_for)
void parallel_for(LongType begin, LongType end, LongType min_chunk,
                  const std::function<void(LongType, LongType)>& fn);

}  // namespace rt
