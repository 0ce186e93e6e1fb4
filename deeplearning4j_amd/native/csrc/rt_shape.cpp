// shapeInfo / TAD math + constant shape/TAD caches.
//
// Role of libnd4j's helpers/shape.h, helpers/TAD.h, ConstantShapeHelper / ConstantTadHelper
// (reference: libnd4j/include/helpers/shape.h, helpers/TAD.h:1-913, helpers/ConstantShapeHelper.h,
// helpers/ConstantTadHelper.h). Descriptor layout is the reference's ABI:
//   LongType[2*rank+4] = [rank, shape.., strides.., extras(flags|dtype), ews, order]
// Own design: descriptors are interned in one process-wide hash set (content-addressed), TAD packs
// are cached by (descriptor pointer, dimension list); both hand out stable pointers so Python /
// device code can hold them without ownership.
#include "rt_common.h"

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace rt {

static thread_local int g_err_code = 0;
static thread_local std::string g_err_msg;

void set_error(int code, const std::string& msg) {
  g_err_code = code;
  g_err_msg = msg;
}

void parallel_for(LongType begin, LongType end, LongType min_chunk,
                  const std::function<void(LongType, LongType)>& fn) {
  LongType n = end - begin;
  if (n <= 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  LongType maxt = hw ? std::min<unsigned>(hw, 16) : 4;
  LongType nt = std::max<LongType>(1, std::min<LongType>(maxt, n / std::max<LongType>(1, min_chunk)));
  if (nt == 1) {
    fn(begin, end);
    return;
  }
  std::vector<std::thread> ts;
  LongType per = (n + nt - 1) / nt;
  for (LongType t = 0; t < nt; ++t) {
    LongType b = begin + t * per, e = std::min(end, b + per);
    if (b >= e) break;
    ts.emplace_back([=, &fn] { fn(b, e); });
  }
  for (auto& t : ts) t.join();
}

static void default_strides(int rank, const LongType* shape, char order, LongType* strides) {
  LongType s = 1;
  if (order == 'f') {
    for (int i = 0; i < rank; ++i) {
      strides[i] = s;
      s *= shape[i];
    }
  } else {
    for (int i = rank - 1; i >= 0; --i) {
      strides[i] = s;
      s *= shape[i];
    }
  }
}

// element-wise stride: 1 for dense c/f buffers, s if the whole array walks memory with one stride, else 0
static LongType compute_ews(int rank, const LongType* shape, const LongType* strides, char order) {
  if (rank == 0) return 1;
  // collapse unit dims
  std::vector<LongType> sh, st;
  for (int i = 0; i < rank; ++i)
    if (shape[i] != 1) {
      sh.push_back(shape[i]);
      st.push_back(strides[i]);
    }
  if (sh.empty()) return 1;
  if (order == 'c') {
    LongType base = st.back();
    LongType expect = base;
    for (int i = (int)sh.size() - 1; i >= 0; --i) {
      if (st[i] != expect) return 0;
      expect *= sh[i];
    }
    return base;
  }
  LongType base = st.front();
  LongType expect = base;
  for (size_t i = 0; i < sh.size(); ++i) {
    if (st[i] != expect) return 0;
    expect *= sh[i];
  }
  return base;
}

struct VecHash {
  size_t operator()(const std::vector<LongType>& v) const {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (LongType x : v) {
      h ^= (uint64_t)x;
      h *= 0x100000001b3ULL;
    }
    return (size_t)h;
  }
};

static std::mutex g_shape_mu;
static std::unordered_map<std::vector<LongType>, std::unique_ptr<std::vector<LongType>>, VecHash> g_shape_cache;

static const LongType* intern(const std::vector<LongType>& si) {
  std::lock_guard<std::mutex> lk(g_shape_mu);
  auto it = g_shape_cache.find(si);
  if (it != g_shape_cache.end()) return it->second->data();
  auto p = std::make_unique<std::vector<LongType>>(si);
  const LongType* out = p->data();
  g_shape_cache.emplace(si, std::move(p));
  return out;
}

struct TadPack {
  std::vector<LongType> tadShapeInfo;
  std::vector<LongType> offsets;
  LongType numTads;
};

static std::mutex g_tad_mu;
static std::unordered_map<std::vector<LongType>, std::unique_ptr<TadPack>, VecHash> g_tad_cache;

}  // namespace rt

using namespace rt;

RT_API int nd4j_last_error_code() { return g_err_code; }
RT_API const char* nd4j_last_error_message() { return g_err_msg.c_str(); }
RT_API void nd4j_clear_error() {
  g_err_code = 0;
  g_err_msg.clear();
}

RT_API int nd4j_shape_info_length(int rank) { return si_len(rank); }

// Build a descriptor with default strides for `order`. `extras` carries dtype/empty flags (ArrayOptions).
RT_API int nd4j_shape_buffer(int rank, const LongType* shape, char order, LongType extras, LongType* out) {
  if (rank < 0 || rank > MAX_RANK) {
    set_error(1, "rank out of range");
    return 1;
  }
  out[0] = rank;
  for (int i = 0; i < rank; ++i) out[1 + i] = shape[i];
  default_strides(rank, shape, order, out + 1 + rank);
  out[1 + 2 * rank] = extras;
  out[2 + 2 * rank] = 1;
  out[3 + 2 * rank] = order;
  return 0;
}

RT_API int nd4j_shape_buffer_ex(int rank, const LongType* shape, const LongType* strides, char order, LongType extras,
                                LongType* out) {
  if (rank < 0 || rank > MAX_RANK) {
    set_error(1, "rank out of range");
    return 1;
  }
  out[0] = rank;
  for (int i = 0; i < rank; ++i) {
    out[1 + i] = shape[i];
    out[1 + rank + i] = strides[i];
  }
  out[1 + 2 * rank] = extras;
  out[2 + 2 * rank] = compute_ews(rank, shape, strides, order);
  out[3 + 2 * rank] = order;
  return 0;
}

// ConstantShapeHelper: returns a process-lifetime pointer, identical for identical descriptors.
RT_API const LongType* nd4j_shape_intern(const LongType* si) {
  std::vector<LongType> v(si, si + si_len(si_rank(si)));
  return intern(v);
}

RT_API LongType nd4j_shape_cache_size() {
  std::lock_guard<std::mutex> lk(g_shape_mu);
  return (LongType)g_shape_cache.size();
}

RT_API LongType nd4j_shape_length(const LongType* si) { return si_length(si); }

RT_API void nd4j_index2coords(LongType index, const LongType* si, LongType* coords) {
  int rank = si_rank(si);
  const LongType* sh = si_shape(si);
  if (si_order(si) == 'f') {
    for (int i = 0; i < rank; ++i) {
      coords[i] = index % sh[i];
      index /= sh[i];
    }
  } else {
    for (int i = rank - 1; i >= 0; --i) {
      coords[i] = index % sh[i];
      index /= sh[i];
    }
  }
}

RT_API LongType nd4j_coords2offset(const LongType* si, const LongType* coords) {
  LongType off = 0;
  const LongType* st = si_strides(si);
  for (int i = 0; i < si_rank(si); ++i) off += coords[i] * st[i];
  return off;
}

RT_API LongType nd4j_coords2index(const LongType* si, const LongType* coords) {
  int rank = si_rank(si);
  const LongType* sh = si_shape(si);
  LongType idx = 0;
  if (si_order(si) == 'f') {
    LongType m = 1;
    for (int i = 0; i < rank; ++i) {
      idx += coords[i] * m;
      m *= sh[i];
    }
  } else {
    for (int i = 0; i < rank; ++i) idx = idx * sh[i] + coords[i];
  }
  return idx;
}

// linear index (in the array's own order) -> buffer offset  (shape::getIndexOffset)
RT_API LongType nd4j_index_offset(LongType index, const LongType* si) {
  LongType ews = si_ews(si);
  if (ews > 0) return index * ews;
  LongType coords[MAX_RANK];
  nd4j_index2coords(index, si, coords);
  return nd4j_coords2offset(si, coords);
}

RT_API void nd4j_index_offsets(const LongType* si, LongType* out) {
  LongType n = si_length(si);
  parallel_for(0, n, 1 << 16, [&](LongType b, LongType e) {
    for (LongType i = b; i < e; ++i) out[i] = nd4j_index_offset(i, si);
  });
}

RT_API int nd4j_permute(const LongType* si, const int* perm, LongType* out) {
  int rank = si_rank(si);
  std::vector<char> seen(rank, 0);
  LongType sh[MAX_RANK], st[MAX_RANK];
  for (int i = 0; i < rank; ++i) {
    int p = perm[i] < 0 ? perm[i] + rank : perm[i];
    if (p < 0 || p >= rank || seen[p]) {
      set_error(2, "invalid permutation");
      return 2;
    }
    seen[p] = 1;
    sh[i] = si_shape(si)[p];
    st[i] = si_strides(si)[p];
  }
  return nd4j_shape_buffer_ex(rank, sh, st, si_order(si), si_extras(si), out);
}

// numpy-style broadcast of two shapes; returns the output rank or -1
RT_API int nd4j_broadcast_shape(int ra, const LongType* a, int rb, const LongType* b, LongType* out) {
  int r = std::max(ra, rb);
  for (int i = 0; i < r; ++i) {
    LongType da = i < r - ra ? 1 : a[i - (r - ra)];
    LongType db = i < r - rb ? 1 : b[i - (r - rb)];
    if (da != db && da != 1 && db != 1) {
      set_error(3, "shapes are not broadcastable");
      return -1;
    }
    out[i] = da == 1 ? db : da;
  }
  return r;
}

// Can `si` be viewed with `newShape` without a copy? On success writes the new strides and returns 1.
// (same question as shape::reshapeC; algorithm: merge contiguous runs of the old view, then split them)
RT_API int nd4j_reshape_nocopy(const LongType* si, int newRank, const LongType* newShape, char order, LongType* newStrides) {
  int rank = si_rank(si);
  std::vector<LongType> osh, ost;
  for (int i = 0; i < rank; ++i)
    if (si_shape(si)[i] != 1) {
      osh.push_back(si_shape(si)[i]);
      ost.push_back(si_strides(si)[i]);
    }
  LongType nOld = 1, nNew = 1;
  for (auto v : osh) nOld *= v;
  for (int i = 0; i < newRank; ++i) nNew *= newShape[i];
  if (nOld != nNew) return 0;
  if (nOld == 0) return 0;
  bool f = order == 'f';
  if (f) {
    std::reverse(osh.begin(), osh.end());
    std::reverse(ost.begin(), ost.end());
  }
  std::vector<LongType> nsh(newShape, newShape + newRank), nst(newRank, 0);
  if (f) std::reverse(nsh.begin(), nsh.end());
  int oi = 0, oj = 1, ni = 0, nj = 1, on = (int)osh.size();
  while (ni < newRank && oi < on) {
    LongType np = nsh[ni], op = osh[oi];
    while (np != op) {
      if (np < op) np *= nsh[nj++];
      else op *= osh[oj++];
    }
    for (int k = oi; k < oj - 1; ++k)
      if (ost[k] != osh[k + 1] * ost[k + 1]) return 0;   // old dims in this group are not mergeable
    nst[nj - 1] = ost[oj - 1];
    for (int k = nj - 1; k > ni; --k) nst[k - 1] = nst[k] * nsh[k];
    ni = nj++;
    oi = oj++;
  }
  LongType last = ni >= 1 ? nst[ni - 1] : 1;
  for (int k = ni; k < newRank; ++k) nst[k] = last;   // trailing unit dims
  if (f) std::reverse(nst.begin(), nst.end());
  for (int i = 0; i < newRank; ++i) newStrides[i] = nst[i];
  return 1;
}

// ---- TAD ("tensor along dimension") packs -------------------------------------------------------
// dims = the dimensions that stay INSIDE each sub-array. Offsets enumerate the remaining dims in c order.
RT_API const void* nd4j_tad_pack(const LongType* si, const int* dims, int ndims) {
  int rank = si_rank(si);
  std::vector<LongType> key(si, si + si_len(rank));
  std::vector<int> d(ndims);
  for (int i = 0; i < ndims; ++i) {
    d[i] = dims[i] < 0 ? dims[i] + rank : dims[i];
    if (d[i] < 0 || d[i] >= rank) {
      set_error(4, "TAD dimension out of range");
      return nullptr;
    }
  }
  std::sort(d.begin(), d.end());
  d.erase(std::unique(d.begin(), d.end()), d.end());
  key.push_back(-1);
  for (int v : d) key.push_back(v);
  {
    std::lock_guard<std::mutex> lk(g_tad_mu);
    auto it = g_tad_cache.find(key);
    if (it != g_tad_cache.end()) return it->second.get();
  }
  auto pack = std::make_unique<TadPack>();
  std::vector<LongType> tsh, tst, osh, ost;
  for (int i = 0; i < rank; ++i) {
    bool in = std::binary_search(d.begin(), d.end(), i);
    (in ? tsh : osh).push_back(si_shape(si)[i]);
    (in ? tst : ost).push_back(si_strides(si)[i]);
  }
  int tr = (int)tsh.size();
  pack->tadShapeInfo.resize(si_len(tr));
  nd4j_shape_buffer_ex(tr, tsh.data(), tst.data(), si_order(si), si_extras(si), pack->tadShapeInfo.data());
  LongType nt = 1;
  for (auto v : osh) nt *= v;
  pack->numTads = nt;
  pack->offsets.resize((size_t)nt);
  int orank = (int)osh.size();
  std::vector<LongType> c(orank, 0);
  LongType off = 0;
  for (LongType t = 0; t < nt; ++t) {
    pack->offsets[(size_t)t] = off;
    for (int k = orank - 1; k >= 0; --k) {       // odometer increment in c order
      off += ost[k];
      if (++c[k] < osh[k]) break;
      off -= ost[k] * osh[k];
      c[k] = 0;
    }
  }
  std::lock_guard<std::mutex> lk(g_tad_mu);
  auto& slot = g_tad_cache[key];
  if (!slot) slot = std::move(pack);
  return slot.get();
}

RT_API LongType nd4j_tad_num(const void* p) { return ((const TadPack*)p)->numTads; }
RT_API const LongType* nd4j_tad_shape_info(const void* p) { return ((const TadPack*)p)->tadShapeInfo.data(); }
RT_API const LongType* nd4j_tad_offsets(const void* p) { return ((const TadPack*)p)->offsets.data(); }
RT_API LongType nd4j_tad_cache_size() {
  std::lock_guard<std::mutex> lk(g_tad_mu);
  return (LongType)g_tad_cache.size();
}

// ravel / unravel (NativeOps.h:1458-1464)
RT_API void nd4j_ravel_multi_index(const LongType* indices, LongType* flat, LongType n, const LongType* shape, int rank,
                                   int clip_mode) {
  for (LongType i = 0; i < n; ++i) {
    LongType idx = 0;
    for (int d = 0; d < rank; ++d) {
      LongType v = indices[d * n + i];
      if (clip_mode == 1) v = ((v % shape[d]) + shape[d]) % shape[d];          // wrap
      else if (clip_mode == 2) v = std::max<LongType>(0, std::min(shape[d] - 1, v));  // clip
      idx = idx * shape[d] + v;
    }
    flat[i] = idx;
  }
}

RT_API void nd4j_unravel_index(LongType* indices, const LongType* flat, LongType n, const LongType* shape, int rank) {
  for (LongType i = 0; i < n; ++i) {
    LongType v = flat[i];
    for (int d = rank - 1; d >= 0; --d) {
      indices[d * n + i] = v % shape[d];
      v /= shape[d];
    }
  }
}
