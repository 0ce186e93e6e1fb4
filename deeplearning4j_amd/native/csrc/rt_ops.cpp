// Op registry: declared contracts of every custom (declarable) op + name hashing.
//
// Reference role: sd::ops::OpRegistrator (ops/declarable/OpRegistrator.h:170-181) — lookup by name
// or by 64-bit name hash — and HashHelper::getLongHash (helpers/impl/helper_hash.cpp; the hash is
// stored as FlatNode.opNum for custom ops, so it must be bit-compatible), plus the
// getAllCustomOps descriptor string parsed by OpExecutioner.getCustomOperations.
// The table itself is generated (tools/gen_op_table.py → op_table.inc).
#include "rt_common.h"

#include <mutex>
#include <unordered_map>

namespace {
struct OpDesc {
  const char* name;
  const char* kind;
  const char* group;
  int nIn, nOut, inplace, nT, nI;
  LongType hash;
};

LongType g_byte_table[256];
std::once_flag g_hash_once;

void init_hash() {
  uint64_t h = 0x544B2FBACAAF1684ULL;
  for (int i = 0; i < 256; i++) {
    for (int j = 0; j < 31; j++) {
      h = (h >> 7) ^ h;
      h = (h << 11) ^ h;
      h = (h >> 10) ^ h;
    }
    g_byte_table[i] = (LongType)h;
  }
}

LongType long_hash(const char* s) {
  std::call_once(g_hash_once, init_hash);
  uint64_t h = 0xBB40E64DA205B064ULL;
  const uint64_t mult = 7664345821815920749ULL;
  for (; *s; ++s) {
    unsigned char c = (unsigned char)*s;
    h = (h * mult) ^ (uint64_t)g_byte_table[c & 0xff];
    h = (h * mult) ^ (uint64_t)g_byte_table[0];   // high byte of an 8-bit char is always 0
  }
  return (LongType)h;
}

std::vector<OpDesc> g_ops;
std::unordered_map<std::string, int> g_by_name;
std::unordered_map<LongType, int> g_by_hash;
std::once_flag g_ops_once;
std::mutex g_ops_mu;
std::string g_all_ops;

void add_op(const OpDesc& d) {
  int idx = (int)g_ops.size();
  g_ops.push_back(d);
  g_ops.back().hash = long_hash(d.name);
  g_by_name[d.name] = idx;
  g_by_hash[g_ops.back().hash] = idx;
}

void init_ops() {
#define OPDEF(n, k, g, a, b, c, d, e) add_op(OpDesc{n, k, g, a, b, c, d, e, 0});
#define OPSYN(alias, target)                                  \
  {                                                           \
    auto it = g_by_name.find(target);                         \
    if (it != g_by_name.end() && !g_by_name.count(alias)) {   \
      g_by_name[alias] = it->second;                          \
      g_by_hash[long_hash(alias)] = it->second;               \
    }                                                         \
  }
#include "op_table.inc"
#undef OPDEF
#undef OPSYN
}

const OpDesc* find(const char* name) {
  std::call_once(g_ops_once, init_ops);
  std::lock_guard<std::mutex> lk(g_ops_mu);
  auto it = g_by_name.find(name);
  return it == g_by_name.end() ? nullptr : &g_ops[(size_t)it->second];
}
}  // namespace

RT_API LongType ops_hash(const char* name) { return long_hash(name); }

RT_API int ops_count() {
  std::call_once(g_ops_once, init_ops);
  return (int)g_ops.size();
}

// out = {nIn, nOut, inplace, nT, nI}; returns 0 if the op is unknown
RT_API int ops_describe(const char* name, int* out, LongType* hash_out) {
  const OpDesc* d = find(name);
  if (!d) return 0;
  out[0] = d->nIn; out[1] = d->nOut; out[2] = d->inplace; out[3] = d->nT; out[4] = d->nI;
  if (hash_out) *hash_out = d->hash;
  return 1;
}

RT_API const char* ops_name_by_hash(LongType hash) {
  std::call_once(g_ops_once, init_ops);
  std::lock_guard<std::mutex> lk(g_ops_mu);
  auto it = g_by_hash.find(hash);
  return it == g_by_hash.end() ? nullptr : g_ops[(size_t)it->second].name;
}

RT_API const char* ops_name_at(int i) {
  std::call_once(g_ops_once, init_ops);
  return i >= 0 && i < (int)g_ops.size() ? g_ops[(size_t)i].name : nullptr;
}
RT_API const char* ops_kind_at(int i) { return i >= 0 && i < ops_count() ? g_ops[(size_t)i].kind : nullptr; }
RT_API const char* ops_group_at(int i) { return i >= 0 && i < ops_count() ? g_ops[(size_t)i].group : nullptr; }

// user-defined ops (UserDefinedCustomOp, ADR 0023) can be registered at run time; name is copied
RT_API int ops_register(const char* name, int nIn, int nOut, int inplace, int nT, int nI) {
  std::call_once(g_ops_once, init_ops);
  std::lock_guard<std::mutex> lk(g_ops_mu);
  if (g_by_name.count(name)) return 0;
  char* copy = new char[std::strlen(name) + 1];
  std::strcpy(copy, name);
  add_op(OpDesc{copy, "UDF", "user", nIn, nOut, inplace, nT, nI, 0});
  g_all_ops.clear();
  return 1;
}

// "name:hash:nIn:nOut:inplace:nT:nI;" per op (format of getAllCustomOps, NativeOps.h:1479)
RT_API const char* ops_all_custom() {
  std::call_once(g_ops_once, init_ops);
  std::lock_guard<std::mutex> lk(g_ops_mu);
  if (g_all_ops.empty()) {
    for (auto& d : g_ops) {
      g_all_ops += d.name;
      g_all_ops += ":" + std::to_string(d.hash) + ":" + std::to_string(d.nIn) + ":" + std::to_string(d.nOut) + ":" +
                   std::to_string(d.inplace) + ":" + std::to_string(d.nT) + ":" + std::to_string(d.nI) + ";";
    }
  }
  return g_all_ops.c_str();
}
