// Arena workspaces + memory accounting.
//
// Role of libnd4j memory/Workspace (memory/cpu/Workspace.cpp, memory/cuda/Workspace.cu:60-239),
// MemoryRegistrator, MemoryTracker, MemoryCounter and of the Java Nd4jWorkspace learning/spill
// policies (nd4j-api linalg/api/memory/**, ADR 0024).
//
// Own design for MI355X: a workspace is a pair of bump arenas (HOST, DEVICE). The arena memory is
// *external*: Python hands in base pointers (pinned host block, and a slab carved out of the torch
// caching allocator for the 288 GB HBM side) so there is exactly one device allocator in the
// process. Allocation = 256-byte aligned bump; over-capacity requests spill to the caller
// (returns 0 -> caller allocates externally and reports it with ws_note_spill so the arena can
// learn its steady-state size). scope_out() applies the learning policy and reports the size the
// arena wants for the next cycle; the owner reallocates between cycles (never inside a scope).
#include "rt_common.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>

namespace {

enum { KIND_HOST = 0, KIND_DEVICE = 1 };
enum { LEARN_NONE = 0, LEARN_FIRST_LOOP = 1, LEARN_OVER_TIME = 2 };

struct Arena {
  uint64_t base = 0;
  LongType capacity = 0;
  LongType offset = 0;
  LongType peak_cycle = 0;      // bytes requested in the current cycle (incl. spilled)
  LongType peak_all = 0;        // max over cycles
  LongType spilled_cycle = 0;
  LongType spilled_total = 0;
};

struct Workspace {
  Arena a[2];
  int learning = LEARN_FIRST_LOOP;
  double over_allocation = 0.0;    // OVERALLOCATE policy: extra fraction
  LongType cycles_before_init = 0; // OVER_TIME: number of cycles to observe
  LongType cycles = 0;
  LongType generation = 0;
  int depth = 0;
  LongType align = 256;
  std::mutex mu;
};

std::atomic<LongType> g_counter[64][2];     // per device, per kind
std::atomic<LongType> g_limit[64][2];

}  // namespace

RT_API void* ws_create(int learning_policy, double over_allocation, LongType cycles_before_init) {
  auto* w = new Workspace();
  w->learning = learning_policy;
  w->over_allocation = over_allocation;
  w->cycles_before_init = cycles_before_init;
  return w;
}

RT_API void ws_destroy(void* h) { delete (Workspace*)h; }

// (Re)bind an arena to externally owned memory. Only legal outside a scope.
RT_API int ws_bind(void* h, int kind, uint64_t base, LongType capacity) {
  auto* w = (Workspace*)h;
  std::lock_guard<std::mutex> lk(w->mu);
  if (w->depth > 0) {
    rt::set_error(10, "ws_bind inside an open scope");
    return 10;
  }
  w->a[kind].base = base;
  w->a[kind].capacity = capacity;
  w->a[kind].offset = 0;
  return 0;
}

// Cyclic (ENDOFBUFFER_REACHED) workspaces wrap to the start of the arena when it is exhausted.
RT_API void ws_rewind(void* h, int kind) {
  auto* w = (Workspace*)h;
  std::lock_guard<std::mutex> lk(w->mu);
  w->a[kind].offset = 0;
  w->generation++;
}

RT_API void ws_scope_in(void* h) {
  auto* w = (Workspace*)h;
  std::lock_guard<std::mutex> lk(w->mu);
  if (w->depth++ == 0) {
    for (auto& a : w->a) {
      a.offset = 0;
      a.peak_cycle = 0;
      a.spilled_cycle = 0;
    }
    w->generation++;
  }
}

// Returns, per kind, the capacity the arena should have for the next cycle (0 = keep current).
RT_API void ws_scope_out(void* h, LongType* want_host, LongType* want_device) {
  auto* w = (Workspace*)h;
  std::lock_guard<std::mutex> lk(w->mu);
  LongType want[2] = {0, 0};
  if (w->depth > 0 && --w->depth == 0) {
    w->cycles++;
    for (int k = 0; k < 2; ++k) {
      Arena& a = w->a[k];
      a.peak_all = std::max(a.peak_all, a.peak_cycle);
      bool learn_now = (w->learning == LEARN_FIRST_LOOP && w->cycles == 1) ||
                       (w->learning == LEARN_OVER_TIME && w->cycles == std::max<LongType>(1, w->cycles_before_init));
      bool grow = a.spilled_cycle > 0 && w->learning != LEARN_NONE && w->cycles > 1 && w->learning == LEARN_OVER_TIME;
      if ((learn_now || grow) && a.peak_all > a.capacity) {
        LongType target = (LongType)((double)a.peak_all * (1.0 + w->over_allocation));
        want[k] = (target + w->align - 1) / w->align * w->align;
      }
    }
  }
  if (want_host) *want_host = want[0];
  if (want_device) *want_device = want[1];
}

// Bump allocation. Returns the address, or 0 when the request does not fit (caller spills).
RT_API uint64_t ws_alloc(void* h, int kind, LongType bytes) {
  auto* w = (Workspace*)h;
  std::lock_guard<std::mutex> lk(w->mu);
  Arena& a = w->a[kind];
  LongType need = (bytes + w->align - 1) / w->align * w->align;
  a.peak_cycle += need;
  if (a.base == 0 || a.offset + need > a.capacity) {
    a.spilled_cycle += need;
    a.spilled_total += need;
    return 0;
  }
  uint64_t p = a.base + (uint64_t)a.offset;
  a.offset += need;
  return p;
}

// stats: [capacity, offset, peak_cycle, peak_all, spilled_cycle, spilled_total, cycles, generation, depth]
RT_API void ws_stats(void* h, int kind, LongType* out) {
  auto* w = (Workspace*)h;
  std::lock_guard<std::mutex> lk(w->mu);
  const Arena& a = w->a[kind];
  LongType v[9] = {a.capacity, a.offset, a.peak_cycle, a.peak_all, a.spilled_cycle, a.spilled_total,
                   w->cycles, w->generation, (LongType)w->depth};
  std::memcpy(out, v, sizeof(v));
}

RT_API int ws_owns(void* h, int kind, uint64_t ptr) {
  auto* w = (Workspace*)h;
  const Arena& a = w->a[kind];
  return a.base != 0 && ptr >= a.base && ptr < a.base + (uint64_t)a.capacity;
}

// ---- MemoryCounter / MemoryTracker (memory/MemoryCounter.h): per-device byte accounting + limits ----
RT_API LongType mem_counter_add(int device, int kind, LongType delta) {
  if (device < 0 || device >= 64) return -1;
  return g_counter[device][kind].fetch_add(delta) + delta;
}
RT_API LongType mem_counter_get(int device, int kind) {
  if (device < 0 || device >= 64) return -1;
  return g_counter[device][kind].load();
}
RT_API void mem_counter_set_limit(int device, int kind, LongType limit) {
  if (device >= 0 && device < 64) g_limit[device][kind].store(limit);
}
RT_API LongType mem_counter_get_limit(int device, int kind) {
  if (device < 0 || device >= 64) return -1;
  return g_limit[device][kind].load();
}
// 1 when allocating `bytes` more would stay within the limit (0 limit = unlimited)
RT_API int mem_counter_validate(int device, int kind, LongType bytes) {
  if (device < 0 || device >= 64) return 0;
  LongType lim = g_limit[device][kind].load();
  return lim <= 0 || g_counter[device][kind].load() + bytes <= lim;
}
