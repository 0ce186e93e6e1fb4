// Native graph runtime: FlatGraph loader, topological layering, GraphHolder registry,
// frame-based dataflow executor with TF-style control flow, per-node profiling.
//
// Reference roles (libnd4j/include/graph/**): Graph (builds layered Nodes from a FlatGraph,
// graph/impl/Graph.cpp), GraphExecutioner::executeFlatGraph (NativeOps.h:1474), GraphHolder
// (registerGraph / executeStoredGraph / unregisterGraph, NativeOps.h:1525-1543), VariableSpace /
// FlowPath / FrameState, graph/execution/Logic{Switch,Merge,Enter,Exit,NextIteration,LoopCond}.cpp and
// graph/profiling/{GraphProfile,NodeProfile}.
//
// Design: the runtime owns the *schedule* — which node runs in which (frame, iteration) context and
// which values are visible there — while tensors stay opaque 64-bit handles owned by the caller.
// Compute ops are dispatched through one callback (the Python op library launches HIP kernels on the
// current stream; nothing here blocks on the GPU except Switch / LoopCond predicates, which need a host
// boolean by definition). Arrays of VARIABLE / CONSTANT nodes are exposed zero-copy from the
// registered buffer.
#include "rt_common.h"
#include "rt_exceptions.h"

#include <algorithm>
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_map>
#include <unordered_set>

namespace {

// ------------------------------------------------------------------ minimal FlatBuffers reader
struct FB {
  const uint8_t* buf;
  size_t len;
  bool ok(size_t off, size_t n) const { return off + n <= len && off + n >= off; }
  template <typename T>
  T rd(size_t off) const {
    T v{};
    if (ok(off, sizeof(T))) std::memcpy(&v, buf + off, sizeof(T));
    return v;
  }
};

struct Tbl {
  const FB* fb = nullptr;
  size_t pos = 0;
  explicit operator bool() const { return fb != nullptr; }
  // byte offset of field `slot` inside the table, 0 when absent
  size_t field(int slot) const {
    int32_t soff = fb->rd<int32_t>(pos);
    size_t vt = (size_t)((int64_t)pos - soff);
    uint16_t vsize = fb->rd<uint16_t>(vt);
    size_t e = 4 + 2 * (size_t)slot;
    if (e + 2 > vsize) return 0;
    uint16_t o = fb->rd<uint16_t>(vt + e);
    return o ? pos + o : 0;
  }
  template <typename T>
  T scalar(int slot, T dflt = T{}) const {
    size_t f = field(slot);
    return f ? fb->rd<T>(f) : dflt;
  }
  size_t indirect(int slot) const {
    size_t f = field(slot);
    return f ? f + fb->rd<uint32_t>(f) : 0;
  }
  Tbl table(int slot) const {
    size_t p = indirect(slot);
    return p ? Tbl{fb, p} : Tbl{};
  }
  std::string str_at(size_t p) const {
    uint32_t n = fb->rd<uint32_t>(p);
    if (!fb->ok(p + 4, n)) return std::string();
    return std::string((const char*)fb->buf + p + 4, n);
  }
  std::string str(int slot) const {
    size_t p = indirect(slot);
    return p ? str_at(p) : std::string();
  }
  // vector: returns element count, *data = first element position
  uint32_t vec(int slot, size_t* data) const {
    size_t p = indirect(slot);
    if (!p) { *data = 0; return 0; }
    *data = p + 4;
    return fb->rd<uint32_t>(p);
  }
  template <typename T>
  std::vector<T> scalars(int slot) const {
    size_t d; uint32_t n = vec(slot, &d);
    std::vector<T> out;
    if (!fb->ok(d, (size_t)n * sizeof(T))) return out;
    out.resize(n);
    if (n) std::memcpy(out.data(), fb->buf + d, (size_t)n * sizeof(T));
    return out;
  }
  std::vector<Tbl> tables(int slot) const {
    size_t d; uint32_t n = vec(slot, &d);
    std::vector<Tbl> out;
    for (uint32_t i = 0; i < n; ++i) {
      size_t e = d + 4 * (size_t)i;
      out.push_back(Tbl{fb, e + fb->rd<uint32_t>(e)});
    }
    return out;
  }
  std::vector<std::string> strings(int slot) const {
    size_t d; uint32_t n = vec(slot, &d);
    std::vector<std::string> out;
    for (uint32_t i = 0; i < n; ++i) {
      size_t e = d + 4 * (size_t)i;
      out.push_back(str_at(e + fb->rd<uint32_t>(e)));
    }
    return out;
  }
};

// ------------------------------------------------------------------ graph model
enum Kind { K_OP = 0, K_SWITCH, K_MERGE, K_ENTER, K_EXIT, K_NEXT, K_LOOPCOND };

struct GVar {
  std::string name;
  int id = 0, idx = 0;
  int vtype = 0;          // 0 VARIABLE 1 CONSTANT 2 ARRAY 3 PLACEHOLDER (variable.fbs VarType)
  int dtype = 0;
  std::vector<LongType> shape;       // declared shape (placeholders) or array shapeInfo-derived
  std::vector<LongType> shape_info;  // FlatArray.shape when an array is attached
  const uint8_t* data = nullptr;     // FlatArray.buffer (zero-copy into the registered buffer)
  LongType data_len = 0;
  int arr_dtype = 0, byte_order = 0;
  int producer = -1;
  std::vector<int> consumers;
};

struct GNode {
  int id = 0;
  std::string name, op_name, frame_name;
  int op_type = 0;
  LongType op_num = 0;
  bool const_enter = false;
  std::vector<int> inputs, outputs, control;   // variable indices
  std::vector<double> targs;
  std::vector<LongType> iargs;
  std::vector<uint8_t> bargs, dargs;
  std::vector<int32_t> dims;
  std::vector<std::string> sargs;
  double scalar = 0;
  bool has_scalar = false;
  Kind kind = K_OP;
  int layer = -1;
  // profile (graph/profiling/NodeProfile)
  LongType exec_count = 0, exec_ns = 0;
};

std::string lower(std::string s) {
  for (auto& c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

struct Graph {
  std::vector<uint8_t> storage;
  std::vector<GVar> vars;
  std::vector<GNode> nodes;
  std::unordered_map<std::string, int> var_by_name;
  std::vector<int> outputs;                      // FlatGraph.outputs
  std::vector<std::string> placeholders, loss_vars;
  std::string training_config;
  std::vector<std::vector<int>> layers;
  int n_layers = 0;
  // profile (GraphProfile)
  LongType executions = 0, total_ns = 0;
  std::mutex mu;

  int var_index(const std::string& n) const {
    auto it = var_by_name.find(n);
    return it == var_by_name.end() ? -1 : it->second;
  }
};

double read_scalar_array(const Tbl& a) {
  size_t d; uint32_t n = a.vec(1, &d);
  int dt = a.scalar<int8_t>(2);
  if (!n) return 0;
  const FB* fb = a.fb;
  switch (dt) {
    case 5: return fb->rd<float>(d);
    case 6: return fb->rd<double>(d);
    case 9: return fb->rd<int32_t>(d);
    case 10: return (double)fb->rd<int64_t>(d);
    case 7: return fb->rd<int8_t>(d);
    case 8: return fb->rd<int16_t>(d);
    case 11: case 1: return fb->rd<uint8_t>(d);
    default: return 0;
  }
}

std::shared_ptr<Graph> parse(const uint8_t* buf, LongType len) {
  auto g = std::make_shared<Graph>();
  g->storage.assign(buf, buf + len);
  static thread_local FB fbs;   // only used during parse
  FB* fb = new FB{g->storage.data(), g->storage.size()};
  std::unique_ptr<FB> guard(fb);
  if (len < 8) throw std::runtime_error("FlatGraph buffer too small");
  Tbl root{fb, (size_t)fb->rd<uint32_t>(0)};

  std::map<std::pair<int, int>, int> by_id;
  for (auto& v : root.tables(1)) {
    GVar gv;
    gv.name = v.str(1);
    Tbl idp = v.table(0);
    int a = idp ? idp.scalar<int32_t>(0) : 0, b = idp ? idp.scalar<int32_t>(1) : 0;
    gv.id = a; gv.idx = b;
    gv.dtype = v.scalar<int8_t>(2);
    gv.shape = v.scalars<LongType>(3);
    gv.vtype = v.scalar<int8_t>(6);
    Tbl arr = v.table(4);
    if (arr) {
      gv.shape_info = arr.scalars<LongType>(0);
      size_t d; uint32_t n = arr.vec(1, &d);
      gv.data = n ? g->storage.data() + d : nullptr;
      gv.data_len = n;
      gv.arr_dtype = arr.scalar<int8_t>(2);
      gv.byte_order = arr.scalar<int8_t>(3);
      if (gv.shape.empty() && !gv.shape_info.empty()) {
        int r = (int)gv.shape_info[0];
        if (r >= 0 && r <= rt::MAX_RANK && (size_t)r + 1 <= gv.shape_info.size())
          gv.shape.assign(gv.shape_info.begin() + 1, gv.shape_info.begin() + 1 + r);
      }
    }
    int index = (int)g->vars.size();
    if (idp) by_id[{a, b}] = index;
    g->var_by_name[gv.name] = index;
    g->vars.push_back(std::move(gv));
  }
  auto resolve = [&](int a, int b) -> int {
    auto it = by_id.find({a, b});
    if (it != by_id.end()) return it->second;
    it = by_id.find({a, 0});
    return it == by_id.end() ? -1 : it->second;
  };

  for (auto& n : root.tables(2)) {
    GNode gn;
    gn.id = n.scalar<int32_t>(0);
    gn.name = n.str(1);
    if (gn.name.empty()) gn.name = "op_" + std::to_string(gn.id);
    gn.op_type = (uint8_t)n.scalar<int8_t>(2);
    gn.op_num = n.scalar<int64_t>(3);
    gn.op_name = n.str(16);
    for (auto& p : n.tables(6)) {
      int vi = resolve(p.scalar<int32_t>(0), p.scalar<int32_t>(1));
      if (vi >= 0) gn.inputs.push_back(vi);
    }
    if (gn.inputs.empty())
      for (int k : n.scalars<int32_t>(5)) {
        int vi = resolve(k, 0);
        if (vi >= 0) gn.inputs.push_back(vi);
      }
    auto outs = n.strings(15);
    if (outs.empty()) {
      int vi = resolve(gn.id, 0);
      if (vi >= 0) gn.outputs.push_back(vi);
    } else {
      for (auto& on : outs) {
        int vi = g->var_index(on);
        if (vi < 0) {   // output variable not declared: synthesise an ARRAY variable
          GVar gv; gv.name = on; gv.vtype = 2; gv.id = gn.id; gv.idx = (int)gn.outputs.size();
          vi = (int)g->vars.size();
          g->var_by_name[on] = vi;
          g->vars.push_back(gv);
        }
        gn.outputs.push_back(vi);
      }
    }
    for (auto& cd : n.strings(19)) {
      int vi = g->var_index(cd);
      if (vi >= 0) gn.control.push_back(vi);
    }
    gn.targs = n.scalars<double>(8);
    gn.iargs = n.scalars<LongType>(9);
    gn.bargs = n.scalars<uint8_t>(10);
    gn.dims = n.scalars<int32_t>(11);
    gn.dargs = n.scalars<uint8_t>(22);
    gn.sargs = n.strings(23);
    Tbl sc = n.table(18);
    if (sc) { gn.scalar = read_scalar_array(sc); gn.has_scalar = true; }
    for (auto& pr : n.tables(4)) {
      std::string pn = pr.str(0);
      if (pn == "frameName" || pn == "frame_name") {
        auto ss = pr.strings(6);
        if (!ss.empty()) gn.frame_name = ss[0];
      } else if (pn == "isConstant" || pn == "is_constant") {
        auto bb = pr.scalars<uint8_t>(5);
        gn.const_enter = !bb.empty() && bb[0];
      }
    }
    std::string ln = lower(gn.op_name);
    if (gn.op_name.empty() && gn.op_type == 119 /*LOGIC*/) {
      switch ((int)gn.op_num) {
        case 30: ln = "switch"; break; case 60: ln = "merge"; break; case 70: ln = "loop_cond"; break;
        case 80: ln = "next_iteration"; break; case 90: ln = "exit"; break; case 100: ln = "enter"; break;
      }
      gn.op_name = ln;
    }
    gn.kind = ln == "switch" ? K_SWITCH : ln == "merge" ? K_MERGE : ln == "enter" ? K_ENTER : ln == "exit" ? K_EXIT
              : ln == "next_iteration" ? K_NEXT : ln == "loop_cond" ? K_LOOPCOND : K_OP;
    g->nodes.push_back(std::move(gn));
  }
  for (size_t i = 0; i < g->nodes.size(); ++i) {
    for (int o : g->nodes[i].outputs) g->vars[(size_t)o].producer = (int)i;
    for (int in : g->nodes[i].inputs) g->vars[(size_t)in].consumers.push_back((int)i);
  }
  for (auto& p : root.tables(3)) {
    int vi = resolve(p.scalar<int32_t>(0), p.scalar<int32_t>(1));
    if (vi >= 0) g->outputs.push_back(vi);
  }
  g->placeholders = root.strings(5);
  g->loss_vars = root.strings(6);
  g->training_config = root.str(7);

  // topological layering (Graph::buildGraph): layer(n) = 1 + max layer of producers; back edges
  // NextIteration -> Merge are ignored so loops do not make the graph cyclic
  size_t N = g->nodes.size();
  std::vector<int> indeg(N, 0);
  std::vector<std::vector<int>> succ(N);
  for (size_t i = 0; i < N; ++i) {
    auto& nd = g->nodes[i];
    std::set<int> preds;
    auto add = [&](int vi) {
      int p = g->vars[(size_t)vi].producer;
      if (p < 0 || p == (int)i) return;
      if (nd.kind == K_MERGE && g->nodes[(size_t)p].kind == K_NEXT) return;
      preds.insert(p);
    };
    for (int in : nd.inputs) add(in);
    for (int c : nd.control) add(c);
    indeg[i] = (int)preds.size();
    for (int p : preds) succ[(size_t)p].push_back((int)i);
  }
  std::deque<int> q;
  for (size_t i = 0; i < N; ++i)
    if (!indeg[i]) { g->nodes[i].layer = 0; q.push_back((int)i); }
  size_t seen = 0;
  while (!q.empty()) {
    int u = q.front(); q.pop_front(); ++seen;
    for (int v : succ[(size_t)u]) {
      g->nodes[(size_t)v].layer = std::max(g->nodes[(size_t)v].layer, g->nodes[(size_t)u].layer + 1);
      if (--indeg[(size_t)v] == 0) q.push_back(v);
    }
  }
  if (seen != N) throw std::runtime_error("FlatGraph has a cycle outside NextIteration->Merge back edges");
  int nl = 0;
  for (auto& nd : g->nodes) nl = std::max(nl, nd.layer + 1);
  g->layers.assign((size_t)nl, {});
  for (size_t i = 0; i < N; ++i) g->layers[(size_t)g->nodes[i].layer].push_back((int)i);
  g->n_layers = nl;
  (void)fbs;
  return g;
}

// ------------------------------------------------------------------ GraphHolder
std::mutex g_mu;
std::unordered_map<LongType, std::shared_ptr<Graph>> g_graphs;

std::shared_ptr<Graph> get(LongType id) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_graphs.find(id);
  return it == g_graphs.end() ? nullptr : it->second;
}

// ------------------------------------------------------------------ executor
typedef int (*ExecFn)(void* user, int node, const LongType* ins, int n_in, LongType* outs, int n_out);
typedef int (*PredFn)(void* user, LongType handle);

struct Key {
  int var, frame, iter;
  bool operator==(const Key& o) const { return var == o.var && frame == o.frame && iter == o.iter; }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = (uint64_t)(uint32_t)k.var * 0x9E3779B97F4A7C15ULL;
    h ^= ((uint64_t)(uint32_t)k.frame + 0x7F4A7C15ULL) * 0xBF58476D1CE4E5B9ULL;
    h ^= ((uint64_t)(uint32_t)k.iter + 0x1CE4E5B9ULL) * 0x94D049BB133111EBULL;
    return (size_t)(h ^ (h >> 29));
  }
};

constexpr LongType NONE = -1;

struct Frame { std::string name; int parent, parent_iter; };

struct Exec {
  Graph& g;
  ExecFn exec; PredFn pred; void* user;
  std::vector<char> need;
  std::unordered_map<Key, LongType, KeyHash> vals;
  std::unordered_set<Key, KeyHash> executed;     // var field holds the node index
  std::vector<Frame> frames{{"", -1, 0}};
  std::unordered_map<std::string, int> frame_by_name{{"", 0}};
  std::set<std::pair<int, int>> live{{0, 0}};
  std::deque<Key> queue;
  std::string error;
  int max_iter;

  LongType lookup(int var, int f, int i) const {
    while (true) {
      auto it = vals.find(Key{var, f, i});
      if (it != vals.end()) return it->second;
      if (f == 0) return NONE;
      it = vals.find(Key{var, f, -1});          // loop-invariant Enter values are visible in every iteration
      if (it != vals.end()) return it->second;
      i = frames[(size_t)f].parent_iter;
      f = frames[(size_t)f].parent;
    }
  }
  void put(const Key& k, LongType h, std::vector<Key>* produced) {
    vals[k] = h;
    if (k.iter >= 0) live.insert({k.frame, k.iter});
    produced->push_back(k);
  }

  // returns false on error
  bool run(int ni, int frame, int it, std::vector<Key>* produced) {
    GNode& nd = g.nodes[(size_t)ni];
    Key ek{ni, frame, it};
    if (executed.count(ek)) return true;
    std::vector<LongType> ins(nd.inputs.size());
    for (size_t k = 0; k < ins.size(); ++k) ins[k] = lookup(nd.inputs[k], frame, it);
    std::vector<LongType> outs(nd.outputs.size(), NONE);
    int oframe = frame, oit = it;
    if (nd.kind == K_MERGE) {
      LongType first = NONE, last = NONE, efirst = NONE, elast = NONE;
      for (size_t k = 0; k < ins.size(); ++k) {
        if (ins[k] != NONE) { if (first == NONE) first = ins[k]; last = ins[k]; }
        auto e = vals.find(Key{nd.inputs[k], frame, it});
        if (e != vals.end()) { if (efirst == NONE) efirst = e->second; elast = e->second; }
      }
      if (first == NONE) return true;
      executed.insert(ek);
      // prefer the value produced in this very iteration (NextIteration) over the loop-entry value
      LongType pick = it > 0 ? (elast != NONE ? elast : last) : (efirst != NONE ? efirst : first);
      if (!outs.empty()) outs[0] = pick;
    } else {
      for (LongType h : ins) if (h == NONE) return true;
      for (int c : nd.control) if (lookup(c, frame, it) == NONE) return true;
      executed.insert(ek);
      switch (nd.kind) {
        case K_ENTER: {
          std::string fname = !nd.frame_name.empty() ? nd.frame_name : (!nd.sargs.empty() ? nd.sargs[0] : nd.name);
          std::string child = frame == 0 ? fname : fname + "@" + std::to_string(frame) + "#" + std::to_string(it);
          auto fi = frame_by_name.find(child);
          int cf;
          if (fi == frame_by_name.end()) {
            cf = (int)frames.size();
            frames.push_back(Frame{child, frame, it});
            frame_by_name[child] = cf;
          } else cf = fi->second;
          bool invariant = nd.const_enter;
          if (!invariant) {
            bool feeds_merge = false;
            if (!nd.outputs.empty())
              for (int c : g.vars[(size_t)nd.outputs[0]].consumers)
                if (g.nodes[(size_t)c].kind == K_MERGE) { feeds_merge = true; break; }
            invariant = !feeds_merge;
          }
          if (!outs.empty()) outs[0] = ins[0];
          oframe = cf; oit = invariant ? -1 : 0;
          break;
        }
        case K_EXIT:
          if (!outs.empty()) outs[0] = ins[0];
          oframe = frame == 0 ? 0 : frames[(size_t)frame].parent;
          oit = frame == 0 ? 0 : frames[(size_t)frame].parent_iter;
          break;
        case K_NEXT:
          if (it > max_iter) { error = "loop did not terminate (node " + nd.name + ")"; return false; }
          if (!outs.empty()) outs[0] = ins[0];
          oit = it + 1;
          break;
        case K_LOOPCOND:
          if (!outs.empty()) outs[0] = ins[0];
          break;
        case K_SWITCH: {
          if (ins.size() < 2) { error = "Switch needs 2 inputs (node " + nd.name + ")"; return false; }
          int p = pred(user, ins[1]);
          if (p < 0) { error = "predicate callback failed at node " + nd.name; return false; }
          if (outs.size() < 2) outs.resize(2, NONE);
          outs[p ? 1 : 0] = ins[0];            // output 0 = false branch, 1 = true branch
          break;
        }
        default: {
          auto t0 = std::chrono::steady_clock::now();
          int rc = exec(user, ni, ins.data(), (int)ins.size(), outs.data(), (int)outs.size());
          auto dt = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
          nd.exec_count++; nd.exec_ns += dt;
          if (rc != 0) { error = "op callback failed at node " + nd.name + " (" + nd.op_name + ")"; return false; }
        }
      }
    }
    for (size_t k = 0; k < nd.outputs.size() && k < outs.size(); ++k)
      if (outs[k] != NONE) put(Key{nd.outputs[k], oframe, oit}, outs[k], produced);
    return true;
  }

  bool drain() {
    std::vector<Key> produced;
    while (!queue.empty()) {
      Key k = queue.front(); queue.pop_front();
      for (int cn : g.vars[(size_t)k.var].consumers) {
        if (!need[(size_t)cn]) continue;
        std::vector<std::pair<int, int>> ctxs;
        if (k.iter >= 0) ctxs.push_back({k.frame, k.iter});
        else for (auto& c : live) if (c.first == k.frame) ctxs.push_back(c);
        for (auto& c : ctxs) {
          produced.clear();
          if (!run(cn, c.first, c.second, &produced)) return false;
          for (auto& p : produced) queue.push_back(p);
        }
      }
    }
    return true;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------- C ABI
RT_API int graph_register(LongType id, const void* buf, LongType len) {
  try {
    auto g = parse((const uint8_t*)buf, len);
    std::lock_guard<std::mutex> lk(g_mu);
    g_graphs[id] = g;
    return 0;
  } catch (...) {
    return rt::report_current_exception("graph_register");       // category code (rt_exceptions.h), message in the error slot
  }
}

RT_API int graph_unregister(LongType id) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_graphs.erase(id) ? 0 : 1;
}

RT_API int graph_registered(LongType id) { return get(id) ? 1 : 0; }
RT_API int graph_count() { std::lock_guard<std::mutex> lk(g_mu); return (int)g_graphs.size(); }

#define GET_OR(ret) auto g = get(id); if (!g) { rt::set_error(rt::CAT_UNKNOWN_GRAPH, rt::unknown_graph_exception(id).what()); return ret; }

RT_API int graph_num_nodes(LongType id) { GET_OR(-1) return (int)g->nodes.size(); }
RT_API int graph_num_vars(LongType id) { GET_OR(-1) return (int)g->vars.size(); }
RT_API int graph_num_layers(LongType id) { GET_OR(-1) return g->n_layers; }
RT_API int graph_layer_size(LongType id, int layer) {
  GET_OR(-1) return layer >= 0 && layer < g->n_layers ? (int)g->layers[(size_t)layer].size() : -1;
}
RT_API int graph_layer_nodes(LongType id, int layer, int* out, int cap) {
  GET_OR(-1)
  if (layer < 0 || layer >= g->n_layers) return -1;
  auto& l = g->layers[(size_t)layer];
  int n = std::min(cap, (int)l.size());
  for (int i = 0; i < n; ++i) out[i] = l[(size_t)i];
  return (int)l.size();
}

RT_API const char* graph_var_name(LongType id, int v) { GET_OR(nullptr) return v >= 0 && v < (int)g->vars.size() ? g->vars[(size_t)v].name.c_str() : nullptr; }
RT_API int graph_var_index(LongType id, const char* name) { GET_OR(-1) return g->var_index(name); }
// out = {vtype, dtype, rank(-1 unknown), producer node, has_array, arr_dtype, byte_order}
RT_API int graph_var_info(LongType id, int v, LongType* out) {
  GET_OR(1)
  if (v < 0 || v >= (int)g->vars.size()) return 1;
  auto& gv = g->vars[(size_t)v];
  out[0] = gv.vtype; out[1] = gv.dtype; out[2] = (gv.shape.empty() && gv.shape_info.empty()) ? -1 : (LongType)gv.shape.size();
  out[3] = gv.producer; out[4] = gv.data || !gv.shape_info.empty() ? 1 : 0; out[5] = gv.arr_dtype; out[6] = gv.byte_order;
  return 0;
}
RT_API int graph_var_shape(LongType id, int v, LongType* out, int cap) {
  GET_OR(-1)
  if (v < 0 || v >= (int)g->vars.size()) return -1;
  auto& s = g->vars[(size_t)v].shape;
  for (int i = 0; i < std::min(cap, (int)s.size()); ++i) out[i] = s[(size_t)i];
  return (int)s.size();
}
RT_API int graph_var_shape_info(LongType id, int v, LongType* out, int cap) {
  GET_OR(-1)
  if (v < 0 || v >= (int)g->vars.size()) return -1;
  auto& s = g->vars[(size_t)v].shape_info;
  for (int i = 0; i < std::min(cap, (int)s.size()); ++i) out[i] = s[(size_t)i];
  return (int)s.size();
}
// zero-copy pointer into the registered buffer (valid until graph_unregister)
RT_API const void* graph_var_data(LongType id, int v, LongType* len) {
  GET_OR(nullptr)
  if (v < 0 || v >= (int)g->vars.size()) return nullptr;
  *len = g->vars[(size_t)v].data_len;
  return g->vars[(size_t)v].data;
}

RT_API const char* graph_node_name(LongType id, int n) { GET_OR(nullptr) return n >= 0 && n < (int)g->nodes.size() ? g->nodes[(size_t)n].name.c_str() : nullptr; }
RT_API const char* graph_node_op(LongType id, int n) { GET_OR(nullptr) return n >= 0 && n < (int)g->nodes.size() ? g->nodes[(size_t)n].op_name.c_str() : nullptr; }
RT_API const char* graph_node_frame(LongType id, int n) { GET_OR(nullptr) return n >= 0 && n < (int)g->nodes.size() ? g->nodes[(size_t)n].frame_name.c_str() : nullptr; }
// out = {opType, opNum, layer, kind, nIn, nOut, nT, nI, nB, nD, nDims, nS, has_scalar, nControl}
RT_API int graph_node_info(LongType id, int n, LongType* out) {
  GET_OR(1)
  if (n < 0 || n >= (int)g->nodes.size()) return 1;
  auto& nd = g->nodes[(size_t)n];
  LongType v[14] = {nd.op_type, nd.op_num, nd.layer, nd.kind, (LongType)nd.inputs.size(), (LongType)nd.outputs.size(),
                    (LongType)nd.targs.size(), (LongType)nd.iargs.size(), (LongType)nd.bargs.size(), (LongType)nd.dargs.size(),
                    (LongType)nd.dims.size(), (LongType)nd.sargs.size(), nd.has_scalar ? 1 : 0, (LongType)nd.control.size()};
  std::memcpy(out, v, sizeof(v));
  return 0;
}
RT_API int graph_node_io(LongType id, int n, int* ins, int* outs) {
  GET_OR(1)
  if (n < 0 || n >= (int)g->nodes.size()) return 1;
  auto& nd = g->nodes[(size_t)n];
  for (size_t i = 0; i < nd.inputs.size(); ++i) ins[i] = nd.inputs[i];
  for (size_t i = 0; i < nd.outputs.size(); ++i) outs[i] = nd.outputs[i];
  return 0;
}
RT_API int graph_node_args(LongType id, int n, double* t, LongType* i, uint8_t* b, uint8_t* d, int* dims, double* scalar) {
  GET_OR(1)
  if (n < 0 || n >= (int)g->nodes.size()) return 1;
  auto& nd = g->nodes[(size_t)n];
  if (t) std::copy(nd.targs.begin(), nd.targs.end(), t);
  if (i) std::copy(nd.iargs.begin(), nd.iargs.end(), i);
  if (b) std::copy(nd.bargs.begin(), nd.bargs.end(), b);
  if (d) std::copy(nd.dargs.begin(), nd.dargs.end(), d);
  if (dims) std::copy(nd.dims.begin(), nd.dims.end(), dims);
  if (scalar) *scalar = nd.scalar;
  return 0;
}
RT_API const char* graph_node_sarg(LongType id, int n, int k) {
  GET_OR(nullptr)
  if (n < 0 || n >= (int)g->nodes.size()) return nullptr;
  auto& s = g->nodes[(size_t)n].sargs;
  return k >= 0 && k < (int)s.size() ? s[(size_t)k].c_str() : nullptr;
}
RT_API int graph_num_outputs(LongType id) { GET_OR(-1) return (int)g->outputs.size(); }
RT_API int graph_output_var(LongType id, int k) { GET_OR(-1) return k >= 0 && k < (int)g->outputs.size() ? g->outputs[(size_t)k] : -1; }
RT_API int graph_num_placeholders(LongType id) { GET_OR(-1) return (int)g->placeholders.size(); }
RT_API const char* graph_placeholder(LongType id, int k) { GET_OR(nullptr) return k >= 0 && k < (int)g->placeholders.size() ? g->placeholders[(size_t)k].c_str() : nullptr; }
RT_API const char* graph_training_config(LongType id) { GET_OR(nullptr) return g->training_config.c_str(); }

// Execute a registered graph.
//   feeds:      (variable index, handle) pairs available in the root frame (placeholders, variables, constants)
//   wanted:     variable indices to compute; results written to out_handles (NONE=-1 if a dead branch)
//   exec / pred callbacks: see the file header.
RT_API int graph_execute(LongType id, const int* feed_vars, const LongType* feed_handles, int n_feeds, const int* wanted, int n_wanted,
                         LongType* out_handles, void* exec_cb, void* pred_cb, void* user, int max_iterations) {
  GET_OR(2)
  std::lock_guard<std::mutex> run_lock(g->mu);
  auto t0 = std::chrono::steady_clock::now();
  Exec ex{*g, (ExecFn)exec_cb, (PredFn)pred_cb, user};
  ex.max_iter = max_iterations > 0 ? max_iterations : 100000;
  size_t N = g->nodes.size();
  // ops required for the requested outputs
  ex.need.assign(N, 0);
  {
    std::vector<int> stack(wanted, wanted + n_wanted);
    std::vector<char> seen(g->vars.size(), 0);
    while (!stack.empty()) {
      int v = stack.back(); stack.pop_back();
      if (v < 0 || v >= (int)g->vars.size() || seen[(size_t)v]) continue;
      seen[(size_t)v] = 1;
      int p = g->vars[(size_t)v].producer;
      if (p < 0 || ex.need[(size_t)p]) continue;
      ex.need[(size_t)p] = 1;
      for (int in : g->nodes[(size_t)p].inputs) stack.push_back(in);
      for (int c : g->nodes[(size_t)p].control) stack.push_back(c);
    }
  }
  bool has_control = false;
  for (size_t i = 0; i < N; ++i) if (ex.need[i] && g->nodes[i].kind != K_OP) has_control = true;
  for (int k = 0; k < n_feeds; ++k) {
    Key key{feed_vars[k], 0, 0};
    ex.vals[key] = feed_handles[k];
    ex.queue.push_back(key);
  }
  bool ok = true;
  std::vector<Key> produced;
  if (!has_control) {
    // static graph: one pass over the topological layers is complete and cheaper than the worklist
    ex.queue.clear();
    for (auto& layer : g->layers) {
      for (int ni : layer) {
        if (!ex.need[(size_t)ni]) continue;
        produced.clear();
        if (!(ok = ex.run(ni, 0, 0, &produced))) break;
      }
      if (!ok) break;
    }
  } else {
    for (size_t i = 0; i < N && ok; ++i)
      if (ex.need[i] && g->nodes[i].inputs.empty()) {
        produced.clear();
        ok = ex.run((int)i, 0, 0, &produced);
        for (auto& p : produced) ex.queue.push_back(p);
      }
    ok = ok && ex.drain();
    // fixpoint over live (frame, iteration) contexts for values that cross frame boundaries
    while (ok) {
      bool progressed = false;
      auto ctxs = std::vector<std::pair<int, int>>(ex.live.begin(), ex.live.end());
      for (auto& c : ctxs) {
        for (size_t i = 0; i < N && ok; ++i) {
          if (!ex.need[i] || ex.executed.count(Key{(int)i, c.first, c.second})) continue;
          auto& nd = g->nodes[i];
          if (!(c.first == 0 && c.second == 0)) {
            bool touches = false;
            for (int in : nd.inputs)
              if (ex.vals.count(Key{in, c.first, c.second}) || ex.vals.count(Key{in, c.first, -1})) { touches = true; break; }
            if (!touches) continue;
          }
          produced.clear();
          ok = ex.run((int)i, c.first, c.second, &produced);
          if (!produced.empty()) {
            for (auto& p : produced) ex.queue.push_back(p);
            ok = ok && ex.drain();
            progressed = true;
          }
        }
        if (!ok) break;
      }
      if (!progressed) break;
    }
  }
  if (!ok) {
    rt::set_error(rt::CAT_GRAPH_EXECUTION, std::string("graph_execute: ") + rt::graph_execution_exception(id, ex.error).what());
    return rt::CAT_GRAPH_EXECUTION;
  }
  for (int k = 0; k < n_wanted; ++k) {
    auto it = ex.vals.find(Key{wanted[k], 0, 0});
    out_handles[k] = it == ex.vals.end() ? NONE : it->second;
  }
  g->executions++;
  g->total_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

// profiling (GraphProfile / NodeProfile): out = {exec_count, total_ns}
RT_API int graph_node_profile(LongType id, int n, LongType* out) {
  GET_OR(1)
  if (n < 0 || n >= (int)g->nodes.size()) return 1;
  out[0] = g->nodes[(size_t)n].exec_count; out[1] = g->nodes[(size_t)n].exec_ns;
  return 0;
}
RT_API int graph_profile(LongType id, LongType* out) { GET_OR(1) out[0] = g->executions; out[1] = g->total_ns; return 0; }
RT_API int graph_profile_reset(LongType id) {
  GET_OR(1)
  g->executions = g->total_ns = 0;
  for (auto& nd : g->nodes) nd.exec_count = nd.exec_ns = 0;
  return 0;
}
