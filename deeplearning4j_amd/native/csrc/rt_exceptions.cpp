// Exception -> error-slot mapping, stack-trace capture, crash handler. See rt_exceptions.h.
#include "rt_exceptions.h"

#include <cxxabi.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <new>
#include <sstream>

namespace rt {

namespace {
std::atomic<int> g_bt_enabled{-1};            // -1: read ND4J_BACKTRACE on first use
thread_local std::string t_throw_site;        // stack captured by RT_THROW, consumed by the next report
thread_local std::string t_last_backtrace;
thread_local int t_last_category = 0;
}

bool backtrace_enabled() {
  int v = g_bt_enabled.load();
  if (v < 0) {
    const char* e = std::getenv("ND4J_BACKTRACE");
    v = (e && *e && *e != '0') ? 1 : 0;
    g_bt_enabled.store(v);
  }
  return v == 1;
}

std::string capture_backtrace(int skip, int max_frames) {
  void* frames[64];
  const int want = std::min(max_frames + skip, 64);
  const int n = ::backtrace(frames, want);
  std::ostringstream os;
  for (int i = skip; i < n; ++i) {
    Dl_info info;
    os << "#" << (i - skip) << " ";
    if (dladdr(frames[i], &info) && info.dli_sname) {
      int status = 0;
      char* dem = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
      os << (status == 0 && dem ? dem : info.dli_sname) << " + " << ((char*)frames[i] - (char*)info.dli_saddr);
      std::free(dem);
      if (info.dli_fname) { const char* b = strrchr(info.dli_fname, '/'); os << " (" << (b ? b + 1 : info.dli_fname) << ")"; }
    } else if (dladdr(frames[i], &info) && info.dli_fname) {
      const char* b = strrchr(info.dli_fname, '/');
      os << frames[i] << " (" << (b ? b + 1 : info.dli_fname) << " + " << ((char*)frames[i] - (char*)info.dli_fbase) << ")";
    } else {
      os << frames[i];
    }
    os << "\n";
  }
  return os.str();
}

void note_throw_site() {
  if (backtrace_enabled()) t_throw_site = capture_backtrace(2);
}

int report_current_exception(const char* where) {
  int cat = CAT_RUNTIME;
  std::string msg;
  try {
    throw;
  } catch (const graph_exception& e) { cat = e.category(); msg = e.what();
  } catch (const device_exception& e) { cat = CAT_DEVICE; msg = e.what();
  } catch (const allocation_exception& e) { cat = CAT_ALLOCATION; msg = e.what();
  } catch (const datatype_exception& e) { cat = CAT_DATATYPE; msg = e.what();
  } catch (const std::bad_alloc& e) { cat = CAT_BAD_ALLOC; msg = e.what();
  } catch (const std::out_of_range& e) { cat = CAT_OUT_OF_RANGE; msg = e.what();
  } catch (const std::invalid_argument& e) { cat = CAT_INVALID_ARGUMENT; msg = e.what();
  } catch (const std::exception& e) { cat = CAT_RUNTIME; msg = e.what();
  } catch (...) { cat = CAT_RUNTIME; msg = "unknown C++ exception"; }
  set_error(cat, std::string(where) + ": " + msg);
  t_last_category = cat;
  if (backtrace_enabled()) {
    t_last_backtrace = t_throw_site.empty() ? capture_backtrace(1) : t_throw_site;
  } else {
    t_last_backtrace.clear();
  }
  t_throw_site.clear();
  return cat;
}

}  // namespace rt

using namespace rt;

RT_API void nd4j_backtrace_toggle(int on) { g_bt_enabled.store(on ? 1 : 0); }
RT_API int nd4j_backtrace_enabled() { return backtrace_enabled() ? 1 : 0; }
RT_API const char* nd4j_last_error_backtrace() { return t_last_backtrace.c_str(); }
// "runtime" / "device" / "allocation" / "datatype" / "graph" / ... for a code returned by a guarded entry point (0 -> "")
RT_API const char* nd4j_error_category(int code) {
  switch (code) {
    case CAT_RUNTIME: return "runtime"; case CAT_DEVICE: return "device"; case CAT_ALLOCATION: return "allocation";
    case CAT_DATATYPE: return "datatype"; case CAT_GRAPH: return "graph"; case CAT_GRAPH_EXECUTION: return "graph_execution";
    case CAT_GRAPH_EXISTS: return "graph_exists"; case CAT_NO_RESULTS: return "no_results"; case CAT_UNKNOWN_GRAPH: return "unknown_graph";
    case CAT_INVALID_ARGUMENT: return "invalid_argument"; case CAT_OUT_OF_RANGE: return "out_of_range"; case CAT_BAD_ALLOC: return "bad_alloc";
    default: return "";
  }
}
// current stack of the caller, demangled (debugging aid: the reference prints it through backward.hpp)
RT_API LongType nd4j_current_backtrace(char* out, LongType cap) {
  const std::string s = capture_backtrace(1);
  if (out && cap > 0) { const LongType n = std::min<LongType>(cap - 1, (LongType)s.size()); memcpy(out, s.data(), n); out[n] = 0; }
  return (LongType)s.size();
}

// exercises the whole path (typed throw inside nested frames -> guarded entry point -> slot); kind: 1 device, 2 allocation (limit form),
// 3 datatype, 4 graph execution, 5 graph exists, 6 no results, 7 unknown graph, 8 invalid_argument, 9 out_of_range, 10 bad_alloc, else runtime
namespace {
__attribute__((noinline)) void raise_kind(int kind, LongType arg) {
  switch (kind) {
    case 1: RT_THROW(device_exception::build("hipMemcpyAsync failed", (int)arg));
    case 2: RT_THROW(allocation_exception::build("Requested amount exceeds device limits", (LongType)1 << 30, arg));
    case 3: RT_THROW(datatype_exception::build("matmul operands must share one data type", "FLOAT32", "BFLOAT16"));
    case 4: RT_THROW(graph_execution_exception(arg, "node 7 (conv2d) has no executor"));
    case 5: RT_THROW(graph_exists_exception(arg));
    case 6: RT_THROW(no_results_exception(arg));
    case 7: RT_THROW(unknown_graph_exception(arg));
    case 8: RT_THROW(std::invalid_argument("rank must be positive"));
    case 9: RT_THROW(std::out_of_range("dimension 5 for rank 2"));
    case 10: RT_THROW(std::bad_alloc());
    default: RT_THROW(std::runtime_error("generic failure"));
  }
}
__attribute__((noinline)) void raise_nested(int kind, LongType arg, int depth) {
  if (depth > 0) { raise_nested(kind, arg, depth - 1); asm volatile("" ::: "memory"); return; }
  raise_kind(kind, arg);
}
}
RT_API int nd4j_raise_for_test(int kind, LongType arg) {
  return guarded("nd4j_raise_for_test", [&] { raise_nested(kind, arg, 2); });
}

// ---- crash handler: SIGSEGV / SIGBUS / SIGFPE / SIGILL / SIGABRT print the native stack to stderr, then the default action runs
namespace {
void crash_handler(int sig) {
  static const char head[] = "\n*** libnd4jrocm_rt: fatal signal, native backtrace follows ***\n";
  (void)!write(STDERR_FILENO, head, sizeof(head) - 1);
  void* frames[64];
  const int n = ::backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, STDERR_FILENO);
  signal(sig, SIG_DFL);
  raise(sig);
}
}
RT_API void nd4j_install_crash_handler() {
  void* warm[2];
  (void)::backtrace(warm, 2);                 // loads libgcc now: not safe to do lazily inside a signal handler
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = crash_handler;
  sigemptyset(&sa.sa_mask);
  sa.sa_flags = SA_RESETHAND | SA_NODEFER;
  for (int s : {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT}) sigaction(s, &sa, nullptr);
}
