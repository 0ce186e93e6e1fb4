// Typed exceptions of the host runtime + stack-trace capture.
//
// Reference: libnd4j/include/exceptions/{cuda_exception,allocation_exception,datatype_exception,graph_exception,
// graph_execution_exception,graph_exists_exception,no_results_exception,unknown_graph_exception}.h (each with `build(...)` message
// factories), the THROW_EXCEPTION macro and the vendored backward.hpp stack-trace printer. Here every entry point of the C ABI turns
// a thrown exception into (category code, message, optional backtrace) in the per-thread error slot (rt::report), so a foreign caller
// sees WHICH kind of failure happened (nd4j_error_category) and, when enabled (ND4J_BACKTRACE=1 / nd4j_backtrace_toggle), where.
#pragma once
#include "rt_common.h"

#include <stdexcept>

namespace rt {

// category codes (stable: nd4j/exceptions.py maps them to Python classes)
enum ErrorCategory { CAT_NONE = 0, CAT_RUNTIME = 100, CAT_DEVICE = 101, CAT_ALLOCATION = 102, CAT_DATATYPE = 103, CAT_GRAPH = 110,
                     CAT_GRAPH_EXECUTION = 111, CAT_GRAPH_EXISTS = 112, CAT_NO_RESULTS = 113, CAT_UNKNOWN_GRAPH = 114,
                     CAT_INVALID_ARGUMENT = 120, CAT_OUT_OF_RANGE = 121, CAT_BAD_ALLOC = 122 };

struct device_exception : std::runtime_error {          // reference: cuda_exception
  int error_code;
  device_exception(const std::string& m, int code = 0) : std::runtime_error(m), error_code(code) {}
  static device_exception build(const std::string& m, int code) { return device_exception(m + "; Error code: [" + std::to_string(code) + "]", code); }
};
struct allocation_exception : std::runtime_error {
  using std::runtime_error::runtime_error;
  static allocation_exception build(const std::string& m, LongType bytes) {
    return allocation_exception(m + "; Requested bytes: [" + std::to_string(bytes) + "]");
  }
  static allocation_exception build(const std::string& m, LongType limit, LongType bytes) {
    return allocation_exception(m + "; Limit bytes: [" + std::to_string(limit) + "]; Requested bytes: [" + std::to_string(bytes) + "]");
  }
};
struct datatype_exception : std::runtime_error {
  using std::runtime_error::runtime_error;
  static datatype_exception build(const std::string& m, const std::string& actual) { return datatype_exception(m + "; Actual: [" + actual + "]"); }
  static datatype_exception build(const std::string& m, const std::string& expected, const std::string& actual) {
    return datatype_exception(m + "; Expected: [" + expected + "]; Actual: [" + actual + "]");
  }
};
struct graph_exception : std::runtime_error {
  LongType graph_id;
  std::string description;
  graph_exception(const std::string& m, LongType id, const std::string& d = "") : std::runtime_error(m), graph_id(id), description(d) {}
  virtual int category() const { return CAT_GRAPH; }
};
struct graph_execution_exception : graph_exception {
  explicit graph_execution_exception(LongType id, const std::string& why = "")
      : graph_exception("Caught exception during graph execution: [" + std::to_string(id) + "]" + (why.empty() ? "" : ": " + why), id, why) {}
  int category() const override { return CAT_GRAPH_EXECUTION; }
};
struct graph_exists_exception : graph_exception {
  explicit graph_exists_exception(LongType id) : graph_exception("Graph with given ID already exists: [" + std::to_string(id) + "]", id) {}
  int category() const override { return CAT_GRAPH_EXISTS; }
};
struct no_results_exception : graph_exception {
  explicit no_results_exception(LongType id) : graph_exception("Got no results after graph execution: [" + std::to_string(id) + "]", id) {}
  int category() const override { return CAT_NO_RESULTS; }
};
struct unknown_graph_exception : graph_exception {
  explicit unknown_graph_exception(LongType id) : graph_exception("Unknown graph: [" + std::to_string(id) + "]", id) {}
  int category() const override { return CAT_UNKNOWN_GRAPH; }
};

// category of the exception in flight (call inside a catch block); records (code, "<where>: <what>", backtrace of the throw site
// when capture is on) in the error slot and returns the category code
int report_current_exception(const char* where);

std::string capture_backtrace(int skip = 1, int max_frames = 48);     // demangled, one frame per line
bool backtrace_enabled();

// run fn; on an exception fill the error slot and return its category code, 0 otherwise
template <typename F>
int guarded(const char* where, F&& fn) {
  try { fn(); return 0; } catch (...) { return report_current_exception(where); }
}

}  // namespace rt

// throw with the stack captured at the throw site (the role of the reference's THROW_EXCEPTION)
#define RT_THROW(ex) do { rt::note_throw_site(); throw ex; } while (0)
namespace rt { void note_throw_site(); }
