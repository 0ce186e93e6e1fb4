// Native smoke test of the host runtime C ABI (reference: libnd4j/tests_cpu gtest layers_tests — shape / ops registry).
#include <cstdint>
#include <cstdio>
#include <cstring>

extern "C" {
int nd4j_shape_info_length(int rank);
int nd4j_shape_buffer(int rank, const int64_t* shape, char order, int64_t dtype, int64_t* out);
int64_t nd4j_shape_length(const int64_t* si);
int ops_count();
int64_t ops_hash(const char* name);
const char* ops_name_by_hash(int64_t h);
int graph_count();
}

#define CHECK(c)                                          \
  if (!(c)) {                                             \
    std::printf("FAILED: %s (line %d)\n", #c, __LINE__);  \
    return 1;                                             \
  }

int main() {
  int64_t shape[3] = {2, 3, 4}, si[16];
  CHECK(nd4j_shape_info_length(3) == 10);
  CHECK(nd4j_shape_buffer(3, shape, 'c', 5, si) == 0);
  CHECK(si[0] == 3 && si[4] == 12 && si[5] == 4 && si[6] == 1);
  CHECK(nd4j_shape_length(si) == 24);
  CHECK(ops_count() > 400);
  const char* n = ops_name_by_hash(ops_hash("conv2d"));
  CHECK(n && std::strcmp(n, "conv2d") == 0);
  CHECK(graph_count() == 0);
  std::printf("rt_smoke OK\n");
  return 0;
}
