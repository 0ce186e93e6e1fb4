// Sanitizer workload for the host runtime (reference: SD_SANITIZE builds of libnd4j run against tests_cpu): built with
// -fsanitize=address,undefined (host code only) it drives the parts of the C ABI that take sizes / buffers from outside:
// shape + TAD math, workspace arenas (bump, spill, rewind, learning), threshold / bitmap codecs, npy IO, CSV parsing, RNG fills,
// the op registry and — the interesting one — the FlatBuffers graph reader, with every .fb file given on the command line parsed
// whole AND truncated at several lengths (a truncated or corrupt file must be rejected, never read out of bounds).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

typedef long long LongType;
extern "C" {
int nd4j_shape_info_length(int rank);
int nd4j_shape_buffer(int rank, const LongType* shape, char order, LongType dtype, LongType* out);
LongType nd4j_shape_length(const LongType* si);
void nd4j_index2coords(LongType index, const LongType* si, LongType* coords);
LongType nd4j_coords2offset(const LongType* si, const LongType* coords);
int nd4j_permute(const LongType* si, const int* perm, LongType* out);
int nd4j_broadcast_shape(int ra, const LongType* a, int rb, const LongType* b, LongType* out);
const void* nd4j_tad_pack(const LongType* si, const int* dims, int ndims);
LongType nd4j_tad_num(const void* p);
const LongType* nd4j_tad_offsets(const void* p);
void* ws_create(int learning_policy, double over_allocation, LongType cycles_before_init);
void ws_destroy(void* h);
int ws_bind(void* h, int kind, uint64_t base, LongType capacity);
void ws_scope_in(void* h);
void ws_scope_out(void* h, LongType* want_host, LongType* want_device);
uint64_t ws_alloc(void* h, int kind, LongType bytes);
void ws_rewind(void* h, int kind);
LongType nd4j_threshold_count_f32(const float* x, LongType n, float tau);
LongType nd4j_threshold_encode_f32(float* x, LongType n, float tau, int* out, LongType max_elems);
int nd4j_threshold_decode_f32(const int* enc, float* z, LongType z_len);
LongType nd4j_bitmap_encode_f32(float* x, LongType n, float tau, int* out);
int nd4j_bitmap_decode_f32(const int* enc, float* z, LongType z_len);
int nd4j_npy_save(const char* path, const void* data, const LongType* shape, int rank, const char* descr, int elem_size, int fortran);
LongType nd4j_npy_header(const char* path, LongType* shape_out, int* rank_out, char* descr_out, int* fortran_out);
int nd4j_npy_read_data(const char* path, LongType offset, void* dst, LongType bytes);
void* dv_csv_parse(const char* path, char delim, int skip_lines);
LongType dv_rows(void* h);
LongType dv_cols(void* h);
void dv_copy(void* h, float* dst);
void dv_free(void* h);
void rng_fill_uniform_f32(float* z, LongType n, uint64_t key, uint64_t stream, uint64_t offset, float lo, float hi);
void rng_fill_gaussian_f32(float* z, LongType n, uint64_t key, uint64_t stream, uint64_t offset, float mean, float stddev);
int ops_count();
const char* ops_name_at(int i);
LongType ops_hash(const char* name);
const char* ops_name_by_hash(LongType h);
int ops_describe(const char* name, int* out, LongType* hash_out);
int graph_register(LongType id, const void* buf, LongType len);
int graph_unregister(LongType id);
int graph_num_nodes(LongType id);
int graph_num_vars(LongType id);
int graph_num_layers(LongType id);
const char* graph_node_op(LongType id, int n);
const char* graph_var_name(LongType id, int v);
int graph_var_shape(LongType id, int v, LongType* out, int cap);
const void* graph_var_data(LongType id, int v, LongType* len);
void nd4j_clear_error();
LongType ops_hash(const char* name);
struct OpaqueContext;
struct OpaqueShapeList;
OpaqueContext* createGraphContext(int nodeId);
void deleteGraphContext(OpaqueContext* ctx);
void setGraphContextInputArrays(OpaqueContext* ctx, int n, void** buffer, void** shapeInfo, void** specialBuffer, void** specialShapeInfo);
void setGraphContextOutputArrays(OpaqueContext* ctx, int n, void** buffer, void** shapeInfo, void** specialBuffer, void** specialShapeInfo);
void setGraphContextIArguments(OpaqueContext* ctx, LongType* arguments, int n);
void setGraphContextTArguments(OpaqueContext* ctx, double* arguments, int n);
int execCustomOp2(void** extra, LongType hash, OpaqueContext* ctx);
OpaqueShapeList* calculateOutputShapes2(void** extra, LongType hash, void** inputBuffers, void** inputShapes, int numInputShapes, double* tArgs,
                                        int numTArgs, LongType* iArgs, int numIArgs, bool* bArgs, int numBArgs, int* dArgs, int numDArgs);
void deleteShapeList(OpaqueShapeList* list);
void* ctxInputBuffer(OpaqueContext* ctx, int k, int special);
const LongType* ctxInputShapeInfo(OpaqueContext* ctx, int k);
void* ctxOutputBuffer(OpaqueContext* ctx, int k, int special);
double ctxTArg(OpaqueContext* ctx, int k);
LongType ctxIArg(OpaqueContext* ctx, int k);
int ctxBArg(OpaqueContext* ctx, int k);
}

#define CHECK(c)                                          \
  if (!(c)) {                                             \
    std::printf("FAILED: %s (line %d)\n", #c, __LINE__);  \
    return 1;                                             \
  }

int main(int argc, char** argv) {
  // ---- shape / TAD
  LongType shape[4] = {3, 4, 5, 6}, si[24], out[24], coords[4];
  CHECK(nd4j_shape_buffer(4, shape, 'c', 5, si) == 0);
  for (LongType i = 0; i < nd4j_shape_length(si); i += 7) {
    nd4j_index2coords(i, si, coords);
    CHECK(nd4j_coords2offset(si, coords) == i);
  }
  int perm[4] = {3, 1, 0, 2};
  CHECK(nd4j_permute(si, perm, out) == 0 && out[1] == 6 && out[3] == 3);
  LongType a[3] = {4, 1, 6}, b[2] = {5, 1}, bc[4];
  CHECK(nd4j_broadcast_shape(3, a, 2, b, bc) == 3 && bc[1] == 5 && bc[2] == 6);
  int dims[2] = {1, 3};
  const void* tp = nd4j_tad_pack(si, dims, 2);
  CHECK(tp && nd4j_tad_num(tp) == 15 && nd4j_tad_offsets(tp)[14] >= 0);

  // ---- workspace arena: fill, spill (returns 0), scope out asks for a bigger arena, rebind, cyclic rewind
  std::vector<char> arena(1 << 16), arena2(1 << 20);
  void* ws = ws_create(1, 0.25, 0);
  CHECK(ws_bind(ws, 0, (uint64_t)(uintptr_t)arena.data(), (LongType)arena.size()) == 0);
  ws_scope_in(ws);
  int spilled = 0;
  for (int i = 0; i < 100; ++i) {
    const uint64_t p = ws_alloc(ws, 0, 1000 + i);
    if (!p) { ++spilled; continue; }
    CHECK(p >= (uint64_t)(uintptr_t)arena.data() && p + 1000 + i <= (uint64_t)(uintptr_t)arena.data() + arena.size());
    std::memset((void*)(uintptr_t)p, 0xAB, 1000 + i);
  }
  CHECK(spilled > 0);
  LongType wh = 0, wd = 0;
  ws_scope_out(ws, &wh, &wd);
  CHECK(wh > (LongType)arena.size());
  CHECK(ws_bind(ws, 0, (uint64_t)(uintptr_t)arena2.data(), (LongType)arena2.size()) == 0);
  ws_scope_in(ws);
  for (int i = 0; i < 100; ++i) CHECK(ws_alloc(ws, 0, 1000 + i) != 0);
  ws_rewind(ws, 0);
  CHECK(ws_alloc(ws, 0, 64) == (uint64_t)(uintptr_t)arena2.data());
  ws_scope_out(ws, &wh, &wd);
  ws_destroy(ws);

  // ---- RNG + codecs round trip (encode mutates x into the residual)
  const LongType n = 100003;
  std::vector<float> x(n), orig, z(n, 0.f);
  rng_fill_gaussian_f32(x.data(), n, 42, 0, 0, 0.f, 1.f);
  orig = x;
  const float tau = 1.5f;
  const LongType cnt = nd4j_threshold_count_f32(x.data(), n, tau);
  CHECK(cnt > 0 && cnt < n);
  std::vector<int> enc((size_t)cnt + 8);
  CHECK(nd4j_threshold_encode_f32(x.data(), n, tau, enc.data(), cnt) == cnt);
  CHECK(nd4j_threshold_decode_f32(enc.data(), z.data(), n) == 0);
  for (LongType i = 0; i < n; ++i) CHECK(std::fabs(z[i] + x[i] - orig[i]) < 1e-5f);
  x = orig;
  std::vector<int> bm((size_t)(n / 16 + 8));
  std::fill(z.begin(), z.end(), 0.f);
  CHECK(nd4j_bitmap_encode_f32(x.data(), n, tau, bm.data()) > 0);
  CHECK(nd4j_bitmap_decode_f32(bm.data(), z.data(), n) == 0);
  for (LongType i = 0; i < n; i += 11) CHECK(std::fabs(z[i] + x[i] - orig[i]) < 1e-5f);
  // an encoded message that claims more elements than the target holds must be refused, not written past the end
  std::vector<float> small(16, 0.f);
  CHECK(nd4j_threshold_decode_f32(enc.data(), small.data(), 16) != 0);
  nd4j_clear_error();

  // ---- npy + csv IO
  const char* dir = std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp";
  const std::string npy = std::string(dir) + "/rt_sanitize.npy", csv = std::string(dir) + "/rt_sanitize.csv";
  LongType nshape[2] = {37, 11}, rshape[8];
  CHECK(nd4j_npy_save(npy.c_str(), orig.data(), nshape, 2, "<f4", 4, 0) == 0);
  int rank = 0, fortran = 0;
  char descr[16] = {0};
  const LongType off = nd4j_npy_header(npy.c_str(), rshape, &rank, descr, &fortran);
  CHECK(off > 0 && rank == 2 && rshape[0] == 37 && rshape[1] == 11);
  std::vector<float> back(37 * 11);
  CHECK(nd4j_npy_read_data(npy.c_str(), off, back.data(), 37 * 11 * 4) == 0);
  CHECK(std::memcmp(back.data(), orig.data(), back.size() * 4) == 0);
  {
    std::ofstream f(csv);
    f << "a,b,c\n";
    for (int r = 0; r < 50; ++r) f << r << "," << r * 0.5 << "," << -r << "\n";
    f << "7,8\n";            // ragged last row
  }
  void* pc = dv_csv_parse(csv.c_str(), ',', 1);
  if (pc) {
    std::vector<float> m((size_t)(dv_rows(pc) * dv_cols(pc)));
    dv_copy(pc, m.data());
    dv_free(pc);
  }
  nd4j_clear_error();

  // ---- op-execution ABI with hostile arguments: out-of-range reduce dimensions, rank > 8 "device" broadcast, bad accessor indices
  {
    const LongType F_FLOAT = 8192;
    LongType xs[2] = {3, 4}, xsi[16], osi[16], o1[1] = {3};
    CHECK(nd4j_shape_buffer(2, xs, 'c', F_FLOAT, xsi) == 0 && nd4j_shape_buffer(1, o1, 'c', F_FLOAT, osi) == 0);
    std::vector<float> xv(12, 1.f), ov(3, 0.f);
    void* ib[1] = {xv.data()}; void* isi[1] = {xsi}; void* ob[1] = {ov.data()}; void* oshp[1] = {osi};
    OpaqueContext* c = createGraphContext(1);
    setGraphContextInputArrays(c, 1, ib, isi, nullptr, nullptr);
    setGraphContextOutputArrays(c, 1, ob, oshp, nullptr, nullptr);
    LongType good[1] = {1}, bad_hi[1] = {2}, bad_lo[1] = {-3}, bad_far[1] = {1000000};
    setGraphContextIArguments(c, good, 1);
    CHECK(execCustomOp2(nullptr, ops_hash("reduce_sum"), c) == 0 && ov[0] == 4.f);
    for (LongType* bad : {bad_hi, bad_lo, bad_far}) {
      setGraphContextIArguments(c, bad, 1);
      CHECK(execCustomOp2(nullptr, ops_hash("reduce_sum"), c) != 0);
      OpaqueShapeList* sl = calculateOutputShapes2(nullptr, ops_hash("reduce_sum"), nullptr, isi, 1, nullptr, 0, bad, 1, nullptr, 0, nullptr, 0);
      CHECK(sl == nullptr);
    }
    // accessor indices outside the context
    CHECK(ctxInputBuffer(c, 5, 0) == nullptr && ctxInputBuffer(c, -1, 1) == nullptr && ctxInputShapeInfo(c, 9) == nullptr);
    CHECK(ctxOutputBuffer(c, 3, 0) == nullptr && ctxTArg(c, 0) == 0.0 && ctxIArg(c, 7) == 0 && ctxBArg(c, 0) == 0);
    deleteGraphContext(c);
    // rank-9 broadcast add with "device" pointers: must be declined before any stride table is written (no device is touched)
    LongType s9[9] = {2, 1, 2, 1, 2, 1, 2, 1, 2}, t9[9] = {1, 2, 1, 2, 1, 2, 1, 2, 1}, o9[9] = {2, 2, 2, 2, 2, 2, 2, 2, 2};
    LongType a9[64], b9[64], z9[64];
    CHECK(nd4j_shape_buffer(9, s9, 'c', F_FLOAT, a9) == 0 && nd4j_shape_buffer(9, t9, 'c', F_FLOAT, b9) == 0 && nd4j_shape_buffer(9, o9, 'c', F_FLOAT, z9) == 0);
    std::vector<float> av(32, 1.f), bv(16, 2.f), zv(512, 0.f);
    void* fake_dev_in[2] = {(void*)0x1000, (void*)0x2000}; void* fake_dev_out[1] = {(void*)0x3000};
    void* hin[2] = {av.data(), bv.data()}; void* sin_[2] = {a9, b9}; void* hout[1] = {zv.data()}; void* sout[1] = {z9};
    OpaqueContext* d = createGraphContext(2);
    setGraphContextInputArrays(d, 2, nullptr, sin_, fake_dev_in, nullptr);
    setGraphContextOutputArrays(d, 1, nullptr, sout, fake_dev_out, nullptr);
    CHECK(execCustomOp2(nullptr, ops_hash("add"), d) != 0);
    // the same shapes on the host are computed (rank 9 < MAX_RANK)
    setGraphContextInputArrays(d, 2, hin, sin_, nullptr, nullptr);
    setGraphContextOutputArrays(d, 1, hout, sout, nullptr, nullptr);
    CHECK(execCustomOp2(nullptr, ops_hash("add"), d) == 0 && zv[0] == 3.f && zv[511] == 3.f);
    deleteGraphContext(d);
    nd4j_clear_error();
  }

  // ---- op registry
  const int nops = ops_count();
  CHECK(nops > 400);
  for (int i = 0; i < nops; ++i) {
    const char* nm = ops_name_at(i);
    CHECK(nm != nullptr);
    int d[8];
    LongType h = 0;
    ops_describe(nm, d, &h);
    const char* back_name = ops_name_by_hash(ops_hash(nm));
    CHECK(back_name != nullptr);
  }

  // ---- FlatBuffers graphs: whole files and truncated prefixes
  int parsed = 0, rejected = 0;
  for (int k = 1; k < argc; ++k) {
    std::ifstream f(argv[k], std::ios::binary);
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.empty()) continue;
    if (graph_register(1000 + k, buf.data(), (LongType)buf.size()) == 0) {
      ++parsed;
      const int nn = graph_num_nodes(1000 + k), nv = graph_num_vars(1000 + k);
      for (int i = 0; i < nn; ++i) CHECK(graph_node_op(1000 + k, i) != nullptr);
      for (int v = 0; v < nv; ++v) {
        LongType shp[32], len = 0;
        graph_var_name(1000 + k, v);
        graph_var_shape(1000 + k, v, shp, 32);
        const char* data = (const char*)graph_var_data(1000 + k, v, &len);
        volatile char sink = 0;
        for (LongType j = 0; data && j < len; j += 97) sink = data[j];       // touch the constant's bytes
        (void)sink;
      }
      CHECK(graph_num_layers(1000 + k) >= 0);
      CHECK(graph_unregister(1000 + k) == 0);
    } else {
      ++rejected;
    }
    nd4j_clear_error();
    const size_t cuts[6] = {0, 3, 7, buf.size() / 3, buf.size() / 2, buf.size() - 1};
    for (size_t c : cuts) {
      std::vector<char> part(buf.begin(), buf.begin() + (std::ptrdiff_t)c);      // exact-size heap copy: any over-read trips ASan
      if (graph_register(5000 + k, part.data(), (LongType)part.size()) == 0) graph_unregister(5000 + k);
      nd4j_clear_error();
    }
  }
  std::printf("rt_sanitize OK: %d graphs parsed, %d rejected, %d ops, codecs/workspace/io clean\n", parsed, rejected, nops);
  return 0;
}
