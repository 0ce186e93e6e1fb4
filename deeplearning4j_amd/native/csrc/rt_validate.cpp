// Launch validation against the gfx950 (MI355X) execution limits + occupancy estimate.
//
// Reference role: libnd4j/include/execution/cuda/DeviceValidator.cu (checks a planned launch against the CUDA device's
// limits: threads per block, shared memory, registers, grid) — here the table is the CDNA4 one: 64-lane wavefronts,
// 1024 threads (16 waves) per workgroup, 160 KB LDS per CU, 512 VGPRs per lane split across resident waves of a SIMD,
// 4 SIMDs per CU, 256 CUs in 8 XCDs.
#include "rt_common.h"

namespace {
constexpr int kWave = 64;
constexpr int kMaxThreads = 1024;
constexpr int kLdsPerCu = 160 * 1024;
constexpr int kVgprsPerSimdLane = 512;
constexpr int kSimdsPerCu = 4;
constexpr int kMaxWavesPerSimd = 8;
constexpr int kNumCu = 256;
constexpr LongType kMaxGridX = 2147483647LL;
constexpr int kMaxGridYZ = 65535;
}  // namespace

// returns 0 if the launch is valid; otherwise a bit mask: 1 block size, 2 LDS, 4 grid, 8 VGPRs, 16 block not a multiple of the wave
// out (optional, 6 values) = {waves_per_block, blocks_per_cu, waves_per_simd, occupancy_percent, full_chip_waves(blocks to fill), lds_limited}
RT_API int device_validate_launch(LongType gx, int gy, int gz, int bx, int by, int bz, LongType smem_bytes, int vgprs_per_thread,
                                  LongType* out) {
  int bad = 0;
  const LongType threads = (LongType)bx * by * bz;
  if (threads <= 0 || threads > kMaxThreads) bad |= 1;
  if (smem_bytes < 0 || smem_bytes > kLdsPerCu) bad |= 2;
  if (gx <= 0 || gx > kMaxGridX || gy <= 0 || gy > kMaxGridYZ || gz <= 0 || gz > kMaxGridYZ) bad |= 4;
  if (vgprs_per_thread > kVgprsPerSimdLane) bad |= 8;
  if (threads > 0 && threads % kWave != 0) bad |= 16;   // legal, but wastes lanes: reported as a warning bit
  if (out) {
    const LongType waves_per_block = threads > 0 ? (threads + kWave - 1) / kWave : 0;
    LongType by_lds = smem_bytes > 0 ? kLdsPerCu / smem_bytes : 64;
    const int v = vgprs_per_thread > 0 ? vgprs_per_thread : 64;
    LongType waves_simd_by_vgpr = kVgprsPerSimdLane / v;
    if (waves_simd_by_vgpr > kMaxWavesPerSimd) waves_simd_by_vgpr = kMaxWavesPerSimd;
    const LongType waves_cu_by_vgpr = waves_simd_by_vgpr * kSimdsPerCu;
    LongType by_vgpr = waves_per_block > 0 ? waves_cu_by_vgpr / waves_per_block : 0;
    LongType by_waves = waves_per_block > 0 ? (LongType)(kMaxWavesPerSimd * kSimdsPerCu) / waves_per_block : 0;
    LongType blocks_per_cu = by_lds < by_vgpr ? by_lds : by_vgpr;
    if (by_waves < blocks_per_cu) blocks_per_cu = by_waves;
    if (blocks_per_cu < 0) blocks_per_cu = 0;
    const LongType waves_per_simd = blocks_per_cu * waves_per_block / kSimdsPerCu;
    out[0] = waves_per_block;
    out[1] = blocks_per_cu;
    out[2] = waves_per_simd;
    out[3] = 100 * waves_per_simd / kMaxWavesPerSimd;
    out[4] = blocks_per_cu * kNumCu;
    out[5] = (by_lds <= by_vgpr && by_lds <= by_waves) ? 1 : 0;
  }
  if (bad & ~16) rt::set_error(10, "launch exceeds gfx950 limits (mask " + std::to_string(bad) + ")");
  return bad;
}

RT_API void device_limits(LongType* out) {
  out[0] = kWave; out[1] = kMaxThreads; out[2] = kLdsPerCu; out[3] = kVgprsPerSimdLane; out[4] = kSimdsPerCu;
  out[5] = kMaxWavesPerSimd; out[6] = kNumCu; out[7] = 8;
}
