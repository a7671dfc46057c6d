// cu_mask_probe.hip — which compute units a stream created with hipExtStreamCreateWithCUMask really runs on (gfx950: 8 XCDs x 32 CUs):
// waves record HW_REG_HW_ID / HW_REG_XCC_ID, the host counts the distinct (XCD, SE, CU) per mask pattern.  Used to choose the masks
// of the split sharded tick's two streams (the boundary launch on CUs the interior launch does not use).
// build: hipcc -O2 --offload-arch=gfx950 tools/cu_mask_probe.hip -o gpurun_out/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where(unsigned* out, long long ticks) {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
  const unsigned xc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // XCC_ID
  const long long t0 = wall_clock64();
  unsigned k = 0;
  while (wall_clock64() - t0 < ticks && k < 100000000u) k++;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xc & 0xFu) | (k == 0xFFFFFFFFu ? 16u : 0u); }
}

static void probe(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  hipError_t  e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%-28s: hipExtStreamCreateWithCUMask: %s\n", name, hipGetErrorString(e)); return; }
  const int blocks = 8192;
  unsigned* d;
  CK(hipMalloc(&d, 8 * blocks));
  hipLaunchKernelGGL(where, dim3(blocks), dim3(64), 0, st, d, 2000ll);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned> h(2 * blocks);
  CK(hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost));
  std::map<unsigned, std::set<unsigned>> per_xcd;
  for (int b = 0; b < blocks; b++) {
    const unsigned hw = h[2 * b], cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcd[h[2 * b + 1] & 0xF].insert(se * 32 + sh * 16 + cu);
  }
  int bits = 0, total = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  printf("%-28s: %3d mask bits ->", name, bits);
  for (auto& kv : per_xcd) { printf(" xcd%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  = %d CUs\n", total);
  if (total <= 40) {
    for (auto& kv : per_xcd) { printf("     xcd%u (se*32+sh*16+cu):", kv.first); for (unsigned c : kv.second) printf(" %u", c); printf("\n"); }
  }
  CK(hipFree(d));
  CK(hipStreamDestroy(st));
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  printf("%s: %d CUs\n", p.gcnArchName, ncu);
  std::vector<uint32_t> m(words, 0xFFFFFFFFu);
  probe("all", m);
  m.assign(words, 0u); m[0] = 0xFFFFu;
  probe("bits 0..15", m);
  m.assign(words, 0u); m[0] = 0xFFu;
  probe("bits 0..7", m);
  m.assign(words, 0u); m[0] = 0xFFFFFFFFu;
  probe("bits 0..31", m);
  m.assign(words, 0u); for (int b = 0; b < ncu; b += 16) m[b / 32] |= 1u << (b % 32);
  probe("every 16th bit", m);
  m.assign(words, 0u); m[words - 1] = 0xFFFF0000u;
  probe("last 16 bits", m);
  m.assign(words, 0xFFFFFFFFu); m[0] = 0xFFFF0000u;
  probe("all but bits 0..15", m);
  m.assign(words, 0xFFFFFFFFu); m[0] = 0u;
  probe("all but bits 0..31", m);
  return 0;
}
