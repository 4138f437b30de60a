// Probe: VGPR-index mode semantics on gfx950 (used by the rANS decode chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void probe(uint32_t* out, uint32_t row, uint32_t lanesel) {
  int lane = threadIdx.x;
  uint32_t r1, r2, r3;
  asm volatile(
    "v_lshlrev_b32 v64, 8, %[l]\n\t"            // v64+k = lane<<8 | k
    "v_or_b32 v65, 1, v64\n\t" "v_or_b32 v66, 2, v64\n\t" "v_or_b32 v67, 3, v64\n\t"
    "v_mov_b32 v70, 0x7777\n\t" "v_mov_b32 v71, 0x7778\n\t" "v_mov_b32 v72, 0x7779\n\t" "v_mov_b32 v73, 0x777a\n\t"
    "s_nop 4\n\t"
    "s_set_gpr_idx_on %[row], 0x1\n\t"          // src0 relative
    "v_mov_b32 v70, v64\n\t"                    // expect v70 = v[64+row]
    "s_nop 1\n\t"
    "v_readlane_b32 %[r1], v70, %[ls]\n\t"      // is src0 (v70) indexed too?  lane select masked to 6 bits?
    "v_readlane_b32 %[r2], v64, %[ls]\n\t"      // indexed readlane straight from the table?
    "s_set_gpr_idx_off\n\t"
    "s_nop 1\n\t"
    "v_readlane_b32 %[r3], v70, %[ls]\n\t"
    : [r1] "=&s"(r1), [r2] "=&s"(r2), [r3] "=&s"(r3)
    : [l] "v"(lane), [row] "s"(row), [ls] "s"(lanesel)
    : "v64","v65","v66","v67","v70","v71","v72","v73","m0","memory");
  if (lane == 0) { out[0] = r1; out[1] = r2; out[2] = r3; }
}
int main() {
  uint32_t* d; hipMalloc(&d, 64);
  for (uint32_t row = 0; row < 4; ++row) for (uint32_t ls : {5u, 64u + 7u, 0xfffff0c9u}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, row, ls);
    uint32_t h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("row %u lanesel 0x%x (lane %u): mov+readlane(in mode)=0x%x  readlane(v64,in mode)=0x%x  readlane(v70, mode off)=0x%x\n", row, ls, ls & 63, h[0], h[1], h[2]);
  }
  return 0;
}
