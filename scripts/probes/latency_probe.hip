// Probe: per-instruction costs of a single wave's dependent chain on gfx950 (informs the rANS chains).
// Each test runs REP copies of a snippet inside a 256-iteration loop and reports shader cycles per copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(S) S S S S S S S S
#define REP32(S) REP8(S) REP8(S) REP8(S) REP8(S)

#define PROBE_KERNEL(NAME, SNIPPET)                                                           \
__global__ __launch_bounds__(64) void NAME(unsigned long long* out, uint32_t seed) {          \
  unsigned long long t0, t1;                                                                  \
  uint32_t cnt;                                                                               \
  asm volatile(                                                                               \
    "s_mov_b32 s40, %[seed]\n\t s_mov_b32 s41, 0x10001\n\t s_mov_b32 s42, 3\n\t s_mov_b32 s43, 0\n\t" \
    "v_mov_b32 v40, %[seed]\n\t v_mov_b32 v41, 0x10001\n\t v_mov_b32 v42, 3\n\t v_mov_b32 v43, 7\n\t" \
    "s_mov_b32 s44, 0x00800000\n\t s_mov_b32 s46, 0\n\t s_mov_b32 s47, 1\n\t"                 \
    "s_set_gpr_idx_on s43, 0x1\n\t"                                                           \
    "s_memtime %[t0]\n\t s_waitcnt lgkmcnt(0)\n\t"                                            \
    "s_mov_b32 %[cnt], 256\n\t"                                                               \
    "1:\n\t" REP32(SNIPPET)                                                                   \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t s_cmp_lg_u32 %[cnt], 0\n\t s_cbranch_scc1 1b\n\t"        \
    "s_memtime %[t1]\n\t s_waitcnt lgkmcnt(0)\n\t"                                            \
    "s_set_gpr_idx_off\n\t"                                                                   \
    : [t0] "=&s"(t0), [t1] "=&s"(t1), [cnt] "=&s"(cnt) : [seed] "s"(seed)                     \
    : "s40","s41","s42","s43","s44","s45","s46","s47","v40","v41","v42","v43","v44","v45","vcc","scc","m0","memory"); \
  if (threadIdx.x == 0) out[0] = t1 - t0;                                                     \
}

PROBE_KERNEL(k_empty, "")
PROBE_KERNEL(k_sadd, "s_add_u32 s40, s40, s41\n\t")
PROBE_KERNEL(k_sadd_indep, "s_add_u32 s45, s41, s42\n\t")
PROBE_KERNEL(k_smul, "s_mul_i32 s40, s40, s41\n\t")
PROBE_KERNEL(k_smulhi, "s_mul_hi_u32 s40, s40, s41\n\t s_or_b32 s40, s40, s41\n\t")
PROBE_KERNEL(k_sbfe, "s_bfe_u32 s40, s40, 0x1f0001\n\t")
PROBE_KERNEL(k_slshl64, "s_lshl_b64 s[40:41], s[40:41], s43\n\t")
PROBE_KERNEL(k_vadd, "v_add_u32 v40, v40, v41\n\t")
PROBE_KERNEL(k_vadd_indep, "v_add_u32 v44, v41, v42\n\t")
PROBE_KERNEL(k_vmulhi, "v_mul_hi_u32 v40, v40, v41\n\t v_or_b32 v40, v40, v41\n\t")
PROBE_KERNEL(k_vmullo, "v_mul_lo_u32 v40, v40, v41\n\t")
PROBE_KERNEL(k_vmad24, "v_mad_u32_u24 v40, v40, v42, v41\n\t")
PROBE_KERNEL(k_vdpp, "v_add_u32_dpp v40, v40, v41 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t")
PROBE_KERNEL(k_vrowdpp, "v_add_u32_dpp v40, v40, v41 row_shr:1 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t")
PROBE_KERNEL(k_vcmp_cnd, "v_cmp_ge_u32_e32 vcc, v40, v41\n\t s_nop 1\n\t v_cndmask_b32_e32 v40, v40, v42, vcc\n\t")
PROBE_KERNEL(k_readlane_salu, "v_readlane_b32 s45, v40, s42\n\t s_add_u32 s40, s45, s40\n\t")
PROBE_KERNEL(k_readlane_chain, "v_readlane_b32 s40, v40, s40\n\t")
PROBE_KERNEL(k_readlane_chain2, "v_readlane_b32 s45, v40, s40\n\t s_and_b32 s40, s45, 63\n\t")
PROBE_KERNEL(k_setidx_readlane, "s_set_gpr_idx_idx s43\n\t v_readlane_b32 s45, v40, s40\n\t s_and_b32 s40, s45, 63\n\t")
PROBE_KERNEL(k_writelane, "v_writelane_b32 v44, s40, 5\n\t")
PROBE_KERNEL(k_branch_nt, "s_cmp_lt_u32 s40, s46\n\t s_cbranch_scc1 9f\n\t 9:\n\t")
PROBE_KERNEL(k_branch_t, "s_cmp_lt_u32 s46, s47\n\t s_cbranch_scc1 9f\n\t s_nop 0\n\t 9:\n\t")
PROBE_KERNEL(k_branch_t_far, "s_cmp_lt_u32 s46, s47\n\t s_cbranch_scc1 9f\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t 9:\n\t")
PROBE_KERNEL(k_snop0, "s_nop 0\n\t")
PROBE_KERNEL(k_snop1, "s_nop 1\n\t")
PROBE_KERNEL(k_cselect, "s_cmp_lt_u32 s40, s44\n\t s_cselect_b32 s40, s41, s42\n\t")
PROBE_KERNEL(k_dec_core, "s_bfe_u32 s45, s40, 0x60006\n\t s_set_gpr_idx_idx s43\n\t s_lshr_b32 s46, s40, 12\n\t v_writelane_b32 v44, s40, 3\n\t v_readlane_b32 s47, v40, s40\n\t s_and_b32 s45, s47, 0xffff\n\t s_lshr_b32 s47, s47, 16\n\t s_mul_i32 s46, s45, s46\n\t s_add_u32 s40, s46, s47\n\t")
PROBE_KERNEL(k_setidx_on, "s_set_gpr_idx_on s43, 0x1\n\t")
PROBE_KERNEL(k_setidx_idx, "s_set_gpr_idx_idx s43\n\t")
PROBE_KERNEL(k_flbit, "s_flbit_i32_b32 s45, s40\n\t")
PROBE_KERNEL(k_flbit_m0_movrels, "s_flbit_i32_b32 m0, s44\n\t s_nop 0\n\t s_movrels_b32 s45, s40\n\t")
PROBE_KERNEL(k_m0_write, "s_mov_b32 m0, s43\n\t")
PROBE_KERNEL(k_dec2_core, "s_bfe_u32 s45, s40, 0x60006\n\t s_set_gpr_idx_on s43, 0x1\n\t v_readlane_b32 s47, v40, s40\n\t v_readlane_b32 s46, v41, s40\n\t s_lshr_b32 s45, s41, 9\n\t s_lshr_b32 s45, s41, 3\n\t s_set_gpr_idx_on s43, 0x1\n\t v_readlane_b32 s45, v42, s41\n\t s_lshr_b32 s45, s40, 12\n\t v_writelane_b32 v44, s40, 3\n\t s_mul_i32 s47, s47, s45\n\t s_add_u32 s40, s47, s46\n\t s_flbit_i32_b32 m0, s44\n\t s_nop 0\n\t s_movrels_b32 s45, s42\n\t s_lshl_b64 s[40:41], s[40:41], s43\n\t s_add_u32 s41, s41, s43\n\t")
PROBE_KERNEL(k_dec2_core_idx, "s_bfe_u32 s45, s40, 0x60006\n\t s_set_gpr_idx_idx s43\n\t v_readlane_b32 s47, v40, s40\n\t v_readlane_b32 s46, v41, s40\n\t s_lshr_b32 s45, s41, 9\n\t s_lshr_b32 s45, s41, 3\n\t s_set_gpr_idx_idx s43\n\t v_readlane_b32 s45, v42, s41\n\t s_lshr_b32 s45, s40, 12\n\t v_writelane_b32 v44, s40, 3\n\t s_mul_i32 s47, s47, s45\n\t s_add_u32 s40, s47, s46\n\t s_cmp_lt_u32 s40, s44\n\t s_cselect_b32 s45, 8, 0\n\t s_cmp_lt_u32 s40, s42\n\t s_cselect_b32 s45, 16, s45\n\t s_lshl_b64 s[40:41], s[40:41], s43\n\t s_add_u32 s41, s41, s43\n\t")
PROBE_KERNEL(k_dec4_core, "s_lshr_b32 s45, s40, 12\n\t s_bfe_u32 s46, s40, 0x60006\n\t s_set_gpr_idx_on s43, 0x1\n\t v_writelane_b32 v44, s40, 3\n\t v_readlane_b32 s47, v40, s40\n\t v_readlane_b32 s46, v41, s40\n\t s_mul_i32 s47, s47, s45\n\t s_add_u32 s40, s47, s46\n\t s_flbit_i32_b32 m0, s44\n\t s_mov_b32 s46, s41\n\t s_movrels_b32 s45, s42\n\t s_lshl_b64 s[40:41], s[40:41], s43\n\t s_lshl_b64 s[46:47], s[46:47], s43\n\t s_sub_u32 s45, s45, s43\n\t")
PROBE_KERNEL(k_dec4_pair, "s_lshr_b32 s45, s40, 12\n\t s_bfe_u32 s46, s40, 0x60006\n\t s_set_gpr_idx_on s43, 0x1\n\t v_writelane_b32 v44, s40, 3\n\t v_readlane_b32 s47, v40, s40\n\t v_readlane_b32 s46, v41, s40\n\t s_mul_i32 s47, s47, s45\n\t s_add_u32 s40, s47, s46\n\t s_flbit_i32_b32 m0, s44\n\t s_mov_b32 s46, s41\n\t s_movrels_b32 s45, s42\n\t s_lshl_b64 s[40:41], s[40:41], s43\n\t s_lshl_b64 s[46:47], s[46:47], s43\n\t s_sub_u32 s45, s45, s43\n\ts_lshr_b32 s45, s40, 12\n\t s_bfe_u32 s46, s40, 0x60006\n\t s_set_gpr_idx_on s43, 0x1\n\t v_writelane_b32 v44, s40, 3\n\t v_readlane_b32 s47, v40, s40\n\t v_readlane_b32 s46, v41, s40\n\t s_mul_i32 s47, s47, s45\n\t s_add_u32 s40, s47, s46\n\t s_flbit_i32_b32 m0, s44\n\t s_mov_b32 s46, s41\n\t s_movrels_b32 s45, s42\n\t s_lshl_b64 s[40:41], s[40:41], s43\n\t s_lshl_b64 s[46:47], s[46:47], s43\n\t s_sub_u32 s45, s45, s43\n\ts_cmp_lt_u32 s42, s43\n\t s_cbranch_scc1 9f\n\t 9:\n\t")
PROBE_KERNEL(k_enc_core, "v_add_u32_dpp v40, v44, v41 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t v_cmp_ge_u32_e64 s[46:47], v40, v41\n\t v_cmp_ge_u32_e32 vcc, v40, v42\n\t s_nop 0\n\t v_cndmask_b32_e64 v45, 0, 8, s[46:47]\n\t v_cndmask_b32_e64 v45, v45, 16, vcc\n\t v_lshrrev_b32_e32 v45, v45, v40\n\t v_mul_hi_u32 v43, v45, v41\n\t v_lshrrev_b32_e32 v43, v42, v43\n\t v_mad_i32_i24 v44, v43, v42, v45\n\t s_nop 1\n\t")

struct T { const char* name; void (*fn)(unsigned long long*, uint32_t); };
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  T tests[] = {{"empty loop", k_empty}, {"s_add dep", k_sadd}, {"s_add indep", k_sadd_indep}, {"s_mul_i32 dep", k_smul},
    {"s_mul_hi+s_or dep", k_smulhi}, {"s_bfe dep", k_sbfe}, {"s_lshl_b64 dep", k_slshl64}, {"v_add dep", k_vadd},
    {"v_add indep", k_vadd_indep}, {"v_mul_hi+v_or dep", k_vmulhi}, {"v_mul_lo dep", k_vmullo}, {"v_mad_u32_u24 dep", k_vmad24},
    {"v_add_dpp wave_shr + s_nop1 dep", k_vdpp}, {"v_add_dpp row_shr + s_nop1 dep", k_vrowdpp},
    {"v_cmp;s_nop1;v_cndmask dep", k_vcmp_cnd}, {"v_readlane;s_add(dep on it)", k_readlane_salu},
    {"v_readlane lane-sel chain", k_readlane_chain}, {"v_readlane;s_and chain", k_readlane_chain2},
    {"set_idx;v_readlane;s_and chain", k_setidx_readlane}, {"v_writelane", k_writelane},
    {"cmp+branch not taken", k_branch_nt}, {"cmp+branch taken (skip 1)", k_branch_t}, {"cmp+branch taken (skip 16)", k_branch_t_far},
    {"s_nop 0", k_snop0}, {"s_nop 1", k_snop1}, {"s_cmp+s_cselect dep", k_cselect},
    {"s_set_gpr_idx_on", k_setidx_on}, {"s_set_gpr_idx_idx", k_setidx_idx}, {"s_flbit", k_flbit}, {"s_flbit m0;nop;s_movrels", k_flbit_m0_movrels}, {"s_mov m0", k_m0_write}, {"decode v3 core (17 instr, idx_on+flbit)", k_dec2_core}, {"decode v3 core (18 instr, idx_idx+cmp/csel)", k_dec2_core_idx}, {"decode v4 symbol (14 instr)", k_dec4_core}, {"decode v4 pair (28 instr + cmp + untaken branch)", k_dec4_pair}, {"decode core (9 instr)", k_dec_core}, {"encode ripple step (11 instr)", k_enc_core}};
  double base = 0;
  for (auto& t : tests) {
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
      hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, d, 12345u + r);
      unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      if (h < best) best = h;
    }
    double per = (double)best / (256.0 * 32.0);
    if (std::string(t.name) == "empty loop") base = (double)best;
    printf("%-40s %8.2f cycles/copy\n", t.name, ((double)best - base) / (256.0 * 32.0) + (std::string(t.name) == "empty loop" ? per : 0));
  }
  return 0;
}
