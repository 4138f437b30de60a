// Issue rate of the integer VALU instructions the transform kernels are made of, per SIMD, on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
// Every wave runs ITER x 64 independent instructions of one kind (8 accumulator chains, so a wave alone is never
// waiting on its own result); the grid puts W waves on every SIMD (W = 1, 2, 4, 8).  Reported: wave-instructions
// per cycle per SIMD at the clock the run held (s_memtime / s_memrealtime), i.e. 0.25 = one per 4 cycles (a 16-lane
// SIMD), 0.5 = one per 2 cycles (a 32-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(int* out, int iters, int a, int b, unsigned long long* clk) {
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define ONE(k)                                                                                              \
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v##k) : "v"(a));                           \
        else if (KIND == 1) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(v##k) : "s"(a), "v"(b));      \
        else if (KIND == 2) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(v##k));                            \
        else if (KIND == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v##k) : "v"(b));                   \
        else if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v##k) : "v"(b));                   \
        else if (KIND == 5) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v##k) : "v"(b));                   \
        else if (KIND == 6) asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(v##k) : "v"(b), "v"(a));     \
        else if (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v##k) : "v"(b), "s"(a));         \
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v##k) : "v"(b), "v"(a));
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int KIND>
void run(const char* name) {
    const int iters = 4000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves, one per SIMD
        int* d_out; unsigned long long* d_clk;
        hipMalloc(&d_out, (size_t)blocks * 256 * 4);
        hipMalloc(&d_clk, (size_t)blocks * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<KIND><<<blocks, 256>>>(d_out, 10, 3, 5, d_clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<KIND><<<blocks, 256>>>(d_out, iters, 3, 5, d_clk);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> clk(2 * blocks);
        hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, ticks = 0;
        for (int i = 0; i < blocks; ++i) { cyc += clk[2 * i]; ticks += clk[2 * i + 1]; }
        const double ghz = cyc / ticks * 0.1;
        const double insts_per_wave = (double)iters * 64;
        // per SIMD: waves_per_simd waves, each insts_per_wave instructions, in (mean cycles of a wave)
        const double per_cycle = waves_per_simd * insts_per_wave / (cyc / blocks);
        printf("%-16s %d wave(s)/SIMD: %.3f wave-instr/cycle/SIMD (%.2f cycles each), kernel %.3f ms, clock %.2f GHz, chip %.1f T lane-ops/s\n",
               name, waves_per_simd, per_cycle, 1.0 / per_cycle, ms, ghz, 1024.0 * waves_per_simd * insts_per_wave * 64 / (ms * 1e-3) / 1e12);
        hipFree(d_out); hipFree(d_clk);
    }
}

int main() {
    run<0>("v_add_u32"); run<1>("v_mad_i32_i24"); run<2>("v_ashrrev_i32"); run<3>("v_mul_hi_u32"); run<4>("v_mul_lo_u32");
    run<5>("v_pk_add_i16"); run<6>("v_dot2_i32_i16"); run<7>("v_perm_b32"); run<8>("v_fma_f32");
    return 0;
}
