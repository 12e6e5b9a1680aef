// Issue cost of the vector instructions the tracing kernels are made of, on the machine they run on: cycles of one SIMD per
// wave64 instruction, with 8 waves per SIMD issuing independent instructions (so that latency is hidden and the rate is the
// SIMD's).  hipcc --offload-arch=gfx950 -O2 -o valu_rates scripts/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                              \
  __global__ void __launch_bounds__(256) name(float *out, int iters) {                   \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;       \
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7; \
    unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3, q4 = u4, q5 = u5, q6 = u6, q7 = u7; \
    for (int i = 0; i < iters; i++) { REP8(body) }                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + \
        (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7); \
  }
#define EIGHT_F(op) asm volatile(op " %0, %0, %0" : "+v"(a0)); asm volatile(op " %0, %0, %0" : "+v"(a1)); asm volatile(op " %0, %0, %0" : "+v"(a2)); asm volatile(op " %0, %0, %0" : "+v"(a3)); \
                    asm volatile(op " %0, %0, %0" : "+v"(a4)); asm volatile(op " %0, %0, %0" : "+v"(a5)); asm volatile(op " %0, %0, %0" : "+v"(a6)); asm volatile(op " %0, %0, %0" : "+v"(a7));
#define EIGHT_F1(op) asm volatile(op " %0, %0" : "+v"(a0)); asm volatile(op " %0, %0" : "+v"(a1)); asm volatile(op " %0, %0" : "+v"(a2)); asm volatile(op " %0, %0" : "+v"(a3)); \
                     asm volatile(op " %0, %0" : "+v"(a4)); asm volatile(op " %0, %0" : "+v"(a5)); asm volatile(op " %0, %0" : "+v"(a6)); asm volatile(op " %0, %0" : "+v"(a7));
#define EIGHT_D(op) asm volatile(op " %0, %0, %0" : "+v"(d0)); asm volatile(op " %0, %0, %0" : "+v"(d1)); asm volatile(op " %0, %0, %0" : "+v"(d2)); asm volatile(op " %0, %0, %0" : "+v"(d3)); \
                    asm volatile(op " %0, %0, %0" : "+v"(d4)); asm volatile(op " %0, %0, %0" : "+v"(d5)); asm volatile(op " %0, %0, %0" : "+v"(d6)); asm volatile(op " %0, %0, %0" : "+v"(d7));
#define EIGHT_D3(op) asm volatile(op " %0, %0, %0, %0" : "+v"(d0)); asm volatile(op " %0, %0, %0, %0" : "+v"(d1)); asm volatile(op " %0, %0, %0, %0" : "+v"(d2)); asm volatile(op " %0, %0, %0, %0" : "+v"(d3)); \
                     asm volatile(op " %0, %0, %0, %0" : "+v"(d4)); asm volatile(op " %0, %0, %0, %0" : "+v"(d5)); asm volatile(op " %0, %0, %0, %0" : "+v"(d6)); asm volatile(op " %0, %0, %0, %0" : "+v"(d7));
#define EIGHT_U(op) asm volatile(op " %0, %0, %0" : "+v"(u0)); asm volatile(op " %0, %0, %0" : "+v"(u1)); asm volatile(op " %0, %0, %0" : "+v"(u2)); asm volatile(op " %0, %0, %0" : "+v"(u3)); \
                    asm volatile(op " %0, %0, %0" : "+v"(u4)); asm volatile(op " %0, %0, %0" : "+v"(u5)); asm volatile(op " %0, %0, %0" : "+v"(u6)); asm volatile(op " %0, %0, %0" : "+v"(u7));
#define MAD64(q, u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, 0" : "+v"(q) : "v"(u) : "vcc");
#define EIGHT_MAD MAD64(q0, u0) MAD64(q1, u1) MAD64(q2, u2) MAD64(q3, u3) MAD64(q4, u4) MAD64(q5, u5) MAD64(q6, u6) MAD64(q7, u7)
#define CVT_FD(d, a) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a));
#define EIGHT_CVTFD CVT_FD(d0, a0) CVT_FD(d1, a1) CVT_FD(d2, a2) CVT_FD(d3, a3) CVT_FD(d4, a4) CVT_FD(d5, a5) CVT_FD(d6, a6) CVT_FD(d7, a7)
#define CVT_DF(a, d) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(d));
#define EIGHT_CVTDF CVT_DF(a0, d0) CVT_DF(a1, d1) CVT_DF(a2, d2) CVT_DF(a3, d3) CVT_DF(a4, d4) CVT_DF(a5, d5) CVT_DF(a6, d6) CVT_DF(a7, d7)

KERNEL(k_fma_f32, EIGHT_F("v_fmac_f32"))
KERNEL(k_mul_f32, EIGHT_F("v_mul_f32"))
KERNEL(k_xor, EIGHT_U("v_xor_b32"))
KERNEL(k_add_u32, EIGHT_U("v_add_u32"))
KERNEL(k_mul_lo, EIGHT_U("v_mul_lo_u32"))
KERNEL(k_mul_hi, EIGHT_U("v_mul_hi_u32"))
KERNEL(k_mad64, EIGHT_MAD)
KERNEL(k_rcp, EIGHT_F1("v_rcp_f32"))
KERNEL(k_log, EIGHT_F1("v_log_f32"))
KERNEL(k_sqrt, EIGHT_F1("v_sqrt_f32"))
KERNEL(k_sin, EIGHT_F1("v_sin_f32"))
KERNEL(k_add_f64, EIGHT_D("v_add_f64"))
KERNEL(k_mul_f64, EIGHT_D("v_mul_f64"))
KERNEL(k_fma_f64, EIGHT_D3("v_fma_f64"))
KERNEL(k_cvt_f64_f32, EIGHT_CVTFD)
KERNEL(k_cvt_f32_f64, EIGHT_CVTDF)

// ---- round 4: the scalar side, and what the two-photons-per-lane design is made of ------------------------------------------
// SALU classes the tracing kernels are full of (ballot bookkeeping: s_and_saveexec / s_or exec restores, s_bcnt1 popcounts,
// s_cbranch around predicated blocks, s_mov / s_cselect of wave-uniform state), alone and interleaved 1:1 with v_fmac_f32:
// is scalar issue a resource of its own, and how much of it does a kernel with 0.49 scalar per vector instruction use?
#define KERNEL_S(name, body)                                                             \
  __global__ void __launch_bounds__(256) name(float *out, int iters) {                   \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7; \
    unsigned long long m0 = ~0ull, m1 = ~0ull, m2 = ~0ull, m3 = ~0ull;                    \
    for (int i = 0; i < iters; i++) { REP8(body) }                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) + (float)(m0 ^ m1 ^ m2 ^ m3); \
  }
#define S_ADD(s) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");
#define EIGHT_SADD S_ADD(s0) S_ADD(s1) S_ADD(s2) S_ADD(s3) S_ADD(s4) S_ADD(s5) S_ADD(s6) S_ADD(s7)
#define S_MOV(s, t) asm volatile("s_mov_b32 %0, %1" : "=s"(s) : "s"(t));
#define EIGHT_SMOV S_MOV(s0, s1) S_MOV(s1, s2) S_MOV(s2, s3) S_MOV(s3, s4) S_MOV(s4, s5) S_MOV(s5, s6) S_MOV(s6, s7) S_MOV(s7, s0)
#define S_BCNT(s, m) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s) : "s"(m) : "scc");
#define EIGHT_BCNT S_BCNT(s0, m0) S_BCNT(s1, m1) S_BCNT(s2, m2) S_BCNT(s3, m3) S_BCNT(s4, m0) S_BCNT(s5, m1) S_BCNT(s6, m2) S_BCNT(s7, m3)
#define S_SAVEEXEC(m) asm volatile("s_and_saveexec_b64 %0, %0\n\ts_mov_b64 exec, %0" : "+s"(m) :: "scc", "exec");  /* (two scalar instructions) */
#define FOUR_SAVEEXEC S_SAVEEXEC(m0) S_SAVEEXEC(m1) S_SAVEEXEC(m2) S_SAVEEXEC(m3)
#define S_BRANCH(s) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc0 1f\n1:" :: "s"(s) : "scc");  /* (two scalar instructions, branch not taken or taken to the next instruction) */
#define FOUR_BRANCH S_BRANCH(s0) S_BRANCH(s1) S_BRANCH(s2) S_BRANCH(s3)
#define V_CMP_BALLOT(a, m) asm volatile("v_cmp_lt_f32 %0, %1, %1" : "=s"(m) : "v"(a));  /* a vector compare that writes a lane mask to scalar registers (every __ballot) */
#define FOUR_BALLOT V_CMP_BALLOT(a0, m0) V_CMP_BALLOT(a1, m1) V_CMP_BALLOT(a2, m2) V_CMP_BALLOT(a3, m3) V_CMP_BALLOT(a4, m0) V_CMP_BALLOT(a5, m1) V_CMP_BALLOT(a6, m2) V_CMP_BALLOT(a7, m3)
#define FMAC(a) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(a));
#define MIX(a, s) FMAC(a) S_ADD(s)
#define EIGHT_MIX MIX(a0, s0) MIX(a1, s1) MIX(a2, s2) MIX(a3, s3) MIX(a4, s4) MIX(a5, s5) MIX(a6, s6) MIX(a7, s7)
#define MIX2(a, s, t) FMAC(a) S_ADD(s) S_ADD(t)
#define EIGHT_MIX2 MIX2(a0, s0, s1) MIX2(a1, s2, s3) MIX2(a2, s4, s5) MIX2(a3, s6, s7) MIX2(a4, s0, s1) MIX2(a5, s2, s3) MIX2(a6, s4, s5) MIX2(a7, s6, s7)
// the exchange of the two-photons-per-lane design: v_swap_b32 against three moves, and a select per register
#define V_SWAP(a, b) asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b));
#define FOUR_SWAP V_SWAP(a0, a1) V_SWAP(a2, a3) V_SWAP(a4, a5) V_SWAP(a6, a7) V_SWAP(a0, a2) V_SWAP(a1, a3) V_SWAP(a4, a6) V_SWAP(a5, a7)
#define V_CNDMASK(a, b) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
#define EIGHT_CND V_CNDMASK(a0, a1) V_CNDMASK(a1, a2) V_CNDMASK(a2, a3) V_CNDMASK(a3, a4) V_CNDMASK(a4, a5) V_CNDMASK(a5, a6) V_CNDMASK(a6, a7) V_CNDMASK(a7, a0)
#define V_ACC(a) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(a) :: "a0");  /* (two instructions: park and fetch) */
#define FOUR_ACC V_ACC(a0) V_ACC(a1) V_ACC(a2) V_ACC(a3)
KERNEL_S(k_s_add, EIGHT_SADD)
KERNEL_S(k_s_mov, EIGHT_SMOV)
KERNEL_S(k_s_bcnt, EIGHT_BCNT)
KERNEL_S(k_s_saveexec, FOUR_SAVEEXEC)
KERNEL_S(k_s_branch, FOUR_BRANCH)
KERNEL_S(k_v_cmp_ballot, FOUR_BALLOT)
KERNEL_S(k_mix_1to1, EIGHT_MIX)    /* 8 v_fmac + 8 s_add per body: counted as 8 (the vector ones) */
KERNEL_S(k_mix_1to2, EIGHT_MIX2)   /* 8 v_fmac + 16 s_add */
KERNEL_S(k_v_swap, FOUR_SWAP)
KERNEL_S(k_v_cndmask, EIGHT_CND)
KERNEL_S(k_v_accvgpr, FOUR_ACC)

template <typename K>
void run(const char *name, K kernel, float *out, int wavesPerSimd) {
  const int iters = 2000;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * wavesPerSimd;  // 256 lanes = 4 waves per workgroup = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  int clockKHz = 0;
  hipDeviceGetAttribute(&clockKHz, hipDeviceAttributeClockRate, 0);
  const double instrPerSimd = (double)iters * 64.0 * wavesPerSimd;  // 8 x 8 per iteration per wave
  printf("%-34s %d waves/SIMD: %.3f ms -> %.2f cycles per wave64 instruction per SIMD (at %.2f GHz)\n", name, wavesPerSimd, ms,
         ms * 1e-3 * clockKHz * 1e3 / instrPerSimd, clockKHz * 1e-6);
}

int main() {
  float *out;
  hipMalloc(&out, sizeof(float) * 256 * 4096);
  for (int w : {1, 4, 6, 8}) {
    run("s_add_u32", k_s_add, out, w); run("s_mov_b32", k_s_mov, out, w); run("s_bcnt1_i32_b64", k_s_bcnt, out, w);
    run("s_and_saveexec+s_mov exec (pairs)", k_s_saveexec, out, w); run("s_cmp+s_cbranch (pairs)", k_s_branch, out, w);
    run("v_cmp -> sgpr mask (ballot)", k_v_cmp_ballot, out, w);
    run("v_fmac + 1 s_add (per v_fmac)", k_mix_1to1, out, w); run("v_fmac + 2 s_add (per v_fmac)", k_mix_1to2, out, w);
    run("v_swap_b32", k_v_swap, out, w); run("v_cndmask_b32", k_v_cndmask, out, w); run("v_accvgpr write+read (pairs)", k_v_accvgpr, out, w);
  }
  for (int w : {1, 8}) {
    run("v_fmac_f32", k_fma_f32, out, w); run("v_mul_f32", k_mul_f32, out, w); run("v_xor_b32", k_xor, out, w); run("v_add_u32", k_add_u32, out, w);
    run("v_mul_lo_u32", k_mul_lo, out, w); run("v_mul_hi_u32", k_mul_hi, out, w); run("v_mad_u64_u32", k_mad64, out, w);
    run("v_rcp_f32", k_rcp, out, w); run("v_log_f32", k_log, out, w); run("v_sqrt_f32", k_sqrt, out, w); run("v_sin_f32", k_sin, out, w);
    run("v_add_f64", k_add_f64, out, w); run("v_mul_f64", k_mul_f64, out, w); run("v_fma_f64", k_fma_f64, out, w);
    run("v_cvt_f64_f32", k_cvt_f64_f32, out, w); run("v_cvt_f32_f64", k_cvt_f32_f64, out, w);
  }
  return 0;
}
