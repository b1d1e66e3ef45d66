// Microbenchmark (development tool): issue cost of the VALU instructions the identity kernel's epilogue
// is made of, one wavefront per SIMD, independent operands (throughput) - ticks of s_memtime per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 0.01f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float c = 0.999f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pc = {c, c};
  unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    if (OP == 1) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (OP == 2) asm volatile(REP8("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (OP == 3) asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
    if (OP == 4) asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n")
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    if (OP == 5) asm volatile(REP8("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    if (OP == 6) asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
    if (OP == 7) asm volatile(REP8("v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n")
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    if (OP == 8) asm volatile(REP8("v_fma_f32 %0, |%0|, %8, %8\n v_sub_f32 %1, %1, %8\n v_add_f32 %2, 1.0, %2\n v_mul_f32 %3, %3, %8\n v_fma_f32 %4, |%4|, %8, %8\n v_sub_f32 %5, %5, %8\n v_add_f32 %6, 1.0, %6\n v_mul_f32 %7, %7, %8\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + u0 + u1 + u2 + u3;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(t1 - t0) / (iters * 64);
}
template <int OP> void run(const char* name, float* d) {
  float h = 0;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, d, 500); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("%-34s %.2f ticks per wave64 instruction\n", name, h);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 256 * 4);
  run<0>("v_fma_f32", d); run<5>("v_max_f32", d); run<8>("fma|abs| / sub / add / mul mix", d); run<1>("v_exp_f32", d); run<2>("v_log_f32", d);
  run<3>("v_pk_fma_f32 (2 values)", d); run<6>("v_pk_mul_f32 (2 values)", d); run<4>("v_cvt_pk_bf16_f32 (2 values)", d); run<7>("v_and_b32 / v_lshlrev_b32", d);
  return 0;
}
