// Dependent-chain latency of the instructions the serial rANS coders are made of, one wave on an idle chip (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lat tools/experiments/lat.hip && /tmp/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256
template <int OP>
__global__ void chain(uint64_t *out, uint32_t seed, long *cycles) {
  __shared__ uint64_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint64_t)((i * 7 + 1) & 1023) * 8;
  __syncthreads();
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = 12345;
  uint64_t x = ((uint64_t)a << 32) | b;
  uint32_t addr = (threadIdx.x * 8) & 8191;
  long t0 = __builtin_readcyclecounter();
  long m0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 2) asm volatile("v_mad_u64_u32 %0, s[0:1], %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "s0", "s1");
      if (OP == 3) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(x));
      if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
      if (OP == 5) { asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr)); addr = (uint32_t)x; }
      if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(x));
      if (OP == 8) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
      if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      if (OP == 10) asm volatile("s_nop 0");
      if (OP == 11) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a) : "v"(b));
      if (OP == 12) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr)); addr = a & 8188; }
    }
  }
  long m1 = clock64();
  long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + a + addr;
  if (threadIdx.x == 0) { cycles[0] = t1 - t0; cycles[1] = m1 - m0; }
}

// (round 4: the same chains with only the lower 32 lanes of the wave active - does a half-empty wave64 issue a dependent
// instruction sooner on the SIMD-32?)
template <int OP>
void run(const char *name) {
  uint64_t *out; long *cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
  for (int threads = 64; threads >= 32; threads -= 32) {
    chain<OP><<<1, threads>>>(out, 7, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    chain<OP><<<1, threads>>>(out, 7, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-30s %2d lanes %7.2f memtime-cycles/op  %7.2f ns/op (event)\n", name, threads, (double)h[1] / (64.0 * REP), 1e6 * ms / (64.0 * REP));
  }
  fflush(stdout);
}

int main() {
  run<0>("v_add_u32 dep");
  run<1>("v_mul_hi_u32 dep");
  run<6>("v_mul_lo_u32 dep");
  run<2>("v_mad_u64_u32 dep");
  run<9>("v_mad_u32_u24 dep");
  run<3>("v_lshrrev_b64 dep");
  run<7>("v_lshl_add_u64 dep");
  run<11>("v_alignbit_b32 dep");
  run<4>("v_cndmask_b32 dep");
  run<8>("v_cmp + v_cndmask dep");
  run<5>("ds_read_b64 dep + wait");
  run<12>("ds_read_b32 dep + wait");
  run<10>("s_nop 0");
  return 0;
}
