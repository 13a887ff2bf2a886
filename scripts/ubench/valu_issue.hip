// Issue cost per wave-instruction of the integer / transcendental operations the photon split is made of, on one
// SIMD of gfx950: chains of independent instructions (8 accumulators), one wave per SIMD, s_memtime around them.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue scripts/ubench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP 64
template <int OP>
__global__ void k(uint64_t *out, uint32_t seed)
{
    uint32_t a[8];
    float f[8];
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = seed * (threadIdx.x + 17 * i + 1);
        f[i] = 1.0f + (float)(a[i] & 1023) * 0.001f;
        w[i] = a[i];
    }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 256; ++r) {
#pragma unroll
        for (int j = 0; j < REP / 8; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "s"(0xD2511F53u) : "vcc");
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(0xD2511F53u));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(0xD2511F53u));
                if (OP == 3) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
                if (OP == 6) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(0x511F53u));
                if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(0x511F53u));
                if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i]));
                if (OP == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(0x511F53u));
                if (OP == 12) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
                if (OP == 13) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(w[i]));
                if (OP == 14) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == 15) asm volatile("v_add_f64 %0, %0, %0" : "+v"(w[i]));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (uint64_t)f[i] + w[i];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s == 0x1234567) out[blockIdx.x * 2 + 1] = s;
}

template <int OP>
void run(const char *name, int waves)
{
    uint64_t *d;
    hipMalloc(&d, 4096 * 16);
    k<OP><<<1, 64 * waves>>>(d, 12345u);
    hipDeviceSynchronize();
    k<OP><<<1, 64 * waves>>>(d, 12345u);
    uint64_t h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz; report ticks per instruction x (clock / 100 MHz) left to the reader: print raw and
    // relative to v_fma_f32
    printf("%-18s waves/CU=%d  memtime ticks per %d instr: %llu  -> %.3f ticks/instr\n", name, waves, 256 * REP,
           (unsigned long long)h[0], (double)h[0] / (256.0 * REP));
    hipFree(d);
}

int main()
{
    for (int waves : {4, 8}) {  // 4 waves in one workgroup = one per SIMD; 8 = two per SIMD (issue cost doubles if bound)
        run<5>("v_fma_f32", waves);
        run<10>("v_pk_fma_f32", waves);
        run<0>("v_mad_u64_u32", waves);
        run<1>("v_mul_lo_u32", waves);
        run<2>("v_mul_hi_u32", waves);
        run<7>("v_mul_u32_u24", waves);
        run<8>("v_mul_hi_u32_u24", waves);
        run<11>("v_mad_u32_u24", waves);
        run<3>("v_bcnt_u32_b32", waves);
        run<6>("v_bitop3_b32", waves);
        run<13>("v_lshl_add_u64", waves);
        run<4>("v_log_f32", waves);
        run<9>("v_rcp_f32", waves);
        run<12>("v_sqrt_f32", waves);
        run<14>("v_cvt_f32_u32", waves);
        run<15>("v_add_f64", waves);
    }
    return 0;
}
