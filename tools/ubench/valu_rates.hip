// Issue cost of the vector instructions the trace kernel is made of, with the kernel's own
// occupancy (six waves per SIMD, every CU busy): cycles of SIMD time per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdint>

#define REP 64
#define LOOPS 2000

template <int OP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6)))
k(uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = b + 77u, d = c * 3u;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v pa = {1.5f, 2.5f}, pb = {0.5f, 0.25f}, pc = {1.0001f, 0.9999f}, pd = {0.125f, 0.375f};
    float fa = (float)(a & 1023) + 1.5f, fb = (float)(b & 1023) + 2.5f, fc = 1.0001f, fd = 0.5f;
    asm volatile("s_mov_b64 s[12:13], 0x5555\n s_mov_b64 vcc, 0x3333" : : : "s12", "s13", "vcc");
    for (int l = 0; l < LOOPS; l++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (OP == 0) {  // v_fma_f32, four independent chains
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(1.0001f), "v"(0.25f));
            } else if (OP == 1) {  // v_mad_u64_u32
                uint64_t p0, p1, p2, p3;
                asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %8, 0\n v_mad_u64_u32 %1, s[10:11], %5, %8, 0\n"
                             "v_mad_u64_u32 %2, s[10:11], %6, %8, 0\n v_mad_u64_u32 %3, s[10:11], %7, %8, 0"
                             : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(0xD2511F53u)
                             : "s10", "s11");
                a = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; b = (uint32_t)(p1 >> 32) ^ (uint32_t)p2;
                c = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; d = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
            } else if (OP == 2) {  // the four xors alone (what OP 1 adds to its multiplies)
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 26 || OP == 27) {  // v_mad_u64_u32 alone, multiplier in a vector (26) or scalar (27) register
                uint64_t p0, p1, p2, p3;
                if (OP == 26)
                    asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %8, 0\n v_mad_u64_u32 %1, s[10:11], %5, %8, 0\n"
                                 "v_mad_u64_u32 %2, s[10:11], %6, %8, 0\n v_mad_u64_u32 %3, s[10:11], %7, %8, 0"
                                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(0xD2511F53u)
                                 : "s10", "s11");
                else
                    asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %8, 0\n v_mad_u64_u32 %1, s[10:11], %5, %8, 0\n"
                                 "v_mad_u64_u32 %2, s[10:11], %6, %8, 0\n v_mad_u64_u32 %3, s[10:11], %7, %8, 0"
                                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a), "v"(b), "v"(c), "v"(d), "s"(0xD2511F53u)
                                 : "s10", "s11");
                a = (uint32_t)(p0 >> 32); b = (uint32_t)(p1 >> 32); c = (uint32_t)(p2 >> 32); d = (uint32_t)(p3 >> 32);
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 28) {  // v_bitop3_b32, three vector operands
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n"
                             "v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 25) {  // v_bitop3_b32 with the parity table and a scalar operand: the three-way xor of a Philox round
                asm volatile("v_bitop3_b32 %0, %0, %1, s12 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, s12 bitop3:0x96\n"
                             "v_bitop3_b32 %2, %2, %3, s12 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, s12 bitop3:0x96"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 3) {  // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0xD2511F53u));
            } else if (OP == 4) {  // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
            } else if (OP == 5) {  // v_sqrt_f32
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
            } else if (OP == 6) {  // v_mul_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0x511F53u));
            } else if (OP == 7) {  // v_cndmask_b32 (VCC)
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "vcc");
            } else if (OP == 9) {  // v_cndmask_b32, mask in an SGPR pair set once
                asm volatile("v_cndmask_b32 %0, %0, %4, s[12:13]\n v_cndmask_b32 %1, %1, %4, s[12:13]\n v_cndmask_b32 %2, %2, %4, s[12:13]\n v_cndmask_b32 %3, %3, %4, s[12:13]"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            } else if (OP == 10) {  // v_cmp_lt_f32 -> vcc, v_cndmask_b32 on it: the select idiom
                asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %5, %6\n v_cndmask_b32 %2, %2, %3, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(fa), "v"(fb), "v"(fc) : "vcc");
            } else if (OP == 11) {  // v_mov_b32
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 12) {  // v_mul_f32 / v_add_f32 mix
                asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(1.0001f));
            } else if (OP == 13) {  // v_and_b32 with a 32-bit literal
                asm volatile("v_and_b32 %0, 0xFFFFFF00, %0\n v_and_b32 %1, 0xFFFFFF00, %1\n v_and_b32 %2, 0xFFFFFF00, %2\n v_and_b32 %3, 0xFFFFFF00, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 14) {  // v_fma_f32 with an SGPR operand
                asm volatile("v_fma_f32 %0, %0, s12, %4\n v_fma_f32 %1, %1, s12, %4\n v_fma_f32 %2, %2, s12, %4\n v_fma_f32 %3, %3, s12, %4"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(0.25f));
            } else if (OP == 15) {  // v_cvt_f32_u32
                asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7"
                             : "=v"(fa), "=v"(fb), "=v"(fc), "=v"(fd) : "v"(a), "v"(b), "v"(c), "v"(d));
            } else if (OP == 16) {  // v_pk_fma_f32, registers only
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3"
                             : "+v"(pa), "+v"(pb) : "v"(pc), "v"(pd));
            } else if (OP == 17) {  // v_pk_fma_f32 with an SGPR pair as one operand
                asm volatile("v_pk_fma_f32 %0, s[12:13], %2, %3\n v_pk_fma_f32 %1, s[12:13], %2, %3\n v_pk_fma_f32 %0, s[12:13], %2, %0\n v_pk_fma_f32 %1, s[12:13], %2, %1"
                             : "+v"(pa), "+v"(pb) : "v"(pc), "v"(pd));
            } else if (OP == 18) {  // v_pk_add_f32 with an SGPR pair
                asm volatile("v_pk_add_f32 %0, s[12:13], %0\n v_pk_add_f32 %1, s[12:13], %1\n v_pk_add_f32 %0, s[12:13], %0\n v_pk_add_f32 %1, s[12:13], %1"
                             : "+v"(pa), "+v"(pb));
            } else if (OP == 19) {  // v_sub_f32 with an SGPR operand
                asm volatile("v_sub_f32 %0, s12, %0\n v_sub_f32 %1, s12, %1\n v_sub_f32 %2, s12, %2\n v_sub_f32 %3, s12, %3"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
            } else if (OP == 20) {  // v_cmpx_le_f32 against an SGPR, mask restored once per four
                asm volatile("s_mov_b64 s[10:11], exec\n v_cmpx_le_f32 vcc, |%0|, s12\n v_cmpx_le_f32 vcc, |%1|, s12\n v_cmpx_le_f32 vcc, |%2|, s12\n v_cmpx_le_f32 vcc, |%3|, s12\n s_mov_b64 exec, s[10:11]"
                             : : "v"(fa), "v"(fb), "v"(fc), "v"(fd) : "s10", "s11", "vcc");
            } else if (OP >= 21 && OP <= 24) {  // v_fma_f32 under a partial execution mask: does the SIMD skip empty lane groups?
                asm volatile("s_mov_b64 s[10:11], exec\n s_mov_b64 exec, %6\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "s_mov_b64 exec, s[10:11]"
                             : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(1.0001f), "v"(0.25f),
                               "s"(OP == 21 ? 0x00000000FFFFFFFFull : OP == 22 ? 0xFFFFull : OP == 23 ? 0x5555555555555555ull : 0xFFFFFFFFFFFFFFFFull)
                             : "s10", "s11");
            } else if (OP == 8) {  // v_cmpx_le_f32 (writes EXEC) + restore
                asm volatile("s_mov_b64 s[10:11], exec\n v_cmpx_le_f32 %0, %1\n v_cmpx_le_f32 %2, %3\n v_cmpx_le_f32 %0, %3\n v_cmpx_le_f32 %1, %2\n s_mov_b64 exec, s[10:11]"
                             : : "v"(fa), "v"(fb), "v"(fc), "v"(fd) : "s10", "s11");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(fa + fb + fc + fd + pa.x + pa.y + pb.x + pb.y);
    if ((threadIdx.x & 63) == 0) {  // shader cycles and 100 MHz ticks of this wave
        atomicAdd(&clk[0], __builtin_amdgcn_s_memtime() - c0);
        atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int OP>
static double run(const char *name, uint32_t *out, int n_cu, unsigned long long *clk)
{
    const int grid = n_cu * 3;  // three 512-thread workgroups per CU = six waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(512), 0, 0, out, 1u, clk);
    hipMemsetAsync(clk, 0, 16, 0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(512), 0, 0, out, 2u, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 6 waves x LOOPS x REP instructions
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = 100.0 * (double)h[0] / (double)h[1];
    // SIMD cycles per wave-instruction: a wave's own cycles / its instructions / the six waves that share the SIMD
    const double inst = (double)LOOPS * REP;
    const double cyc = ms * 1e-3 * mhz * 1e6 / (inst * 6.0);   // six waves share a SIMD
    (void)grid;
    printf("%-28s %7.3f ms  %5.2f SIMD cycles per wave-instruction (kernel time x clock), shader clock %.0f MHz\n", name, ms, cyc, mhz);
    return cyc;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    uint32_t *out;
    unsigned long long *clk;
    hipMalloc(&out, 4 * 512 * 3 * n_cu);
    hipMalloc(&clk, 16);
    printf("%d CUs, six waves per SIMD, %d instructions per wave\n", n_cu, LOOPS * REP);
    run<0>("v_fma_f32", out, n_cu, clk);
    run<2>("v_xor_b32", out, n_cu, clk);
    run<25>("v_bitop3_b32 (xor3, sgpr)", out, n_cu, clk);
    run<28>("v_bitop3_b32 (xor3, vgprs)", out, n_cu, clk);
    run<26>("v_mad_u64_u32, vgpr multiplier", out, n_cu, clk);
    run<27>("v_mad_u64_u32, sgpr multiplier", out, n_cu, clk);
    run<6>("v_mul_u32_u24", out, n_cu, clk);
    run<3>("v_mul_lo_u32", out, n_cu, clk);
    run<1>("v_mad_u64_u32 + v_xor_b32", out, n_cu, clk);
    run<4>("v_rcp_f32", out, n_cu, clk);
    run<5>("v_sqrt_f32", out, n_cu, clk);
    run<8>("v_cmpx_le_f32 (+2 s_mov / 4)", out, n_cu, clk);
    run<9>("v_cndmask_b32 (sgpr pair)", out, n_cu, clk);
    run<10>("v_cmp_lt_f32 + v_cndmask", out, n_cu, clk);
    run<11>("v_mov_b32", out, n_cu, clk);
    run<12>("v_mul_f32 / v_add_f32", out, n_cu, clk);
    run<13>("v_and_b32 literal", out, n_cu, clk);
    run<14>("v_fma_f32 with sgpr", out, n_cu, clk);
    run<15>("v_cvt_f32_u32", out, n_cu, clk);
    run<16>("v_pk_fma_f32", out, n_cu, clk);
    run<17>("v_pk_fma_f32 with sgpr pair", out, n_cu, clk);
    run<18>("v_pk_add_f32 with sgpr pair", out, n_cu, clk);
    run<19>("v_sub_f32 with sgpr", out, n_cu, clk);
    run<20>("v_cmpx_le_f32 |v|, sgpr", out, n_cu, clk);
    run<7>("v_cndmask_b32 (vcc by s_mov)", out, n_cu, clk);
    run<24>("v_fma_f32, exec all (+2 s_mov / 4)", out, n_cu, clk);
    run<21>("v_fma_f32, exec low 32 lanes", out, n_cu, clk);
    run<22>("v_fma_f32, exec low 16 lanes", out, n_cu, clk);
    run<23>("v_fma_f32, exec every other lane", out, n_cu, clk);
    hipFree(out);
    return 0;
}
