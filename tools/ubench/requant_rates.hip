// Cost of one exact requantisation (int32 accumulator -> clamped int8 value) in the forms the INT8 kernels could use, in shader
// cycles per SIMD, at 1 / 2 / 4 waves per SIMD on one CU:
//     hipcc --offload-arch=gfx950 -O3 requant_rates.hip -o requant_rates && ./requant_rates
//   int      : (x*m + 2^30) >> 31 on the 64-bit product, (v + c1 + (v >> 31)) >> e, clamp            (bn_i8_strip.hip: rq + med3)
//   int_relu : the same without the sign term (exact whenever negative results clamp to the zero point anyway)
//   f64      : x -> double, fma(x, M, c) rounded down, + (1.5 * 2^52 + zp) rounded down, clamp the low dword
//   single instructions for reference
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 2048
#define NCH 8

__device__ __forceinline__ int med3(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

template <int OP>
__global__ void k(int* out, int seed, long long* cyc) {
    int x[NCH], q[NCH];
    for (int i = 0; i < NCH; ++i) { x[i] = (threadIdx.x * 7919 + i * 104729 + seed) & 0xfffff; q[i] = 0; }
    const int m = 1518500250 + seed, c1 = (1 << 8) + (-128 << 9), e = 9, lo = -128, hi = 127;
    const double M = (double)m / 2199023255552.0 / 512.0 * 2.0, c = 0.5 + 1.0 / 1024.0, magic = 6755399441055744.0 - 128.0;
    const float Mf = (float)M, magicf = 12582912.0f - 128.0f;
    if (OP == 2) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2");  // f64/f16 rounding: toward -inf
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int a = x[i];
            int v;
            if (OP == 0) {
                const int h = (int)(((long long)a * m + (1ll << 30)) >> 31);
                v = med3((h + c1 + (h >> 31)) >> e, lo, hi);
            } else if (OP == 1) {
                const int h = (int)(((long long)a * m + (1ll << 30)) >> 31);
                v = med3((h + c1) >> e, lo, hi);
            } else if (OP == 2) {
                const double t = __builtin_fma((double)a, M, c);
                const double y = t + magic;
                v = med3((int)__double_as_longlong(y), lo, hi);
            } else if (OP == 3) {
                const float y = __builtin_fmaf((float)a, Mf, magicf);
                v = med3(__float_as_int(y), 0x4B400000 + lo, 0x4B400000 + hi);
            } else if (OP == 4) {
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(v) : "v"(a), "v"(m));
            } else if (OP == 5) {
                double d;
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(a));
                v = (int)__double_as_longlong(d);
            } else if (OP == 6) {
                double d = __longlong_as_double(((long long)a << 32) | (unsigned)m);
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(magic));
                v = (int)__double_as_longlong(d);
            } else if (OP == 7) {
                double d = __longlong_as_double(((long long)a << 32) | (unsigned)m);
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(M), "v"(c));
                v = (int)__double_as_longlong(d);
            } else {
                float f;
                asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f) : "v"(a));
                v = __float_as_int(f);
            }
            q[i] += v;
            x[i] = a + 977;  // a new accumulator every round (one v_add, counted in every variant)
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < NCH; ++i) s += q[i] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int* d_out, long long* d_cyc) {
    printf("%-28s", name);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int threads = 256 * waves_per_simd;  // one CU: 4 SIMDs
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d_out, 1, d_cyc);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d_out, 2, d_cyc);
        long long c;
        hipMemcpy(&c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
        // cycles one SIMD spends per requantisation: elapsed / (rounds * chains * waves on that SIMD)
        printf("  %dw/SIMD: %6.2f cyc", waves_per_simd, (double)c / (REP * (double)NCH * waves_per_simd));
    }
    printf("   (per element, incl. one v_add_u32)\n");
}

int main() {
    int* d_out;
    long long* d_cyc;
    hipMalloc(&d_out, 1024 * 4);
    hipMalloc(&d_cyc, 8);
    run<4>("v_add_u32 (+1 add)", d_out, d_cyc);
    run<0>("requant int (6 instr)", d_out, d_cyc);
    run<1>("requant int, no sign (5)", d_out, d_cyc);
    run<2>("requant f64 (cvt,fma,add,med3)", d_out, d_cyc);
    run<3>("requant f32 (cvt,fma,med3)", d_out, d_cyc);
    run<5>("v_cvt_f64_i32", d_out, d_cyc);
    run<6>("v_add_f64", d_out, d_cyc);
    run<7>("v_fma_f64", d_out, d_cyc);
    run<8>("v_cvt_f32_i32", d_out, d_cyc);
    return 0;
}
