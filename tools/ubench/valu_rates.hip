// Issue rate of the instructions the INT8 requantisation is built from (per wave64, cycles per instruction), measured with
// 8 independent chains per lane so that latency is hidden:   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 4096
template <int OP>
__global__ void k(int* out, int seed, long long* cyc) {
    int x[8];
    long long acc64[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 7 + i + seed; acc64[i] = x[i]; d[i] = x[i]; }
    const int m = 1234567891 + seed;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
            if (OP == 1) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(acc64[i]) : "v"(x[i]), "v"(m) : "s10", "s11");
            if (OP == 2) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
            if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
            if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (OP == 5) asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(x[i]) : "v"(m));
            if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(m));
            if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[i]) : "v"(m));
            if (OP == 8) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(m));
            if (OP == 9) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[i]) : "v"(m));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (int)acc64[i] + (int)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int* d_out, long long* d_cyc) {
    // one wave per SIMD on one CU (256 threads = 4 waves): the instruction stream of a wave is the only user of its SIMD
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256), 0, 0, d_out, 1, d_cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256), 0, 0, d_out, 2, d_cyc);
    long long c;
    hipMemcpy(&c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz; the shader clock is ~2.4 GHz
    printf("%-16s %8.2f memtime ticks per instruction x 1000 (ratio to v_add_u32 is what matters)\n", name, 1000.0 * c / (REP * 8.0));
}

int main() {
    int* d_out;
    long long* d_cyc;
    hipMalloc(&d_out, 256 * 4);
    hipMalloc(&d_cyc, 8);
    run<0>("v_add_u32", d_out, d_cyc);
    run<1>("v_mad_i64_i32", d_out, d_cyc);
    run<2>("v_mul_hi_i32", d_out, d_cyc);
    run<3>("v_mul_lo_u32", d_out, d_cyc);
    run<4>("v_fma_f64", d_out, d_cyc);
    run<5>("v_dot4_i32_i8", d_out, d_cyc);
    run<6>("v_perm_b32", d_out, d_cyc);
    run<7>("v_alignbit_b32", d_out, d_cyc);
    run<8>("v_med3_i32", d_out, d_cyc);
    run<9>("v_mul_i32_i24", d_out, d_cyc);
    return 0;
}
