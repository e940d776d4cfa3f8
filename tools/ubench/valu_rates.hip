// tools/ubench/valu_rates.hip — issue cost of the vector instructions the kernels of this path are made of, on the GPU it runs on.
//
// Every kernel of the hot path is bound by vector-instruction issue, and the SQ counters weigh every instruction the
// same.  This measures what one wave64 instruction of each kind costs a SIMD: 8 independent dependency chains per lane, one instruction
// per chain and loop trip, `wps` waves per SIMD on every SIMD of the chip; cycles per instruction and SIMD = wall time of the launch
// (HIP events) x 2.4 GHz x SIMDs / wave-instructions issued.  Results: profiles/r03_valu_rates.md.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates [waves per SIMD = 4]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kChains = 8, kTrips = 32768;
typedef float v2f __attribute__((ext_vector_type(2)));

// X(case, printed name, instruction text, outputs, inputs [: clobbers]); operands of chain i: a[i] int, w[i] long, f[i] float, d[i] double,
// p[i] float2; loop-invariant sources b, c (int), fb, fc, db, dc, pb, pc, mask (SGPR pair)
#define OUTS(...) __VA_ARGS__
#define INS(...) __VA_ARGS__
#define OPS(X) \
    X(0, "v_mov_b32", "v_mov_b32 %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(1, "v_add_u32", "v_add_u32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(2, "v_sub_u32", "v_sub_u32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(3, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(4, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(5, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(6, "v_and_b32", "v_and_b32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(7, "v_or_b32", "v_or_b32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(8, "v_xor_b32", "v_xor_b32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(9, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(10, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(11, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(12, "v_lshlrev_b32 (constant count)", "v_lshlrev_b32 %0, 1, %0", OUTS("+v"(a[i])), INS()) \
    X(13, "v_lshlrev_b32 (register count)", "v_lshlrev_b32 %0, %1, %0", OUTS("+v"(a[i])), INS("v"(b))) \
    X(14, "v_lshrrev_b32", "v_lshrrev_b32 %0, 1, %0", OUTS("+v"(a[i])), INS()) \
    X(15, "v_ashrrev_i32 (constant count)", "v_ashrrev_i32 %0, 1, %0", OUTS("+v"(a[i])), INS()) \
    X(16, "v_ashrrev_i32 (register count)", "v_ashrrev_i32 %0, %1, %0", OUTS("+v"(a[i])), INS("v"(b))) \
    X(17, "v_ashrrev_i32_sdwa (count = byte 1)", "v_ashrrev_i32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", OUTS("+v"(a[i])), INS("v"(b))) \
    X(18, "v_add_u32_sdwa", "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", OUTS("+v"(a[i])), INS("v"(b))) \
    X(19, "v_max_i32", "v_max_i32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(20, "v_min_i32", "v_min_i32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(21, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(22, "v_max3_i32", "v_max3_i32 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(23, "v_cndmask_b32 (mask in SGPRs)", "v_cndmask_b32_e64 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "s"(mask))) \
    X(24, "v_cmp_lt_i32 -> SGPR pair", "v_cmp_lt_i32_e64 %1, %0, %2", OUTS("+v"(a[i]), "=s"(cmp[i])), INS("v"(b))) \
    X(25, "v_bfe_i32", "v_bfe_i32 %0, %0, 8, 8", OUTS("+v"(a[i])), INS()) \
    X(26, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(27, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 8", OUTS("+v"(a[i])), INS("v"(b))) \
    X(28, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(29, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(30, "v_mul_hi_i32_i24", "v_mul_hi_i32_i24 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(31, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(32, "v_mul_hi_i32", "v_mul_hi_i32 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(33, "v_mad_i64_i32", "v_mad_i64_i32 %0, vcc, %1, %2, %0", OUTS("+v"(w[i])), INS("v"(a[i]), "v"(b)) : "vcc") \
    X(34, "v_mad_i32_i16", "v_mad_i32_i16 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(35, "v_pk_mad_i16", "v_pk_mad_i16 %0, %0, %1, %2", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(36, "v_pk_add_i16", "v_pk_add_i16 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(37, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1", OUTS("+v"(a[i])), INS("v"(b))) \
    X(38, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, %1, %0", OUTS("+v"(a[i])), INS("v"(b))) \
    X(39, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %1, %2, %0", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(40, "v_dot2_i32_i16", "v_dot2_i32_i16 %0, %1, %2, %0", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(41, "v_dot8_i32_i4", "v_dot8_i32_i4 %0, %1, %2, %0", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(42, "v_sad_u8", "v_sad_u8 %0, %1, %2, %0", OUTS("+v"(a[i])), INS("v"(b), "v"(c))) \
    X(43, "v_lshlrev_b64", "v_lshlrev_b64 %0, 1, %0", OUTS("+v"(w[i])), INS()) \
    X(44, "v_ashrrev_i64", "v_ashrrev_i64 %0, 1, %0", OUTS("+v"(w[i])), INS()) \
    X(45, "v_mov_b32_dpp quad_perm", "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", OUTS("+v"(a[i])), INS()) \
    X(46, "v_add_u32_dpp row_ror", "v_add_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf", OUTS("+v"(a[i])), INS()) \
    X(47, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0", OUTS("+v"(a[i])), INS()) \
    X(48, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0", OUTS("+v"(a[i])), INS()) \
    X(49, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %0", OUTS("+v"(a[i])), INS()) \
    X(50, "v_add_f32", "v_add_f32 %0, %0, %1", OUTS("+v"(f[i])), INS("v"(fb))) \
    X(51, "v_mul_f32", "v_mul_f32 %0, %0, %1", OUTS("+v"(f[i])), INS("v"(fb))) \
    X(52, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", OUTS("+v"(f[i])), INS("v"(fb), "v"(fc))) \
    X(53, "v_fmac_f32", "v_fmac_f32 %0, %1, %2", OUTS("+v"(f[i])), INS("v"(fb), "v"(fc))) \
    X(54, "v_max_f32", "v_max_f32 %0, %0, %1", OUTS("+v"(f[i])), INS("v"(fb))) \
    X(55, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2", OUTS("+v"(f[i])), INS("v"(fb), "v"(fc))) \
    X(56, "v_rndne_f32", "v_rndne_f32 %0, %0", OUTS("+v"(f[i])), INS()) \
    X(57, "v_fract_f32", "v_fract_f32 %0, %0", OUTS("+v"(f[i])), INS()) \
    X(58, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", OUTS("+v"(p[i])), INS("v"(pb), "v"(pc))) \
    X(59, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", OUTS("+v"(p[i])), INS("v"(pb))) \
    X(60, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", OUTS("+v"(p[i])), INS("v"(pb))) \
    X(61, "v_rcp_f32", "v_rcp_f32 %0, %0", OUTS("+v"(f[i])), INS()) \
    X(62, "v_sqrt_f32", "v_sqrt_f32 %0, %0", OUTS("+v"(f[i])), INS()) \
    X(63, "v_exp_f32", "v_exp_f32 %0, %0", OUTS("+v"(f[i])), INS()) \
    X(64, "v_fma_f64", "v_fma_f64 %0, %0, %1, %2", OUTS("+v"(d[i])), INS("v"(db), "v"(dc))) \
    X(65, "v_add_f64", "v_add_f64 %0, %0, %1", OUTS("+v"(d[i])), INS("v"(db))) \
    X(66, "v_mul_f64", "v_mul_f64 %0, %0, %1", OUTS("+v"(d[i])), INS("v"(db))) \
    X(67, "v_cvt_f32_f64", "v_cvt_f32_f64 %0, %1", OUTS("+v"(a[i])), INS("v"(db))) \
    X(68, "v_sqrt_f64", "v_sqrt_f64 %0, %0", OUTS("+v"(d[i])), INS()) \
    /* end */

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(int* out, int seed) {
    int a[kChains], b = seed | 1, c = seed * 3 + 7;
    long w[kChains], cmp[kChains] = {};
    float f[kChains];
    double d[kChains];
    v2f p[kChains];
#pragma unroll
    for (int i = 0; i < kChains; ++i) {
        a[i] = threadIdx.x * 17 + i + seed;
        w[i] = a[i];
        f[i] = (float)a[i];
        d[i] = (double)a[i];
        p[i] = v2f{f[i], f[i] + 1.f};
    }
    const float fb = (float)b * 1e-9f, fc = (float)c * 1e-9f;
    const double db = (double)b * 1e-9, dc = (double)c * 1e-9;
    const v2f pb = v2f{fb, fb}, pc = v2f{fc, fc};
    const long mask = __builtin_amdgcn_readfirstlane(seed) * 0x0101010101010101L;
    for (int t = 0; t < kTrips; ++t) {
#define X(n, name, text, outs, ...)                                                \
        if constexpr (OP == n) {                                                   \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(text : outs : __VA_ARGS__); \
        }
        OPS(X)
#undef X
    }
    int acc = 0;
#pragma unroll
    for (int i = 0; i < kChains; ++i) acc += a[i] + (int)w[i] + (int)f[i] + (int)d[i] + (int)p[i].x + (int)p[i].y + (int)cmp[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static int g_cus = 0;

template <int OP>
static void run(const char* name, int* out, int wps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = g_cus * wps;  // workgroups of 4 waves: one wave per SIMD each
    rate_kernel<OP><<<blocks, 256>>>(out, 3);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(out, 5);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts = (double)kChains * kTrips * blocks * 4;
    printf("| `%s` | %.3f | %.2f |\n", name, ms, 2.4e9 * (g_cus * 4.0) * (ms * 1e-3) / wave_insts);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    const int wps = argc > 1 ? atoi(argv[1]) : 4;
    if (wps < 1 || wps > 8) { fprintf(stderr, "waves per SIMD: 1..8\n"); return 2; }
    printf("%s: %d CUs, %d waves per SIMD, %d instructions per wave (8 independent chains)\n\n", prop.gcnArchName, g_cus, wps, kChains * kTrips);
    printf("| instruction | launch ms | cycles per wave64 instruction and SIMD (2.4 GHz) |\n|---|---|---|\n");
    int* out;
    CHECK(hipMalloc(&out, sizeof(int) * g_cus * 8 * 256));
#define X(n, name, text, outs, ...) run<n>(name, out, wps);
    OPS(X)
#undef X
    return 0;
}
