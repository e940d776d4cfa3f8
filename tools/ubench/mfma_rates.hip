// tools/ubench/mfma_rates.hip — cycles per v_mfma_f32_16x16x4_f32 and per v_mfma_i32_16x16x64_i8 on one SIMD with 1, 2, 3 or 4 waves issuing them
// (12 independent accumulators per wave, the arrangement of the 1x1-convolution kernels).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rates tools/ubench/mfma_rates.hip && /tmp/mfma_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kTrips = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, float seed) {
    f32x4 acc[12];
    v4i iacc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
        iacc[i] = (v4i){(int)seed, 0, 0, 0};
    }
    float a[4], b[3];
    v4i ia = {(int)threadIdx.x, 1, 2, 3}, ib = {4, 5, (int)threadIdx.x, 7};
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = seed * (i + 1) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = seed * (i + 5);
    for (int t = 0; t < kTrips; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if constexpr (KIND == 0) acc[g * 3 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b[c], acc[g * 3 + c], 0, 0, 0);
                    else iacc[g * 3 + c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, iacc[g * 3 + c], 0, 0, 0);
                }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) r += acc[i][0] + acc[i][3] + (float)iacc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static int run(const char* name, float* out, int cus, int wps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = cus * wps;  // 256-thread workgroups: one wave per SIMD each
    mfma_kernel<KIND><<<blocks, 256>>>(out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    mfma_kernel<KIND><<<blocks, 256>>>(out, 2.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = 48.0 * kTrips * wps;
    printf("| `%s` | %d | %.3f | %.1f |\n", name, wps, ms, 2.4e9 * ms * 1e-3 / per_simd);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 256));
    printf("| instruction (12 independent accumulators per wave) | waves per SIMD | launch ms | cycles per instruction and SIMD (2.4 GHz) |\n|---|---|---|---|\n");
    for (int w = 1; w <= 4; ++w)
        if (run<0>("v_mfma_f32_16x16x4_f32", out, cus, w)) return 1;
    for (int w = 1; w <= 4; ++w)
        if (run<1>("v_mfma_i32_16x16x64_i8", out, cus, w)) return 1;
    return 0;
}
