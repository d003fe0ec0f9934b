// Probe: operand / result layout and issue rate of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per
// instruction), used for the 8x8 inverse DCT as a matrix product.
//   expectation: lane = 4 * blk + i supplies A_blk[i] and B_blk[i]; result register r of lane 4 * blk + j = D_blk[r][j]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[l * 4 + r] = acc[r];
}
__global__ void rate(float *out, int iters)
{
    float4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main()
{
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; l++) { a[l] = 1 + l; b[l] = 100 + l; }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; blk++)
        for (int j = 0; j < 4; j++)
            for (int r = 0; r < 4; r++) {
                const float want = a[4 * blk + r] * b[4 * blk + j], got = d[(4 * blk + j) * 4 + r];
                if (want != got) { if (bad < 8) printf("blk %d j %d r %d want %g got %g\n", blk, j, r, want, got); bad++; }
            }
    printf("layout mismatches: %d\n", bad);
    float *dout; hipMalloc(&dout, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    rate<<<1024, 256>>>(dout, 10);
    hipEventRecord(e0); rate<<<1024, 256>>>(dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 1024.0 * 4 * iters * 4;      // wave-instructions
    printf("%.3f ms, %.2f cycles per instruction per SIMD at 2.4 GHz (1024 SIMDs)\n", ms, ms * 1e-3 * 2.4e9 / (insts / 1024));
    return 0;
}
