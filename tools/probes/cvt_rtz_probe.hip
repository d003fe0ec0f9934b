// Probe: with MODE.FP_ROUND = toward-zero, is v_cvt_pk_u8_f32 equal to clamp(x,0,255) + truncation (decoder.rs:382-390)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, unsigned *a, unsigned *b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    unsigned r;
    // hwreg(HW_REG_MODE = 1, offset 0, size 2) = FP_ROUND for f32: 0 nearest-even, 3 toward zero
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 2\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_nop 0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 2"
                 : "=v"(r) : "v"(v));
    a[i] = r;
    float c = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
    b[i] = (unsigned)c;
}
int main()
{
    std::vector<float> h;
    for (int i = -600; i < 70000; i++) { float f = i / 256.0f; h.push_back(f); h.push_back(nextafterf(f, 1e9f)); h.push_back(nextafterf(f, -1e9f)); }
    h.push_back(1e30f); h.push_back(-1e30f); h.push_back(-0.0f); h.push_back(255.99998f); h.push_back(256.0f);
    int n = h.size();
    float *dx; unsigned *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, da, db, n);
    std::vector<unsigned> a(n), b(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (a[i] != b[i]) { if (bad < 10) printf("x=%.9g cvt_pk(rtz)=%u trunc=%u\n", h[i], a[i], b[i]); bad++; }
    printf("n=%d mismatches=%d\n", n, bad);
    return 0;
}
