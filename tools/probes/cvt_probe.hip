// Probe: does v_cvt_pk_u8_f32 equal clamp(x,0,255) followed by truncation (decoder.rs:382-390 f32_to_u8)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float *x, unsigned *a, unsigned *b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    a[i] = __builtin_amdgcn_cvt_pk_u8_f32(v, 0, 0);
    float c = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
    b[i] = (unsigned)c;
}
int main()
{
    std::vector<float> h;
    for (int i = -600; i < 70000; i++) { float f = i / 256.0f; h.push_back(f); h.push_back(nextafterf(f, 1e9f)); h.push_back(nextafterf(f, -1e9f)); }
    h.push_back(1e30f); h.push_back(-1e30f); h.push_back(NAN); h.push_back(INFINITY); h.push_back(-INFINITY); h.push_back(-0.0f);
    int n = h.size();
    float *dx; unsigned *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, da, db, n);
    std::vector<unsigned> a(n), b(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (a[i] != b[i]) { if (bad < 10) printf("x=%.9g cvt_pk=%u trunc=%u\n", h[i], a[i], b[i]); bad++; }
    printf("n=%d mismatches=%d\n", n, bad);
    return 0;
}
