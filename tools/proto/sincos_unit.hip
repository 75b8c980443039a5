// sincos_unit.hip -- how accurate is the transcendental unit itself?  v_sin_f32 / v_cos_f32 take REVOLUTIONS; the fast bf16 mode
// feeds them fract(fp32(angle / 2 pi)) and owes its 2e-4 rad at 512 x to the ROUNDING of that product, not to the unit.  This
// sweeps 2^24 fp32 fractions in [0, 1) and prints the unit's max / rms absolute error against double-precision sin / cos of
// 2 pi f -- what an encoding with an exact (double-float) range reduction in front of the unit would inherit.
// build: hipcc --offload-arch=gfx950 -O3 -o sincos_unit tools/proto/sincos_unit.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(float* s, float* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float f = (float)i / (float)n;          // exact: n = 2^24
  s[i] = __builtin_amdgcn_sinf(f);
  c[i] = __builtin_amdgcn_cosf(f);
}

int main() {
  const int n = 1 << 24;
  float *ds, *dc;
  hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, ds, dc, n);
  std::vector<float> s(n), c(n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double ms = 0, mc = 0, qs = 0, qc = 0; int is = 0, ic = 0;
  for (int i = 0; i < n; ++i) {
    const double a = 2.0 * M_PI * (double)i / n;
    const double es = fabs((double)s[i] - sin(a)), ec = fabs((double)c[i] - cos(a));
    if (es > ms) { ms = es; is = i; }
    if (ec > mc) { mc = ec; ic = i; }
    qs += es * es; qc += ec * ec;
  }
  printf("v_sin_f32: max abs err %.3e at f = %.8f, rms %.3e\n", ms, (double)is / n, sqrt(qs / n));
  printf("v_cos_f32: max abs err %.3e at f = %.8f, rms %.3e\n", mc, (double)ic / n, sqrt(qc / n));
  return 0;
}
