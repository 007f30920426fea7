#include <math.h>
#include <stdint.h>
// candidate orders of OpenBLAS 0.3.29 sdot, any n.  tail: 0 = mul then add, 1 = fma
static float tail_seq(float dot, const float* a, const float* x, int64_t i, int64_t n, int fma) {
  // the elements behind the last whole 32: their f32 products summed in DOUBLE, the kernel's f32 result added last, rounded once
  double t = 0.0;
  for (; i < n; ++i) {
    if (fma) t = __builtin_fma((double)a[i], (double)x[i], t);          // (variant: exact products)
    else { volatile float p = a[i] * x[i]; t += (double)p; }
  }
  return (float)(t + (double)dot);
}
float sdot_skx(const float* a, const float* x, int64_t n, int fma) {
  int64_t n1 = n & -32, i = 0;
  float dot = 0.f;
  if (n1) {
    float acc[4][8] = {{0}};
    int64_t n64 = n1 & ~63LL;
    if (n64) {
      float w[4][16] = {{0}};
      for (; i < n64; i += 64)
        for (int j = 0; j < 4; ++j) for (int l = 0; l < 16; ++l) w[j][l] = fmaf(a[i + 16 * j + l], x[i + 16 * j + l], w[j][l]);
      for (int j = 0; j < 4; ++j) for (int l = 0; l < 8; ++l) acc[j][l] = w[j][l] + w[j][l + 8];
    }
    for (; i < n1; i += 32)
      for (int j = 0; j < 4; ++j) for (int l = 0; l < 8; ++l) acc[j][l] = fmaf(a[i + 8 * j + l], x[i + 8 * j + l], acc[j][l]);
    float s[8];
    for (int l = 0; l < 8; ++l) s[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
    float h[4];
    for (int l = 0; l < 4; ++l) h[l] = s[l] + s[l + 4];
    dot = (h[0] + h[1]) + (h[2] + h[3]);
  }
  return tail_seq(dot, a, x, n1, n, fma);
}
float sdot_hsw(const float* a, const float* x, int64_t n, int fma) {
  int64_t n1 = n & -32, i = 0;
  float dot = 0.f;
  if (n1) {
    float acc[4][8] = {{0}};
    for (; i < n1; i += 32)
      for (int j = 0; j < 4; ++j) for (int l = 0; l < 8; ++l) acc[j][l] = fmaf(a[i + 8 * j + l], x[i + 8 * j + l], acc[j][l]);
    float q[4][4];
    for (int j = 0; j < 4; ++j) for (int l = 0; l < 4; ++l) q[j][l] = acc[j][l] + acc[j][l + 4];
    float h[4];
    for (int l = 0; l < 4; ++l) h[l] = (q[0][l] + q[1][l]) + (q[2][l] + q[3][l]);
    dot = (h[0] + h[1]) + (h[2] + h[3]);
  }
  return tail_seq(dot, a, x, n1, n, fma);
}
