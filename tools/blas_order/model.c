#include <math.h>
#include <stdint.h>
#include <string.h>
// dot(a, x) of length n as L-lane x U-accumulator FMA kernel, then reduction.
// acc_tree: 0 sequential ((a0+a1)+a2)+a3, 1 pairwise (a0+a1)+(a2+a3)
// lane_tree: 0 halves (v[i]+v[i+L/2] repeatedly), 1 adjacent pairs (hadd), 2 sequential
// blk: K block size (0 = none): partial results of each block are reduced and added to the running y
float model_dot(const float* a, const float* x, int n, int L, int U, int use_fma, int acc_tree, int lane_tree, int blk) {
  float y = 0.f; int first = 1;
  if (blk <= 0) blk = n;
  for (int k0 = 0; k0 < n; k0 += blk) {
    int kn = n - k0 < blk ? n - k0 : blk;
    float acc[8][64];
    for (int u = 0; u < U; ++u) for (int l = 0; l < L; ++l) acc[u][l] = 0.f;
    int step = L * U, k = 0;
    for (; k + step <= kn; k += step)
      for (int u = 0; u < U; ++u)
        for (int l = 0; l < L; ++l) {
          int idx = k0 + k + u * L + l;
          acc[u][l] = use_fma ? fmaf(a[idx], x[idx], acc[u][l]) : acc[u][l] + a[idx] * x[idx];
        }
    // remaining full vectors go to accumulator 0.. in order
    int u = 0;
    for (; k + L <= kn; k += L, u = (u + 1) % U)
      for (int l = 0; l < L; ++l) { int idx = k0 + k + l; acc[u][l] = use_fma ? fmaf(a[idx], x[idx], acc[u][l]) : acc[u][l] + a[idx] * x[idx]; }
    float v[64];
    if (acc_tree == 0) { for (int l = 0; l < L; ++l) { float s = acc[0][l]; for (int q = 1; q < U; ++q) s += acc[q][l]; v[l] = s; } }
    else { for (int l = 0; l < L; ++l) { float t[8]; for (int q = 0; q < U; ++q) t[q] = acc[q][l]; int m = U; while (m > 1) { for (int q = 0; q < m / 2; ++q) t[q] = t[2*q] + t[2*q+1]; m /= 2; } v[l] = t[0]; } }
    float s;
    if (lane_tree == 0) { int m = L; while (m > 1) { m /= 2; for (int l = 0; l < m; ++l) v[l] = v[l] + v[l + m]; } s = v[0]; }
    else if (lane_tree == 1) { int m = L; while (m > 1) { for (int l = 0; l < m / 2; ++l) v[l] = v[2*l] + v[2*l+1]; m /= 2; } s = v[0]; }
    else { s = v[0]; for (int l = 1; l < L; ++l) s += v[l]; }
    // scalar tail
    for (; k < kn; ++k) s = use_fma ? fmaf(a[k0 + k], x[k0 + k], s) : s + a[k0 + k] * x[k0 + k];
    if (first) { y = s; first = 0; } else y += s;
  }
  return y;
}
