// Phase timing of the blocked Cholesky + inverse (dense.hip) for l = 60: shader-clock stamps at the
// phase boundaries, and the kernel's duration from HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSAPCA_CHOL_TIMING -I single-algebra_amd/csrc -I include \
//         tools/ubench/chol_phases.hip -o /tmp/chol_phases -ldl
#include "../../single-algebra_amd/csrc/dense.hip"
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  const int l = argc > 1 ? atoi(argv[1]) : 60, ld = l <= 64 ? 64 : 128;
  std::mt19937_64 g(5);
  std::normal_distribution<double> nd;
  std::vector<double> P(400 * l), G((size_t)ld * ld, 0.0);
  for (auto& x : P) x = nd(g);
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) {
      double s = 0;
      for (int r = 0; r < 400; ++r) s += P[r * l + i] * P[r * l + j];
      G[(size_t)i * ld + j] = s;
    }
  double *dG, *dR, *dX;
  int* dinfo;
  hipMalloc(&dG, G.size() * 8); hipMalloc(&dR, G.size() * 8); hipMalloc(&dX, G.size() * 8); hipMalloc(&dinfo, 4);
  hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
  hipMemset(dinfo, 0, 4);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 5; ++i) sapca::k::chol_inv(dG, l, ld, dR, dX, dinfo, s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) sapca::k::chol_inv(dG, l, ld, dR, dX, dinfo, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("l=%d: %.1f us per call (20 back-to-back launches)\n", l, ms * 1000 / 20);
  }
  unsigned long long st[32];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(sapca::k::sapca_chol_stamps), sizeof(st));
  const char* names[] = {"load G", "factor b0", "trail b0", "factor b1", "trail b1", "factor b2", "trail b2", "factor b3", "trail b3",
                         "diag inverses", "off-diag 1", "off-diag 2", "off-diag 3", "store"};
  const int idx[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14};
  for (int k = 0; k < 14; ++k) printf("%-14s %8llu cycles\n", names[k], st[idx[k + 1]] - st[idx[k]]);
  printf("total          %8llu cycles\n", st[14] - st[0]);
  // check
  std::vector<double> R(G.size()), X(G.size());
  hipMemcpy(R.data(), dR, G.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(X.data(), dX, G.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, errx = 0;
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < l; ++k) { s1 += R[k * ld + i] * R[k * ld + j]; s2 += R[i * ld + k] * X[k * ld + j]; }
      err = fmax(err, fabs(s1 - G[(size_t)i * ld + j]) / G[0]);
      errx = fmax(errx, fabs(s2 - (i == j)));
    }
  printf("max |R^T R - G| / G00 = %.2e, max |R Rinv - I| = %.2e\n", err, errx);
  return 0;
}
