// fft_host_check.cpp — TEST HARNESS (tests/test_fir_fft_host.py, -m "not gpu"): runs the __host__ __device__ core of
// iac_amd/csrc/render_fir_fft.hpp lane by lane on the CPU (64 "lanes", an array as the wave's LDS scratch) and checks
//   1. the staged 1024-point forward transform against a float64 DFT (bin order through fft_bin),
//   2. inverse(forward(x)) = N x,
//   3. one whole overlap-save hop (pair packing, U / V accumulation with the host-built tables, mirror, inverse)
//      against a float64 direct convolution, for M = 16, 9, 1 channels and 256 / 33 taps.
// Prints the maximum errors; exit code 1 if any is out of bounds.
#define FFT_HD inline
#define IAMF_FFT_HOST_TABLES
#include <math.h>

#include <vector>

#include "../../iac_amd/csrc/render_fir_fft.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <complex>
#include <random>

typedef std::complex<double> cd;

static void forward(fft_c32 z[64][16], const std::vector<float> &tw, fft_c32 *S) {
  for (int l = 0; l < 64; ++l) {
    fft_c32 tw1[16];
    for (int k = 0; k < 16; ++k) tw1[k] = fft_mk(tw[(k * 64 + l) * 2], tw[(k * 64 + l) * 2 + 1]);
    fft_fwd_a(z[l], tw1);
  }
  for (int l = 0; l < 64; ++l) fft_x1_write(z[l], l, S);
  for (int l = 0; l < 64; ++l) fft_x1_read(z[l], l, S);
  for (int l = 0; l < 64; ++l) {
    fft_c32 tw2[16];
    for (int k = 0; k < 16; ++k) tw2[k] = fft_mk(tw[(16 * 64 + k * 4 + (l & 3)) * 2], tw[(16 * 64 + k * 4 + (l & 3)) * 2 + 1]);
    fft_fwd_b(z[l], tw2);
  }
  for (int l = 0; l < 64; ++l) fft_x2_write(z[l], l, S);
  for (int l = 0; l < 64; ++l) fft_x2_read(z[l], l, S);
  for (int l = 0; l < 64; ++l) fft_fwd_c(z[l]);
}

static void inverse(fft_c32 z[64][16], const std::vector<float> &tw, fft_c32 *S) {
  for (int l = 0; l < 64; ++l) fft_inv_c(z[l]);
  for (int l = 0; l < 64; ++l) fft_x2_write_back(z[l], l, S);
  for (int l = 0; l < 64; ++l) fft_x2_read_back(z[l], l, S);
  for (int l = 0; l < 64; ++l) {
    fft_c32 tw2[16];
    for (int k = 0; k < 16; ++k) tw2[k] = fft_mk(tw[(16 * 64 + k * 4 + (l & 3)) * 2], tw[(16 * 64 + k * 4 + (l & 3)) * 2 + 1]);
    fft_inv_b(z[l], tw2);
  }
  for (int l = 0; l < 64; ++l) fft_x1_write_back(z[l], l, S);
  for (int l = 0; l < 64; ++l) fft_x1_read_back(z[l], l, S);
  for (int l = 0; l < 64; ++l) {
    fft_c32 tw1[16];
    for (int k = 0; k < 16; ++k) tw1[k] = fft_mk(tw[(k * 64 + l) * 2], tw[(k * 64 + l) * 2 + 1]);
    fft_inv_a(z[l], tw1);
  }
}

int main() {
  std::mt19937 rng(12345);
  std::normal_distribution<float> nd(0.f, 1.f);
  int bad = 0;
  std::vector<float> pq, tw;
  {
    std::vector<float> h0(2 * 2 * 4, 0.f);
    fft_build_tables(h0.data(), 2, 4, pq, tw);
  }
  static fft_c32 z[64][16];
  std::vector<fft_c32> S(kFftScratch);
  // 1. forward against a float64 DFT
  std::vector<cd> x(kFftN), X(kFftN);
  for (auto &v : x) v = cd(nd(rng), nd(rng));
  for (int k = 0; k < kFftN; ++k) {
    cd a = 0;
    for (int n = 0; n < kFftN; ++n) a += x[n] * std::polar(1.0, -2 * M_PI * ((n * k) % kFftN) / kFftN);
    X[k] = a;
  }
  for (int l = 0; l < 64; ++l)
    for (int n1 = 0; n1 < 16; ++n1) z[l][n1] = fft_mk((float)x[l + 64 * n1].real(), (float)x[l + 64 * n1].imag());
  forward(z, tw, S.data());
  double e1 = 0, seen = 0;
  std::vector<int> hit(kFftN, 0);
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int k = fft_bin(l, r);
      hit[k]++;
      e1 = fmax(e1, std::abs(cd(z[l][r].x, z[l][r].y) - X[k]));
    }
  for (int k = 0; k < kFftN; ++k) seen += hit[k] == 1;
  printf("forward: max |err| %.3e (spectrum magnitude ~%.1f), bins covered once %d / %d\n", e1, sqrt(2.0 * kFftN), (int)seen, kFftN);
  if (e1 > 2e-4 || seen != kFftN) bad = 1;
  // 2. the inverse undoes it
  inverse(z, tw, S.data());
  double e2 = 0;
  for (int l = 0; l < 64; ++l)
    for (int n1 = 0; n1 < 16; ++n1)
      e2 = fmax(e2, std::abs(cd(z[l][n1].x, z[l][n1].y) / (double)kFftN - x[l + 64 * n1]));
  printf("inverse(forward(x)) / N - x: max |err| %.3e\n", e2);
  if (e2 > 2e-6) bad = 1;
  // 3. one overlap-save hop
  const int cases[4][2] = {{16, 256}, {9, 33}, {1, 256}, {12, 200}};
  for (auto &cs : cases) {
    const int M = cs[0], taps = cs[1];
    std::vector<float> h((size_t)2 * M * taps), xin((size_t)M * kFftN);
    for (size_t i = 0; i < h.size(); ++i) h[i] = nd(rng) * 0.08f * expf(-(float)(i % taps) / (taps / 6.0f));
    for (auto &v : xin) v = nd(rng) * 0.25f;
    const int pairs = fft_build_tables(h.data(), M, taps, pq, tw);
    fft_c32 u[64][16], v[64][16];
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) u[l][r] = v[l][r] = fft_mk(0.f, 0.f);
    for (int p = 0; p < pairs; ++p) {
      const int a = 2 * p, b = 2 * p + 1;
      for (int l = 0; l < 64; ++l)
        for (int n1 = 0; n1 < 16; ++n1)
          z[l][n1] = fft_mk(xin[(size_t)a * kFftN + l + 64 * n1], b < M ? xin[(size_t)b * kFftN + l + 64 * n1] : 0.f);
      forward(z, tw, S.data());
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
          const float *t = &pq[(((size_t)p * 16 + r) * 64 + l) * 4];
          u[l][r] = fft_cmac(u[l][r], z[l][r], fft_mk(t[0], t[1]));
          v[l][r] = fft_cmac(v[l][r], z[l][r], fft_mk(t[2], t[3]));
        }
    }
    for (int l = 0; l < 64; ++l) fft_mirror_write(v[l], l, S.data());
    for (int l = 0; l < 64; ++l) fft_mirror_read_add(u[l], l, S.data());
    inverse(u, tw, S.data());
    double e3 = 0, ymax = 0;
    for (int n = kFftN - kFftHop; n < kFftN; ++n) {
      double yl = 0, yr = 0;
      for (int c = 0; c < M; ++c)
        for (int k = 0; k < taps; ++k) {
          yl += (double)h[((size_t)0 * M + c) * taps + k] * xin[(size_t)c * kFftN + n - k];
          yr += (double)h[((size_t)1 * M + c) * taps + k] * xin[(size_t)c * kFftN + n - k];
        }
      const fft_c32 g = u[n & 63][n >> 6];
      e3 = fmax(e3, fmax(fabs(g.x - yl), fabs(g.y - yr)));
      ymax = fmax(ymax, fmax(fabs(yl), fabs(yr)));
    }
    printf("hop M=%d taps=%d: max |err| %.3e (max |y| %.3f)\n", M, taps, e3, ymax);
    if (e3 > 1.9e-6 * fmax(1.0, ymax)) bad = 1;   // a quarter of the stated 2^-17
  }
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
