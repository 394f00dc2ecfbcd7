// Exhaustive check of the descriptor kernel's short double-precision sincos (sincos_small in
// orb_slam3_amd/csrc/orbx_extract.hip, restated here) against the C library: for every float in
// [0, 6.2832] taken as the angle in radians, and for every float angle in [0, 360] degrees times
// the float factor pi/180, (float)cos((double)rad) and (float)sin((double)rad) must be the same
// floats.  g++ -O2 -ffp-contract=off -mfma tools/sincos_exhaustive.cpp && ./a.out  (about 40 s)
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
static void sincos_small(double x, double& s, double& c) {
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(z * r, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const int n = (int)k & 3;
  const double s0 = (n & 1) ? cr : sr, c0 = (n & 1) ? sr : cr;
  s = (n & 2) ? -s0 : s0;
  c = ((n + 1) & 2) ? -c0 : c0;
}
int main() {
  // every float in [0, 6.2832] as rad: too many (1.1e9)? floats in [2^-20, 6.3]: ~ 23*8M = 190M; do all.
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);
  long bad = 0, total = 0; double maxulp = 0;
  uint32_t lo, hi; float f0 = 0.0f, f1 = 6.2832f; memcpy(&lo, &f0, 4); memcpy(&hi, &f1, 4);
  for (uint32_t u = lo; u <= hi; u += 1) {
    float rad; memcpy(&rad, &u, 4);
    double s, c; sincos_small((double)rad, s, c);
    float a = (float)c, b = (float)s;
    float ra = (float)cos((double)rad), rb = (float)sin((double)rad);
    ++total;
    if (a != ra || b != rb) { if (bad < 10) printf("mismatch rad=%.9g a=%.9g ra=%.9g b=%.9g rb=%.9g\n", rad, a, ra, b, rb); ++bad; }
  }
  // also the exact set of inputs the kernel can see: angle floats in [0,360] times factorPI
  long bad2 = 0, total2 = 0;
  float a0 = 0.0f, a1 = 360.0f; memcpy(&lo, &a0, 4); memcpy(&hi, &a1, 4);
  for (uint32_t u = lo; u <= hi; u += 1) {
    float ang; memcpy(&ang, &u, 4);
    float rad = ang * factorPI;
    double s, c; sincos_small((double)rad, s, c);
    if ((float)c != (float)cos((double)rad) || (float)s != (float)sin((double)rad)) ++bad2;
    ++total2;
  }
  printf("rad sweep: %ld mismatches of %ld; angle sweep: %ld of %ld\n", bad, total, bad2, total2);
  return 0;
}
