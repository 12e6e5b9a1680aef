// mcbrat_host.cpp -- host-side set-up routines that sit on the hot path's boundary:
// the inverse phase-function table builder and the thermal-emission weighting.  They run
// once per domain on the CPU in the reference too (src/inversePhaseFunctions.f95,
// src/emissionAndBroadBandWeights.f95); the GPU kernels only consume their outputs.
// Arithmetic kinds follow the reference: `real` -> float, `real(8)` -> double, evaluated
// in the written order (build with -ffp-contract=off).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mcbrat.h"

namespace {

inline float ulp_spacing(float x) {  // Fortran SPACING for default real
  if (x == 0.0f) return FLT_MIN;
  int e;
  std::frexp(std::fabs(x), &e);
  const float s = std::ldexp(1.0f, e - 24);
  return s < FLT_MIN ? FLT_MIN : s;
}

// computeLegendrePolynomials, src/numericUtilities.f95:187-205.  P[j*(maxL+1) + l].
void legendre_polynomials(int maxL, const std::vector<float> &mu, std::vector<float> &P) {
  const int ld = maxL + 1;
  P.assign((size_t)ld * mu.size(), 0.0f);
  for (size_t j = 0; j < mu.size(); ++j) {
    float *p = &P[j * ld];
    p[0] = 1.0f;
    if (maxL >= 1) p[1] = mu[j];
    for (int l = 1; l < maxL; ++l)
      p[l + 1] = (((float)(2 * l + 1) * mu[j]) * p[l] - (float)l * p[l - 1]) / (float)(l + 1);
  }
}

// computeLobattoTerms, src/numericUtilities.f95:27-114 (abscissas only are used downstream).
std::vector<float> lobatto_abscissas(int n) {
  const int mid = (n + 1) / 2, m = mid - 1;
  const float pi = std::acos(-1.0f);
  const float c1 = (n % 2 == 1) ? 1.0f : 0.5f;
  const float denom = ((float)n - 1.0f) + 0.5f;
  const float nm1 = (float)(n - 1), nn1 = (float)(n * (n - 1));
  std::vector<float> x(m), prev(m), d1(m), d2(m), P;
  for (int i = 1; i <= m; ++i) x[i - 1] = std::sin((pi * ((float)i - c1)) / denom);
  auto newton = [&](int j) {
    const float *p = &P[(size_t)j * n];
    d1[j] = (nm1 * (x[j] * p[n - 1] - p[n - 2])) / (x[j] * x[j] - 1.0f);
    d2[j] = ((2.0f * x[j]) * d1[j] - (nn1 * p[n - 1])) / (1.0f - x[j] * x[j]);
    prev[j] = x[j];
    x[j] = x[j] - d1[j] / d2[j];
  };
  legendre_polynomials(n - 1, x, P);
  for (int j = 0; j < m; ++j) newton(j);
  for (int it = 0;; ++it) {
    bool done = true;
    for (int j = 0; j < m; ++j)
      if (!(std::fabs(x[j] - prev[j]) <= 3.0f * ulp_spacing(x[j]))) done = false;
    if (done) break;
    legendre_polynomials(n - 1, x, P);
    for (int j = 0; j < m; ++j)
      if (std::fabs(x[j] - prev[j]) > 3.0f * ulp_spacing(x[j])) newton(j);
    if (it + 1 > 25) break;
  }
  std::vector<float> mus(n, 0.0f);
  mus[0] = -1.0f;
  for (int j = 0; j < m; ++j) mus[mid - 1 - j] = -x[j];
  if (n % 2 == 0) {
    for (int k = 0; k < mid; ++k) mus[mid + k] = -mus[mid - 1 - k];
  } else {
    std::vector<float> t(mid);
    for (int k = 0; k < mid; ++k) t[k] = -mus[mid - 1 - k];
    for (int k = 0; k < mid; ++k) mus[mid - 1 + k] = t[k];
  }
  return mus;
}

// findIndexReal, src/numericUtilities.f95:417-470; t is 1-based through T().
int find_index(float v, const std::vector<float> &t, int guess) {
  const int n = (int)t.size();
  auto T = [&](int i) { return t[i - 1]; };
  int lo, hi;
  if (guess > 0) {
    lo = guess;
    int inc = 1;
    for (;;) {
      hi = std::min(lo + inc, n);
      if (lo == n || (T(lo) <= v && T(hi) > v)) break;
      if (T(lo) > v) { hi = lo; lo = std::max(hi - inc, 1); }
      else lo = hi;
      inc *= 2;
    }
  } else { lo = 0; hi = n; }
  while (!(lo == n || hi <= lo + 1)) {
    const int mid = (lo + hi) / 2;
    if (v >= T(mid)) lo = mid; else hi = mid;
  }
  return lo;
}

// The CDF build and its analytic inversion, src/inversePhaseFunctions.f95:114-169.
void invert_cdf(const std::vector<float> &mu, const std::vector<float> &val, int nSteps, float *table) {
  const int n = (int)mu.size();
  std::vector<float> cdf(n);
  cdf[0] = 0.0f;
  for (int i = 1; i < n; ++i) cdf[i] = cdf[i - 1] + ((mu[i] - mu[i - 1]) * 0.5f) * (val[i] + val[i - 1]);
  const float total = cdf[n - 1];
  for (int i = 0; i < n; ++i) cdf[i] = cdf[i] / total;
  int where = find_index(0.0f, cdf, 0);
  for (int i = 1; i <= nSteps - 1; ++i) {
    const float prob = (float)(i - 1) / (float)(nSteps - 1);
    if (i > 1) where = find_index(prob, cdf, where);
    int k = where;
    if (k >= n) k = n - 1;
    const float c0 = cdf[k - 1], c1 = cdf[k], m0 = mu[k - 1], m1 = mu[k], v0 = val[k - 1], v1 = val[k];
    float cosine;
    if (c1 - c0 <= ulp_spacing(c0)) cosine = m0;
    else if (std::fabs(v0 - v1) <= ulp_spacing(v0)) cosine = m0 + ((m1 - m0) * (prob - c0)) / (c1 - c0);
    else {
      const float root = std::sqrt(((c1 - prob) * (v0 * v0) + (prob - c0) * (v1 * v1)) / (c1 - c0));
      cosine = m0 + ((m1 - m0) / (v0 - v1)) * (v0 - root);
    }
    table[i - 1] = std::acos(cosine);
  }
  // the reference also locates prob = 1 (:132-135) but never uses it
  table[nSteps - 1] = 0.0f;
}

}  // namespace

extern "C" int mcbrat_inverse_table_legendre(int32_t nCoef, const float *coef, int32_t nSteps, float *table) {
  if (nCoef < 0 || nSteps < 2 || !table || (nCoef > 0 && !coef)) return 1;
  const int n = nCoef > 2 ? nCoef : 2;  // :107
  const std::vector<float> mu = lobatto_abscissas(n);
  // getPhaseFunctionValues at acos(mu) in decreasing-mu order (:111), Legendre sum
  // (src/scatteringPhaseFunctions.f95:485-497), then flipped back (:112).
  std::vector<float> val(n);
  if (nCoef == 0) {
    for (auto &v : val) v = 0.5f;
  } else {
    std::vector<float> c(n), P;
    for (int j = 0; j < n; ++j) c[j] = std::cos(std::acos(mu[j]));
    legendre_polynomials(nCoef, c, P);
    std::vector<float> wgt(nCoef + 1);
    for (int l = 0; l <= nCoef; ++l) wgt[l] = (l == 0 ? 1.0f : coef[l - 1]) * (float)(2 * l + 1);
    for (int j = 0; j < n; ++j) {
      float s = 0.0f;
      for (int l = 0; l <= nCoef; ++l) s += wgt[l] * P[(size_t)j * (nCoef + 1) + l];
      val[j] = s;
    }
  }
  invert_cdf(mu, val, nSteps, table);
  return 0;
}

extern "C" int mcbrat_inverse_table_tabulated(int32_t nAngles, const float *angle, const float *value, int32_t nSteps,
                                              float *table) {
  if (nAngles < 2 || nSteps < 2 || !angle || !value || !table) return 1;
  // normalizePhaseFunction, src/scatteringPhaseFunctions.f95:1520-1536
  float dot = 0.0f;
  for (int i = 0; i < nAngles - 1; ++i)
    dot += (std::cos(angle[i + 1]) - std::cos(angle[i])) * (0.5f * (value[i + 1] + value[i]));
  std::vector<float> stored(nAngles);
  for (int i = 0; i < nAngles; ++i) stored[i] = (-value[i] * 2.0f) / dot;
  // getPhaseFunctionValues at the native angles (:500-527), then increasing mu (:98)
  std::vector<float> ang(angle, angle + nAngles), mu(nAngles), val(nAngles);
  for (int l = 0; l < nAngles; ++l) {
    const int ti = find_index(ang[l], ang, 0);
    int tp = ti + 1;
    float dMu;
    if (ti < nAngles) dMu = std::cos(ang[tp - 1]) - std::cos(ang[ti - 1]);
    else { dMu = FLT_MAX; tp = ti; }
    const float wt = 1.0f - (std::cos(ang[l]) - std::cos(ang[ti - 1])) / dMu;
    const float v = wt * stored[ti - 1] + (1.0f - wt) * stored[tp - 1];
    val[nAngles - 1 - l] = v;
    mu[nAngles - 1 - l] = std::cos(ang[l]);
  }
  invert_cdf(mu, val, nSteps, table);
  return 0;
}

// ---- forward phase-function tables for radiance (tabulateForwardPhaseFunctions,
// src/opticalProperties.f95:1872-1935): values at nAngles equally spaced scattering angles ----
namespace {
constexpr float kPiF = 3.14159265358979312f;

std::vector<float> forward_angles(int nAngles) {  // :1914
  std::vector<float> a(nAngles);
  for (int j = 0; j < nAngles; ++j) a[j] = ((float)j / (float)(nAngles - 1)) * kPiF;
  return a;
}

// computeNormalization (:2026-2050) and phaseFuncDiff (:2011-2024); t is the 1-based transition index
float hybrid_norm(const std::vector<float> &cosA, const float *v, const std::vector<float> &g, int t) {
  const int n = (int)cosA.size();
  float ig = 0.0f, io = 0.0f;
  for (int k = 1; k <= t - 1; ++k) ig += (0.5f * (g[k - 1] + g[k])) * (cosA[k - 1] - cosA[k]);
  for (int k = t; k <= n - 1; ++k) io += (0.5f * (v[k - 1] + v[k])) * (cosA[k - 1] - cosA[k]);
  return io >= 2.0f ? 1.0f / ig : (2.0f - io) / ig;
}
float hybrid_diff(const std::vector<float> &cosA, const float *v, const std::vector<float> &g, int t) {
  return hybrid_norm(cosA, v, g, t) * g[t - 1] - v[t - 1];
}
}  // namespace

extern "C" int mcbrat_forward_table_legendre(int32_t nCoef, const float *coef, int32_t nAngles, float *table) {
  if (nCoef < 0 || nAngles < 2 || !table || (nCoef > 0 && !coef)) return 1;
  if (nCoef == 0) {  // isotropic: P0 only (scatteringPhaseFunctions.f95:486-491)
    for (int j = 0; j < nAngles; ++j) table[j] = 0.5f;
    return 0;
  }
  const std::vector<float> ang = forward_angles(nAngles);
  std::vector<float> c(nAngles), P;
  for (int j = 0; j < nAngles; ++j) c[j] = std::cos(ang[j]);
  legendre_polynomials(nCoef, c, P);
  for (int j = 0; j < nAngles; ++j) {  // :493-496
    float s = 0.0f;
    for (int l = 0; l <= nCoef; ++l) s += ((l == 0 ? 1.0f : coef[l - 1]) * (float)(2 * l + 1)) * P[(size_t)j * (nCoef + 1) + l];
    table[j] = s;
  }
  return 0;
}

extern "C" int mcbrat_forward_table_tabulated(int32_t nStored, const float *angle, const float *value, int32_t nAngles,
                                              float *table) {
  if (nStored < 2 || nAngles < 2 || !angle || !value || !table) return 1;
  float dot = 0.0f;  // normalizePhaseFunction, scatteringPhaseFunctions.f95:1520-1536
  for (int i = 0; i < nStored - 1; ++i) dot += (std::cos(angle[i + 1]) - std::cos(angle[i])) * (0.5f * (value[i + 1] + value[i]));
  std::vector<float> stored(nStored), st(angle, angle + nStored);
  for (int i = 0; i < nStored; ++i) stored[i] = (-value[i] * 2.0f) / dot;
  const std::vector<float> ang = forward_angles(nAngles);
  for (int l = 0; l < nAngles; ++l) {  // interpolation in the cosine of the angle, :500-527
    const int ti = std::max(1, find_index(ang[l], st, 0));
    int tp = ti + 1;
    float dMu;
    if (ti < nStored) dMu = std::cos(st[tp - 1]) - std::cos(st[ti - 1]);
    else { dMu = FLT_MAX; tp = ti; }
    const float wt = 1.0f - (std::cos(ang[l]) - std::cos(st[ti - 1])) / dMu;
    table[l] = wt * stored[ti - 1] + (1.0f - wt) * stored[tp - 1];
  }
  return 0;
}

// computeHybridPhaseFunctions (:1937-2009): the forward peak of each entry is replaced by a Gaussian of the given
// width, continuous with the original at the transition angle and normalised together with it.
extern "C" int mcbrat_hybrid_phase_functions(int32_t nAngles, int32_t nEntries, const float *values, float widthDeg,
                                             float *out) {
  if (nAngles < 4 || nEntries < 1 || !values || !out || !(widthDeg > 0.0f)) return 1;
  const std::vector<float> ang = forward_angles(nAngles);
  std::vector<float> g(nAngles), cosA(nAngles);
  for (int k = 0; k < nAngles; ++k) {
    cosA[k] = std::cos(ang[k]);
    const float x = ang[k] / (widthDeg * kPiF / 180.0f);
    g[k] = std::exp(-(x * x));
  }
  std::memcpy(out, values, sizeof(float) * (size_t)nAngles * nEntries);
  for (int e = 0; e < nEntries; ++e) {
    const float *v = values + (size_t)e * nAngles;
    int lo = find_index(widthDeg * kPiF / 180.0f, ang, 0) + 1;
    if (lo >= nAngles - 2) break;
    float dLo = hybrid_diff(cosA, v, g, lo), dUp = 0.0f;
    int inc = 1, up = lo;
    bool root = true;
    for (;;) {  // hunt for a sign change
      up = std::min(lo + inc, nAngles - 1);
      dUp = hybrid_diff(cosA, v, g, up);
      if (lo == nAngles - 1) { root = false; break; }
      if (dLo * dUp < 0.0f) break;
      lo = up; dLo = dUp; inc *= 2;
    }
    if (!root) continue;  // no transition angle: the original phase function stays
    while (up > lo + 1) {  // bisection
      const int mid = (lo + up) / 2;
      const float dMid = hybrid_diff(cosA, v, g, mid);
      if (dMid * dUp < 0.0f) { lo = mid; dLo = dMid; } else { up = mid; dUp = dMid; }
    }
    const float P0 = hybrid_norm(cosA, v, g, lo);
    for (int k = 0; k < lo; ++k) out[(size_t)e * nAngles + k] = P0 * g[k];
  }
  return 0;
}

// emission_weightingNEW, src/emissionAndBroadBandWeights.f95:424-550.
extern "C" int mcbrat_emission_weighting(int32_t nx, int32_t ny, int32_t nz, int32_t nc, const double *xe,
                                         const double *ye, const double *ze, const double *temps,
                                         const double *totalExt, const double *cumExt, const double *ssa,
                                         double albedo, double lambdaMicrons, double sfcTemp, double dLambda,
                                         double *voxelWeights, double *fracAtmsPower, double *totalFlux) {
  if (nx < 1 || ny < 1 || nz < 1 || nc < 1 || !voxelWeights || !fracAtmsPower) return 1;
  const double planckH = (double)6.62606957e-34f, lightC = (double)2.99792458e+8f, boltzK = (double)1.3806488e-23f;
  const double twoHC2 = 2.0 * planckH * std::pow(lightC, 2.0);
  const double pi = 4.0 * std::atan(1.0);
  const size_t nvox = (size_t)nx * ny * nz;
  const double lambda = lambdaMicrons / std::pow(10.0, 6.0);
  const double hck = planckH * lightC / (boltzK * lambda);
  const double lam5 = std::pow(lambda, 5.0);
  auto planck = [&](double T) { return (twoHC2 / (lam5 * (std::exp(hck / T) - 1.0))) / std::pow(10.0, 6.0); };
  const double emiss = 1.0 - albedo;
  const double wx = xe[nx] - xe[0], wy = ye[ny] - ye[0], km2 = std::pow(1000.0, 2.0);
  double sfcPower = 0.0;
  if (!(emiss == 0.0 || sfcTemp == 0.0)) sfcPower = pi * emiss * planck(sfcTemp) * wx * wy * km2;
  bool positive = true;
  for (size_t i = 0; i < nvox; ++i) positive = positive && temps[i] > 0.0;
  std::memset(voxelWeights, 0, sizeof(double) * nvox);
  if (positive) {
    double running = 0.0, comp = 0.0;  // compensated running sum :499-509
    for (size_t v = 0; v < nvox; ++v) {
      const int iz = (int)(v / ((size_t)nx * ny));
      double scat = 0.0;
      for (int j = 0; j < nc; ++j) {
        const double cj = cumExt[v + nvox * j];
        const double ej = j == 0 ? totalExt[v] * cj : totalExt[v] * (cj - cumExt[v + nvox * (j - 1)]);
        scat += ssa[v + nvox * j] * ej;
      }
      const double absorb = totalExt[v] - scat;
      const double term = (4.0 * pi * planck(temps[v]) * absorb * (ze[iz + 1] - ze[iz])) - comp;
      const double next = running + term;
      comp = (next - running) - term;
      running = next;
      voxelWeights[v] = running;
    }
  }
  double atmsPower = 0.0;
  *fracAtmsPower = 0.0;
  const double last = voxelWeights[nvox - 1];
  if (last > 0.0) {
    atmsPower = last * wx * wy * km2 / (double)(nx * ny);
    for (size_t v = 0; v < nvox; ++v) voxelWeights[v] = voxelWeights[v] / last;
    voxelWeights[nvox - 1] = 1.0;
    *fracAtmsPower = atmsPower / (atmsPower + sfcPower);
  }
  const double power = atmsPower + sfcPower;
  if (power == 0.0) return 2;  // "Neither surface nor atmosphere will emit photons"
  if (totalFlux) *totalFlux = (power / (wx * wy * km2)) * dLambda;
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Block decomposition for the block walk (mcbrat_blockwalk.hip): axis-aligned boxes of cells that carry one
// extinction value.  Greedy: the first cell without a block (x fastest) grows along x, then the row grows along y,
// then the slab along z, as long as every new cell has the same extinction and no block yet.  Any partition into
// such boxes is valid for the walk; this one finds the two slabs of the I3RC step cloud, the one slab of a
// plane-parallel medium and the clear air around clouds.  Host arithmetic only (no device needed).
//   blockOf [nx*ny*nz]   block of each cell (x fastest)
//   blockRec[4*nBlocks]  per block {x0 | x1 << 16, y0 | y1 << 16, z0 | z1 << 16, flags}: cell range [lo, hi) per axis;
//                        flag bit 0 / 1: the block spans the whole (periodic) x / y axis
// Returns 0, or 1 when the grid has more than 65535 blocks or an axis longer than 65535 cells.
// ------------------------------------------------------------------------------------------------
extern "C" int mcbrat_block_decomposition(int32_t nx, int32_t ny, int32_t nz, const float *ext, uint16_t *blockOf,
                                          uint32_t *blockRec, int32_t *nBlocks) {
  if (nx < 1 || ny < 1 || nz < 1 || nx > 65535 || ny > 65535 || nz > 65535 || !ext || !blockOf || !blockRec || !nBlocks) return 1;
  const size_t nvox = (size_t)nx * ny * nz;
  std::vector<int> of(nvox, -1);
  auto at = [&](int i, int j, int k) { return (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k); };
  size_t nb = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        if (of[at(i, j, k)] >= 0) continue;
        if (nb >= 65535) return 1;
        const float v = ext[at(i, j, k)];
        int x1 = i + 1, y1 = j + 1, z1 = k + 1;
        while (x1 < nx && of[at(x1, j, k)] < 0 && ext[at(x1, j, k)] == v) ++x1;
        for (; y1 < ny; ++y1) {
          bool ok = true;
          for (int ii = i; ii < x1 && ok; ++ii) ok = of[at(ii, y1, k)] < 0 && ext[at(ii, y1, k)] == v;
          if (!ok) break;
        }
        for (; z1 < nz; ++z1) {
          bool ok = true;
          for (int jj = j; jj < y1 && ok; ++jj)
            for (int ii = i; ii < x1 && ok; ++ii) ok = of[at(ii, jj, z1)] < 0 && ext[at(ii, jj, z1)] == v;
          if (!ok) break;
        }
        for (int kk = k; kk < z1; ++kk)
          for (int jj = j; jj < y1; ++jj)
            for (int ii = i; ii < x1; ++ii) of[at(ii, jj, kk)] = (int)nb;
        blockRec[4 * nb + 0] = (uint32_t)i | ((uint32_t)x1 << 16);
        blockRec[4 * nb + 1] = (uint32_t)j | ((uint32_t)y1 << 16);
        blockRec[4 * nb + 2] = (uint32_t)k | ((uint32_t)z1 << 16);
        blockRec[4 * nb + 3] = (i == 0 && x1 == nx ? 1u : 0u) | (j == 0 && y1 == ny ? 2u : 0u);
        ++nb;
      }
  for (size_t v = 0; v < nvox; ++v) blockOf[v] = (uint16_t)of[v];
  *nBlocks = (int32_t)nb;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Tables of the layer-skipping walk and the clear-air flight (mcbrat_kernels.hip).
//   background[nz]        the most common extinction of every layer (ties: the smallest such value)
//   range[(ny/4)*(nx/4)]  per brick column (4 x 4 columns) lo | hi << 8: the layers [lo, hi) that hold a cell whose
//                         extinction differs from its layer's background (lo = nz, hi = 0: none)
//   walk[nx*ny*nz]        the extinction, with the sign bit set in every cell outside the range of its brick column
//                         (such a cell holds its layer's background value)
//   depth[nz+1]           vertical optical depth of the background below every face, over the layers a flight can
//                         cross -- those in which some brick column lies outside its range -- and the layers of one
//                         extinction value (runs of the layer-skipping walk); without brick columns: the latter only
//   *flights              1 when brick columns exist (column counts multiples of four, 2..255 layers) and some range
//                         is not empty (a medium of one-extinction layers only is served by the runs alone)
// range and walk are only written when brick columns exist (they may be null otherwise).  Host arithmetic only.
// ------------------------------------------------------------------------------------------------
extern "C" int mcbrat_flight_tables(int32_t nx, int32_t ny, int32_t nz, const float *ext, const double *zEdges,
                                    float *background, uint16_t *range, float *walk, double *depth, int32_t *flights) {
  if (nx < 1 || ny < 1 || nz < 1 || !ext || !zEdges || !background || !depth || !flights) return 1;
  const size_t ncol = (size_t)nx * ny;
  std::vector<float> layer(ncol);
  std::vector<char> flyable(nz, 0), uniform(nz, 0);
  for (int k = 0; k < nz; ++k) {
    std::copy(ext + ncol * k, ext + ncol * (k + 1), layer.begin());
    std::sort(layer.begin(), layer.end());
    float best = layer[0];
    size_t bestRun = 0;
    for (size_t i = 0; i < ncol;) {  // mode of the layer
      size_t j = i;
      while (j < ncol && layer[j] == layer[i]) ++j;
      if (j - i > bestRun) { bestRun = j - i; best = layer[i]; }
      i = j;
    }
    background[k] = best;
    uniform[k] = bestRun == ncol ? 1 : 0;
    flyable[k] = uniform[k];
  }
  *flights = 0;
  if (nx % 4 == 0 && ny % 4 == 0 && nz >= 2 && nz <= 255) {
    if (!range || !walk) return 1;
    const int fbx = nx / 4, fby = ny / 4;
    std::copy(ext, ext + ncol * nz, walk);
    std::fill(flyable.begin(), flyable.end(), 0);
    bool anyRange = false;
    for (int by = 0; by < fby; ++by)
      for (int bx = 0; bx < fbx; ++bx) {
        int lo = nz, hi = 0;
        for (int k = 0; k < nz; ++k) {
          bool differs = false;
          for (int j = by * 4; j < by * 4 + 4 && !differs; ++j)
            for (int i = bx * 4; i < bx * 4 + 4 && !differs; ++i)
              differs = ext[(size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k)] != background[k];
          if (differs) { lo = std::min(lo, k); hi = k + 1; }
        }
        range[(size_t)bx + (size_t)fbx * by] = (uint16_t)(lo | (hi << 8));
        anyRange = anyRange || hi > lo;
        for (int k = 0; k < nz; ++k) {
          if (k >= lo && k < hi) continue;
          flyable[k] = 1;
          for (int j = by * 4; j < by * 4 + 4; ++j)
            for (int i = bx * 4; i < bx * 4 + 4; ++i) {
              float &w = walk[(size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k)];
              uint32_t u;
              std::memcpy(&u, &w, 4);
              u |= 0x80000000u;
              std::memcpy(&w, &u, 4);
            }
        }
      }
    *flights = anyRange ? 1 : 0;
  }
  // (a layer of one extinction value always counts: the runs of the layer-skipping walk take their optical depth from
  // this table, also where such a layer lies between two cloud decks, inside the range of every brick column)
  depth[0] = 0.0;
  for (int k = 0; k < nz; ++k) depth[k + 1] = depth[k] + ((flyable[k] || uniform[k]) ? (double)background[k] * (zEdges[k + 1] - zEdges[k]) : 0.0);
  return 0;
}
