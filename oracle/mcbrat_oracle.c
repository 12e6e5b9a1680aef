/*
 * mcbrat_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of MCBRaT3D's photon-tracing path.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference).
 * Arithmetic kinds follow the Fortran: `real` = float, `real(8)` = double,
 * default integer = int32.  Build with -O2 -ffp-contract=off and without
 * -ffast-math so that float/double expression order is what is written here.
 *
 * Who may use this: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline.
 *
 * Pinned (tests/test_oracle_pin.py): the findIndex family, Lobatto quadrature and Legendre
 * polynomials bit for bit against the reference's own numericUtilities.f95 compiled here (oracle/_ref, fixtures in
 * tests/golden/ref_numeric.json); MT19937 against its known answers and the reference stream recorded in SURVEY.md;
 * the whole photon loop (solar source, regular grid, Legendre phase function, roulette) against the reference
 * Fortran's own step-cloud outputs at 1e5 and 1e6 photons recorded in SURVEY.md section 8c / BASELINE.md, to all six
 * printed digits.  Parity unpinned (restated from the source text only, no reference output recorded): thermal
 * emission source and weighting, angle/value phase functions, irregular-grid launch, radiance by local estimation,
 * the surface description.
 */
#include "mcbrat_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Fortran intrinsics                                                        */
/* ------------------------------------------------------------------------ */
static double spacing_d(double x) { /* SPACING(real(8)) */
  if (x == 0.0) return DBL_MIN;
  int e;
  frexp(fabs(x), &e);
  double s = ldexp(1.0, e - 53);
  return s < DBL_MIN ? DBL_MIN : s;
}
static float spacing_f(float x) { /* SPACING(real) */
  if (x == 0.0f) return FLT_MIN;
  int e;
  frexpf(fabsf(x), &e);
  float s = ldexpf(1.0f, e - 24);
  return s < FLT_MIN ? FLT_MIN : s;
}

/* ------------------------------------------------------------------------ */
/* src/RandomNumbersForMC.f95                                                */
/* ------------------------------------------------------------------------ */
#define MT_N 624
#define MT_M 397

void orc_mt_init_scalar(orc_rng *r, int32_t seed) { /* :171-187 */
  r->mode = 0;
  r->mt[0] = (uint32_t)seed;
  for (int i = 1; i < MT_N; i++)
    r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->mti = MT_N;
  r->ndraws = 0;
}

void orc_mt_init_vector(orc_rng *r, const int32_t *seed, int n) { /* :189-241 */
  orc_mt_init_scalar(r, 19650218);
  int nWraps = 0;
  int nFirstLoop = MT_N > n ? MT_N : n;
  uint32_t *s = r->mt;
  for (int k = 1; k <= nFirstLoop; k++) {
    int i = (k + nWraps) % MT_N;
    int j = (k - 1) % n;
    if (i == 0) { /* :202-209: wrap, then the update lands on element 1 */
      s[0] = s[MT_N - 1];
      s[1] = (s[1] ^ ((s[0] ^ (s[0] >> 30)) * 1664525u)) + (uint32_t)seed[j] + (uint32_t)j;
      nWraps++;
    } else {
      s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1664525u)) + (uint32_t)seed[j] + (uint32_t)j;
    }
  }
  for (int i = nFirstLoop % MT_N + nWraps + 1; i <= MT_N - 1; i++) /* :222-227 */
    s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
  s[0] = s[MT_N - 1];
  for (int i = 1; i <= nFirstLoop % MT_N + nWraps; i++) /* :231-236 */
    s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
  s[0] = 0x80000000u; /* :238 UMASK */
  r->mti = MT_N;
}

static void mt_next_state(orc_rng *r) { /* :118-154 */
  uint32_t *s = r->mt;
  int k;
#define MT_TWIST(u, v) (((((u) & 0x80000000u) | ((v) & 0x7fffffffu)) >> 1) ^ (((v) & 1u) ? 0x9908b0dfu : 0u))
  for (k = 0; k < MT_N - MT_M; k++) s[k] = s[k + MT_M] ^ MT_TWIST(s[k], s[k + 1]);
  for (; k < MT_N - 1; k++) s[k] = s[k + MT_M - MT_N] ^ MT_TWIST(s[k], s[k + 1]);
  s[MT_N - 1] = s[MT_M - 1] ^ MT_TWIST(s[MT_N - 1], s[0]);
  r->mti = 0;
}

uint32_t orc_mt_next_u32(orc_rng *r) { /* getRandomInt :245-260, temper :156-167 */
  if (r->mti >= MT_N) mt_next_state(r);
  uint32_t y = r->mt[r->mti++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* Philox4x32-10 (Salmon et al., SC'11): the counter-based generator the
 * north_star substitutes for the stateful MT stream. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int i = 0; i < 10; i++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox_init(orc_rng *r, uint64_t seed, uint64_t firstPhoton) {
  memset(r, 0, sizeof(*r));
  r->mode = 1;
  r->seed = seed;
  r->firstPhoton = firstPhoton;
}

static void philox_start_photon(orc_rng *r, uint64_t photon) {
  r->photon = photon;
  r->event = 0;
  r->cachedBlock = 0xffffffffu;
}
static void philox_next_event(orc_rng *r) {
  r->event++;
  r->cachedBlock = 0xffffffffu;
}

/* getRandomDouble / getRandomReal :277-301: u32/(2^32-1) in double, then to
 * float; the closed interval [0,1] (both ends attainable).  The Philox mode
 * rounds u to float and scales by 2^-32 (what the HIP kernel does; within one
 * float ulp of the reference's map, also on [0,1] with both ends attainable). */
float orc_random_real(orc_rng *r) { /* sequential (MT) stream */
  r->ndraws++;
  return (float)((double)orc_mt_next_u32(r) / 4294967295.0);
}

/* One uniform for a given role.  MT mode: the next number of the stream (block
 * and elem are ignored -- call order IS the reference's draw order).  Philox
 * mode: element `elem` of block `block` of the current event of this photon. */
static float draw(orc_rng *r, uint32_t block, uint32_t elem) {
  if (r->mode == 0) return orc_random_real(r);
  r->ndraws++;
  if (r->cachedBlock != block) {
    uint32_t ctr[4] = {r->event, block, (uint32_t)r->photon, (uint32_t)(r->photon >> 32)};
    uint32_t key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)};
    orc_philox4x32_10(ctr, key, r->buf);
    r->cachedBlock = block;
  }
  return (float)r->buf[elem] * 2.3283064365386963e-10f; /* float(u) * 2^-32: what the HIP kernel does */
}

/* ------------------------------------------------------------------------ */
/* src/numericUtilities.f95                                                  */
/* ------------------------------------------------------------------------ */
void orc_legendre(int maxL, int nmu, const float *mus, float *out) { /* :187-205 */
  int ld = maxL + 1;
  for (int j = 0; j < nmu; j++) {
    float *P = out + (size_t)j * ld;
    P[0] = 1.0f;
    if (maxL >= 1) P[1] = mus[j];
    for (int l = 1; l <= maxL - 1; l++)
      P[l + 1] = (((float)(2 * l + 1) * mus[j]) * P[l] - (float)l * P[l - 1]) / (float)(l + 1);
  }
}

void orc_lobatto(int n, float *mus, float *weights) { /* :27-114 */
  const float relativeAccuracy = 3.0f;
  const int maxIterations = 25;
  float pi = acosf(-1.0f);
  int nTerms = n;
  int midPoint = (nTerms + 1) / 2;
  int m = midPoint - 1;
  float *trial = (float *)calloc(m > 0 ? m : 1, sizeof(float));
  float *last = (float *)calloc(m > 0 ? m : 1, sizeof(float));
  float *der = (float *)calloc(m > 0 ? m : 1, sizeof(float));
  float *sec = (float *)calloc(m > 0 ? m : 1, sizeof(float));
  int ld = nTerms; /* legendreP(0:nTerms-1, m) */
  float *P = (float *)calloc((size_t)ld * (m > 0 ? m : 1), sizeof(float));
  float c1 = (nTerms % 2 == 1) ? 1.0f : 0.5f;
  float denom = ((float)nTerms - 1.0f) + 0.5f;
  float nm1 = (float)(nTerms - 1);
  float nn1 = (float)(nTerms * (nTerms - 1));
  for (int i = 1; i <= m; i++) trial[i - 1] = sinf((pi * ((float)i - c1)) / denom);

  orc_legendre(nTerms - 1, m, trial, P);
  for (int j = 0; j < m; j++) { /* first Newton step :59-69 */
    const float *Pj = P + (size_t)j * ld;
    der[j] = (nm1 * (trial[j] * Pj[nTerms - 1] - Pj[nTerms - 2])) / (trial[j] * trial[j] - 1.0f);
    sec[j] = ((2.0f * trial[j]) * der[j] - (nn1 * Pj[nTerms - 1])) / (1.0f - trial[j] * trial[j]);
    last[j] = trial[j];
    trial[j] = trial[j] - der[j] / sec[j];
  }
  int it = 0;
  for (;;) { /* :73-95 */
    int allDone = 1;
    for (int j = 0; j < m; j++)
      if (!(fabsf(trial[j] - last[j]) <= relativeAccuracy * spacing_f(trial[j]))) allDone = 0;
    if (allDone) break;
    orc_legendre(nTerms - 1, m, trial, P);
    for (int j = 0; j < m; j++) {
      if (fabsf(trial[j] - last[j]) > relativeAccuracy * spacing_f(trial[j])) {
        const float *Pj = P + (size_t)j * ld;
        der[j] = (nm1 * (trial[j] * Pj[nTerms - 1] - Pj[nTerms - 2])) / (trial[j] * trial[j] - 1.0f);
        sec[j] = ((2.0f * trial[j]) * der[j] - (nn1 * Pj[nTerms - 1])) / (1.0f - trial[j] * trial[j]);
        last[j] = trial[j];
        trial[j] = trial[j] - der[j] / sec[j];
      }
    }
    it++;
    if (it > maxIterations) break;
  }
  /* :98-111, 1-based m(i) = mus[i-1] */
  mus[0] = -1.0f;
  weights[0] = 2.0f / nn1;
  for (int j = 0; j < m; j++) { /* mus(midPoint:2:-1) = -trialMus(:) */
    const float *Pj = P + (size_t)j * ld;
    mus[midPoint - 1 - j] = -trial[j];
    weights[midPoint - 1 - j] = 2.0f / (nn1 * (Pj[nTerms - 1] * Pj[nTerms - 1]));
  }
  if (nTerms % 2 == 0) {
    for (int k = 0; k < midPoint; k++) { /* mus(mid+1:n) = -mus(mid:1:-1) */
      mus[midPoint + k] = -mus[midPoint - 1 - k];
      weights[midPoint + k] = weights[midPoint - 1 - k];
    }
  } else {
    float *tm = (float *)malloc(sizeof(float) * midPoint), *tw = (float *)malloc(sizeof(float) * midPoint);
    for (int k = 0; k < midPoint; k++) { tm[k] = -mus[midPoint - 1 - k]; tw[k] = weights[midPoint - 1 - k]; }
    for (int k = 0; k < midPoint; k++) { mus[midPoint - 1 + k] = tm[k]; weights[midPoint - 1 + k] = tw[k]; }
    free(tm); free(tw);
  }
  free(trial); free(last); free(der); free(sec); free(P);
}

/* findIndex family.  Tables are 1-based in the reference: t(i) == t[i-1]. */
#define FIND_INDEX_BODY(T1)                                                         \
  int lowerBound, upperBound, midPoint, increment;                                  \
  if (firstGuess > 0) {                                                             \
    lowerBound = firstGuess; increment = 1;                                         \
    for (;;) {                                                                      \
      upperBound = lowerBound + increment < n ? lowerBound + increment : n;         \
      if (lowerBound == n || (T1(lowerBound) <= v && T1(upperBound) > v)) break;    \
      if (T1(lowerBound) > v) {                                                     \
        upperBound = lowerBound;                                                    \
        lowerBound = upperBound - increment > 1 ? upperBound - increment : 1;       \
      } else lowerBound = upperBound;                                               \
      increment *= 2;                                                               \
    }                                                                               \
  } else { lowerBound = 0; upperBound = n; }                                        \
  for (;;) {                                                                        \
    if (lowerBound == n || upperBound <= lowerBound + 1) break;                     \
    midPoint = (lowerBound + upperBound) / 2;                                       \
    if (v >= T1(midPoint)) lowerBound = midPoint; else upperBound = midPoint;       \
  }                                                                                 \
  return lowerBound;

int orc_find_index_real(float v, const float *t, int n, int firstGuess) { /* :417-470 */
#define T1(i) (t[(i)-1])
  FIND_INDEX_BODY(T1)
#undef T1
}
int orc_find_index_double(double v, const double *t, int n, int firstGuess) { /* :207-260 */
#define T1(i) (t[(i)-1])
  FIND_INDEX_BODY(T1)
#undef T1
}
int orc_find_index_mixed(float vf, const double *t, int n, int firstGuess) { /* :262-315 */
  double v = (double)vf;
#define T1(i) (t[(i)-1])
  FIND_INDEX_BODY(T1)
#undef T1
}

/* makePeriodic, src/surfaceProperties.f95:211-230 (result kind is default real: the position is rounded to float) */
static float make_periodic(double a, double aMin, double aMax) {
  float r = (float)a;
  for (int guard = 0; guard < 65536; ++guard) { /* (unbounded in the reference: endless where the period is below the float spacing of r) */
    if ((double)r <= aMax && (double)r > aMin) break;
    if ((double)r > aMax) r = (float)((double)r - (aMax - aMin));
    else if ((double)r == aMin) r = (float)aMax;
    else r = (float)((double)r + (aMax - aMin));
  }
  return r;
}

/* computeSurfaceReflectance, src/surfaceProperties.f95:119-147, with R = BRDFParameters(1) (:153-161).  A position
 * exactly on the lower edge is sent to the upper one (:224-225), where findIndex answers size(xPosition): one past
 * the last patch in the reference; the last patch here. */
static float surface_reflectance(const orc_problem *P, double xPos, double yPos) {
  const int nX = P->surfNumX, nY = P->surfNumY;
  int xi = orc_find_index_mixed(make_periodic(xPos, P->surfXPosition[0], P->surfXPosition[nX - 1]), P->surfXPosition, nX, 0);
  int yi = orc_find_index_mixed(make_periodic(yPos, P->surfYPosition[0], P->surfYPosition[nY - 1]), P->surfYPosition, nY, 0);
  if (xi < 1) xi = 1;
  if (xi > nX - 1) xi = nX - 1;
  if (yi < 1) yi = 1;
  if (yi > nY - 1) yi = nY - 1;
  return P->surfReflectance[(size_t)(xi - 1) + (size_t)(nX - 1) * (yi - 1)];
}
/* getFrequencyDistrNEW, src/emissionAndBroadBandWeights.f95:552-572: one uniform per photon, findCDFIndex against the
 * power CDF.  MT mode: the uniforms are the next numbers of the stream (the reference).  Philox mode: draw d is
 * element d % 4 of block (0xFFFFFFFF, 0, d / 4) under the key, d counted from firstDraw (what the HIP kernel does). */
void orc_frequency_distribution(orc_rng *r, uint64_t firstDraw, int numLambda, const double *cdf, int64_t totalPhotons,
                                int64_t *distribution) {
  for (int i = 0; i < numLambda; i++) distribution[i] = 0;
  uint32_t buf[4];
  uint64_t cached = ~(uint64_t)0;
  for (int64_t n = 0; n < totalPhotons; n++) {
    float RN;
    if (r->mode == 0) RN = orc_random_real(r);
    else {
      const uint64_t d = firstDraw + (uint64_t)n, blk = d / 4;
      if (blk != cached) {
        uint32_t ctr[4] = {0xFFFFFFFFu, 0u, (uint32_t)blk, (uint32_t)(blk >> 32)};
        uint32_t key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)};
        orc_philox4x32_10(ctr, key, buf);
        cached = blk;
      }
      RN = (float)buf[d % 4] * 2.3283064365386963e-10f;
    }
    int i = orc_find_cdf_index(RN, cdf, numLambda);
    if (i > numLambda) i = numLambda;
    distribution[i - 1]++;
  }
}

/* exported for the pin against the reference's own computeSurfaceReflectance (tests/golden/ref_surface.json) */
float orc_surface_reflectance(const orc_problem *P, double xPos, double yPos) { return surface_reflectance(P, xPos, yPos); }
int orc_find_cdf_index(float vf, const double *t, int n) { /* :317-348 */
  double v = (double)vf;
  int lowerBound = 0, upperBound = n, midPoint;
  for (;;) {
    if (lowerBound == n || upperBound <= lowerBound + 1) break;
    midPoint = (lowerBound + upperBound) / 2;
    if (v > t[midPoint - 1]) lowerBound = midPoint; else upperBound = midPoint;
  }
  return upperBound;
}
/* strided variant for colWeights / voxelWeights slices of the running CDF */
static int find_cdf_index_strided(float vf, const double *t, int n, int64_t stride) {
  double v = (double)vf;
  int lowerBound = 0, upperBound = n, midPoint;
  for (;;) {
    if (lowerBound == n || upperBound <= lowerBound + 1) break;
    midPoint = (lowerBound + upperBound) / 2;
    if (v > t[(int64_t)(midPoint - 1) * stride]) lowerBound = midPoint; else upperBound = midPoint;
  }
  return upperBound;
}

/* ------------------------------------------------------------------------ */
/* src/scatteringPhaseFunctions.f95                                          */
/* ------------------------------------------------------------------------ */
void orc_phase_values_legendre(int ncoef, const float *coef, int nang, const float *angles,
                               float *values) { /* getPhaseFunctionValues_one :480-498 */
  int maxL = ncoef;
  if (maxL == 0) { for (int j = 0; j < nang; j++) values[j] = 0.5f; return; }
  float *cosA = (float *)malloc(sizeof(float) * nang);
  float *P = (float *)malloc(sizeof(float) * (size_t)(maxL + 1) * nang);
  float *v = (float *)malloc(sizeof(float) * (maxL + 1));
  for (int j = 0; j < nang; j++) cosA[j] = cosf(angles[j]);
  orc_legendre(maxL, nang, cosA, P);
  for (int l = 0; l <= maxL; l++) v[l] = (l == 0 ? 1.0f : coef[l - 1]) * (float)(2 * l + 1);
  for (int j = 0; j < nang; j++) { /* matmul(vector, matrix): sum over l ascending */
    float s = 0.0f;
    const float *Pj = P + (size_t)j * (maxL + 1);
    for (int l = 0; l <= maxL; l++) s += v[l] * Pj[l];
    values[j] = s;
  }
  free(cosA); free(P); free(v);
}

void orc_normalize_phase_function(int n, const float *angles, const float *vin, float *vout) { /* :1520-1536 */
  float dot = 0.0f;
  for (int i = 0; i < n - 1; i++)
    dot += (cosf(angles[i + 1]) - cosf(angles[i])) * (0.5f * (vin[i + 1] + vin[i]));
  for (int i = 0; i < n; i++) vout[i] = (-vin[i] * 2.0f) / dot;
}

void orc_phase_values_tabulated(int nst, const float *stA, const float *stV, int nang,
                                const float *angles, float *values) { /* :500-527 */
  for (int l = 0; l < nang; l++) {
    int ti = orc_find_index_real(angles[l], stA, nst, 0);
    int tp = ti + 1;
    float dMu;
    if (ti < nst) dMu = cosf(stA[tp - 1]) - cosf(stA[ti - 1]);
    else { dMu = FLT_MAX; tp = ti; }
    float w = 1.0f - (cosf(angles[l]) - cosf(stA[ti - 1])) / dMu;
    values[l] = w * stV[ti - 1] + (1.0f - w) * stV[tp - 1];
  }
}

/* ------------------------------------------------------------------------ */
/* src/inversePhaseFunctions.f95:66-174                                      */
/* ------------------------------------------------------------------------ */
static int inverse_from_cdf_inputs(int nAngles, const float *mus, const float *values, int nSteps,
                                   float *table) { /* :114-169 */
  float *cdf = (float *)malloc(sizeof(float) * nAngles);
  int *ind = (int *)malloc(sizeof(int) * nSteps);
  cdf[0] = 0.0f;
  for (int i = 1; i < nAngles; i++)
    cdf[i] = cdf[i - 1] + ((mus[i] - mus[i - 1]) * 0.5f) * (values[i] + values[i - 1]);
  float tot = cdf[nAngles - 1];
  for (int i = 0; i < nAngles; i++) cdf[i] = cdf[i] / tot;
  ind[0] = orc_find_index_real(0.0f, cdf, nAngles, 0);
  for (int i = 2; i <= nSteps; i++) {
    float p = (float)(i - 1) / (float)(nSteps - 1);
    ind[i - 1] = orc_find_index_real(p, cdf, nAngles, ind[i - 2]);
  }
  for (int i = 1; i <= nSteps - 1; i++) {
    float p = (float)(i - 1) / (float)(nSteps - 1);
    int k = ind[i - 1]; /* 1-based */
    if (k >= nAngles) k = nAngles - 1; /* guard: the reference would index past the table */
    float c0 = cdf[k - 1], c1 = cdf[k], m0 = mus[k - 1], m1 = mus[k], v0 = values[k - 1], v1 = values[k];
    if (c1 - c0 <= spacing_f(c0)) {
      table[i - 1] = acosf(m0);
    } else if (fabsf(v0 - v1) <= spacing_f(v0)) {
      table[i - 1] = acosf(m0 + ((m1 - m0) * (p - c0)) / (c1 - c0));
    } else {
      float root = sqrtf(((c1 - p) * (v0 * v0) + (p - c0) * (v1 * v1)) / (c1 - c0));
      table[i - 1] = acosf(m0 + ((m1 - m0) / (v0 - v1)) * (v0 - root));
    }
  }
  table[nSteps - 1] = 0.0f;
  free(cdf); free(ind);
  return 0;
}

int orc_inverse_table_legendre(int ncoef, const float *coef, int nSteps, float *table) {
  int nAngles = ncoef > 2 ? ncoef : 2; /* :107 */
  float *mus = (float *)malloc(sizeof(float) * nAngles);
  float *wts = (float *)malloc(sizeof(float) * nAngles);
  float *ang = (float *)malloc(sizeof(float) * nAngles);
  float *val = (float *)malloc(sizeof(float) * nAngles);
  float *valr = (float *)malloc(sizeof(float) * nAngles);
  orc_lobatto(nAngles, mus, wts); /* :110 */
  for (int i = 0; i < nAngles; i++) ang[i] = acosf(mus[nAngles - 1 - i]); /* :111 */
  orc_phase_values_legendre(ncoef, coef, nAngles, ang, val);
  for (int i = 0; i < nAngles; i++) valr[i] = val[nAngles - 1 - i]; /* :112 */
  int rc = inverse_from_cdf_inputs(nAngles, mus, valr, nSteps, table);
  free(mus); free(wts); free(ang); free(val); free(valr);
  return rc;
}

int orc_inverse_table_tabulated(int nAngles, const float *angles, const float *values, int nSteps,
                                float *table) { /* :88-98; values are the (already normalised) stored ones */
  float *mus = (float *)malloc(sizeof(float) * nAngles);
  float *val = (float *)malloc(sizeof(float) * nAngles);
  float *valr = (float *)malloc(sizeof(float) * nAngles);
  orc_phase_values_tabulated(nAngles, angles, values, nAngles, angles, val); /* :97 */
  for (int i = 0; i < nAngles; i++) { mus[i] = cosf(angles[nAngles - 1 - i]); valr[i] = val[nAngles - 1 - i]; }
  int rc = inverse_from_cdf_inputs(nAngles, mus, valr, nSteps, table);
  free(mus); free(val); free(valr);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* src/opticalProperties.f95:1656-1815 accumulateExtinctionAlongPath         */
/* ------------------------------------------------------------------------ */
#define IDX3(P, ix, iy, iz) ((size_t)((ix)-1) + (size_t)(P)->nx * ((size_t)((iy)-1) + (size_t)(P)->ny * (size_t)((iz)-1)))
#define IDX4(P, ix, iy, iz, ic) (IDX3(P, ix, iy, iz) + (size_t)(P)->nx * (P)->ny * (P)->nz * (size_t)((ic)-1))

float orc_accumulate_extinction(const orc_problem *P, const float dir[3], double pos[3], int32_t idx[3],
                                int hasTarget, float extToAccumulate, int64_t *crossings) {
  const double *edge[3] = {P->xe, P->ye, P->ze};
  const int ncell[3] = {P->nx, P->ny, P->nz};
  float extAccumulated = 0.0f;
  int side[3], inc[3];
  for (int a = 0; a < 3; a++) { /* :1690-1692 */
    side[a] = dir[a] >= 0.0f ? 1 : 0;
    inc[a] = dir[a] >= 0.0f ? 1 : -1;
  }
  double z0 = P->ze[0], zMax = P->ze[P->nz];
  for (;;) {
    double step[3];
    for (int a = 0; a < 3; a++) { /* :1705-1712 */
      if (fabsf(dir[a]) >= 2.0f * FLT_MIN)
        step[a] = (edge[a][idx[a] + side[a] - 1] - pos[a]) / (double)dir[a];
      else
        step[a] = DBL_MAX;
    }
    double thisStep = step[0];
    if (step[1] < thisStep) thisStep = step[1];
    if (step[2] < thisStep) thisStep = step[2];
    if (thisStep <= 0.0) { extAccumulated = -2.0f; break; } /* :1719-1722 */
    double thisCellExt = P->totalExt[IDX3(P, idx[0], idx[1], idx[2])];
    if (hasTarget) { /* :1729-1739 */
      if ((double)extAccumulated + thisStep * thisCellExt > (double)extToAccumulate) {
        thisStep = (double)(extToAccumulate - extAccumulated) / thisCellExt;
        pos[0] = pos[0] + thisStep * (double)dir[0];
        pos[1] = pos[1] + thisStep * (double)dir[1];
        pos[2] = pos[2] + thisStep * (double)dir[2];
        extAccumulated = extToAccumulate;
        break;
      }
    }
    extAccumulated = (float)((double)extAccumulated + thisStep * thisCellExt); /* :1743 */
    if (crossings) (*crossings)++;
    for (int a = 0; a < 3; a++) { /* :1752-1777 */
      if (step[a] <= thisStep) {
        pos[a] = edge[a][idx[a] + side[a] - 1];
        idx[a] += inc[a];
      } else {
        pos[a] = pos[a] + thisStep * (double)dir[a];
        if (fabs(edge[a][idx[a] + side[a] - 1] - pos[a]) <= 2.0 * spacing_d(pos[a])) idx[a] += inc[a];
      }
    }
    /* periodicity :1782-1796 (y uses cellIncrement(1), sic) */
    for (int a = 0; a < 2; a++) {
      if (idx[a] <= 0) {
        idx[a] = ncell[a];
        pos[a] = edge[a][idx[a]] + (double)(inc[0] * 2) * spacing_d(pos[a]);
      } else if (idx[a] >= ncell[a] + 1) {
        idx[a] = 1;
        pos[a] = edge[a][0] + (double)(inc[0] * 2) * spacing_d(pos[a]);
      }
    }
    if (idx[2] > P->nz) { pos[2] = zMax + 2.0 * spacing_d(zMax); break; } /* :1801-1804 */
    if (idx[2] < 1) { pos[2] = z0; break; }                               /* :1809-1812 */
  }
  return extAccumulated;
}

/* ------------------------------------------------------------------------ */
/* src/emissionAndBroadBandWeights.f95:424-550 emission_weightingNEW          */
/* ------------------------------------------------------------------------ */
int orc_emission_weighting(int nx, int ny, int nz, int nc, const double *xe, const double *ye,
                           const double *ze, const double *temps, const double *totalExt,
                           const double *cumExt, const double *ssa, double albedo, double lambda_um,
                           double sfcTemp, double dLambda, double *voxelWeights, double *fracAtmsPower,
                           double *totalFlux) {
  const double h = (double)6.62606957e-34f; /* default-real literals, as written (:448-450) */
  const double c = (double)2.99792458e+8f;
  const double k = (double)1.3806488e-23f;
  const double a = 2.0 * h * pow(c, 2.0);
  const double Pi = 4.0 * atan(1.0);
  size_t nvox = (size_t)nx * ny * nz;
  double emiss = 1.0 - albedo;
  double lambda = lambda_um / pow(10.0, 6.0);
  double b = h * c / (k * lambda);
  double area = (xe[nx] - xe[0]) * (ye[ny] - ye[0]) * pow(1000.0, 2.0);
  double sfcPower;
  if (emiss == 0.0 || sfcTemp == 0.0) sfcPower = 0.0;
  else {
    double sfcPlanckRad = (a / (pow(lambda, 5.0) * (exp(b / sfcTemp) - 1.0))) / pow(10.0, 6.0);
    sfcPower = Pi * emiss * sfcPlanckRad * (xe[nx] - xe[0]) * (ye[ny] - ye[0]) * pow(1000.0, 2.0);
  }
  double atmsPower = 0.0, previous = 0.0, corr = 0.0;
  int allPositive = 1;
  for (size_t i = 0; i < nvox; i++) if (temps[i] <= 0.0) allPositive = 0;
  memset(voxelWeights, 0, sizeof(double) * nvox);
  if (allPositive) {
    for (int iz = 0; iz < nz; iz++) {
      double dz = ze[iz + 1] - ze[iz];
      for (int iy = 0; iy < ny; iy++)
        for (int ix = 0; ix < nx; ix++) {
          size_t v = (size_t)ix + (size_t)nx * ((size_t)iy + (size_t)ny * iz);
          double planck = (a / (pow(lambda, 5.0) * (exp(b / temps[v]) - 1.0))) / pow(10.0, 6.0);
          /* ext(:,:,:,j) = totalExt*(cumExt(j)-cumExt(j-1)) (opticalProperties.f95 getInfo_Domain) */
          double s = 0.0;
          for (int j = 0; j < nc; j++) {
            double cj = cumExt[v + nvox * j], cjm = j > 0 ? cumExt[v + nvox * (j - 1)] : 0.0;
            double extj = j == 0 ? totalExt[v] * cj : totalExt[v] * (cj - cjm);
            s += ssa[v + nvox * j] * extj;
          }
          double totalAbsCoef = totalExt[v] - s;
          double corr_contrib = (4.0 * Pi * planck * totalAbsCoef * dz) - corr; /* Kahan :503-507 */
          double temp_sum = previous + corr_contrib;
          corr = (temp_sum - previous) - corr_contrib;
          previous = temp_sum;
          voxelWeights[v] = previous;
        }
    }
  }
  *fracAtmsPower = 0.0;
  double last = voxelWeights[nvox - 1];
  if (last > 0.0) { /* :513-520 */
    atmsPower = last * (xe[nx] - xe[0]) * (ye[ny] - ye[0]) * pow(1000.0, 2.0) / (double)(nx * ny);
    for (size_t i = 0; i < nvox; i++) voxelWeights[i] = voxelWeights[i] / last;
    voxelWeights[nvox - 1] = 1.0;
    *fracAtmsPower = atmsPower / (atmsPower + sfcPower);
  }
  double totalPower = atmsPower + sfcPower;
  if (totalPower == 0.0) return 1;
  if (totalFlux) *totalFlux = (totalPower / area) * dLambda; /* :536-543 */
  return 0;
}

/* ------------------------------------------------------------------------ */
/* src/monteCarloIllumination.f95: one photon of a stream                    */
/* ------------------------------------------------------------------------ */
typedef struct { double x, y, z; float mu, phi; } launch_t;

static void launch_directional(const orc_source *S, orc_rng *R, launch_t *L) { /* :88-96 */
  L->x = (double)draw(R, 0, 0);
  L->y = (double)draw(R, 0, 1);
  L->z = (double)(1.0f - spacing_f(1.0f));
  L->mu = -fabsf(S->solarMu);
  L->phi = (S->solarAzimuthDeg * acosf(-1.0f)) / 180.0f;
}

static void launch_bbemission(const orc_problem *P, const orc_source *S, orc_rng *R, launch_t *L) { /* :481-516 */
  int nx = P->nx, ny = P->ny, nz = P->nz;
  /* Philox slots (event 0): block 0 = [select, r1, r2, r3];
   * surface: x = r1, y = r2, mu attempts from block 1.., phi = r3;
   * atmosphere: cdf = r1, z = r2, x = r3, block 1 = [y, phi, mu0, mu1], more mu from block 2.. */
  float RN = draw(R, 0, 0);
  if ((double)RN > S->fracAtmsPower) { /* surface :484-493 */
    L->x = (double)draw(R, 0, 1);
    L->y = (double)draw(R, 0, 2);
    for (uint32_t j = 0;; j++) {
      L->mu = sqrtf(draw(R, 1 + j / 4, j % 4));
      if (fabsf(L->mu) > 2.0f * FLT_MIN) break;
    }
    L->phi = (draw(R, 0, 3) * 2.0f) * acosf(-1.0f);
    L->z = 0.0;
  } else { /* atmosphere :495-510 */
    RN = draw(R, 0, 1);
    const double *vw = S->voxelWeights;
    /* levelWeights(k)=vW(nx,ny,k); colWeights(j,k)=vW(nx,j,k) (emissionAndBroadBandWeights.f95:56-57) */
    int ik = find_cdf_index_strided(RN, vw + ((size_t)nx - 1) + (size_t)nx * ((size_t)ny - 1), nz, (int64_t)nx * ny);
    int ij = find_cdf_index_strided(RN, vw + ((size_t)nx - 1) + (size_t)nx * ny * ((size_t)ik - 1), ny, nx);
    int ii = find_cdf_index_strided(RN, vw + (size_t)nx * (((size_t)ij - 1) + (size_t)ny * ((size_t)ik - 1)), nx, 1);
    L->z = ((double)(ik - 1) * 1.0 / (double)nz) + (double)(draw(R, 0, 2) / (float)nz);
    if (ik == 1 && L->z == 0.0) L->z = 0.0 + spacing_d(1.0);
    if (ik == nz && L->z > 1.0 - 2.0 * spacing_d(1.0)) L->z = L->z - (2.0 * spacing_d(1.0));
    L->x = ((double)(ii - 1) * 1.0 / (double)nx) + (double)(draw(R, 0, 3) * (1.0f / (float)nx));
    L->y = ((double)(ij - 1) * 1.0 / (double)ny) + (double)(draw(R, 1, 0) * (1.0f / (float)ny));
    for (uint32_t j = 0;; j++) {
      L->mu = 1.0f - (2.0f * (j < 2 ? draw(R, 1, 2 + j) : draw(R, 2 + (j - 2) / 4, (j - 2) % 4)));
      if (fabsf(L->mu) > 2.0f * FLT_MIN) break;
    }
    L->phi = (draw(R, 1, 1) * 2.0f) * acosf(-1.0f);
  }
}

static void launch_one(const orc_problem *P, const orc_source *S, orc_rng *R, launch_t *L) {
  if (S->kind == 0) launch_directional(S, R, L);
  else launch_bbemission(P, S, R, L);
}

/* ------------------------------------------------------------------------ */
/* Integrators/monteCarloRadiativeTransfer.f95 helpers                       */
/* ------------------------------------------------------------------------ */
void orc_grid_flags(const orc_problem *P, int *xyRegular, int *zRegular, double *deltaX, double *deltaY,
                    double *deltaZ) { /* new_Integrator :163-181; note the float locals at :140 */
  float dX = (float)(P->xe[1] - P->xe[0]);
  float dY = (float)(P->ye[1] - P->ye[0]);
  float dZ = (float)(P->ze[1] - P->ze[0]);
  int xr = 1, yr = 1, zr = 1;
  for (int i = 0; i < P->nx; i++)
    if (!(fabs((P->xe[i + 1] - P->xe[i]) - (double)dX) <= 2.0 * spacing_d(P->xe[i + 1]))) xr = 0;
  for (int i = 0; i < P->ny; i++)
    if (!(fabs((P->ye[i + 1] - P->ye[i]) - (double)dY) <= 2.0 * spacing_d(P->ye[i + 1]))) yr = 0;
  for (int i = 0; i < P->nz; i++)
    if (!(fabs((P->ze[i + 1] - P->ze[i]) - (double)dZ) <= spacing_d(P->ze[i + 1]))) zr = 0;
  *xyRegular = xr && yr;
  *zRegular = zr;
  *deltaX = *xyRegular ? (double)dX : 0.0;
  *deltaY = *xyRegular ? (double)dY : 0.0;
  *deltaZ = zr ? (double)dZ : 0.0;
}

static void make_direction_cosines(float mu, float phi, float d[3]) { /* :1876-1894 */
  float sinTheta = sqrtf(1.0f - mu * mu);
  float cosPhi = cosf(phi);
  float sinPhi = sinf(phi);
  d[0] = sinTheta * cosPhi;
  d[1] = sinTheta * sinPhi;
  d[2] = mu;
}

static float compute_scattering_angle(float u, const float *tbl, int n) { /* :1594-1621 */
  int angleIndex = (int)(u * (float)n) + 1;
  if (angleIndex < n) {
    float leftOver = u - (float)(angleIndex - 1) / (float)n;
    return (1.0f - leftOver) * tbl[angleIndex - 1] + leftOver * tbl[angleIndex];
  }
  return tbl[n - 1];
}

/* (cos, sin) of 2 pi u, u in [0,1]: the expression the HIP kernel evaluates (sincos_2pi), term for term */
static void sincos_2pi(float u, float *c, float *s) {
  const float q = rintf(4.0f * u);
  const float r = (u - 0.25f * q) * 6.28318548202514648f;
  const float z = r * r;
  const float cc = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f),
                                        4.16666233237390631894e-2f), -0.499999997251031003120f), 1.0f);
  const float ss = fmaf(r * z, fmaf(z, fmaf(z, fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f),
                                            8.3333293858894631756e-3f), -0.166666666416265235595f), r);
  const int qi = (int)q & 3;
  *c = qi == 0 ? cc : (qi == 1 ? -ss : (qi == 2 ? -cc : ss));
  *s = qi == 0 ? ss : (qi == 1 ? cc : (qi == 2 ? -ss : -cc));
}

static void next_direct(orc_rng *R, float scatteringCosine, float S[3]) { /* :1921-1948 */
  float D = 2.0f, AX = 0.f, AY = 0.f, B;
  if (R->mode == 1) {
    /* Philox mode (what the HIP kernel does): the uniformly distributed unit vector (AX, AY)/sqrt(D) of the
     * rejection loop below is (cos, sin) of a uniform azimuth, taken from ONE number: slot Y of the leg's block */
    sincos_2pi(draw(R, 0, 2), &AX, &AY);
    D = 1.0f;
  } else {
    while (D > 1.0f) { /* MT mode: the reference's loop and draw order */
      AX = 1.0f - 2.0f * draw(R, 0, 0);
      AY = 1.0f - 2.0f * draw(R, 0, 0);
      D = AX * AX + AY * AY;
    }
  }
  B = sqrtf((1.0f - scatteringCosine * scatteringCosine) / D);
  AX = AX * B;
  AY = AY * B;
  B = S[0] * AX - S[1] * AY;
  D = scatteringCosine - B / (1.0f + fabsf(S[2]));
  S[0] = S[0] * D + AX;
  S[1] = S[1] * D - AY;
  S[2] = S[2] * scatteringCosine - copysignf(fabsf(B), S[2] * B);
}

typedef struct { int xyRegular, zRegular; double dX, dY, dZ; } grid_flags;

static void find_xy_indices(const orc_problem *P, const grid_flags *G, double xPos, double yPos,
                            int *xIndex, int *yIndex) { /* :1551-1578 */
  int nx = P->nx, ny = P->ny;
  if (G->xyRegular) {
    int xi = (int)((xPos - P->xe[0]) / G->dX) + 1; if (xi > nx) xi = nx;
    int yi = (int)((yPos - P->ye[0]) / G->dY) + 1; if (yi > ny) yi = ny;
    if (fabs(P->xe[xi] - xPos) < spacing_d(xPos)) xi++;
    if (fabs(P->ye[yi] - yPos) < spacing_d(yPos)) yi++;
    if (xi == nx + 1) xi = 1;
    if (yi == ny + 1) yi = 1;
    *xIndex = xi; *yIndex = yi;
  } else {
    int xi = orc_find_index_double(xPos, P->xe, nx + 1, *xIndex);
    int yi = orc_find_index_double(yPos, P->ye, ny + 1, *yIndex);
    if (fabs(P->xe[xi - 1] - xPos) < spacing_d(xPos)) xi++;
    if (fabs(P->ye[yi - 1] - yPos) < spacing_d(yPos)) yi++;
    if (xi >= nx + 1) xi = 1;
    if (yi >= ny + 1) yi = 1;
    *xIndex = xi; *yIndex = yi;
  }
}

/* ------------------------------------------------------------------------ */
/* radiance by local estimation                                              */
/* ------------------------------------------------------------------------ */
void orc_intensity_directions(int n, const float *mus, const float *phisDeg, float *dirs) { /* :1270-1272 */
  const float Pi = 3.14159265358979312f;
  for (int i = 0; i < n; i++) make_direction_cosines(mus[i], phisDeg[i] * Pi / 180.0f, dirs + 3 * i);
}

void orc_forward_angles(int nAngles, float *angles) { /* opticalProperties.f95:1914 */
  const float Pi = 3.14159265358979312f;
  for (int j = 0; j < nAngles; j++) angles[j] = ((float)j / (float)(nAngles - 1)) * Pi;
}

/* computeNormalization, opticalProperties.f95:2026-2050 (1-based transitionIndex) */
static float hybrid_normalization(int nAngles, const float *angleCosines, const float *values,
                                  const float *gaussianValues, int transitionIndex) {
  float integralGaus = 0.0f, integralOrig = 0.0f;
  for (int k = 1; k <= transitionIndex - 1; k++)
    integralGaus += (0.5f * (gaussianValues[k - 1] + gaussianValues[k])) * (angleCosines[k - 1] - angleCosines[k]);
  for (int k = transitionIndex; k <= nAngles - 1; k++)
    integralOrig += (0.5f * (values[k - 1] + values[k])) * (angleCosines[k - 1] - angleCosines[k]);
  if (integralOrig >= 2.0f) return 1.0f / integralGaus;
  return (2.0f - integralOrig) / integralGaus;
}
static float hybrid_diff(int nAngles, const float *angleCosines, const float *values, const float *gaussianValues,
                         int transitionIndex) { /* phaseFuncDiff :2011-2024 */
  float P0 = hybrid_normalization(nAngles, angleCosines, values, gaussianValues, transitionIndex);
  return P0 * gaussianValues[transitionIndex - 1] - values[transitionIndex - 1];
}

void orc_hybrid_phase_functions(int nAngles, int nEntries, const float *angles, const float *values,
                                float gaussianWidthDeg, float *newValues) { /* :1937-2009 */
  const float Pi = 3.14159265358979312f;
  float *gaus = (float *)malloc(sizeof(float) * nAngles), *cosA = (float *)malloc(sizeof(float) * nAngles);
  for (int k = 0; k < nAngles; k++) {
    cosA[k] = cosf(angles[k]);
    float x = angles[k] / (gaussianWidthDeg * Pi / 180.0f);
    gaus[k] = expf(-(x * x));
  }
  memcpy(newValues, values, sizeof(float) * (size_t)nAngles * nEntries);
  for (int e = 0; e < nEntries; e++) {
    const float *v = values + (size_t)e * nAngles;
    float *nv = newValues + (size_t)e * nAngles;
    int lowerBound = orc_find_index_real(gaussianWidthDeg * Pi / 180.0f, angles, nAngles, 0) + 1;
    if (lowerBound >= nAngles - 2) break; /* exit entryLoop */
    float lowDiff = hybrid_diff(nAngles, cosA, v, gaus, lowerBound), upDiff = 0.0f;
    int increment = 1, upperBound = lowerBound, noRoot = 0;
    for (;;) { /* huntingLoop */
      upperBound = lowerBound + increment < nAngles - 1 ? lowerBound + increment : nAngles - 1;
      upDiff = hybrid_diff(nAngles, cosA, v, gaus, upperBound);
      if (lowerBound == nAngles - 1) { noRoot = 1; break; }
      if (lowDiff * upDiff < 0.0f) break;
      lowerBound = upperBound;
      lowDiff = upDiff;
      increment *= 2;
    }
    if (noRoot) continue; /* cycle entryLoop: keep the original phase function */
    for (;;) { /* bisectionLoop */
      if (upperBound <= lowerBound + 1) break;
      int midPoint = (lowerBound + upperBound) / 2;
      float midDiff = hybrid_diff(nAngles, cosA, v, gaus, midPoint);
      if (midDiff * upDiff < 0.0f) { lowerBound = midPoint; lowDiff = midDiff; }
      else { upperBound = midPoint; upDiff = midDiff; }
    }
    int transitionIndex = lowerBound;
    float P0 = hybrid_normalization(nAngles, cosA, v, gaus, transitionIndex);
    for (int k = 0; k < transitionIndex; k++) nv[k] = P0 * gaus[k];
  }
  free(gaus); free(cosA);
}

float orc_lookup_phase_value(const float *table, int nAngleSteps, float scatteringAngle) { /* :1835-1870 */
  const float Pi = 3.14159265358979312f;
  float deltaTheta = Pi / (float)(nAngleSteps - 1);
  int angleIndex = (int)(scatteringAngle / deltaTheta) + 1;
  if (angleIndex < nAngleSteps) {
    float weight = 1.0f - (scatteringAngle - (float)(angleIndex - 1) * deltaTheta) / deltaTheta;
    return weight * table[angleIndex - 1] + (1.0f - weight) * table[angleIndex];
  }
  return table[nAngleSteps - 1];
}

/* computeIntensityContribution :1623-1832.  Philox slots: direction i of event e uses block
 * 0x100 + i = [tauFree, roulette, -, -]. */
static void intensity_contribution(const orc_problem *P, const orc_intensity *I, orc_rng *R, float photonWeight,
                                   double xPos, double yPos, double zPos, int xIndex, int yIndex, int zIndex,
                                   const float dir[3], int component, int scatteringOrder, float *contributions,
                                   int *xIndexF, int *yIndexF, float *intensityExcess) {
  const float Pi = 3.14159265358979312f;
  const int nDir = I->nDirections;
  const int zIndexMax = P->nz + 1; /* size(zPosition) */
  for (int i = 0; i < nDir; i++) {
    const float *D = I->directions + 3 * i;
    float normalizedPhaseFunc;
    if (component == 0) normalizedPhaseFunc = 1.0f / Pi;                            /* :1691 Lambertian surface */
    else if (component < 0) normalizedPhaseFunc = 1.0f / ((4.0f * Pi) * fabsf(D[2])); /* :1696 isotropic emission */
    else {
      float projection = dir[0] * D[0] + dir[1] * D[1] + dir[2] * D[2]; /* :1704 */
      if (fabsf(projection) > 1.0f) projection = copysignf(1.0f, projection);
      float scatteringAngle = acosf(projection);
      int pfi = P->pfIndex[IDX4(P, xIndex, yIndex, zIndex, component)];
      int nA = I->fwdNAngles[component - 1];
      const float *base = (I->useHybrid && scatteringOrder <= I->numOrdersOrig) ? I->fwdOrigTables : I->fwdTables;
      float val = orc_lookup_phase_value(base + I->fwdOffset[component - 1] + (size_t)(pfi - 1) * nA, nA, scatteringAngle);
      normalizedPhaseFunc = val / ((4.0f * Pi) * fabsf(D[2])); /* :1726 */
    }
    double pos[3] = {xPos, yPos, zPos};
    int32_t idx[3] = {xIndex, yIndex, zIndex};
    float contribution;
    if (!I->useRussianRoulette) { /* :1729-1752 */
      float tau = orc_accumulate_extinction(P, D, pos, idx, 0, 0.0f, NULL);
      contribution = tau >= 0.0f ? (photonWeight * normalizedPhaseFunc) * expf(-tau) : 0.0f;
    } else { /* Iwabuchi 2006 :1753-1813 */
      float u = draw(R, 0x100u + (uint32_t)i, 0);
      float tauFree = -logf(u > FLT_MIN ? u : FLT_MIN);
      if (Pi * normalizedPhaseFunc <= I->zetaMin) {
        (void)orc_accumulate_extinction(P, D, pos, idx, 1, tauFree, NULL);
        float u2 = draw(R, 0x100u + (uint32_t)i, 1);
        if (u2 <= Pi * normalizedPhaseFunc / I->zetaMin && idx[2] >= zIndexMax) contribution = photonWeight * I->zetaMin / Pi;
        else contribution = 0.0f;
      } else {
        float m = Pi * normalizedPhaseFunc;
        float tauMax = -logf(I->zetaMin / (m > FLT_MIN ? m : FLT_MIN));
        float tau = orc_accumulate_extinction(P, D, pos, idx, 1, tauMax, NULL);
        if (idx[2] >= zIndexMax && tau >= 0.0f) {
          contribution = (photonWeight * normalizedPhaseFunc) * expf(-tau);
        } else if (tau >= 0.0f && idx[2] < 1) {
          /* a downward direction left through the surface: the reference would restart the walk with
           * zIndex = 0 (out of bounds); roulette is only meaningful for upward directions */
          contribution = 0.0f;
        } else if (tau >= 0.0f) {
          (void)orc_accumulate_extinction(P, D, pos, idx, 1, tauFree, NULL);
          contribution = idx[2] >= zIndexMax ? photonWeight * I->zetaMin / Pi : 0.0f;
        } else {
          contribution = 0.0f;
        }
      }
    }
    if (I->limitContributions && contribution > I->maxContribution) { /* :1815-1826 */
      const int cc = component < 0 ? 0 : component; /* (the reference indexes intensityExcess(:, -1) out of bounds) */
      if (intensityExcess) intensityExcess[(size_t)cc * nDir + i] += contribution - I->maxContribution;
      contribution = I->maxContribution;
    }
    contributions[i] = contribution;
    /* the column the ray leaves through; a ray stopped inside keeps its last cell */
    xIndexF[i] = idx[0] < 1 ? 1 : (idx[0] > P->nx ? P->nx : idx[0]);
    yIndexF[i] = idx[1] < 1 ? 1 : (idx[1] > P->ny ? P->ny : idx[1]);
  }
}

static void add_intensity(const orc_problem *P, const orc_intensity *I, const float *contributions, const int *xF,
                          const int *yF, int component, float *intensity, float *byComponent) {
  const size_t ncol = (size_t)P->nx * P->ny;
  for (int i = 0; i < I->nDirections; i++) { /* :535-540, :696-701, :785-790 */
    size_t b = (size_t)(xF[i] - 1) + (size_t)P->nx * (yF[i] - 1) + ncol * i;
    intensity[b] += contributions[i];
    if (byComponent) byComponent[(size_t)component * I->nDirections * ncol + b] += contributions[i];
  }
}

/* ------------------------------------------------------------------------ */
/* computeRT :393-841                                                        */
/* ------------------------------------------------------------------------ */
int64_t orc_compute_rt(const orc_problem *P, const orc_source *S, orc_rng *R, int64_t numPhotons,
                       float *fluxUp, float *fluxDown, float *fluxAbsorbed, float *volumeAbsorption,
                       orc_counters *C, orc_fate *fates) {
  return orc_compute_rt_intensity(P, S, R, numPhotons, fluxUp, fluxDown, fluxAbsorbed, volumeAbsorption, C, fates,
                                  NULL, NULL, NULL, NULL);
}

int64_t orc_compute_rt_intensity(const orc_problem *P, const orc_source *S, orc_rng *R, int64_t numPhotons,
                                 float *fluxUp, float *fluxDown, float *fluxAbsorbed, float *volumeAbsorption,
                                 orc_counters *C, orc_fate *fates, const orc_intensity *I, float *intensity,
                                 float *intensityByComponent, float *intensityExcess) {
  const int nx = P->nx, ny = P->ny, nz = P->nz, nc = P->nc;
  const size_t ncol = (size_t)nx * ny, nvox = ncol * nz;
  const float Pi = 3.14159265358979312f; /* :31 */
  const float RussianRouletteW = 1.0f;   /* :56 */
  orc_counters cnt;
  memset(&cnt, 0, sizeof(cnt));
  memset(fluxUp, 0, sizeof(float) * ncol);        /* computeRadiativeTransfer :248-252 */
  memset(fluxDown, 0, sizeof(float) * ncol);
  memset(fluxAbsorbed, 0, sizeof(float) * ncol);
  memset(volumeAbsorption, 0, sizeof(float) * nvox);
  const int nDir = (I && intensity) ? I->nDirections : 0;
  float *contributions = NULL;
  int *xIndexF = NULL, *yIndexF = NULL;
  if (nDir > 0) { /* :257-261, :452-455 */
    memset(intensity, 0, sizeof(float) * ncol * nDir);
    if (intensityByComponent) memset(intensityByComponent, 0, sizeof(float) * ncol * nDir * (nc + 1));
    if (intensityExcess) memset(intensityExcess, 0, sizeof(float) * nDir * (nc + 1));
    contributions = (float *)malloc(sizeof(float) * nDir);
    xIndexF = (int *)malloc(sizeof(int) * nDir);
    yIndexF = (int *)malloc(sizeof(int) * nDir);
  }

  grid_flags G;
  orc_grid_flags(P, &G.xyRegular, &G.zRegular, &G.dX, &G.dY, &G.dZ);
  const double x0 = P->xe[0], xMax = P->xe[nx];
  const double y0 = P->ye[0], yMax = P->ye[ny];
  const double z0 = P->ze[0], zMax = P->ze[nz];
  const double albedo = P->albedo;

  /* MT mode: the whole stream is generated first (new_PhotonStream), as the
   * driver does (monteCarloDriver.f95:950-957), then traced. */
  launch_t *stream = NULL;
  if (R->mode == 0) {
    stream = (launch_t *)malloc(sizeof(launch_t) * (size_t)(numPhotons > 0 ? numPhotons : 1));
    for (int64_t i = 0; i < numPhotons; i++) launch_one(P, S, R, &stream[i]);
  }
  uint64_t drawsAtStart = R->ndraws;

  const char *traceEnv = getenv("ORC_TRACE_PHOTON");
  const int64_t tracePhoton = traceEnv ? (int64_t)strtoll(traceEnv, NULL, 10) : -1;
  int64_t nPhotons = 0;
  for (int64_t ip = 0; ip < numPhotons; ip++) { /* photonLoop :463 */
    launch_t L;
    uint64_t photonDraws0;
    if (R->mode == 0) { L = stream[ip]; photonDraws0 = R->ndraws; }
    else {
      philox_start_photon(R, R->firstPhoton + (uint64_t)ip);
      photonDraws0 = R->ndraws;
      launch_one(P, S, R, &L);
    }
    double xPos = L.x, yPos = L.y, zPos = L.z;
    float mu = L.mu, phi = L.phi;
    int scatteringOrder = 0;
    float dir[3];
    make_direction_cosines(mu, phi, dir); /* :471 */
    float photonWeight = 1.0f;
    nPhotons++;
    int xIndex = 1, yIndex = 1, zIndex = 1; /* :478 */
    xPos = x0 + xPos * (xMax - x0);
    yPos = y0 + yPos * (yMax - y0);
    find_xy_indices(P, &G, xPos, yPos, &xIndex, &yIndex);
    if (G.zRegular) { /* :484-486, findZIndex :1580-1592 */
      zPos = z0 + zPos * (zMax - z0);
      zIndex = (int)((zPos - z0) / G.dZ) + 1; if (zIndex > nz) zIndex = nz;
      if (fabs(P->ze[zIndex] - zPos) < spacing_d(zPos)) zIndex++;
    } else { /* :491-493 */
      double t = (zPos - z0) * (double)nz;
      double remainder = t - floor(t);
      zIndex = (int)floor(t) + 1; if (zIndex > nz) zIndex = nz;
      zPos = P->ze[zIndex - 1] + remainder * (P->ze[zIndex] - P->ze[zIndex - 1]);
    }
    if (P->lwFlag > 0.0f) { /* :504-508 */
      if (zPos > 0.0) {
        size_t c2 = (size_t)(xIndex - 1) + (size_t)nx * (yIndex - 1);
        fluxAbsorbed[c2] = fluxAbsorbed[c2] - 1.0f;
        size_t v3 = IDX3(P, xIndex, yIndex, zIndex);
        volumeAbsorption[v3] = volumeAbsorption[v3] - 1.0f;
      }
      if (nDir > 0) { /* :510-541 emission seen directly: surface (component 0) or atmosphere (-1) */
        intensity_contribution(P, I, R, photonWeight, xPos, yPos, zPos, xIndex, yIndex, zIndex, dir,
                               zPos == 0.0 ? 0 : -1, scatteringOrder, contributions, xIndexF, yIndexF, intensityExcess);
        add_intensity(P, I, contributions, xIndexF, yIndexF, 0, intensity, intensityByComponent);
      }
    }
    int fate = -1;
    float fateWeight = 0.0f;
    for (;;) { /* scatteringLoop :548 */
      if (R->mode == 1) philox_next_event(R); /* event e: block 0 = [tau, X, Y, Z], block 1 = [-, roulette, -, -] (one component: roulette = Z); scattering angle = X, azimuth = Y, component = Z */
      float u = draw(R, 0, 0);
      float tauToTravel = -logf(u > FLT_MIN ? u : FLT_MIN); /* :554 */
      cnt.legs++;
      double pos[3] = {xPos, yPos, zPos};
      int32_t idx[3] = {xIndex, yIndex, zIndex};
      float tauAccumulated = orc_accumulate_extinction(P, dir, pos, idx, 1, tauToTravel, &cnt.crossings);
      xPos = pos[0]; yPos = pos[1]; zPos = pos[2];
      xIndex = idx[0]; yIndex = idx[1]; zIndex = idx[2];
      if (tauAccumulated < 0.0f) { cnt.badPhotons++; fate = 3; break; } /* :562-563 */

      if (zPos >= zMax) { /* :573-617 */
        size_t c2 = (size_t)(xIndex - 1) + (size_t)nx * (yIndex - 1);
        fluxUp[c2] = fluxUp[c2] + photonWeight;
        cnt.topExits++;
        fate = 0; fateWeight = photonWeight;
        break;
      } else if (zPos <= z0 + spacing_d(z0)) { /* :619-676 */
        zIndex = 1;
        zPos = z0 + spacing_d(z0);
        size_t c2 = (size_t)(xIndex - 1) + (size_t)nx * (yIndex - 1);
        fluxDown[c2] = fluxDown[c2] + photonWeight;
        cnt.surfaceHits++;
        scatteringOrder++;
        for (uint32_t j = 0;; j++) { /* Philox slots: mu attempt 0 = X, 1 = Z, then block 2.. ; phi = Y */
          mu = sqrtf(j == 0 ? draw(R, 0, 1) : j == 1 ? draw(R, 0, 3) : draw(R, 2 + (j - 2) / 4, (j - 2) % 4));
          if (fabsf(mu) > 2.0f * FLT_MIN) break;
        }
        phi = (2.0f * Pi) * draw(R, 0, 2);
        float wBefore = photonWeight;
        if (P->surfNumX > 0) /* useSurfaceBDRF :667-670 (incoming / outgoing angles are not used by the reference's R) */
          photonWeight = photonWeight * surface_reflectance(P, xPos, yPos);
        else
          photonWeight = (float)((double)photonWeight * albedo); /* :673 */
        if (photonWeight <= FLT_MIN) { cnt.surfaceAbsorbed++; fate = 1; fateWeight = wBefore; break; }
        if (R->mode == 1) { /* Philox mode: cos / sin of 2 pi Y by the kernel's expression (sincos_2pi), as in next_direct */
          float sinTheta = sqrtf(1.0f - mu * mu), c, s;
          sincos_2pi(draw(R, 0, 2), &c, &s);
          dir[0] = sinTheta * c; dir[1] = sinTheta * s; dir[2] = mu;
        } else {
          make_direction_cosines(mu, phi, dir);
        }
        if (nDir > 0) { /* :680-702 */
          intensity_contribution(P, I, R, photonWeight, xPos, yPos, zPos, xIndex, yIndex, zIndex, dir, 0,
                                 scatteringOrder, contributions, xIndexF, yIndexF, intensityExcess);
          add_intensity(P, I, contributions, xIndexF, yIndexF, 0, intensity, intensityByComponent);
        }
      } else { /* scattering event :703-821 */
        if (tracePhoton >= 0 && ip == tracePhoton) /* development aid: ORC_TRACE_PHOTON=<index> */
          fprintf(stderr, "ORCTRACE %d ev %u cell %d %d %d pos %.17g %.17g %.17g dir %.9g %.9g %.9g tau %.9g w %.9g\n", scatteringOrder,
                  R->event, xIndex, yIndex, zIndex, xPos, yPos, zPos, dir[0], dir[1], dir[2], tauToTravel, photonWeight);
        scatteringOrder++;
        cnt.collisions++;
        if (P->totalExt[IDX3(P, xIndex, yIndex, zIndex)] <= 0.0) { /* :728-754 */
          if (xPos - P->xe[xIndex - 1] <= 0.0 && dir[0] > 0.0f) {
            xPos = xPos - spacing_d(xPos);
            xIndex = xIndex - 1;
            if (xIndex <= 0) {
              xIndex = nx;
              xPos = P->xe[xIndex - 1];
              xPos = xPos - 2.0 * spacing_d(xPos);
            }
          }
          if (yPos - P->ye[yIndex - 1] <= 0.0 && dir[1] > 0.0f) {
            yPos = yPos - spacing_d(yPos);
            yIndex = yIndex - 1;
            if (yIndex <= 0) {
              yIndex = ny;
              /* :743 uses xPosition(yIndex) (sic); guarded so the C does not read out of bounds */
              yPos = (yIndex <= nx + 1) ? P->xe[yIndex - 1] : P->ye[yIndex - 1];
              yPos = yPos - 2.0 * spacing_d(yPos);
            }
          }
          if (zPos - P->ze[zIndex - 1] <= 0.0 && dir[2] > 0.0f) {
            zPos = zPos - spacing_d(zPos);
            zIndex = zIndex - 1;
          }
          if (zIndex < 1) { cnt.badPhotons++; fate = 3; break; } /* reference would index out of bounds */
        }
        double tbl[ORC_MAX_COMPONENTS + 1];
        tbl[0] = 0.0;
        for (int c = 1; c <= nc; c++) tbl[c] = P->cumExt[IDX4(P, xIndex, yIndex, zIndex, c)];
        int component = orc_find_index_mixed(draw(R, 0, 3), tbl, nc + 1, 0); /* slot Z of the leg's block */ /* :759-760 */
        if (component < 1) component = 1;
        if (component > nc) component = nc;
        float ssa = (float)P->ssa[IDX4(P, xIndex, yIndex, zIndex, component)]; /* :764 */
        if ((double)ssa < 1.0) { /* :765-771 */
          size_t c2 = (size_t)(xIndex - 1) + (size_t)nx * (yIndex - 1);
          size_t v3 = IDX3(P, xIndex, yIndex, zIndex);
          fluxAbsorbed[c2] = (float)((double)fluxAbsorbed[c2] + (double)photonWeight * (1.0 - (double)ssa));
          volumeAbsorption[v3] = (float)((double)volumeAbsorption[v3] + (double)photonWeight * (1.0 - (double)ssa));
          photonWeight = photonWeight * ssa;
          cnt.absorbEvents++;
        }
        if (nDir > 0) { /* :776-790 */
          intensity_contribution(P, I, R, photonWeight, xPos, yPos, zPos, xIndex, yIndex, zIndex, dir, component,
                                 scatteringOrder, contributions, xIndexF, yIndexF, intensityExcess);
          add_intensity(P, I, contributions, xIndexF, yIndexF, component, intensity, intensityByComponent);
        }
        if (P->useRussianRoulette && photonWeight < RussianRouletteW / 2.0f) { /* :805-811 */
          /* Philox mode: block 1 elem 1 -- or, in a domain of ONE component, elem 3 of the leg's own block (Z: the uniform of the
           * component pick, which with one component decides nothing), so that the kernels need no second block for it */
          if (draw(R, P->nc == 1 ? 0 : 1, P->nc == 1 ? 3 : 1) >= photonWeight / RussianRouletteW) { photonWeight = 0.0f; cnt.rouletteKills++; }
          else { photonWeight = RussianRouletteW; cnt.rouletteSurvivals++; }
        }
        if (photonWeight <= FLT_MIN) { fate = 2; break; } /* :812 */
        int pfi = P->pfIndex[IDX4(P, xIndex, yIndex, zIndex, component)]; /* :816 */
        int ns = P->invNSteps[component - 1];
        const float *tblf = P->invTables + P->invOffset[component - 1] + (size_t)(pfi - 1) * ns;
        float scatteringAngle = compute_scattering_angle(draw(R, 0, 1), tblf, ns); /* slot X */
        next_direct(R, cosf(scatteringAngle), dir); /* :819 */
      }
    }
    if (fates) {
      fates[ip].fate = fate;
      fates[ip].ix = xIndex; fates[ip].iy = yIndex; fates[ip].iz = zIndex;
      fates[ip].nScatter = scatteringOrder;
      fates[ip].nDraws = (int32_t)(R->ndraws - photonDraws0);
      fates[ip].weight = fateWeight;
    }
  }
  cnt.draws = R->ndraws - drawsAtStart;
  if (C) *C = cnt;
  free(stream);
  free(contributions); free(xIndexF); free(yIndexF);
  return nPhotons;
}

void orc_normalize_intensity(const orc_problem *P, const orc_intensity *I, int64_t numPhotonsProcessed,
                             float *intensity, float *intensityByComponent, const float *intensityExcess) {
  const int nx = P->nx, ny = P->ny, nc = P->nc, nDir = I->nDirections;
  const size_t ncol = (size_t)nx * ny;
  if (I->limitContributions && intensityByComponent && intensityExcess) { /* :294-320 */
    for (int j = 0; j <= nc; j++)
      for (int d = 0; d < nDir; d++) {
        float ex = intensityExcess[(size_t)j * nDir + d];
        if (ex > 0.0f) {
          float *bc = intensityByComponent + ((size_t)j * nDir + d) * ncol;
          float sum = 0.0f;
          for (size_t i = 0; i < ncol; i++) sum += bc[i];
          for (size_t i = 0; i < ncol; i++) intensity[(size_t)d * ncol + i] += (bc[i] / sum) * ex;
          for (size_t i = 0; i < ncol; i++) bc[i] = bc[i] + (bc[i] / sum) * ex;
        }
      }
  }
  grid_flags G;
  orc_grid_flags(P, &G.xyRegular, &G.zRegular, &G.dX, &G.dY, &G.dZ);
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) { /* :331-342, :369-379 */
      float nppc;
      if (G.xyRegular) nppc = (float)numPhotonsProcessed / (float)(nx * ny);
      else {
        double rel = ((P->ye[j + 1] - P->ye[j]) * (P->xe[i + 1] - P->xe[i])) / ((P->xe[nx] - P->xe[0]) * (P->ye[ny] - P->ye[0]));
        nppc = (float)rel * (float)numPhotonsProcessed;
      }
      size_t c2 = (size_t)i + (size_t)nx * j;
      for (int d = 0; d < nDir; d++) intensity[(size_t)d * ncol + c2] /= nppc;
      if (intensityByComponent)
        for (int k = 0; k < (nc + 1) * nDir; k++) intensityByComponent[(size_t)k * ncol + c2] /= nppc;
    }
}

/* computeRadiativeTransfer :328-364 */
void orc_normalize(const orc_problem *P, int64_t numPhotonsProcessed, float *fluxUp, float *fluxDown,
                   float *fluxAbsorbed, float *volumeAbsorption) {
  int nx = P->nx, ny = P->ny, nz = P->nz;
  size_t ncol = (size_t)nx * ny;
  grid_flags G;
  orc_grid_flags(P, &G.xyRegular, &G.zRegular, &G.dX, &G.dY, &G.dZ);
  float *nppc = (float *)malloc(sizeof(float) * ncol);
  if (G.xyRegular) {
    float v = (float)numPhotonsProcessed / (float)(nx * ny); /* :331 */
    for (size_t i = 0; i < ncol; i++) nppc[i] = v;
  } else {
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) { /* :334-342 */
        double rel = ((P->ye[j + 1] - P->ye[j]) * (P->xe[i + 1] - P->xe[i])) /
                     ((P->xe[nx] - P->xe[0]) * (P->ye[ny] - P->ye[0]));
        float r = (float)rel;
        nppc[(size_t)i + (size_t)nx * j] = r * (float)numPhotonsProcessed;
      }
  }
  for (size_t i = 0; i < ncol; i++) {
    fluxUp[i] = fluxUp[i] / nppc[i];
    fluxDown[i] = fluxDown[i] / nppc[i];
    fluxAbsorbed[i] = fluxAbsorbed[i] / nppc[i];
  }
  for (int k = 0; k < nz; k++) { /* :361-364 */
    double dz = P->ze[k + 1] - P->ze[k];
    for (size_t i = 0; i < ncol; i++) {
      size_t v = i + ncol * k;
      volumeAbsorption[v] = (float)((double)volumeAbsorption[v] / (((double)nppc[i] * dz) * 1000.0));
    }
  }
  free(nppc);
}

/* reportResults :877-884, :966 */
void orc_report_means(const orc_problem *P, const float *fluxUp, const float *fluxDown,
                      const float *fluxAbsorbed, const float *volumeAbsorption, float *meanUp,
                      float *meanDown, float *meanAbs, float *absorbedProfile) {
  size_t ncol = (size_t)P->nx * P->ny;
  float su = 0.f, sd = 0.f, sa = 0.f;
  for (size_t i = 0; i < ncol; i++) { su += fluxUp[i]; sd += fluxDown[i]; sa += fluxAbsorbed[i]; }
  if (meanUp) *meanUp = su / (float)ncol;
  if (meanDown) *meanDown = sd / (float)ncol;
  if (meanAbs) *meanAbs = sa / (float)ncol;
  if (absorbedProfile)
    for (int k = 0; k < P->nz; k++) {
      float s = 0.f;
      for (size_t i = 0; i < ncol; i++) s += volumeAbsorption[i + ncol * k];
      absorbedProfile[k] = s / (float)ncol;
    }
}
