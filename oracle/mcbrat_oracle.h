/*
 * mcbrat_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the photon-tracing hot path of MCBRaT3D
 * (Integrators/monteCarloRadiativeTransfer.f95 computeRT and the routines it
 * calls).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.  The product path
 * (mcbrat3d_amd/, include/mcbrat.h) never links or calls it.
 *
 * Pinning status: see oracle/README.md.  Short form: the utility layer
 * (findIndex / findCDFIndex / Lobatto / Legendre) is pinned bit-for-bit
 * against the reference's own numericUtilities.f95 compiled from
 * /root/reference (oracle/_ref); MT19937 is pinned against the canonical
 * known answers and the stream recorded in SURVEY.md section 8c; the full
 * photon loop is pinned against the reference outputs recorded in SURVEY.md
 * section 8c (step cloud, 1e5 and 1e6 photons) by replaying the same MT
 * stream.  The reference integrator itself cannot be rebuilt in this image
 * without a stand-in netcdf module, so no _ref binary exists for the loop.
 */
#ifndef MCBRAT_ORACLE_H
#define MCBRAT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_COMPONENTS 8

/* Domain + integrator parameters: the argument list computeRT pulls out of
 * the domain with getInfo_Domain (monteCarloRadiativeTransfer.f95:434-443).
 * All 3-D arrays are Fortran order, x fastest, component slowest. */
typedef struct {
  int32_t nx, ny, nz, nc;
  const double *xe, *ye, *ze;       /* cell edges, nx+1 / ny+1 / nz+1 (km)  */
  const double *totalExt;            /* [nz][ny][nx]                          */
  const double *cumExt;              /* [nc][nz][ny][nx]                      */
  const double *ssa;                 /* [nc][nz][ny][nx]                      */
  const int32_t *pfIndex;            /* [nc][nz][ny][nx], 1-based entry       */
  double albedo;                     /* Lambertian surface albedo             */
  const float *invTables;            /* all inverse tables, concatenated      */
  const int64_t *invOffset;          /* [nc] float offset of component table  */
  const int32_t *invNSteps;          /* [nc] points per entry                 */
  const int32_t *invNEntries;        /* [nc] entries                          */
  int32_t useRussianRoulette;
  float lwFlag;                      /* > 0: thermal emission bookkeeping     */
  /* surfaceDescription (src/surfaceProperties.f95:33-36), set by specifyParameters(surfaceBDRF=)
   * (monteCarloRadiativeTransfer.f95:1173-1176); surfNumX == 0: the domain's Lambertian albedo is used */
  int32_t surfNumX, surfNumY;        /* size(xPosition), size(yPosition)      */
  const double *surfXPosition, *surfYPosition;
  const float *surfReflectance;      /* BRDFParameters(1, :, :): [numY-1][numX-1], x fastest */
} orc_problem;

/* Photon source (src/monteCarloIllumination.f95). */
typedef struct {
  int32_t kind;                      /* 0 = Directional, 1 = BBEmission       */
  float solarMu, solarAzimuthDeg;    /* kind 0                                */
  const double *voxelWeights;        /* kind 1: running CDF [nz][ny][nx]      */
  double fracAtmsPower;              /* kind 1                                */
} orc_source;

/* Random numbers.  mode 0: MT19937 exactly as src/RandomNumbersForMC.f95
 * (one sequential stream, state carried across calls; the draw ORDER is the
 * reference's).  mode 1: Philox4x32-10, the counter-based generator the
 * BASELINE north_star substitutes.  key = seed, counter = (event, block,
 * photon id lo, photon id hi).  event 0 is the launch, event e >= 1 the e-th
 * leg; each role has a fixed slot (see draw() call sites in mcbrat_oracle.c and
 * the same table in mcbrat3d_amd/csrc/mcbrat_kernels.hip), so the HIP kernel and
 * this oracle consume identical uniforms for identical roles. */
typedef struct {
  int32_t mode;
  int32_t mti;
  uint32_t mt[624];
  uint64_t seed;                     /* philox key                            */
  uint64_t firstPhoton;              /* philox: id of photon 0 of this batch  */
  /* scratch */
  uint64_t photon;
  uint32_t event;
  uint32_t cachedBlock;              /* block index held in buf, or 0xffffffff */
  uint32_t buf[4];
  uint64_t ndraws;
} orc_rng;

typedef struct {
  int64_t legs, crossings, collisions, absorbEvents, topExits, surfaceHits,
          rouletteKills, rouletteSurvivals, badPhotons, surfaceAbsorbed;
  uint64_t draws;
} orc_counters;

/* Optional per-photon record (debug / parity): what became of photon i. */
typedef struct {
  int32_t fate;      /* 0 top exit, 1 absorbed by surface, 2 roulette kill, 3 dropped (tau<0) */
  int32_t ix, iy, iz;/* 1-based cell of the final event */
  int32_t nScatter;  /* scatteringOrder at the end */
  int32_t nDraws;    /* uniforms consumed while tracing (philox mode: incl. launch) */
  float weight;      /* weight tallied at the final event */
} orc_fate;

/* ---- RandomNumbersForMC.f95 ------------------------------------------- */
void orc_mt_init_scalar(orc_rng *r, int32_t seed);                 /* :171-187 */
void orc_mt_init_vector(orc_rng *r, const int32_t *seed, int n);   /* :189-241 */
uint32_t orc_mt_next_u32(orc_rng *r);                              /* :245-260 */
void orc_philox_init(orc_rng *r, uint64_t seed, uint64_t firstPhoton);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_random_real(orc_rng *r);                                 /* :277-301 */

/* ---- numericUtilities.f95 ---------------------------------------------- */
void orc_lobatto(int n, float *mus, float *weights);               /* :27-114  */
void orc_legendre(int maxL, int nmu, const float *mus, float *out);/* :187-205; out[(maxL+1)*nmu], l fastest */
int  orc_find_index_real(float v, const float *t, int n, int firstGuess);     /* :417-470 (0 = no guess) */
int  orc_find_index_double(double v, const double *t, int n, int firstGuess); /* :207-260 */
int  orc_find_index_mixed(float v, const double *t, int n, int firstGuess);   /* :262-315 */
void orc_frequency_distribution(orc_rng *r, uint64_t firstDraw, int numLambda, const double *cdf, int64_t totalPhotons, int64_t *distribution); /* emissionAndBroadBandWeights.f95:552-572 */
float orc_surface_reflectance(const orc_problem *P, double xPos, double yPos); /* surfaceProperties.f95:119-147 */
int  orc_find_cdf_index(float v, const double *t, int n);                     /* :317-348 */

/* ---- scatteringPhaseFunctions.f95 / inversePhaseFunctions.f95 ----------- */
void orc_phase_values_legendre(int ncoef, const float *coef, int nang,
                               const float *angles, float *values);      /* sPF:480-498 */
void orc_normalize_phase_function(int n, const float *angles, const float *vin,
                                  float *vout);                          /* sPF:1520-1536 */
void orc_phase_values_tabulated(int nst, const float *stAngles, const float *stValues,
                                int nang, const float *angles, float *values); /* sPF:500-527 */
int orc_inverse_table_legendre(int ncoef, const float *coef, int nsteps, float *table); /* iPF:66-174 */
int orc_inverse_table_tabulated(int nang, const float *angles, const float *values,
                                int nsteps, float *table);

/* ---- opticalProperties.f95 --------------------------------------------- */
float orc_accumulate_extinction(const orc_problem *P, const float dir[3], double pos[3],
                                int32_t idx[3], int hasTarget, float target,
                                int64_t *crossings);                     /* :1656-1815 */

/* ---- emissionAndBroadBandWeights.f95:424-550 ---------------------------- */
int orc_emission_weighting(int nx, int ny, int nz, int nc, const double *xe, const double *ye,
                           const double *ze, const double *temps, const double *totalExt,
                           const double *cumExt, const double *ssa, double albedo,
                           double lambda_um, double sfcTemp, double dLambda,
                           double *voxelWeights, double *fracAtmsPower, double *totalFlux);

/* ---- radiance by local estimation (monteCarloRadiativeTransfer.f95:1623-1870) ---- */
/* What specifyParameters (:1236-1292) and tabulateForwardPhaseFunctions
 * (opticalProperties.f95:1872-1935) leave in the integrator / domain. */
typedef struct {
  int32_t nDirections;
  const float *directions;           /* [nDir][3] makeDirectionCosines(mu, phi*Pi/180) :1270-1272 */
  const float *fwdTables;            /* tabulatedPhaseFunctions: components concatenated, [entry][angle] */
  const float *fwdOrigTables;        /* tabulatedOrigPhaseFunctions, same layout */
  const int64_t *fwdOffset;          /* [nc] float offset of a component's table */
  const int32_t *fwdNAngles;         /* [nc] angles per entry (equally spaced 0..pi) */
  int32_t useHybrid;                 /* useHybridPhaseFunsForIntenCalcs */
  int32_t numOrdersOrig;             /* numOrdersOrigPhaseFunIntenCalcs */
  int32_t useRussianRoulette;        /* useRussianRouletteForIntensity */
  float zetaMin;
  int32_t limitContributions;        /* limitIntensityContributions */
  float maxContribution;
} orc_intensity;

/* makeDirectionCosines(mu, phiDegrees * Pi/180) for the intensity directions (:1270-1272). */
void orc_intensity_directions(int n, const float *mus, const float *phisDeg, float *dirs);
/* tabulateForwardPhaseFunctions: values of nEntries phase functions at nAngles equally spaced
 * angles (opticalProperties.f95:1914-1916).  Legendre storage (coefficients chi_1.. per entry,
 * ragged via start/length) or angle-value storage share orc_phase_values_*; this helper only
 * makes the angle grid. */
void orc_forward_angles(int nAngles, float *angles);
/* computeHybridPhaseFunctions (opticalProperties.f95:1937-2009) on [entry][angle] tables. */
void orc_hybrid_phase_functions(int nAngles, int nEntries, const float *angles, const float *values,
                                float gaussianWidthDeg, float *newValues);
/* lookUpPhaseFuncValsFromTable (:1835-1870) for one angle. */
float orc_lookup_phase_value(const float *table, int nAngles, float scatteringAngle);
/* computeRT with intensity: intensity [nDir][ny][nx] raw sums; intensityByComponent
 * [(nc+1)][nDir][ny][nx] and intensityExcess [(nc+1)][nDir] may be NULL unless
 * limitContributions is set. */
int64_t orc_compute_rt_intensity(const orc_problem *P, const orc_source *S, orc_rng *R,
                                 int64_t numPhotons, float *fluxUp, float *fluxDown,
                                 float *fluxAbsorbed, float *volumeAbsorption, orc_counters *C,
                                 orc_fate *fates, const orc_intensity *I, float *intensity,
                                 float *intensityByComponent, float *intensityExcess);
/* computeRadiativeTransfer :294-320 (redistribution of excess) and :367-379 (normalisation). */
void orc_normalize_intensity(const orc_problem *P, const orc_intensity *I, int64_t numPhotonsProcessed,
                             float *intensity, float *intensityByComponent, const float *intensityExcess);

/* ---- monteCarloRadiativeTransfer.f95 ------------------------------------ */
/* computeRT (:393-841): raw tallies (zeroed here as :248-252 does). */
int64_t orc_compute_rt(const orc_problem *P, const orc_source *S, orc_rng *R,
                       int64_t numPhotons, float *fluxUp, float *fluxDown,
                       float *fluxAbsorbed, float *volumeAbsorption,
                       orc_counters *C, orc_fate *fates);
/* normalisation of computeRadiativeTransfer (:328-364), in place. */
void orc_normalize(const orc_problem *P, int64_t numPhotonsProcessed, float *fluxUp,
                   float *fluxDown, float *fluxAbsorbed, float *volumeAbsorption);
/* reportResults (:877-884, :966) domain means and absorption profile. */
void orc_report_means(const orc_problem *P, const float *fluxUp, const float *fluxDown,
                      const float *fluxAbsorbed, const float *volumeAbsorption,
                      float *meanUp, float *meanDown, float *meanAbs, float *absorbedProfile);
/* new_Integrator regular-spacing detection (:163-181). */
void orc_grid_flags(const orc_problem *P, int *xyRegular, int *zRegular,
                    double *deltaX, double *deltaY, double *deltaZ);

#ifdef __cplusplus
}
#endif
#endif
