/*
 * mcbrat.h -- C ABI of the MI355X-native photon-tracing integrator.
 *
 * This is the drop-in boundary for ONE path of MCBRaT3D: the integrator module
 * `monteCarloRadiativeTransfer` (Integrators/monteCarloRadiativeTransfer.f95,
 * public list :121-123).  Each entry point names the reference interface it
 * replaces.  Plain C types only: pointers, sizes, scalars.  All 3-D arrays are
 * Fortran order (x fastest, then y, z, component slowest) exactly as the
 * reference's getInfo_Domain hands them to computeRT (:434-443), so a Fortran
 * caller passes its arrays unchanged (see INTEGRATION.md for the
 * ISO_C_BINDING module).
 *
 * Error convention: every call returns 0 on success, non-zero on failure;
 * mcbrat_last_error() then holds the text the Fortran shim pushes with
 * setStateToFailure (src/ErrorMessages.f95:171-245).
 *
 * Threading: one context per GPU, not shared between host threads (the
 * reference integrator holds mutable tallies and is not thread-safe either).
 * There is no CPU fallback: mcbrat_create fails if no HIP device is usable.
 */
#ifndef MCBRAT_H
#define MCBRAT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCBRAT_MAX_COMPONENTS 8
#define MCBRAT_MAX_DIRECTIONS 64 /* intensity directions per run */
#define MCBRAT_ABI_VERSION 3

typedef struct mcbrat_ctx mcbrat_ctx;

/* Per-photon record written by mcbrat_trace_fates (parity/debug only). */
typedef struct {
  int32_t fate;     /* 0 top exit, 1 absorbed by surface, 2 roulette kill, 3 dropped by a loop bound (counters.badPhotons) */
  int32_t ix, iy, iz; /* 1-based cell of the final event */
  int32_t nScatter; /* scattering order at the end (:469, :643, :713) */
  int32_t nEvents;  /* legs traced */
  float weight;     /* weight tallied at the final event */
} mcbrat_fate;

/* Event counters of the last run (sums over photons), for the roofline figure. */
typedef struct {
  int64_t legs, crossings, collisions, absorbEvents, topExits, surfaceHits,
          rouletteKills, rouletteSurvivals;
  /* wave-level loop statistics: iterations of the walk loop and lanes walking in them, event
   * phases and lanes served in them, phases that launched photons / reflected off the surface */
  int64_t walkIterations, walkLanes, eventPhases, eventLanes, launchPhases, surfacePhases;
  /* Photons (and radiance rays) dropped since the context was created because they exceeded a loop bound of the
   * kernels -- the analogue of computeRT's nBad (Integrators/monteCarloRadiativeTransfer.f95:562-563: a photon whose
   * step is not positive is dropped and counted).  The reference's walk ends because it marches by cell index; the
   * kernels here bound every loop instead (legs per photon, loop iterations of a wave without any lane starting a
   * leg, length of a view ray), so that no input can keep a kernel from ending.  Always collected; 0 in every run
   * the tests and bench.py make.  (ABI version 2.) */
  int64_t badPhotons;
} mcbrat_counters;

int mcbrat_abi_version(void);

/* new_Integrator (:129-201) / finalize_Integrator (:1486-1547).
 * device: HIP device ordinal. */
mcbrat_ctx *mcbrat_create(int device);
void mcbrat_destroy(mcbrat_ctx *ctx);
const char *mcbrat_last_error(const mcbrat_ctx *ctx);

/* Domain geometry: getInfo_Domain(xPosition, yPosition, zPosition) as used by
 * new_Integrator :147-181.  Edges in km, nx+1 / ny+1 / nz+1 values. */
int mcbrat_set_grid(mcbrat_ctx *ctx, int32_t nx, int32_t ny, int32_t nz,
                    const double *xEdges, const double *yEdges, const double *zEdges);

/* Optical properties: getInfo_Domain(albedo, totalExt, cumExt, ssa, phaseFuncI)
 * as pulled by computeRT :441-443 (built by getOpticalPropertiesByComponent,
 * src/opticalProperties.f95:966-1072).  phaseFuncIndex is 1-based. */
int mcbrat_set_optics(mcbrat_ctx *ctx, int32_t nComponents, const double *totalExt,
                      const double *cumExt, const double *ssa, const int32_t *phaseFuncIndex,
                      double surfaceAlbedo);

/* inversePhaseFuncs(component)%values(nSteps, nEntries) (computeRT :443, :816-818;
 * built by tabulateInversePhaseFunctions, opticalProperties.f95:1817-1870).
 * component is 1-based; table is column-major (nSteps fastest). */
int mcbrat_set_inverse_table(mcbrat_ctx *ctx, int32_t component, int32_t nSteps,
                             int32_t nEntries, const float *table);

/* specifyParameters (:1046-1484): the keywords the driver passes
 * (monteCarloDriver.f95:540-572).  useRayTracing must be non-zero: the
 * max-cross-section branch is broken in the reference
 * and is rejected here. */
int mcbrat_specify_parameters(mcbrat_ctx *ctx, int32_t useRayTracing, int32_t useRussianRoulette,
                              float LW_flag);

/* new_PhotonStream, Directional form (src/monteCarloIllumination.f95:62-101). */
int mcbrat_set_source_solar(mcbrat_ctx *ctx, float solarMu, float solarAzimuthDeg);
/* new_PhotonStream, BBEmission form (:431-522): the running voxel CDF and
 * atmosphere fraction from emission_weighting / getInfo_Weights. */
int mcbrat_set_source_emission(mcbrat_ctx *ctx, const double *voxelWeights, double fracAtmsPower);

/* computeRadiativeTransfer (:209-391) for nBatches consecutive batches of
 * photonsPerBatch photons, followed on the device by reportResults (:845-1042)
 * and the driver's moment accumulation (monteCarloDriver.f95:1023-1050) for each
 * batch.  Photon i of batch b has the global id firstPhotonId + b*photonsPerBatch + i;
 * its random numbers are Philox4x32-10(key = seed, counter = (event, block, id)),
 * so results do not depend on how photons are spread over GPUs.
 * numPhotonsProcessed (may be NULL) receives the total. */
int mcbrat_compute_radiative_transfer(mcbrat_ctx *ctx, uint64_t seed, uint64_t firstPhotonId,
                                      int64_t photonsPerBatch, int32_t nBatches,
                                      int64_t *numPhotonsProcessed);

/* reportResults (:845-1042) for the LAST batch traced: normalised exactly as
 * computeRadiativeTransfer :328-364.  Any pointer may be NULL. */
int mcbrat_report_results(mcbrat_ctx *ctx, float *meanFluxUp, float *meanFluxDown,
                          float *meanFluxAbsorbed, float *fluxUp, float *fluxDown,
                          float *fluxAbsorbed, float *absorbedProfile, float *volumeAbsorption);

/* ---- radiance by local estimation (computeIntensityContribution :1623-1832) ---------------
 * specifyParameters(intensityMus, intensityPhis, computeIntensity, useRussianRouletteForIntensity,
 * zetaMin, useHybridPhaseFunsForIntenCalcs, numOrdersOrigPhaseFunIntenCalcs,
 * limitIntensityContributions, maxIntensityContribution) (:1046-1292).  nDirections = 0 turns the
 * intensity calculation off.  mus in [-1, 1] \ {0}, phis in degrees [0, 360].  Every emitted-photon
 * launch, surface reflection and scattering event then adds weight * phase function /
 * (4 pi |mu|) * transmission to the pixel where the view ray leaves the domain.
 * Roulette (Iwabuchi 2006, the driver's default) is accepted for upward directions only: for
 * mu < 0 the reference restarts its walk below the surface.  With limitIntensityContributions each
 * local estimate is clipped at maxIntensityContribution and the clipped excess of a (component,
 * direction) is spread over the pixels in proportion to that component's radiance field at the end
 * of the batch (:294-320, :1815-1826).  Changing the number of directions changes
 * mcbrat_moments_length(): a caller-bound moment buffer must be bound again. */
int mcbrat_specify_intensity(mcbrat_ctx *ctx, int32_t nDirections, const float *intensityMus,
                             const float *intensityPhisDeg, int32_t useRussianRouletteForIntensity,
                             float zetaMin, int32_t useHybridPhaseFunsForIntenCalcs,
                             int32_t numOrdersOrigPhaseFunIntenCalcs,
                             int32_t limitIntensityContributions, float maxIntensityContribution);
/* tabulatedPhaseFunctions(component)%values(nAngles, nEntries) and tabulatedOrigPhaseFunctions
 * (tabulateForwardPhaseFunctions, opticalProperties.f95:1872-1935; pulled by
 * computeIntensityContribution :1672): phase function values at nAngles scattering angles equally
 * spaced on [0, pi], angle fastest.  origTable = NULL when no hybrid tables are used. */
int mcbrat_set_forward_table(mcbrat_ctx *ctx, int32_t component, int32_t nAngles, int32_t nEntries,
                             const float *table, const float *origTable);
/* reportResults(meanIntensity, intensity) (:980-1010) for the LAST batch: intensity is
 * [nDirections][ny][nx] (x fastest).  Either pointer may be NULL. */
int mcbrat_report_intensity(mcbrat_ctx *ctx, float *meanIntensity, float *intensity);

/* Batch moments: what the driver keeps in *Stats(...,1:2)
 * (monteCarloDriver.f95:603-616) and reduces with sumAcrossProcesses
 * (:1151-1166).  One double array:
 *   [0] total photons  [1] batches completed  [2..7] reserved
 *   then S1 = sum n*x and S2 = sum n*x^2, each of length mcbrat_moments_length():
 *   meanFluxUp, meanFluxDown, meanFluxAbsorbed, fluxUp[nx*ny], fluxDown[nx*ny],
 *   fluxAbsorbed[nx*ny], absorbedProfile[nz], absorbedVolume[nx*ny*nz],
 *   intensity[nDirections*nx*ny] (RadianceStats, monteCarloDriver.f95:1047-1050).
 * Total doubles = 8 + 2*length.  The buffer is device memory; a caller that
 * wants to all-reduce it with RCCL binds its own device buffer. */
int64_t mcbrat_moments_length(const mcbrat_ctx *ctx);
int mcbrat_bind_moments(mcbrat_ctx *ctx, double *deviceBuffer); /* NULL: library-owned */
/* The device buffer the moments are accumulated in (library-owned unless one was bound).  Several contexts on one
 * grid -- one per wavelength of a spectrally integrated run, each with its optical properties resident -- accumulate
 * into ONE array when the others bind the first one's pointer (monteCarloDriver.f95:1023-1050 keeps one set of
 * *Stats arrays over all wavelengths).  Calls into contexts that share a buffer must not overlap (synchronous mode). */
double *mcbrat_moments_device_pointer(mcbrat_ctx *ctx);

/* getFrequencyDistr (src/emissionAndBroadBandWeights.f95:552-572, driver :438-449): the number of photons each of
 * numLambda wavelengths receives out of totalPhotons, one uniform per photon against the running power CDF
 * (findCDFIndex: smallest i with U <= CDF(i)).  The reference draws the uniforms from its MT stream in a host loop
 * over all photons; here draw d of the run is a Philox uniform keyed by `seed` (counter (0xFFFFFFFF, 0, d / 4),
 * element d % 4; d counts from firstDraw), drawn and counted on the device.  cdf and distribution are host arrays. */
int mcbrat_frequency_distribution(mcbrat_ctx *ctx, uint64_t seed, uint64_t firstDraw, int32_t numLambda, const double *cdf,
                                  int64_t totalPhotons, int64_t *distribution);
int mcbrat_reset_moments(mcbrat_ctx *ctx); /* stream-ordered: enqueued before whatever the context does next; a caller
                                              that reads a bound buffer itself calls mcbrat_synchronize first */
int mcbrat_get_moments(mcbrat_ctx *ctx, double *hostBuffer);

/* Measurement: HIP-event duration of the tracing kernel(s) of the last
 * mcbrat_compute_radiative_transfer call, and its event counters (counters
 * are only collected when enabled; they cost a few percent). */
int mcbrat_enable_counters(mcbrat_ctx *ctx, int32_t enable);
int mcbrat_get_counters(mcbrat_ctx *ctx, mcbrat_counters *out); /* synchronises; badPhotons is valid whether or not counters are enabled */
/* What was dropped FIRST since the context was created (badPhotons > 0): which bound fired, in which kernel, the photon's id,
 * state and leg count -- one occurrence is then enough to trace that photon again alone and find the cause (computeRT only counts, :562-563).  An empty
 * string while nothing was dropped.  Synchronises; the text lives until the next call of this function on the context.  ABI 3. */
const char *mcbrat_first_drop(mcbrat_ctx *ctx);
float mcbrat_last_trace_ms(const mcbrat_ctx *ctx);
/* Asynchronous mode.  A launch ends with its longest photon history, so every call carries a fixed
 * drain time; a caller that issues many calls -- the reference's driver calls
 * computeRadiativeTransfer once per batch, monteCarloDriver.f95:1008 -- can let consecutive calls
 * overlap: with mcbrat_set_async(ctx, 1), mcbrat_compute_radiative_transfer and mcbrat_reset_moments
 * only enqueue work (on a small set of HIP streams owned by the context; per-batch results and
 * moments are still folded in call order, so results are bitwise the same as in synchronous mode)
 * and return.  Every call that reads results or replaces inputs (report_results, get_moments,
 * set_*, bind_moments, trace_fates, destroy) synchronises first; mcbrat_synchronize does so
 * explicitly, after which mcbrat_last_trace_ms is the summed tracing-kernel time of the calls since
 * the previous synchronisation. */
int mcbrat_set_async(mcbrat_ctx *ctx, int32_t enable);
int mcbrat_synchronize(mcbrat_ctx *ctx);
/* Stream interop for a caller that reduces the bound moment buffer on its own HIP stream (RCCL):
 * mcbrat_stream_wait_done makes `hipStream` (a hipStream_t) wait for everything enqueued so far;
 * mcbrat_wait_stream makes the context's next write to the moments wait for what `hipStream` has
 * enqueued so far.  Neither blocks the host. */
int mcbrat_stream_wait_done(mcbrat_ctx *ctx, void *hipStream);
int mcbrat_wait_stream(mcbrat_ctx *ctx, void *hipStream);
/* Several contexts that accumulate into ONE moment array (mcbrat_bind_moments with another context's
 * mcbrat_moments_device_pointer: one context per wavelength of a spectrally integrated run, monteCarloDriver.f95:889-1085) must
 * fold their batches one context after the other.  In synchronous mode the host does that by waiting for every call.  With
 * mcbrat_set_async on each of them, mcbrat_chain_after(ctx, previous) orders the device instead: the finish kernels (and the
 * reset) of ctx's NEXT call run after everything `previous` has enqueued so far, while the tracing kernels of the two contexts
 * overlap -- the tail of one wavelength's launch is filled by the next wavelength's photons.  Results are bitwise those of
 * synchronous calls in the same order.  Both contexts must live on the same device.  "So far" is fixed when this function is
 * called: ctx records an event of its OWN on `previous`'s stream and waits for that, so `previous` may be destroyed, or go on
 * to further calls, before ctx's next one. */
int mcbrat_chain_after(mcbrat_ctx *ctx, mcbrat_ctx *previous);
/* Tuning knobs (negative = leave unchanged): workgroups per CU (0 = occupancy query), number of
 * walking lanes below which a wave serves its waiting lanes (0 = choose by timing short trial
 * launches, the default), batches in flight per launch
 * (0 = memory bound), privateTallies (0 global atomics; 1 the library's plan: tallies private to a workgroup in LDS where the
 * slab fits beside a second workgroup, else the wide plan -- one workgroup of 1024 lanes per compute unit with up to its
 * whole 160 KB of LDS -- else global atomics; 2 private tallies without the optical grid in LDS; 3 as 1 without the wide plan;
 * 4 the wide plan even where the shared one would do; 5 as 4 with the per-cell optics left in global memory; 6 as 4 without
 * the per-cell optics in LDS -- per block in LDS where every block is uniform in them, else as 5),
 * workgroup size (0 = automatic, 256, 512, 768; a fixed size rules the wide plan out), and how
 * many idle / surface lanes queue up before launches / surface reflections are served; brickLayout:
 * 0 dense optical grids, 1 4x4x4 bricks with unstored background bricks, 2 automatic (default). */
int mcbrat_set_tuning(mcbrat_ctx *ctx, int32_t blocksPerCU, int32_t eventThreshold, int32_t maxBatchesInFlight,
                      int32_t privateTallies, int32_t blockSize, int32_t launchThreshold, int32_t surfaceThreshold,
                      int32_t brickLayout);

/* Scheduling options by name (none of them changes a result: the same photons, the same arithmetic per photon, integer
 * tallies -- tests/test_gpu_tunings.py holds random choices against the defaults bit for bit).  ABI 3.
 *   "jumpThreshold", "crossThreshold"  lanes queued before transitions of the layer-skipping walk / block crossings are served
 * The reference has no counterpart (its loop is one photon at a time, monteCarloRadiativeTransfer.f95:463-466). */
int mcbrat_set_option(mcbrat_ctx *ctx, const char *name, int32_t value);

/* specifyParameters(surfaceBDRF = new_SurfaceDescription(surfaceParameters, xPosition, yPosition))
 * (Integrators/monteCarloRadiativeTransfer.f95:1173-1176, src/surfaceProperties.f95:58-94): a reflecting surface whose
 * Lambertian reflectance varies from patch to patch on its own horizontal positions (numX x-positions bound numX-1
 * patches; `reflectance` is BRDFParameters(1, :, :), x fastest).  A photon that reaches the surface is multiplied by
 * the reflectance of the patch under it (computeSurfaceReflectance :119-147) instead of the domain's albedo
 * (computeRT :667-673).  The uniform surface of newSurfaceUniform is numX = numY = 2 with positions (0, huge(1.)).
 * numX <= 0 returns to the domain's albedo.  Error texts are the reference's. */
int mcbrat_set_surface_description(mcbrat_ctx *ctx, int32_t numX, int32_t numY, const double *xPosition,
                                   const double *yPosition, const float *reflectance);

/* Walk options (negative = leave unchanged).
 * layerSkip (default 1): inside a horizontal layer whose cells all carry one extinction value -- the clear air
 * above and below a cloud field -- a photon crosses z faces only; its column is found again from its position when
 * it collides, leaves the domain or reaches a layer with cell-to-cell extinction.  The reference's
 * accumulateExtinctionAlongPath (src/opticalProperties.f95:1697-1814) stops at every x and y face too, which changes
 * nothing inside such a layer but the float rounding of the accumulated optical depth; 0 restores that face-by-face
 * walk (used by the per-photon identity tests).
 * With layerSkip = 1 the same idea is applied inside the other layers too (the clear-air flight): columns are grouped
 * 4 x 4 into brick columns; outside the layers in which a brick column holds a cell that differs from its layer's most
 * common ("background") extinction, a photon whose remaining optical depth cannot be used up by the background
 * between its height and the domain boundary steps from brick column to brick column, and takes its optical depth
 * from the background's vertical optical depth when that flight ends (at a cloud, or at the domain boundary).  Grids
 * whose column counts are multiples of four, at most 255 layers, fluxes only, and a background whose vertical optical
 * depth is at most 0.5 (in a haze most flights would be refused, and asking costs a turn in a queue).  layerSkip = 2:
 * the one-extinction layers only, no flight; 3: flights whatever the optical depth of the background (tests).
 * blockWalk (default 1): domains whose optical grid is resident in LDS (I3RC step cloud, plane-parallel and other
 * small domains): the grid is cut into axis-aligned blocks of cells that carry one extinction value, and a leg goes
 * from block face to block face instead of from cell face to cell face; the cell of a collision or an exit is found
 * from the position.  Same argument as for layerSkip: inside a block the reference's stops at cell faces only add
 * `segment x the same extinction` again.  0 restores the face-by-face walk; 2 uses the block walk even where blocks
 * hold fewer than four cells on average (a medium that differs from cell to cell: slower, meant for tests). */
int mcbrat_set_walk_options(mcbrat_ctx *ctx, int32_t layerSkip, int32_t blockWalk);
/* The walk a flux run of the loaded grid and optics uses, decided by the same plan as the launch itself: bit 0
 * layerSkip, bit 1 the block walk (asked for AND applicable: grid, tallies and tables fit in LDS, blocks hold four
 * cells or more on average, no radiance directions), bit 2 the clear-air flight (asked for and possible: brick columns
 * exist, the background is thin enough or layerSkip = 3, no radiance directions, the grid is not held in LDS and the
 * flight's tables fit beside the rest), bit 3 the blockWalk option as set, bit 6 tallies private to a workgroup in LDS,
 * bit 4 the wide plan (a tally slab too large to share a compute unit's LDS: one workgroup of 1024 lanes per compute unit
 * owns up to its whole 160 KB), bit 5 the block walk with the per-cell optics left in global memory (extinction per block
 * in LDS), bit 7 the block walk with the optics per BLOCK in LDS (every block uniform in them), bit 8 the thermal source's level
 * and row sums of the emission CDF staged in LDS.  Before grid and optics are loaded: the options. */
int mcbrat_get_walk_mode(const mcbrat_ctx *ctx);

/* The event threshold in use (after the first call of a domain: the one chosen by the trial launches). */
int mcbrat_get_event_threshold(const mcbrat_ctx *ctx);

/* Parity/debug: trace n photons (ids firstPhotonId..) and record what became
 * of each one.  Tallies and moments of the context are left untouched. */
int mcbrat_trace_fates(mcbrat_ctx *ctx, uint64_t seed, uint64_t firstPhotonId, int64_t n,
                       mcbrat_fate *fates);

/* ---- host-side set-up routines on the path (no GPU needed) ------------- */
/* computeInversePhaseFunction (src/inversePhaseFunctions.f95:66-174) for a
 * phase function stored as Legendre coefficients chi_1..chi_n (P0 = 1 implied). */
int mcbrat_inverse_table_legendre(int32_t nCoefficients, const float *coefficients,
                                  int32_t nSteps, float *table);
/* ...and for one stored as (scatteringAngle, value) pairs; values are
 * normalised as new_PhaseFunction does (scatteringPhaseFunctions.f95:156). */
int mcbrat_inverse_table_tabulated(int32_t nAngles, const float *scatteringAngle,
                                   const float *value, int32_t nSteps, float *table);
/* tabulateForwardPhaseFunctions for one phase function (opticalProperties.f95:1914-1916 ->
 * getPhaseFunctionValues, scatteringPhaseFunctions.f95:480-527): values at nAngles angles equally
 * spaced on [0, pi], from Legendre coefficients or from (angle, value) pairs. */
int mcbrat_forward_table_legendre(int32_t nCoefficients, const float *coefficients, int32_t nAngles,
                                  float *table);
int mcbrat_forward_table_tabulated(int32_t nStored, const float *scatteringAngle, const float *value,
                                   int32_t nAngles, float *table);
/* computeHybridPhaseFunctions (opticalProperties.f95:1937-2009) on values[nEntries][nAngles]. */
int mcbrat_hybrid_phase_functions(int32_t nAngles, int32_t nEntries, const float *values,
                                  float gaussianWidthDeg, float *hybridValues);
/* The decomposition the block walk uses (mcbrat_set_walk_options, blockWalk): axis-aligned boxes of cells with one
 * extinction value, found greedily (x, then y, then z).  extinction[nx*ny*nz] (x fastest, as the kernels hold it:
 * float); blockOf[nx*ny*nz]: the block of each cell; blockRec[4*nBlocks] (room for 4*nx*ny*nz): per block
 * {x0 | x1 << 16, y0 | y1 << 16, z0 | z1 << 16, flags}, cell range [lo, hi) per axis, flag bit 0 / 1 = the block
 * spans the whole periodic x / y axis.  Host arithmetic only.  Returns 1 for more than 65535 blocks / cells per axis. */
int mcbrat_block_decomposition(int32_t nx, int32_t ny, int32_t nz, const float *extinction, uint16_t *blockOf,
                               uint32_t *blockRec, int32_t *nBlocks);
/* The tables of the layer-skipping walk and the clear-air flight (mcbrat_set_walk_options, layerSkip), as the library
 * builds them when the optics are set.  extinction[nx*ny*nz] (x fastest, float), zEdges[nz+1].  background[nz]: the
 * most common extinction of every layer; range[(ny/4)*(nx/4)]: per brick column of 4 x 4 columns lo | hi << 8, the
 * layers [lo, hi) that hold a cell differing from its layer's background (lo = nz, hi = 0: none); walk[nx*ny*nz]: the
 * extinction with the sign bit set in every cell outside the range of its brick column; depth[nz+1]: vertical optical
 * depth of the background below every face, over the layers a flight can cross; *flights: 1 when brick columns exist
 * (nx, ny multiples of four, 2..255 layers) and some range is not empty.  range / walk are written only when brick
 * columns exist.  Host arithmetic only. */
int mcbrat_flight_tables(int32_t nx, int32_t ny, int32_t nz, const float *extinction, const double *zEdges,
                         float *background, uint16_t *range, float *walk, double *depth, int32_t *flights);

/* emission_weighting (src/emissionAndBroadBandWeights.f95:424-550): builds
 * voxelWeights (running CDF, x fastest), fracAtmsPower and the emitted flux. */
int mcbrat_emission_weighting(int32_t nx, int32_t ny, int32_t nz, int32_t nComponents,
                              const double *xEdges, const double *yEdges, const double *zEdges,
                              const double *temps, const double *totalExt, const double *cumExt,
                              const double *ssa, double surfaceAlbedo, double lambdaMicrons,
                              double surfaceTemp, double dLambda, double *voxelWeights,
                              double *fracAtmsPower, double *totalFlux);

#ifdef __cplusplus
}
#endif
#endif
