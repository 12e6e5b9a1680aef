// mcbrat_device.h -- parameter block shared by host launch code and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcbrat.h"

namespace mcbrat {

constexpr int kWave = 64;                     // CDNA wavefront
constexpr double kTallyScale = 4294967296.0;  // tallies are signed 64-bit fixed point, 2^-32 resolution
constexpr double kTallyInv = 1.0 / 4294967296.0;

// Loop bounds of the production kernels -- compile-time constants there: as kernel parameters they
// lived in scalar registers across the whole loop and cost the step cloud 3 % through spilled SGPRs.  The instrumented
// instantiations (mcbrat_trace_fates, counters) take them from DevParams, so that tests can reach a bound quickly.
constexpr unsigned kMaxEvents = 1u << 24;     // legs per photon
constexpr unsigned kMaxEventsNaN = 1u << 20;  // legs of a photon with a NaN direction that is still at full weight
constexpr unsigned kWatchdog = 1u << 20;      // trace_kernel: event phases of a wave in which none of its lanes makes progress; block walk: block crossings of one leg

// TEST ONLY build (-DMCBRAT_TEST_BOUNDS_IN_PRODUCTION, loaded through MCBRAT_LIB by tests/test_gpu_bounds.py): the PRODUCTION
// instantiations take their loop bounds from DevParams too and honour the legacy tie switch, so that the code that keeps a
// production kernel from hanging -- which differs from the instrumented instantiation's (compile-time constants, per-lane drop
// counter) -- is itself driven into its bounds once.  The shipped library is built without it.
#ifdef MCBRAT_TEST_BOUNDS_IN_PRODUCTION
constexpr bool kTestBounds = true;
#else
constexpr bool kTestBounds = false;
#endif

// What the loop bounds dropped, beyond the count: so that one occurrence can be diagnosed from the run it happened in.
// Layout of the device buffer behind DevParams::bad: [0] count, [1] claim word of the record (0 = none yet), [2..3] the record,
// [7] which bounds have fired (bit kind - 1).
//  * every kernel keeps the KINDS: per lane in the upper bits of its drop counter (a register it has anyway, no pointer and no
//    atomic in the loop), OR-ed into [7] when the kernel ends;
//  * the instrumented instantiations (and the test-only build of the production ones) also record the FIRST drop: kind, photon
//    id, state, leg.  Not the production kernels: the record's pointer and atomic in the rare branch cost the 128x128x64 kernel
//    5 % through register allocation (same-box A/B, profiles/r04_experiments.txt); photon ids are reproducible, so the same
//    photons traced through mcbrat_trace_fates name the photon.
enum : unsigned { DROP_LEGS = 1, DROP_NAN_LEGS = 2, DROP_WATCHDOG = 3, DROP_CROSSINGS = 4, DROP_VACUUM = 5, DROP_RAY = 6 };
constexpr int kBadWords = 8;
constexpr unsigned kBadCountMask = 0x03ffffffu;  // per-lane drop counter: count in the low 26 bits, kinds above
__device__ __forceinline__ unsigned count_drop(unsigned nBadLane, unsigned kind) {
  return (((nBadLane & kBadCountMask) + 1u) & kBadCountMask) | (nBadLane & ~kBadCountMask) | (1u << (25u + kind));
}
__device__ __forceinline__ void publish_drops(unsigned long long *bad, unsigned nBadLane) {  // when the kernel ends
  if (nBadLane != 0u) { atomicAdd(bad, (unsigned long long)(nBadLane & kBadCountMask)); atomicOr(bad + 7, (unsigned long long)(nBadLane >> 26)); }
}
__device__ __forceinline__ void record_first_drop(unsigned long long *bad, unsigned kind, unsigned kernel, uint32_t idLo, uint32_t idHi, int state, uint32_t event) {
  if (atomicCAS(bad + 1, 0ull, 1ull) != 0ull) return;  // (somebody else was first)
  bad[2] = (unsigned long long)kind | ((unsigned long long)kernel << 8) | ((unsigned long long)(unsigned)state << 16) | ((unsigned long long)event << 32);
  bad[3] = ((unsigned long long)idHi << 32) | idLo;
}

// States of a lane in the tracing loop.
// ST_ENTER: a lane on the layer-skipping walk has reached a layer whose extinction varies from cell to cell and
// waits for the event phase to bring its x/y state up to date.
// ST_JUMP: a lane has entered a run of such layers and waits for the event phase to take the run in one step -- or
// has entered a cell outside its brick column's cloud range and waits for the event phase to start a clear-air flight.
enum : int { ST_DEAD = 0, ST_WALK = 1, ST_COLLIDE = 2, ST_SURFACE = 3, ST_TOP = 4, ST_ENTER = 5, ST_JUMP = 6 };

struct DevParams {
  // grid (set_grid)
  int nx, ny, nz, nc;
  int xyRegular, zRegular;        // new_Integrator :163-181 flags (launch cell lookup only)
  double x0, y0, z0, xMax, yMax, zMax, Lx, Ly;
  double zSurf;                   // z0 + spacing(z0): where a reflected photon restarts (:633)
  double invDX, invDY;            // 1/deltaX, 1/deltaY when xyRegular
  int xyRegularWalk, zRegularWalk; // equally spaced axes: the walk steps their face distances by dXf|1/dir| etc.
  float dXf, dYf, dZf;
  // layer-skipping walk (dense layout): in a layer with one extinction value only z faces are crossed; the x/y
  // cell is found again from the position (periodic fold + cell lookup) when the lane leaves such layers
  int layerSkip;
  const int *layerRun;            // [nz] run of one-extinction layers around layer k: face where it ends upwards << 16 | downwards
  const double *layerRunT;        // [nz+1] vertical optical depth of the layers' background extinction below face f
  // clear-air flight (dense layout, flux runs; mcbrat_kernels.hip): columns are grouped 4 x 4 into brick columns; outside the
  // layers [lo, hi) in which a brick column holds a cell that differs from its layer's background extinction, a lane
  // whose optical depth cannot be used up by the background alone steps from brick column to brick column
  int fly;
  int flyNbx, flyNby;             // brick columns along x / y (0: off)
  float flyInvBrickX, flyInvBrickY;  // 1 / mean width of a brick column
  const uint16_t *flyRange;       // [flyNby][flyNbx] lo | hi << 8 (lo = nz, hi = 0: the brick column is background throughout)
  const float *extWalk;           // [nvox] what the photons' walk reads: ext, with the sign bit set in the cells outside their brick column's range (fly off: ext itself)
  const float *bgVal;             // [nz] background extinction of every layer (its most common value)
  int xyNearUniform;              // x and y edges equally spaced to 1e-6 of a cell: cell guess by division, table decides
  int zNearUniform;               // the same for z (block walk, SIMPLE = 3)
  double invLx, invLy;            // 1 / domain length
  double invCellX, invCellY;      // nx / Lx, ny / Ly
  const double *edges;            // [xe(nx+1) | ye(ny+1) | ze(nz+1)]
  // optics (set_optics), float copies of the reference's real(8) arrays
  const float *ext;               // [nvox]
  const float *cum;               // [nc][nvox] (read only when nc > 1)
  const float *ssa;               // [nc][nvox]
  const uint16_t *pfi;            // [nc][nvox] 0-based entry
  const uint4 *rec;               // dense layout, nc <= 2: per cell {cum[0], ssa[0], ssa[1], pfi[0] | pfi[1] << 16}, else null
  float albedo;
  // surfaceDescription (specifyParameters(surfaceBDRF=), src/surfaceProperties.f95): reflectance per surface patch on its
  // own x/y positions; surfNumX == 0: the domain's Lambertian albedo
  int surfNumX, surfNumY;
  const double *surfX, *surfY;
  const float *surfRefl;          // [numY-1][numX-1]
  // brick layout of the optics (large, mostly-background domains): ext/cum/ssa/pfi then hold the
  // STORED bricks only (64 cells each, [component][nStored]); background cells use bg* [component][nz]
  const uint32_t *brickTable;     // [nbz][nby][nbx] offset of the brick's 64 cells, 0xffffffff = background
  int nbx, nby, nbz;
  long long nStored;              // stored cells = 64 * stored bricks
  const float *bgExt;             // [nz]
  const float *bgCum, *bgSsa;     // [nc][nz]
  const uint16_t *bgPfi;          // [nc][nz]
  // block walk (LDS-resident grids, mcbrat_blockwalk.hip): axis-aligned blocks of cells with one extinction value
  int nBlocks;
  const uint4 *blockRec;          // [nBlocks] {x0 | x1 << 16, y0 | y1 << 16, z0 | z1 << 16, flags}: cell range [lo, hi) per axis;
                                  // flag bit 0 / 1: the block spans the whole periodic x / y axis (no face on it)
  const uint16_t *blockOf;        // [nvox] block of each cell
  const float *blockExt;          // [nBlocks] extinction of each block (what LDS holds when the per-cell optics stay in global memory)
  const float *blockSsa, *blockCum;  // [nc][nBlocks] optics per block, where every block is uniform in them too (trace_block_kernel<..., OPT = 2>)
  const uint16_t *blockPfi;
  int cdfTopLds;                  // thermal source, block walk: the level and row sums of the emission CDF are staged in LDS
  int crossThreshold;             // lanes queued before block crossings are served
  // inverse phase-function tables (set_inverse_table)
  const float *tables;            // all components, concatenated, entry-major
  int tblOffset[MCBRAT_MAX_COMPONENTS];
  int tblNSteps[MCBRAT_MAX_COMPONENTS];
  float tblInvN[MCBRAT_MAX_COMPONENTS];   // 1/nSteps
  int tblTotalFloats;
  // parameters / source
  int useRR;
  int lwFlag;                     // LW_flag > 0
  int srcKind;                    // 0 directional, 1 BB emission
  float dir0[3];                  // directional: launch direction cosines
  double zLaunch;                 // directional: launch height
  int izLaunch;                   // directional: 0-based launch layer
  const double *voxelCDF;         // emission: running CDF [nvox]
  double fracAtms;
  // radiance by local estimation (computeIntensityContribution :1623-1832); nDir == 0: fluxes only
  int nDir;
  const float *dirData;           // [nDir][8]: direction cosines, 4 pi |mu|, 1/dx, 1/dy, 1/dz (0 where |d| is tiny), pad
  const float *fwdTables;         // tabulatedPhaseFunctions: components concatenated, [entry][angle]
  const float *fwdOrig;           // tabulatedOrigPhaseFunctions (hybrid runs), same layout
  int fwdOffset[MCBRAT_MAX_COMPONENTS];
  int fwdNAngles[MCBRAT_MAX_COMPONENTS];
  int useHybrid, numOrdersOrig;   // original tables up to this scattering order, hybrid ones beyond
  int useRRIntensity;             // Iwabuchi (2006) roulette on the local estimates
  float zetaMin;
  int rayShort, rayPassIters, rayPassAt;  // iterations a ray gets inside its event phase / in one pass over the buffer; rays in the buffer that start a pass
  int rayCap;                     // unfinished long rays a wave can put aside in LDS (0: every ray is finished inside its event phase)
  int limitContrib;               // limitIntensityContributions: clip each local estimate, redistribute the excess (:294-320, :1815-1826)
  float maxContrib;
  // work
  unsigned long long *counter;    // next global photon index
  unsigned long long total;       // photons in this launch
  unsigned long long ppb;         // photons per batch
  unsigned long long firstPhoton; // global id of photon index 0
  uint32_t seedLo, seedHi;
  long long *slabs;               // per batch: [fluxUp(ncol) | fluxDown(ncol) | volume(nvox)]
  unsigned long long slabStride;  // in elements
  unsigned long long nUnits;      // PRIV mode: work units (each inside one batch) ...
  unsigned long long unitsPerBatch;  // ... and how many of them a batch is cut into
  int eventThreshold;             // process events when fewer than this many lanes are walking
  int launchThreshold;            // idle lanes queued before new photons are launched
  int surfaceThreshold;           // lanes queued before exits (top / surface) are served
  int jumpThreshold;              // lanes queued before transitions of the layer-skipping walk are served
  // Termination guarantees.  The reference's walk marches by cell INDEX and drops a photon whose step
  // is not positive (opticalProperties.f95:1719-1722, counted in nBad, monteCarloRadiativeTransfer.f95:562-563); the kernels
  // here find cells from positions in places (block walk, clear-air flight, layer skipping), keep face distances in float,
  // and a table may hand out a NaN (section 8) -- so every loop carries a bound of its own instead of an argument about
  // rounding: a photon that exceeds one is dropped and counted in *bad (fate 3), as the reference counts nBad.
  unsigned long long *bad;        // photons (and radiance rays) dropped by a bound (counted per lane, one atomic per lane that dropped any at the kernel's end)
  unsigned maxEvents;             // instrumented instantiations only (production: kMaxEvents ...): legs per photon
  unsigned maxEventsNaN;          // ... legs a photon at full weight may go on with a direction that is NaN (<= maxEvents)
  unsigned watchdog;              // ... trace_kernel: event phases of a wave in which no lane started a leg, took a photon or was refused one; block walk: block crossings of one leg
  float rayMaxLen;                // radiance: no view ray is longer than (zMax - z0) / min |mu| (geometry)
  int legacyTies;                 // TEST ONLY (MCBRAT_TEST_LEGACY_TIES; instrumented instantiation only): bit 0 / 1 / 2 re-enable the block walk's tie handling from before the
                                  // fixes of soak seeds 168 / 71 / 763 (no clamp to the block left / move along a NaN / span bits outlive the fold)
  unsigned ldsBytes;              // dynamic LDS of the launch (-DMCBRAT_POISON fills it before the kernel initialises its part)
  // debug / measurement
  mcbrat_fate *fates;             // non-null: record per-photon fate (index = photon index)
  unsigned long long *counters;   // non-null: event counters
  double *traceBuf;               // DEBUG: per-collision records of one photon (12 doubles each)
  unsigned long long traceIndex;  // photon index to trace
  int traceCap;
};

}  // namespace mcbrat
