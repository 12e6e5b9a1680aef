// mcbrat_api.hip -- C ABI (include/mcbrat.h) over the gfx950 kernels.  Host plumbing only:
// device buffers, launches on one HIP stream per context, HIP-event timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcbrat_kernels.hip"
#include "mcbrat_blockwalk.hip"

using namespace mcbrat;

struct mcbrat_ctx {
  int device = 0;
  // Per-launch resources.  Synchronous mode uses lane 0 only; asynchronous mode (mcbrat_set_async) rotates
  // over kLanes so that the drain of one call's tracing kernel overlaps the next call's (a launch ends
  // with its longest photon history: DESIGN.md section 5, "launch tail").
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the tracing kernel (synchronous mode)
    hipEvent_t evDone = nullptr;              // after the finish kernels of the lane's latest call
    unsigned long long *dCounter = nullptr;
    long long *dSlabs = nullptr;
    size_t slabCapacity = 0;  // ELEMENTS allocated (not batches: the stride changes with nc, nDir and the grid)
    float *dColVals = nullptr, *dScalVals = nullptr;
    size_t colCapacity = 0, scalCapacity = 0;  // elements allocated
  };
  static constexpr int kLanes = 4;
  Lane lane[kLanes];
  int cur = 0, nextLane = 0;
  bool asyncOn = false;
  hipEvent_t lastDone = nullptr;   // evDone of the most recent call: the next finish chain waits for it
  hipEvent_t evExternal = nullptr; // recorded on a caller's stream by mcbrat_wait_stream
  bool externalPending = false;
  bool chainAfter = false;          // mcbrat_chain_after: this context's next finish chain follows another context's
  hipEvent_t chainSnapshot = nullptr; // ... namely what that context had enqueued when the chain was asked for: an event of THIS context,
                                    // recorded on the other's stream at that moment (no handle of the other context is kept: it may be
                                    // destroyed, or go on to other calls, before this context's next one)
  hipEvent_t evChain = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timing, eventPool;  // asynchronous mode: kernel brackets not yet read
  Lane &L() { return lane[cur]; }
  std::string err, dropText;
  int numCUs = 256;
  size_t ldsPerCU = 160 * 1024;
  // grid
  int nx = 0, ny = 0, nz = 0, nc = 0;
  bool haveGrid = false, haveOptics = false, haveSource = false;
  std::vector<double> xe, ye, ze;
  int xyRegular = 0, zRegular = 0;
  double dX = 0, dY = 0, dZ = 0;
  // device buffers
  double *dEdges = nullptr;
  float *dExt = nullptr, *dCum = nullptr, *dSsa = nullptr, *dRelArea = nullptr;
  uint16_t *dPfi = nullptr;
  // brick layout (see mcbrat_kernels.hip locate_cell)
  uint32_t *dBrickTable = nullptr;
  uint32_t *dRec = nullptr;    // packed collision record per cell (nc <= 2), see DevParams::rec
  float *dLayerExt = nullptr;  // [nz] extinction of a horizontally uniform layer, -1 otherwise
  int *dLayerRun = nullptr;    // [nz] runs of such layers (DevParams::layerRun)
  double *dLayerRunT = nullptr;  // [nz+1]
  // clear-air flight (DevParams::fly...)
  float *dExtWalk = nullptr, *dBgVal = nullptr;
  uint16_t *dFlyRange = nullptr;
  int flyNbx = 0, flyNby = 0;
  bool flyBuilt = false;
  double flyDepth = 0.0;         // vertical optical depth of the background through the layers a flight can cross
  double flightMaxDepth = 0.5;   // MCBRAT_FLIGHT_MAX_DEPTH: largest vertical optical depth of the background with which flights are used
  // block walk (mcbrat_blockwalk.hip)
  uint32_t *dBlockRec = nullptr;
  uint16_t *dBlockOf = nullptr;
  float *dBlockExt = nullptr;  // [nBlocks] extinction of each block
  float *dBlockSsa = nullptr, *dBlockCum = nullptr;  // [nc][nBlocks] optics per block, where uniform (blockOpticsUniform)
  uint16_t *dBlockPfi = nullptr;
  bool blockOpticsUniform = false;
  int nBlocks = 0;
  int crossThreshold = 8;      // MCBRAT_CROSS_THRESHOLD
  int jumpThreshold = 8;       // MCBRAT_JUMP_THRESHOLD
  float *dExtB = nullptr, *dCumB = nullptr, *dSsaB = nullptr, *dBgExt = nullptr, *dBgCum = nullptr, *dBgSsa = nullptr;
  uint16_t *dPfiB = nullptr, *dBgPfi = nullptr;
  int nbx = 0, nby = 0, nbz = 0;
  long long nStored = 0;
  double backgroundFraction = 0.0;  // share of bricks that are pure background
  bool bricksBuilt = false;
  int brickMode = 2;               // 0 dense, 1 bricks, 2 automatic
  float *dTables = nullptr;
  // surface description (specifyParameters(surfaceBDRF=)); surfNumX == 0: the domain's albedo
  int surfNumX = 0, surfNumY = 0;
  double *dSurfX = nullptr, *dSurfY = nullptr;
  float *dSurfRefl = nullptr;
  double *dVoxelCDF = nullptr;
  unsigned long long *dEventCounters = nullptr;
  float *dLast = nullptr;
  double *dMomentsOwned = nullptr, *dMoments = nullptr;
  // tables (host, per component)
  std::vector<std::vector<float>> tables;
  std::vector<int> tblNSteps, tblNEntries;
  std::vector<int> maxPfi;  // largest 0-based entry referenced per component
  bool tablesDirty = true;
  int tblTotalFloats = 0;
  int tblOffset[MCBRAT_MAX_COMPONENTS] = {0};
  // radiance by local estimation
  int nDir = 0;
  std::vector<float> dirData;  // [nDir][8], see DevParams
  float *dDirData = nullptr, *dFwd = nullptr, *dFwdOrig = nullptr;
  std::vector<std::vector<float>> fwd, fwdOrig;  // per component [nEntries][nAngles]
  std::vector<int> fwdNAngles, fwdNEntries;
  int fwdOffset[MCBRAT_MAX_COMPONENTS] = {0};
  bool fwdDirty = true;
  int useHybrid = 0, numOrdersOrig = 0, useRRIntensity = 0, limitContrib = 0;
  float zetaMin = 0.3f, maxContrib = FLT_MAX;
  // parameters
  float albedo = 0.f;
  int useRR = 1;
  float lwFlag = -1.f;
  int srcKind = 0;
  float dir0[3] = {0, 0, -1};
  double zLaunch = 0;
  int izLaunch = 0;
  double fracAtms = 0;
  // tuning / measurement
  int blocksPerCU = 0;  // 0: from the occupancy query
  int eventThreshold = 16;     // measured optimum 16 (step cloud) .. 32 (128x128x64); see DESIGN.md
  int launchThreshold = 8, surfaceThreshold = 12;  // (exits -- top and surface -- queue together: 12-16 measured best)
  bool surfaceThresholdSet = false;  // by the caller: the block walk otherwise uses its own measured default
  bool launchThresholdSet = false;   // by the caller: the thermal source otherwise launches 32 lanes at a time (below)
  bool autoTune = true;        // pick eventThreshold by timing short trial launches (once per domain/source)
  bool tuned = false;
  int maxBatchesInFlight = 0;  // 0: bounded by memory
  int privMode = 1;            // 1: LDS-private tallies when the slab fits, 0: always global atomics
  int wideDefault = 1;         // MCBRAT_WIDE
  int wideMode = 1;            // 1: a tally slab too large for the shared plan may take a compute unit's whole LDS (one workgroup of 1024 lanes per CU;
                               //    MCBRAT_WIDE=0 / set_tuning(privateTallies = 3) turn it off: such domains then tally with global atomics)
  int blockSize = 0;           // 0: chosen by plan_launch
  unsigned long long tuneTrialPhotons = 1ull << 26;  // MCBRAT_TUNE_PHOTONS
  int regularWalk = 1;         // equally spaced grids: incremental face distances (MCBRAT_REGULAR_WALK=0 turns it off)
  int gridLdsMode = 1;         // stage the optical grid in LDS when it fits (private-tally mode)
  int rayShort = 0, rayPassIters = 0, rayPassAt = 0;  // 0: chosen in launch_trace_b (MCBRAT_RAY_SHORT, MCBRAT_RAY_PASS_ITERS, MCBRAT_RAY_PASS_AT)
  int rayDefer = 1;            // radiance: long rays are put aside and finished in dense passes (MCBRAT_RAY_DEFER=0: inside their event phase)
  int blockWalk = 1;           // LDS-resident grids: blocks of cells with one extinction value are crossed in one step (MCBRAT_BLOCK_WALK=0 / mcbrat_set_walk_options)
  int layerSkip = 1;           // layers of one extinction value: cross z faces only; clear-air flight outside the brick columns' cloud
                               // ranges (MCBRAT_LAYER_SKIP / mcbrat_set_walk_options: 0 off, 1 both, 2 the layers only)
  // loop bounds of the kernels (DevParams::bad ...) and the count of what they dropped
  unsigned long long *dBad = nullptr;   // device: photons / rays dropped since the context was created
  unsigned long long *hBad = nullptr;   // pinned host copy, written by the last finish kernel of every call (hBadDev: the same word as the device sees it)
  unsigned long long *hBadDev = nullptr;
  unsigned maxEvents = 1u << 24, maxEventsNaN = 1u << 20, watchdog = 1u << 20;  // MCBRAT_MAX_EVENTS, MCBRAT_MAX_EVENTS_NAN, MCBRAT_WATCHDOG
  int legacyTies = 0;
  float testRayMaxLen = 0.0f;  // TEST ONLY (MCBRAT_TEST_RAY_MAX_LEN, km): replaces the geometric bound on a view ray's length                   // MCBRAT_TEST_LEGACY_TIES (tests only)
  // mcbrat_frequency_distribution's device buffers, kept and grown with numLambda
  double *dFreqCdf = nullptr;
  unsigned long long *dFreqCounts = nullptr;
  int freqCapacity = 0;
  bool countersOn = false;
  float lastTraceMs = 0.f;
  mcbrat_counters lastCounters{};
  bool haveLast = false;
};

namespace {

int fail(mcbrat_ctx *c, const std::string &msg) {
  if (c) c->err = msg;
  return 1;
}
#define HIP_OK(c, call)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) return fail(c, std::string(#call) + ": " + hipGetErrorString(e_));              \
  } while (0)

// Every device allocation of the library.  -DMCBRAT_POISON (audit build): fresh memory is filled with 0xff, so that a
// kernel that reads what nothing wrote reads NaNs / huge indices instead of whatever an earlier context left there.
hipError_t dev_malloc(void **ptr, size_t bytes) {
  hipError_t e = hipMalloc(ptr, bytes);
#ifdef MCBRAT_POISON
  if (e == hipSuccess) e = hipMemset(*ptr, 0xff, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
#endif
  return e;
}

int init_lane(mcbrat_ctx *c, int i) {
  mcbrat_ctx::Lane &L = c->lane[i];
  if (L.stream) return 0;
  HIP_OK(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
  HIP_OK(c, hipEventCreate(&L.ev0));
  HIP_OK(c, hipEventCreate(&L.ev1));
  HIP_OK(c, hipEventCreateWithFlags(&L.evDone, hipEventDisableTiming));
  HIP_OK(c, dev_malloc((void **)&L.dCounter, sizeof(unsigned long long)));
  return 0;
}

// Waits for everything the context has enqueued and reads the kernel brackets of asynchronous calls.
int sync_all(mcbrat_ctx *c) {
  for (int i = 0; i < mcbrat_ctx::kLanes; ++i)
    if (c->lane[i].stream) HIP_OK(c, hipStreamSynchronize(c->lane[i].stream));
  if (!c->timing.empty()) {
    float total = 0.f;
    for (auto &pr : c->timing) {
      float ms = 0.f;
      HIP_OK(c, hipEventElapsedTime(&ms, pr.first, pr.second));
      total += ms;
      c->eventPool.push_back(pr);
    }
    c->timing.clear();
    c->lastTraceMs = total;
  }
  if (c->hBad) c->lastCounters.badPhotons = (int64_t)*c->hBad;
  return 0;
}

int event_pair(mcbrat_ctx *c, std::pair<hipEvent_t, hipEvent_t> &pr) {
  if (!c->eventPool.empty()) { pr = c->eventPool.back(); c->eventPool.pop_back(); return 0; }
  HIP_OK(c, hipEventCreate(&pr.first));
  HIP_OK(c, hipEventCreate(&pr.second));
  return 0;
}

double spacing_d(double x) {
  if (x == 0.0) return DBL_MIN;
  int e;
  std::frexp(std::fabs(x), &e);
  const double s = std::ldexp(1.0, e - 53);
  return s < DBL_MIN ? DBL_MIN : s;
}

template <typename T>
int upload(mcbrat_ctx *c, T **dst, const T *src, size_t n) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  HIP_OK(c, dev_malloc((void **)dst, sizeof(T) * std::max<size_t>(n, 1)));
  if (n) HIP_OK(c, hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
  return 0;
}

long long moments_len(const mcbrat_ctx *c) {
  const long long ncol = (long long)c->nx * c->ny;
  return 3 + 3 * ncol + c->nz + ncol * c->nz + (long long)c->nDir * ncol;
}

int ensure_moments(mcbrat_ctx *c) {
  if (c->dMoments) return 0;
  const size_t n = 8 + 2 * (size_t)moments_len(c);
  HIP_OK(c, dev_malloc((void **)&c->dMomentsOwned, sizeof(double) * n));
  // (on the stream the finish kernels run on: hipMemset on the null stream is asynchronous for device memory and the
  // context's streams are non-blocking, so nothing would order a null-stream memset before the first finish kernels)
  if (init_lane(c, c->cur)) return 1;
  HIP_OK(c, hipMemsetAsync(c->dMomentsOwned, 0, sizeof(double) * n, c->L().stream));
  HIP_OK(c, hipStreamSynchronize(c->L().stream));
  c->dMoments = c->dMomentsOwned;
  return 0;
}

int sync_tables(mcbrat_ctx *c) {
  if (!c->tablesDirty) return 0;
  if (sync_all(c)) return 1;
  std::vector<float> all;
  for (int k = 0; k < c->nc; ++k) {
    if ((int)c->tables.size() <= k || c->tables[k].empty())
      return fail(c, "computeRadiativeTransfer: no inverse phase function table for component " + std::to_string(k + 1));
    if (c->maxPfi[k] >= c->tblNEntries[k])
      return fail(c, "computeRadiativeTransfer: phaseFunctionIndex exceeds table entries for component " + std::to_string(k + 1));
    c->tblOffset[k] = (int)all.size();
    all.insert(all.end(), c->tables[k].begin(), c->tables[k].end());
  }
  c->tblTotalFloats = (int)all.size();
  if (upload(c, &c->dTables, all.data(), all.size())) return 1;
  c->tablesDirty = false;
  return 0;
}

// forward tables for radiance, uploaded before the first launch that needs them
int sync_forward_tables(mcbrat_ctx *c) {
  if (c->nDir == 0 || !c->fwdDirty) return 0;
  if (sync_all(c)) return 1;
  std::vector<float> all, allOrig;
  for (int k = 0; k < c->nc; ++k) {
    if ((int)c->fwd.size() <= k || c->fwd[k].empty())
      return fail(c, "tabulateForwardPhaseFunctions: no forward phase function table for component " + std::to_string(k + 1));
    if (c->maxPfi[k] >= c->fwdNEntries[k])
      return fail(c, "tabulateForwardPhaseFunctions: phaseFunctionIndex exceeds table entries for component " + std::to_string(k + 1));
    c->fwdOffset[k] = (int)all.size();
    all.insert(all.end(), c->fwd[k].begin(), c->fwd[k].end());
    allOrig.insert(allOrig.end(), c->fwdOrig[k].begin(), c->fwdOrig[k].end());
  }
  if (upload(c, &c->dFwd, all.data(), all.size()) || upload(c, &c->dFwdOrig, allOrig.data(), allOrig.size())) return 1;
  if (upload(c, &c->dDirData, c->dirData.data(), c->dirData.size())) return 1;
  c->fwdDirty = false;
  return 0;
}

// Brick layout of the optical grids: 4x4x4 bricks; a brick is "background" when each of its cells
// carries exactly its layer's most common record (extinction + every component's cumExt/ssa/phase
// index).  Background bricks are not stored.  Lossless: the kernel reads the same floats either way.
int build_bricks(mcbrat_ctx *c, const std::vector<float> &e, const std::vector<float> &cu, const std::vector<float> &s,
                 const std::vector<uint16_t> &pf, int nc) {
  const int nx = c->nx, ny = c->ny, nz = c->nz;
  const size_t ncol = (size_t)nx * ny, nvox = ncol * nz;
  std::vector<float> bgExt(nz), bgCum((size_t)nc * nz), bgSsa((size_t)nc * nz), layerExt(nz);
  std::vector<uint16_t> bgPfi((size_t)nc * nz);
  // the layers' background extinction, the brick columns' cloud ranges, the walk's marked copy of the extinction and the
  // vertical optical depth of the background (layer-skipping walk, clear-air flight): mcbrat_flight_tables, mcbrat_host.cpp
  const bool bricks4 = nx % 4 == 0 && ny % 4 == 0 && nz >= 2 && nz <= 255;
  std::vector<uint16_t> range(bricks4 ? (size_t)(nx / 4) * (ny / 4) : 0);
  std::vector<float> walk(bricks4 ? nvox : 0);
  std::vector<double> runT(nz + 1, 0.0);
  int32_t flights = 0;
  if (mcbrat_flight_tables(nx, ny, nz, e.data(), c->ze.data(), bgExt.data(), bricks4 ? range.data() : nullptr,
                           bricks4 ? walk.data() : nullptr, runT.data(), &flights))
    return fail(c, "getOpticalPropertiesByComponent: could not build the tables of the walk.");
  for (int k = 0; k < nz; ++k) {
    const float best = bgExt[k];
    bool uniform = true;
    for (size_t v = ncol * k; v < ncol * (k + 1) && uniform; ++v) uniform = e[v] == best;
    layerExt[k] = uniform ? best : -1.0f;  // a layer of one extinction value needs no gather in the walk
    size_t rep = ncol * k;
    while (e[rep] != best) ++rep;  // a representative background cell of this layer
    for (int q = 0; q < nc; ++q) {
      bgCum[(size_t)q * nz + k] = cu[(size_t)q * nvox + rep];
      bgSsa[(size_t)q * nz + k] = s[(size_t)q * nvox + rep];
      bgPfi[(size_t)q * nz + k] = pf[(size_t)q * nvox + rep];
    }
  }
  if (upload(c, &c->dLayerExt, layerExt.data(), layerExt.size())) return 1;
  c->flyBuilt = false;
  c->flyNbx = c->flyNby = 0;
  if (bricks4) {
    if (upload(c, &c->dFlyRange, range.data(), range.size()) || upload(c, &c->dExtWalk, walk.data(), walk.size()) ||
        upload(c, &c->dBgVal, bgExt.data(), bgExt.size()))
      return 1;
    c->flyNbx = nx / 4; c->flyNby = ny / 4;
    // A flight is only granted to a lane the background cannot stop before the domain boundary.  In a haze (vertical
    // optical depth of the background not small against 1) most requests are refused, and asking costs a turn in the
    // queue: measured -10 % on the 128x128x64 field in a haze of optical depth 2.3; Rayleigh air (0.02) is what the
    // flight is for.  In between (same field, 5 / 10 / 20 / 40 times the Rayleigh extinction = 0.12 / 0.23 / 0.46 / 0.92):
    // +13 / +8 / +2 / -5 %, so the limit is 0.5 (flight_wanted).
    c->flyBuilt = flights != 0;
    c->flyDepth = runT[nz];
  }
  {
    // runs of consecutive one-extinction layers (layer-skipping walk): for layer k the face where its run ends
    // upwards / downwards.  (runT: inside a run of one-extinction layers its differences are the run's own; a layer no
    // flight can cross adds nothing to it -- differences across such a layer are only used as the bound on what the
    // background can take from a lane before its flight ends.)
    std::vector<int> run(nz, 0);
    for (int k = 0; k < nz; ++k) {
      int up = k + 1, down = k;
      while (up < nz && layerExt[up] >= 0.f) ++up;
      while (down > 0 && layerExt[down - 1] >= 0.f) --down;
      run[k] = (up << 16) | down;
    }
    if (upload(c, &c->dLayerRun, run.data(), run.size()) || upload(c, &c->dLayerRunT, runT.data(), runT.size())) return 1;
  }
  auto isBackground = [&](size_t v, int k) {
    if (e[v] != bgExt[k]) return false;
    for (int q = 0; q < nc; ++q)
      if (cu[(size_t)q * nvox + v] != bgCum[(size_t)q * nz + k] || s[(size_t)q * nvox + v] != bgSsa[(size_t)q * nz + k] ||
          pf[(size_t)q * nvox + v] != bgPfi[(size_t)q * nz + k])
        return false;
    return true;
  };
  const int nbx = (nx + 3) / 4, nby = (ny + 3) / 4, nbz = (nz + 3) / 4;
  std::vector<uint32_t> table((size_t)nbx * nby * nbz, 0xffffffffu);
  size_t stored = 0;
  for (int bz = 0; bz < nbz; ++bz)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) {
        bool bg = true;
        for (int k = bz * 4; k < std::min(nz, bz * 4 + 4) && bg; ++k)
          for (int j = by * 4; j < std::min(ny, by * 4 + 4) && bg; ++j)
            for (int i = bx * 4; i < std::min(nx, bx * 4 + 4) && bg; ++i)
              bg = isBackground((size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k), k);
        if (!bg) table[(size_t)bx + (size_t)nbx * ((size_t)by + (size_t)nby * bz)] = (uint32_t)(64 * stored++);
      }
  const size_t nS = std::max<size_t>(64 * stored, 64);
  std::vector<float> eB(nS, 0.f), cuB(nS * nc, 0.f), sB(nS * nc, 0.f);
  std::vector<uint16_t> pfB(nS * nc, 0);
  for (int bz = 0; bz < nbz; ++bz)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) {
        const uint32_t base = table[(size_t)bx + (size_t)nbx * ((size_t)by + (size_t)nby * bz)];
        if (base == 0xffffffffu) continue;
        for (int lk = 0; lk < 4; ++lk)
          for (int lj = 0; lj < 4; ++lj)
            for (int li = 0; li < 4; ++li) {
              const int i = bx * 4 + li, j = by * 4 + lj, k = bz * 4 + lk;
              if (i >= nx || j >= ny || k >= nz) continue;
              const size_t v = (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k), o = base + li + 4 * (lj + 4 * lk);
              eB[o] = e[v];
              for (int q = 0; q < nc; ++q) {
                cuB[(size_t)q * nS + o] = cu[(size_t)q * nvox + v];
                sB[(size_t)q * nS + o] = s[(size_t)q * nvox + v];
                pfB[(size_t)q * nS + o] = pf[(size_t)q * nvox + v];
              }
            }
      }
  c->nbx = nbx; c->nby = nby; c->nbz = nbz;
  c->nStored = (long long)nS;
  c->backgroundFraction = 1.0 - (double)stored / (double)table.size();
  if (upload(c, &c->dBrickTable, table.data(), table.size()) || upload(c, &c->dExtB, eB.data(), eB.size()) ||
      upload(c, &c->dCumB, cuB.data(), cuB.size()) || upload(c, &c->dSsaB, sB.data(), sB.size()) ||
      upload(c, &c->dPfiB, pfB.data(), pfB.size()) || upload(c, &c->dBgExt, bgExt.data(), bgExt.size()) ||
      upload(c, &c->dBgCum, bgCum.data(), bgCum.size()) || upload(c, &c->dBgSsa, bgSsa.data(), bgSsa.size()) ||
      upload(c, &c->dBgPfi, bgPfi.data(), bgPfi.size()))
    return 1;
  c->bricksBuilt = true;
  return 0;
}

// Block walk (mcbrat_blockwalk.hip): the grid cut into axis-aligned blocks of cells that carry one extinction value
// (mcbrat_block_decomposition, mcbrat_host.cpp), uploaded for the kernel.
int build_blocks(mcbrat_ctx *c, const std::vector<float> &e, const std::vector<float> &cu, const std::vector<float> &ssa,
                 const std::vector<uint16_t> &pf, int nc) {
  const int nx = c->nx, ny = c->ny, nz = c->nz;
  const size_t nvox = (size_t)nx * ny * nz;
  c->nBlocks = 0;
  if (c->dBlockRec) { (void)hipFree(c->dBlockRec); c->dBlockRec = nullptr; }
  if (c->dBlockOf) { (void)hipFree(c->dBlockOf); c->dBlockOf = nullptr; }
  if (c->dBlockExt) { (void)hipFree(c->dBlockExt); c->dBlockExt = nullptr; }
  if (c->dBlockSsa) { (void)hipFree(c->dBlockSsa); c->dBlockSsa = nullptr; }
  if (c->dBlockCum) { (void)hipFree(c->dBlockCum); c->dBlockCum = nullptr; }
  if (c->dBlockPfi) { (void)hipFree(c->dBlockPfi); c->dBlockPfi = nullptr; }
  c->blockOpticsUniform = false;
  // only grids that can live in LDS are walked this way (plan_launch decides); bounds are packed in 16 bits
  if (nvox > 65536 || nx > 65535 || ny > 65535 || nz > 65535) return 0;
  std::vector<uint16_t> of(nvox);
  std::vector<uint32_t> rec(4 * nvox);
  int32_t nb = 0;
  if (mcbrat_block_decomposition(nx, ny, nz, e.data(), of.data(), rec.data(), &nb) != 0) return 0;  // (more than 65535 blocks: face-by-face walk)
  std::vector<float> blockExt((size_t)nb, 0.0f);
  for (size_t v = 0; v < nvox; ++v) blockExt[of[v]] = e[v];  // (one value per block by construction)
  if (upload(c, &c->dBlockRec, rec.data(), (size_t)4 * nb) || upload(c, &c->dBlockOf, of.data(), of.size()) ||
      upload(c, &c->dBlockExt, blockExt.data(), blockExt.size())) return 1;
  // Blocks are cut by extinction; where every block is uniform in the other optics too (a homogeneous medium, slabs: the
  // broadband and I3RC test domains) a collision needs no per-cell record at all -- optics per block, in LDS
  std::vector<float> bSsa((size_t)nc * nb), bCum((size_t)nc * nb);
  std::vector<uint16_t> bPfi((size_t)nc * nb);
  std::vector<char> seen((size_t)nb, 0);
  bool uniform = true;
  for (size_t v = 0; v < nvox && uniform; ++v) {
    const size_t b = of[v];
    for (int k = 0; k < nc && uniform; ++k) {
      const size_t i = (size_t)k * nvox + v, j = (size_t)k * nb + b;
      if (!seen[b]) { bSsa[j] = ssa[i]; bCum[j] = cu[i]; bPfi[j] = pf[i]; }
      else uniform = std::memcmp(&bSsa[j], &ssa[i], 4) == 0 && std::memcmp(&bCum[j], &cu[i], 4) == 0 && bPfi[j] == pf[i];
    }
    seen[b] = 1;
  }
  if (uniform) {
    if (upload(c, &c->dBlockSsa, bSsa.data(), bSsa.size()) || upload(c, &c->dBlockCum, bCum.data(), bCum.size()) ||
        upload(c, &c->dBlockPfi, bPfi.data(), bPfi.size())) return 1;
    c->blockOpticsUniform = true;
  }
  c->nBlocks = nb;
  return 0;
}

// Bricks are for grids that no longer fit the cache hierarchy (32 MiB of L2 + 256 MiB Infinity
// Cache) and are mostly background.  Measured on MI355X: at 128x128x64 (4 MiB of extinction) the
// dense grid is as fast or 1-3 % faster, so the automatic rule only switches
// for grids of 64 MiB and more.
bool use_bricks(const mcbrat_ctx *c) {
  if (!c->bricksBuilt || c->brickMode == 0 || c->nDir > 0) return false;  // (radiance rays read the dense grid)
  if (c->brickMode == 1) return true;
  const size_t nvox = (size_t)c->nx * c->ny * c->nz;
  // automatic: the dense layout is the faster one wherever measured (128x128x64: equal; 512x512x128: 83 vs 105 ms per
  // 2e7 photons), so bricks are a memory saver for the very largest grids only (dense optics: 24 B per cell and component)
  return nvox * sizeof(float) >= ((size_t)2 << 30) && c->backgroundFraction >= 0.5;
}

void fill_params(mcbrat_ctx *c, DevParams &p) {
  std::memset(&p, 0, sizeof(p));
  p.nx = c->nx; p.ny = c->ny; p.nz = c->nz; p.nc = c->nc;
  p.xyRegular = c->xyRegular; p.zRegular = c->zRegular;
  p.x0 = c->xe.front(); p.xMax = c->xe.back();
  p.y0 = c->ye.front(); p.yMax = c->ye.back();
  p.z0 = c->ze.front(); p.zMax = c->ze.back();
  p.Lx = p.xMax - p.x0; p.Ly = p.yMax - p.y0;
  p.zSurf = p.z0 + spacing_d(p.z0);
  p.invDX = c->xyRegular ? 1.0 / c->dX : 0.0;
  p.invDY = c->xyRegular ? 1.0 / c->dY : 0.0;
  // (only where the reference itself calls the axis regular -- its test uses a float spacing, new_Integrator
  // :163-181.  Float steps accumulate rounding over a leg: on 0.03 km cells, which that test calls irregular, long
  // clear-air legs lost 0.7 % of the per-photon identity with the oracle for a 2 % gain, so those keep the table.)
  p.xyRegularWalk = (c->xyRegular && c->regularWalk) ? 1 : 0;
  p.zRegularWalk = (c->zRegular && c->regularWalk) ? 1 : 0;
  p.dXf = (float)((p.xMax - p.x0) / c->nx); p.dYf = (float)((p.yMax - p.y0) / c->ny); p.dZf = (float)((p.zMax - p.z0) / c->nz);
  p.layerSkip = c->layerSkip ? 1 : 0;
  p.layerRun = c->dLayerRun; p.layerRunT = c->dLayerRunT;
  p.fly = 0; p.flyNbx = p.flyNby = 0;  // (switched on by launch_trace where the plan allows it)
  p.flyRange = c->dFlyRange; p.bgVal = c->dBgVal;
  p.invLx = 1.0 / p.Lx; p.invLy = 1.0 / p.Ly;
  p.invCellX = (double)c->nx / p.Lx; p.invCellY = (double)c->ny / p.Ly;
  {
    bool u = true;  // equally spaced to a millionth of a cell: the cell of a position is guessed by division
    for (int i = 0; i <= c->nx && u; ++i) u = std::fabs((c->xe[i] - p.x0) * p.invCellX - i) <= 1e-6;
    for (int i = 0; i <= c->ny && u; ++i) u = std::fabs((c->ye[i] - p.y0) * p.invCellY - i) <= 1e-6;
    p.xyNearUniform = u ? 1 : 0;
    bool uz = true;
    const double invCellZ = (double)c->nz / (p.zMax - p.z0);
    for (int i = 0; i <= c->nz && uz; ++i) uz = std::fabs((c->ze[i] - p.z0) * invCellZ - i) <= 1e-6;
    p.zNearUniform = uz ? 1 : 0;
  }
  p.edges = c->dEdges;
  if (use_bricks(c)) {
    p.ext = c->dExtB; p.cum = c->dCumB; p.ssa = c->dSsaB; p.pfi = c->dPfiB;
    p.extWalk = p.ext;
    p.brickTable = c->dBrickTable; p.nbx = c->nbx; p.nby = c->nby; p.nbz = c->nbz; p.nStored = c->nStored;
    p.bgExt = c->dBgExt; p.bgCum = c->dBgCum; p.bgSsa = c->dBgSsa; p.bgPfi = c->dBgPfi;
  } else {
    p.ext = c->dExt; p.cum = c->dCum; p.ssa = c->dSsa; p.pfi = c->dPfi;
    p.extWalk = p.ext;
    p.bgExt = c->dLayerExt;  // dense layout: the per-layer slot holds the uniform-layer shortcut
    p.rec = reinterpret_cast<const uint4 *>(c->dRec);
  }
  p.albedo = c->albedo;
  p.surfNumX = c->surfNumX; p.surfNumY = c->surfNumY; p.surfX = c->dSurfX; p.surfY = c->dSurfY; p.surfRefl = c->dSurfRefl;
  p.tables = c->dTables;
  for (int k = 0; k < c->nc; ++k) { p.tblOffset[k] = c->tblOffset[k]; p.tblNSteps[k] = c->tblNSteps[k]; p.tblInvN[k] = 1.0f / (float)c->tblNSteps[k]; }
  p.tblTotalFloats = c->tblTotalFloats;
  p.useRR = c->useRR;
  p.lwFlag = c->lwFlag > 0.f ? 1 : 0;
  p.srcKind = c->srcKind;
  p.dir0[0] = c->dir0[0]; p.dir0[1] = c->dir0[1]; p.dir0[2] = c->dir0[2];
  p.zLaunch = c->zLaunch; p.izLaunch = c->izLaunch;
  p.voxelCDF = c->dVoxelCDF; p.fracAtms = c->fracAtms;
  p.nDir = c->nDir;
  p.dirData = c->dDirData; p.fwdTables = c->dFwd; p.fwdOrig = c->dFwdOrig;
  for (int k = 0; k < c->nc && c->nDir > 0; ++k) { p.fwdOffset[k] = c->fwdOffset[k]; p.fwdNAngles[k] = c->fwdNAngles[k]; }
  p.useHybrid = c->useHybrid; p.numOrdersOrig = c->numOrdersOrig; p.useRRIntensity = c->useRRIntensity; p.zetaMin = c->zetaMin;
  p.limitContrib = c->limitContrib; p.maxContrib = c->maxContrib;
  p.nBlocks = c->nBlocks; p.blockRec = reinterpret_cast<const uint4 *>(c->dBlockRec); p.blockOf = c->dBlockOf; p.blockExt = c->dBlockExt;
  p.blockSsa = c->dBlockSsa; p.blockCum = c->dBlockCum; p.blockPfi = c->dBlockPfi;
  p.crossThreshold = std::max(1, std::min(64, c->crossThreshold));
  p.jumpThreshold = std::max(1, std::min(64, c->jumpThreshold));
  p.bad = c->dBad;
  p.maxEvents = c->maxEvents; p.maxEventsNaN = std::min(c->maxEventsNaN, c->maxEvents); p.watchdog = c->watchdog;  // (the kernels test the smaller one first)
  p.legacyTies = c->legacyTies;
  {
    // no view ray is longer than the domain's height over the smallest |mu| of the views (and a hair for rounding)
    float muMin = 1.0f;
    for (int i = 0; i < c->nDir; ++i) muMin = std::min(muMin, std::fabs(c->dirData[(size_t)8 * i + 2]));
    p.rayMaxLen = (float)((p.zMax - p.z0) / (double)std::max(muMin, FLT_MIN)) * 1.0001f + 1e-6f;
    if (c->testRayMaxLen > 0.0f) p.rayMaxLen = c->testRayMaxLen;  // TEST ONLY (MCBRAT_TEST_RAY_MAX_LEN): drive the bound on purpose
  }
  p.counter = c->L().dCounter;
  p.eventThreshold = std::max(1, std::min(64, c->eventThreshold));
  // The thermal source's launch is several hundred instructions (three searches of the emission CDF, double divisions,
  // sine and cosine in full precision) for photons that live two or three legs: served eight lanes at a time it was half
  // of config 4's kernel time.  Dead lanes wait for company instead: 32 (config 4: 3.8 -> 5.1e9 photons/s; 24-40 are level).
  p.launchThreshold = std::max(1, std::min(64, c->launchThresholdSet ? c->launchThreshold : (c->srcKind != 0 ? 32 : c->launchThreshold)));
  p.surfaceThreshold = std::max(1, std::min(64, c->surfaceThreshold));
}

constexpr size_t kLdsBudget = 64 * 1024;      // dynamic LDS of a workgroup that shares its compute unit with another one (two workgroups of 768 lanes, four of 256)
constexpr size_t kTableLdsLimit = 48 * 1024;   // tables above this stay in L2 (shared plans)
constexpr size_t kPrivSlabLimit = 32 * 1024;   // private tally slab above this -> the wide plan, else global atomics
constexpr size_t kStaticLds = 6 * 1024;        // room kept for the kernels' static __shared__ arrays (descriptors; trace_kernel's s_flyTC is 4 KB at 1024 lanes)

// LDS of the per-layer tables: extinction (float), run (int), cumulative optical depth (double, nz + 1 padded to even);
// clear-air flight: background extinction (float), the brick columns' cloud ranges as heights (2 floats) and layers (u16)
size_t per_layer_lds(int nz, int flyCols = 0) {
  return 2 * sizeof(float) * (size_t)((nz + 3) & ~3) + sizeof(double) * (size_t)((nz + 2) & ~1) +
         (flyCols ? sizeof(float) * (size_t)((nz + 3) & ~3) + 2 * sizeof(float) * (size_t)flyCols + sizeof(uint16_t) * (size_t)((flyCols + 3) & ~3) : 0);
}

// clear-air flight: asked for (layerSkip 1: where the background is thin enough for it to pay; 3: regardless) and possible
bool flight_wanted(const mcbrat_ctx *c) {
  return c->flyBuilt && (c->layerSkip == 3 || (c->layerSkip == 1 && c->flyDepth <= c->flightMaxDepth));
}

struct LaunchPlan {
  bool tblLds, priv, brick, gridLds;
  bool fly;  // clear-air flight: dense grid in global memory, flux run, tables built, room in LDS
  // Wide plan (MI355X: 160 KB of LDS per compute unit): a tally slab too large to share a compute unit's LDS with a second
  // workgroup -- the broadband 20 x 20 x 20 domain's is 70 KB -- gets the compute unit to itself: ONE workgroup of 1024 lanes
  // (16 waves, 4 per SIMD) with up to the whole LDS, tallies in LDS as for the small domains instead of one memory-side
  // atomic per deposit (config 4 waited 57 % of its wave time behind them).  Flux runs only.
  bool wide;
  bool blockLite;  // wide plan, block walk without the per-cell optics in LDS (trace_block_kernel<..., OPT = 1 or 2>)
  int optics;      // block walk: where a collision finds its cell's optics: 0 per cell in LDS, 1 per cell in global memory, 2 per block in LDS
  bool cdfTop;     // block walk, thermal source: level and row sums of the emission CDF staged in LDS
  int block;
  size_t lds;
};

bool blocks_worth_it(const mcbrat_ctx *c) {
  if (!c->blockWalk || c->nDir > 0 || c->nBlocks <= 0) return false;
  const size_t nvox = (size_t)c->nx * c->ny * c->nz;
  return (size_t)c->nBlocks * 4 <= nvox || c->blockWalk == 2;  // (2: forced, for tests of heterogeneous media)
}

LaunchPlan plan_launch(const mcbrat_ctx *c, size_t slabStride) {
  LaunchPlan L;
  L.wide = false; L.blockLite = false; L.optics = 0; L.cdfTop = false;
  const size_t edges = sizeof(double) * (size_t)(c->nx + c->ny + c->nz + 3);
  const size_t tbl = sizeof(float) * (size_t)c->tblTotalFloats;
  const size_t slab = sizeof(long long) * slabStride + 16;
  L.priv = c->privMode != 0 && slab <= kPrivSlabLimit;
  L.brick = use_bricks(c);
  const size_t bg = per_layer_lds(c->nz);  // per-layer extinction (background / one-extinction layers) and the runs of such layers
  L.tblLds = tbl <= kTableLdsLimit && edges + bg + tbl + (L.priv ? slab : 0) <= kLdsBudget;
  if (L.priv && edges + bg + slab + (L.tblLds ? tbl : 0) > kLdsBudget) L.priv = false;
  // small domains: the optical grid itself (ext + per-component cum, ssa, phase index) goes to LDS when it fits
  const size_t nvox = (size_t)c->nx * c->ny * c->nz;
  const size_t grid = nvox * 4 + (size_t)c->nc * nvox * (4 + 4) + (((size_t)c->nc * nvox + 1) & ~(size_t)1) * 2;
  L.gridLds = L.priv && c->gridLdsMode != 0 && edges + bg + slab + grid + (L.tblLds ? tbl : 0) <= kLdsBudget;
  L.lds = edges + bg + (L.priv ? slab : 0) + (L.gridLds ? grid : 0) + (L.tblLds ? tbl : 0);
  // the wide plan: the slab did not fit beside another workgroup, but it fits a compute unit
  const size_t cuLds = c->ldsPerCU > kStaticLds ? c->ldsPerCU - kStaticLds : 0;
  if ((!L.priv || c->wideMode == 2) && c->privMode != 0 && c->wideMode != 0 && c->nDir == 0 && !L.brick && c->blockSize == 0 && edges + bg + slab <= cuLds) {
    L.wide = true; L.priv = true; L.fly = false;
    size_t need = edges + bg + slab;
    L.tblLds = need + tbl <= cuLds;
    if (L.tblLds) need += tbl;
    L.gridLds = c->gridLdsMode == 1 && need + grid <= cuLds;
    if (L.gridLds) need += grid;
    L.lds = need;
    L.block = 1024;
    // block walk with the per-cell optics left in global memory: what a medium made of blocks whose grid does not fit still gets
    if (!L.gridLds && blocks_worth_it(c) && c->gridLdsMode != 0) {
      const int opt = (c->blockOpticsUniform && c->gridLdsMode != 2) ? 2 : 1;  // (gridLdsMode 2: per-cell optics in global memory asked for, tests)
      const BlockLds B = block_lds_layout(c->nx, c->ny, c->nz, c->nc, slabStride, c->nBlocks, (size_t)c->tblTotalFloats, opt);
      const BlockLds B0 = block_lds_layout(c->nx, c->ny, c->nz, c->nc, slabStride, c->nBlocks, 0, opt);
      if (B.total <= cuLds) { L.blockLite = true; L.tblLds = true; L.optics = opt; }
      else if (B0.total <= cuLds) { L.blockLite = true; L.tblLds = false; L.optics = opt; }
    }
    if (blocks_worth_it(c) && (L.gridLds || L.blockLite) && c->srcKind != 0)  // the emission CDF's level and row sums beside them, where they fit
      L.cdfTop = block_lds_layout(c->nx, c->ny, c->nz, c->nc, slabStride, c->nBlocks, L.tblLds ? (size_t)c->tblTotalFloats : 0, L.optics, true).total <= cuLds;
    return L;
  }
  const size_t flyLds = per_layer_lds(c->nz, c->flyNbx * c->flyNby) - bg;
  L.fly = flight_wanted(c) && !L.gridLds && !L.brick && c->nDir == 0 && L.lds + flyLds <= kLdsBudget;
  if (L.fly) L.lds += flyLds;
  L.block = c->blockSize > 0 ? c->blockSize : (L.gridLds ? 768 : ((L.tblLds || L.priv) && L.lds > 16 * 1024 ? 512 : 256));
  if (blocks_worth_it(c) && L.priv && L.gridLds && c->srcKind != 0)
    L.cdfTop = block_lds_layout(c->nx, c->ny, c->nz, c->nc, slabStride, c->nBlocks, L.tblLds ? (size_t)c->tblTotalFloats : 0, 0, true).total <= kLdsBudget;
  return L;
}

size_t plan_launch_lds(const mcbrat_ctx *c, const LaunchPlan &L) {
  return sizeof(double) * (size_t)(c->nx + c->ny + c->nz + 3) + per_layer_lds(c->nz, L.fly ? c->flyNbx * c->flyNby : 0) +
         (L.tblLds ? sizeof(float) * (size_t)c->tblTotalFloats : 0);
}

// Units per batch for the kernels whose workgroups keep one batch's tallies in LDS (a workgroup traces photons of one
// batch at a time).  A unit ends with a drain -- its lanes fall idle while its last histories finish, about 0.3 ms on
// the step cloud, the time of ~4000 photons of a workgroup's throughput -- and units are dealt out to `blocks` resident
// workgroups in rounds: the choice minimises rounds x (photons per unit + drain).  (The former rule, blocks / batches
// rounded down, left a fifth of the workgroup slots empty for 200 or 400 batches: 4.7-4.9 instead of 5.3e9 photons/s.)
inline unsigned long long units_per_batch(unsigned long long blocks, unsigned long long ppb, int nBatches, int blockSize) {
  const unsigned long long maxUpb = std::max<unsigned long long>(1, ppb / (unsigned long long)(blockSize * 8));
  const double drainPhotons = 4000.0;
  unsigned long long best = 1;
  double bestCost = 1e300;
  // (up to one unit per resident workgroup and a few rounds of them: a call of ONE large batch -- a wavelength's whole share
  // of a broadband run -- must still reach every compute unit; the cap was 64, a quarter of the wide plan's 256 workgroups)
  for (unsigned long long upb = 1; upb <= std::min<unsigned long long>(maxUpb, std::max<unsigned long long>(64, 4 * blocks)); ++upb) {
    const unsigned long long units = upb * (unsigned long long)nBatches;
    const unsigned long long rounds = (units + blocks - 1) / blocks;
    const double cost = (double)rounds * ((double)ppb / (double)upb + drainPhotons);
    if (cost < bestCost * (1.0 - 1e-9)) { bestCost = cost; best = upb; }
  }
  return best;
}

template <int BLOCK, bool TBL, int PRIV, bool BRICK, bool DBG, bool INTEN, bool EMIT, int SPEC = 0>
int launch_trace_e(mcbrat_ctx *c, DevParams &p, size_t lds, int nBatches) {
  if (lds + kStaticLds > c->ldsPerCU && lds > kLdsBudget) return fail(c, "computeRadiativeTransfer: the grid's edge and layer tables do not fit the LDS of a compute unit.");
  if (lds > kLdsBudget)  // (very tall grids: the per-layer tables alone can pass the default limit of a workgroup)
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(trace_kernel<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, EMIT, SPEC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int perCU = c->blocksPerCU;
  if (perCU <= 0) {
    HIP_OK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, trace_kernel<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, EMIT, SPEC>, BLOCK, lds));
    perCU = std::max(1, std::min(perCU, 8));
  }
  unsigned long long blocks = (unsigned long long)perCU * c->numCUs;
  if (PRIV) {
    // every workgroup traces photons of one batch at a time: cut each batch into units so that
    // there is about one unit per resident workgroup
    const unsigned long long upb = units_per_batch(blocks, p.ppb, nBatches, BLOCK);
    p.unitsPerBatch = upb;
    p.nUnits = upb * (unsigned long long)nBatches;
    blocks = std::min(blocks, p.nUnits);
  } else {
    blocks = std::min(blocks, (p.total + BLOCK - 1) / BLOCK);
  }
  const unsigned grid = (unsigned)std::max<unsigned long long>(1, blocks);
  p.ldsBytes = (unsigned)lds;
  hipLaunchKernelGGL((trace_kernel<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, EMIT, SPEC>), dim3(grid), dim3(BLOCK), lds, c->L().stream, p);
  HIP_OK(c, hipGetLastError());
  return 0;
}

template <int BLOCK, bool TBL, int PRIV, bool BRICK, bool DBG, bool INTEN = false>
int launch_trace_t(mcbrat_ctx *c, DevParams &p, size_t lds, int nBatches) {
  // (the source kind is a template parameter: the emission launch code costs the solar instantiations registers)
  if constexpr (BLOCK == 256 && PRIV == 0 && !BRICK && !DBG && !INTEN) {
    // the large flux runs: dense grid in global memory, collision records (nc <= 2), layer-skipping walk, albedo
    // surface -- with the walk's spacing flags decided at compile time too (SPEC, mcbrat_kernels.hip)
    // A solar run with roulette (every bench workload) also has its component count and the roulette decided at compile
    // time (SPEC bits 2-3 and 4); without roulette it runs the general kernel.
    if (p.rec != nullptr && p.layerSkip && p.fly && p.surfNumX == 0 && (p.xyRegularWalk || !p.zRegularWalk)) {
      const int walk = p.xyRegularWalk ? (p.zRegularWalk ? 2 : 3) : 1;
      if (c->srcKind != 0) {
        if (walk == 1) return launch_trace_e<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, true, 1>(c, p, lds, nBatches);
        if (walk == 2) return launch_trace_e<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, true, 2>(c, p, lds, nBatches);
        return launch_trace_e<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, true, 3>(c, p, lds, nBatches);
      }
      if (p.useRR && (p.nc == 1 || p.nc == 2)) {
        const int spec = walk | p.nc << 2 | 16;
        switch (spec) {
#define MCBRAT_SPEC_CASE(S) case S: return launch_trace_e<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, false, S>(c, p, lds, nBatches)
          MCBRAT_SPEC_CASE(1 | 1 << 2 | 16); MCBRAT_SPEC_CASE(2 | 1 << 2 | 16); MCBRAT_SPEC_CASE(3 | 1 << 2 | 16);
          MCBRAT_SPEC_CASE(1 | 2 << 2 | 16); MCBRAT_SPEC_CASE(2 | 2 << 2 | 16); MCBRAT_SPEC_CASE(3 | 2 << 2 | 16);
#undef MCBRAT_SPEC_CASE
        }
      }
    }
  }
  return c->srcKind == 0 ? launch_trace_e<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, false>(c, p, lds, nBatches)
                         : launch_trace_e<BLOCK, TBL, PRIV, BRICK, DBG, INTEN, true>(c, p, lds, nBatches);
}

template <int BLOCK, bool DBG>
int launch_trace_b(mcbrat_ctx *c, DevParams &p, const LaunchPlan &L, int nBatches) {
  // radiance runs: dense grids, no instrumentation
  if (c->nDir > 0) {
    if (DBG) return fail(c, "computeRadiativeTransfer: event counters / photon fates are not available together with intensity directions.");
    // the waves' buffers of unfinished long rays (80 B per ray) take what LDS is left at the residency the kernel is built
    // for: 4 workgroups of 256 lanes per CU on grids in global memory, 2 workgroups otherwise
    const size_t waves = BLOCK / 64, base = (L.lds + 15) & ~(size_t)15;
    const size_t budget = c->ldsPerCU / (L.priv ? 2 : (BLOCK == 256 ? 4 : 2));
    size_t cap = (c->rayDefer && budget > base + 64) ? std::min<size_t>(64, (budget - base - 64) / (waves * 80)) : 0;
    if (cap < 24 || c->nx + c->ny + c->nz + 3 > 0xffff || c->nDir > 0xffff) cap = 0;  // (a ray record packs edge-table indices and the direction in 16 bits)
    p.rayCap = (int)cap;
    // measured on the 128x128x64 cloud field, 4 directions: with roulette most rays end within a few cells and the long
    // ones are best served in short, dense passes; without it every ray runs to the boundary and long passes pay
    p.rayShort = c->rayShort > 0 ? c->rayShort : (c->useRRIntensity ? 4 : 8);
    p.rayPassIters = c->rayPassIters > 0 ? c->rayPassIters : (c->useRRIntensity ? 16 : 64);
    p.rayPassAt = std::min<int>(c->rayPassAt > 0 ? c->rayPassAt : (c->useRRIntensity ? 56 : 40), (int)cap);
    const size_t lds = base + waves * cap * 80;
    if (L.priv && L.gridLds) return L.tblLds ? launch_trace_t<BLOCK, true, 2, false, false, true>(c, p, lds, nBatches)
                                             : launch_trace_t<BLOCK, false, 2, false, false, true>(c, p, lds, nBatches);
    if (L.priv) return L.tblLds ? launch_trace_t<BLOCK, true, 1, false, false, true>(c, p, lds, nBatches)
                                : launch_trace_t<BLOCK, false, 1, false, false, true>(c, p, lds, nBatches);
    return L.tblLds ? launch_trace_t<BLOCK, true, 0, false, false, true>(c, p, lds, nBatches)
                    : launch_trace_t<BLOCK, false, 0, false, false, true>(c, p, lds, nBatches);
  }
  // instantiated combinations: private tallies (small domains) and bricks (large ones) never coincide
  if (L.priv && L.gridLds) return L.tblLds ? launch_trace_t<BLOCK, true, 2, false, DBG>(c, p, L.lds, nBatches)
                                           : launch_trace_t<BLOCK, false, 2, false, DBG>(c, p, L.lds, nBatches);
  if (L.priv) return L.tblLds ? launch_trace_t<BLOCK, true, 1, false, DBG>(c, p, L.lds, nBatches)
                              : launch_trace_t<BLOCK, false, 1, false, DBG>(c, p, L.lds, nBatches);
  if (L.brick) return L.tblLds ? launch_trace_t<BLOCK, true, 0, true, DBG>(c, p, L.lds, nBatches)
                               : launch_trace_t<BLOCK, false, 0, true, DBG>(c, p, L.lds, nBatches);
  return L.tblLds ? launch_trace_t<BLOCK, true, 0, false, DBG>(c, p, L.lds, nBatches)
                  : launch_trace_t<BLOCK, false, 0, false, DBG>(c, p, L.lds, nBatches);
}

// The block walk applies where the face-by-face plan already keeps grid, tallies (and tables) in LDS -- or, in the wide
// plan, tallies and the cells' block numbers (blockLite) -- radiance is off, and blocks are worth it: at least four cells
// per block on average (a medium that differs from cell to cell would pay a position look-up at every face for nothing).
bool block_walk_applies(const mcbrat_ctx *c, const LaunchPlan &L) {
  return blocks_worth_it(c) && ((L.priv && L.gridLds) || L.blockLite);
}

template <int BLOCK, bool TBL, bool DBG, bool EMIT, int SIMPLE, int OPT = 0>
int launch_block_s(mcbrat_ctx *c, DevParams &p, size_t lds, int nBatches) {
  auto kernel = trace_block_kernel<BLOCK, TBL, DBG, EMIT, SIMPLE, OPT>;
  if (lds + 512 > c->ldsPerCU) return fail(c, "computeRadiativeTransfer: the block-walk tables do not fit the LDS of a compute unit.");
  if (lds > kLdsBudget)
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int perCU = c->blocksPerCU;
  if (perCU <= 0) {
    HIP_OK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, BLOCK, lds));
    perCU = std::max(1, std::min(perCU, 8));
  }
  unsigned long long blocks = (unsigned long long)perCU * c->numCUs;
  const unsigned long long upb = units_per_batch(blocks, p.ppb, nBatches, BLOCK);
  p.unitsPerBatch = upb;
  p.nUnits = upb * (unsigned long long)nBatches;
  blocks = std::min(blocks, p.nUnits);
  p.ldsBytes = (unsigned)lds;
  hipLaunchKernelGGL(kernel, dim3((unsigned)std::max<unsigned long long>(1, blocks)), dim3(BLOCK), lds, c->L().stream, p);
  HIP_OK(c, hipGetLastError());
  return 0;
}

template <int BLOCK, bool TBL, bool DBG, bool EMIT, int OPT = 0>
int launch_block_e(mcbrat_ctx *c, DevParams &p, size_t lds, int nBatches) {
  // equally spaced axes, one component, no surface description: the instantiation with those decided at compile time
  const bool simple = c->xyRegular && c->zRegular && c->nc == 1 && c->surfNumX == 0;
  if constexpr (OPT == 0 && BLOCK != 1024) {
    if (simple && c->ny == 1 && !DBG) return launch_block_s<BLOCK, TBL, DBG, EMIT, 2, OPT>(c, p, lds, nBatches);  // an x-z problem
  }
  if constexpr (!DBG && (BLOCK == 1024 || BLOCK == 768)) {
    // no axis equally spaced by the reference's single-precision test, every axis equally spaced to 1e-6 of a cell (0.1 km cells):
    // the same cells as the general instantiation finds, with nothing left to decide at run time (SIMPLE = 3)
    if (!c->xyRegular && !c->zRegular && p.xyNearUniform && p.zNearUniform && c->nc == 1 && c->surfNumX == 0 && !getenv("MCBRAT_NO_SIMPLE3"))
      return launch_block_s<BLOCK, TBL, DBG, EMIT, 3, OPT>(c, p, lds, nBatches);
  }
  return simple ? launch_block_s<BLOCK, TBL, DBG, EMIT, 1, OPT>(c, p, lds, nBatches) : launch_block_s<BLOCK, TBL, DBG, EMIT, 0, OPT>(c, p, lds, nBatches);
}

template <int BLOCK, bool TBL, bool DBG, int OPT = 0>
int launch_block_t(mcbrat_ctx *c, DevParams &p, size_t lds, int nBatches) {
  return c->srcKind == 0 ? launch_block_e<BLOCK, TBL, DBG, false, OPT>(c, p, lds, nBatches) : launch_block_e<BLOCK, TBL, DBG, true, OPT>(c, p, lds, nBatches);
}

int launch_block(mcbrat_ctx *c, DevParams &p, const LaunchPlan &L, bool debug, int nBatches) {
  p.cdfTopLds = L.cdfTop ? 1 : 0;
  const size_t lds = block_lds_layout(c->nx, c->ny, c->nz, c->nc, (size_t)p.slabStride, c->nBlocks,
                                      L.tblLds ? (size_t)c->tblTotalFloats : 0, L.blockLite ? L.optics : 0, L.cdfTop).total;
  if (L.wide) {  // one workgroup per compute unit: 1024 lanes (the instrumented instantiation: 512)
    if (L.blockLite && L.optics == 2) {
      if (debug) return L.tblLds ? launch_block_t<512, true, true, 2>(c, p, lds, nBatches) : launch_block_t<512, false, true, 2>(c, p, lds, nBatches);
      return L.tblLds ? launch_block_t<1024, true, false, 2>(c, p, lds, nBatches) : launch_block_t<1024, false, false, 2>(c, p, lds, nBatches);
    }
    if (L.blockLite) {
      if (debug) return L.tblLds ? launch_block_t<512, true, true, 1>(c, p, lds, nBatches) : launch_block_t<512, false, true, 1>(c, p, lds, nBatches);
      return L.tblLds ? launch_block_t<1024, true, false, 1>(c, p, lds, nBatches) : launch_block_t<1024, false, false, 1>(c, p, lds, nBatches);
    }
    if (debug) return L.tblLds ? launch_block_t<512, true, true>(c, p, lds, nBatches) : launch_block_t<512, false, true>(c, p, lds, nBatches);
    return L.tblLds ? launch_block_t<1024, true, false>(c, p, lds, nBatches) : launch_block_t<1024, false, false>(c, p, lds, nBatches);
  }
  const int block = c->blockSize > 0 ? c->blockSize : 768;
  if (debug) return L.tblLds ? launch_block_t<512, true, true>(c, p, lds, nBatches) : launch_block_t<512, false, true>(c, p, lds, nBatches);
  if (block == 768) return L.tblLds ? launch_block_t<768, true, false>(c, p, lds, nBatches) : launch_block_t<768, false, false>(c, p, lds, nBatches);
  if (block == 256) return L.tblLds ? launch_block_t<256, true, false>(c, p, lds, nBatches) : launch_block_t<256, false, false>(c, p, lds, nBatches);
  return L.tblLds ? launch_block_t<512, true, false>(c, p, lds, nBatches) : launch_block_t<512, false, false>(c, p, lds, nBatches);
}

int launch_trace(mcbrat_ctx *c, DevParams &p, bool debug, int nBatches) {
  LaunchPlan L = plan_launch(c, (size_t)p.slabStride);
  if (L.priv && L.brick) {  // fill_params chose the brick arrays: private tallies give way
    L.priv = false;
    L.gridLds = false;
    L.lds = plan_launch_lds(c, L);
  }
  if (L.fly) {
    p.fly = 1; p.flyNbx = c->flyNbx; p.flyNby = c->flyNby; p.extWalk = c->dExtWalk;
    p.flyInvBrickX = (float)(c->flyNbx / p.Lx); p.flyInvBrickY = (float)(c->flyNby / p.Ly);
  }
#ifdef MCBRAT_DEV_MAIN_ONLY  // development builds (seconds instead of minutes, scripts/kernel_resources.py --main): only the two
  // instantiations that carry the bench workloads exist -- the step cloud's block walk and the 128x128x64 flux kernel
  if (block_walk_applies(c, L))
    return launch_block_s<768, true, false, false, 2>(c, p, block_lds_layout(c->nx, c->ny, c->nz, c->nc, (size_t)p.slabStride, c->nBlocks, (size_t)c->tblTotalFloats).total, nBatches);
  if (debug) return launch_trace_e<256, false, 0, false, true, false, false, 0>(c, p, L.lds, nBatches);
  if (!(p.rec != nullptr && p.layerSkip && p.fly && p.surfNumX == 0)) return fail(c, "development build: only the bench workloads' kernels exist");
  if (p.xyRegularWalk && !p.zRegularWalk) return launch_trace_e<256, false, 0, false, false, false, false, 3>(c, p, L.lds, nBatches);
  if (!p.xyRegularWalk && !p.zRegularWalk) return launch_trace_e<256, false, 0, false, false, false, false, 1>(c, p, L.lds, nBatches);  // (the 128x128x64 bench fields)
  return fail(c, "development build: only the bench workloads' kernels exist");
#else
  if (block_walk_applies(c, L)) return launch_block(c, p, L, debug, nBatches);
  if (L.wide) {  // one workgroup of 1024 lanes per compute unit, tallies (and what else fits) in its LDS; instrumented: 512 lanes
    if (debug) return launch_trace_b<512, true>(c, p, L, nBatches);
    if (L.gridLds) return L.tblLds ? launch_trace_t<1024, true, 2, false, false>(c, p, L.lds, nBatches) : launch_trace_t<1024, false, 2, false, false>(c, p, L.lds, nBatches);
    return L.tblLds ? launch_trace_t<1024, true, 1, false, false>(c, p, L.lds, nBatches) : launch_trace_t<1024, false, 1, false, false>(c, p, L.lds, nBatches);
  }
  // small domains (grid, tables and tallies in LDS): LDS holds two workgroups per CU, and two workgroups of 12 waves
  // (6 per SIMD, 80 VGPRs) beat two of 8 (4 per SIMD, no spills) by 10 % on the step cloud (640 and 896 lanes lose)
  // (radiance on LDS-resident domains keeps 512 lanes: 768 lanes at 80 VGPRs lose 20 % there)
  if (L.block == 768 && L.priv && L.gridLds && c->nDir == 0 && !debug)
    return L.tblLds ? launch_trace_t<768, true, 2, false, false>(c, p, L.lds, nBatches) : launch_trace_t<768, false, 2, false, false>(c, p, L.lds, nBatches);
  if (L.block >= 512) return debug ? launch_trace_b<512, true>(c, p, L, nBatches) : launch_trace_b<512, false>(c, p, L, nBatches);
  return debug ? launch_trace_b<256, true>(c, p, L, nBatches) : launch_trace_b<256, false>(c, p, L, nBatches);
#endif
}

int check_ready(mcbrat_ctx *c) {
  if (!c) return 1;
  if (!c->haveGrid || !c->haveOptics) return fail(c, "computeRadiativeTransfer: problem not completely specified.");
  if (!c->haveSource) return fail(c, "computeRadiativeTransfer: no photon source set.");
  if (sync_tables(c)) return 1;
  return sync_forward_tables(c);
}


// Times short trial launches at a few event thresholds and keeps the fastest.  The best value
// depends on how many voxel faces a leg crosses (step cloud ~3, 128x128x64 cloud field ~14).
int autotune(mcbrat_ctx *c, DevParams p, unsigned long long ppb, int nBatches) {
  // about 170 photons per resident lane (6.7e7): with fewer the synchronised start and the drain of the launch favour too
  // low a threshold -- measured on the two 128x128x64 workloads, trials of 1.7e7 / 3.4e7 / 6.7e7 / 1.3e8 photons chose
  // 8 / 16 / 20-24 / 24 on the radar-like field (8 costs 13 % of the rate of 24) and 20-32 / 20 / 24 / 24 on the cloud field
  // (radiance runs cost several times more per photon: fewer trial photons, so that the ten trial launches stay well under a second)
  const unsigned long long want = c->tuneTrialPhotons / (unsigned long long)(1 + 2 * c->nDir);
  const unsigned long long total = ppb * (unsigned long long)nBatches;
  if (total < want) {
    // too few photons for a meaningful trial: a guess by domain size (few faces per leg on small grids, many on
    // large ones), and the trial is left for a later, larger call
    // (the thermal source: photons start inside the medium and most end at their first roulette -- few lanes are ever on a
    // long walk, and waiting for 24 of them starves the event phase: 4-8 measured best on config 4, 24 costs 8 %)
    // (the block walk has no walk loop: its threshold only says when a wave has little else to do than its queued work --
    // 16-32 measured best on config 4's wide kernel too, 8 costs it 2.5 %)
    const LaunchPlan L0 = plan_launch(c, (size_t)p.slabStride);
    c->eventThreshold = block_walk_applies(c, L0) ? 16 : (c->srcKind != 0 ? 8 : (L0.gridLds ? 16 : 24));
    return 0;
  }
  const int nb = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)nBatches, want / std::max<unsigned long long>(1, ppb)));
  p.total = std::min(want, ppb * (unsigned long long)nb);
  p.fates = nullptr; p.counters = nullptr;
  const int candidates[] = {8, 16, 20, 24, 32, 48};
  float best = 1e30f;
  int bestThr = c->eventThreshold;
  for (int thr : candidates) {
    p.eventThreshold = thr;
    float ms = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {  // first repetition warms caches / code
      HIP_OK(c, hipMemsetAsync(c->L().dSlabs, 0, sizeof(long long) * p.slabStride * nb, c->L().stream));
      HIP_OK(c, hipMemsetAsync(c->L().dCounter, 0, sizeof(unsigned long long), c->L().stream));
      HIP_OK(c, hipEventRecord(c->L().ev0, c->L().stream));
      if (launch_trace(c, p, false, nb)) return 1;
      HIP_OK(c, hipEventRecord(c->L().ev1, c->L().stream));
      HIP_OK(c, hipStreamSynchronize(c->L().stream));
      HIP_OK(c, hipEventElapsedTime(&ms, c->L().ev0, c->L().ev1));
    }
    if (ms < best) { best = ms; bestThr = thr; }
  }
  c->eventThreshold = bestThr;
  c->tuned = true;
  return 0;
}

}  // namespace

extern "C" {

int mcbrat_abi_version(void) { return MCBRAT_ABI_VERSION; }

mcbrat_ctx *mcbrat_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  mcbrat_ctx *c = new mcbrat_ctx();
  c->device = device;
  if (const char *e = getenv("MCBRAT_REGULAR_WALK")) c->regularWalk = atoi(e);
  if (const char *e = getenv("MCBRAT_LAYER_SKIP")) c->layerSkip = atoi(e);
  if (const char *e = getenv("MCBRAT_BLOCK_WALK")) c->blockWalk = atoi(e);
  if (const char *e = getenv("MCBRAT_FLIGHT_MAX_DEPTH")) c->flightMaxDepth = atof(e);
  if (const char *e = getenv("MCBRAT_WIDE")) c->wideDefault = c->wideMode = std::max(0, std::min(2, atoi(e)));
  if (const char *e = getenv("MCBRAT_JUMP_THRESHOLD")) c->jumpThreshold = std::max(1, std::min(64, atoi(e)));
  if (const char *e = getenv("MCBRAT_CROSS_THRESHOLD")) c->crossThreshold = std::max(1, std::min(64, atoi(e)));
  if (const char *e = getenv("MCBRAT_RAY_DEFER")) c->rayDefer = atoi(e);
  if (const char *e = getenv("MCBRAT_RAY_SHORT")) c->rayShort = std::max(1, atoi(e));
  if (const char *e = getenv("MCBRAT_RAY_PASS_ITERS")) c->rayPassIters = std::max(1, atoi(e));
  if (const char *e = getenv("MCBRAT_RAY_PASS_AT")) c->rayPassAt = std::max(1, atoi(e));
  if (const char *e = getenv("MCBRAT_TUNE_PHOTONS")) c->tuneTrialPhotons = strtoull(e, nullptr, 10);
  if (const char *e = getenv("MCBRAT_WATCHDOG")) c->watchdog = (unsigned)std::max(16ll, atoll(e));
  if (const char *e = getenv("MCBRAT_MAX_EVENTS")) c->maxEvents = (unsigned)std::max(16ll, atoll(e));
  if (const char *e = getenv("MCBRAT_MAX_EVENTS_NAN")) c->maxEventsNaN = (unsigned)std::max(16ll, atoll(e));
  if (const char *e = getenv("MCBRAT_TEST_LEGACY_TIES")) c->legacyTies = atoi(e);
  if (const char *e = getenv("MCBRAT_TEST_RAY_MAX_LEN")) c->testRayMaxLen = (float)atof(e);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    c->numCUs = prop.multiProcessorCount;
    if (prop.maxSharedMemoryPerMultiProcessor > 0) c->ldsPerCU = prop.maxSharedMemoryPerMultiProcessor;
  }
  if (init_lane(c, 0) || hipEventCreateWithFlags(&c->evExternal, hipEventDisableTiming) != hipSuccess ||
      dev_malloc((void **)&c->dEventCounters, 32 * sizeof(unsigned long long)) != hipSuccess ||
      dev_malloc((void **)&c->dBad, kBadWords * sizeof(unsigned long long)) != hipSuccess ||  // count, claim word, record of the first drop
      hipMemsetAsync(c->dBad, 0, kBadWords * sizeof(unsigned long long), c->lane[0].stream) != hipSuccess ||
      hipStreamSynchronize(c->lane[0].stream) != hipSuccess ||
      hipHostMalloc((void **)&c->hBad, sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&c->hBadDev, c->hBad, 0) != hipSuccess) {
    mcbrat_destroy(c);
    return nullptr;
  }
  *c->hBad = 0ull;
  return c;
}

void mcbrat_destroy(mcbrat_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)sync_all(c);
  void *bufs[] = {c->dEdges, c->dExt, c->dCum, c->dSsa, c->dRelArea, c->dPfi, c->dTables, c->dVoxelCDF,
                  c->dEventCounters, c->dLast, c->dMomentsOwned, c->dBrickTable, c->dExtB,
                  c->dBlockRec, c->dBlockOf, c->dBlockExt, c->dBlockSsa, c->dBlockCum, c->dBlockPfi, c->dCumB, c->dSsaB, c->dPfiB, c->dBgExt, c->dBgCum, c->dBgSsa, c->dBgPfi, c->dLayerExt, c->dRec, c->dLayerRun, c->dLayerRunT, c->dExtWalk, c->dBgVal, c->dFlyRange, c->dSurfX, c->dSurfY, c->dSurfRefl,
                  c->dBad, c->dFreqCdf, c->dFreqCounts, c->dDirData, c->dFwd, c->dFwdOrig};
  if (c->hBad) (void)hipHostFree(c->hBad);
  for (void *b : bufs) if (b) (void)hipFree(b);
  for (mcbrat_ctx::Lane &L : c->lane) {
    void *lb[] = {L.dCounter, L.dSlabs, L.dColVals, L.dScalVals};
    for (void *b : lb) if (b) (void)hipFree(b);
    if (L.ev0) (void)hipEventDestroy(L.ev0);
    if (L.ev1) (void)hipEventDestroy(L.ev1);
    if (L.evDone) (void)hipEventDestroy(L.evDone);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  for (auto &pr : c->eventPool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (c->evExternal) (void)hipEventDestroy(c->evExternal);
  if (c->evChain) (void)hipEventDestroy(c->evChain);
  delete c;
}

const char *mcbrat_last_error(const mcbrat_ctx *c) { return c ? c->err.c_str() : "mcbrat: no context (no usable HIP device)"; }

int mcbrat_set_grid(mcbrat_ctx *c, int32_t nx, int32_t ny, int32_t nz, const double *xe, const double *ye,
                    const double *ze) {
  if (!c) return 1;
  if (nx < 1 || ny < 1 || nz < 1 || !xe || !ye || !ze) return fail(c, "new_Integrator: Problems reading domain.");
  for (int i = 0; i < nx; ++i) if (!(xe[i + 1] > xe[i])) return fail(c, "new_Domain: x edges must be increasing, unique.");
  for (int i = 0; i < ny; ++i) if (!(ye[i + 1] > ye[i])) return fail(c, "new_Domain: y edges must be increasing, unique.");
  for (int i = 0; i < nz; ++i) if (!(ze[i + 1] > ze[i])) return fail(c, "new_Domain: z edges must be increasing, unique.");
  if ((long long)nx * ny * nz > 0x7fffffffLL / 2) return fail(c, "new_Integrator: more than 2^30 cells are not supported.");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  c->nx = nx; c->ny = ny; c->nz = nz;
  c->xe.assign(xe, xe + nx + 1); c->ye.assign(ye, ye + ny + 1); c->ze.assign(ze, ze + nz + 1);
  // regular-spacing flags exactly as new_Integrator :163-181 (deltaX/Y/Z are default-real locals, :140)
  const float fX = (float)(xe[1] - xe[0]), fY = (float)(ye[1] - ye[0]), fZ = (float)(ze[1] - ze[0]);
  bool xr = true, yr = true, zr = true;
  for (int i = 0; i < nx; ++i) xr = xr && std::fabs((xe[i + 1] - xe[i]) - (double)fX) <= 2.0 * spacing_d(xe[i + 1]);
  for (int i = 0; i < ny; ++i) yr = yr && std::fabs((ye[i + 1] - ye[i]) - (double)fY) <= 2.0 * spacing_d(ye[i + 1]);
  for (int i = 0; i < nz; ++i) zr = zr && std::fabs((ze[i + 1] - ze[i]) - (double)fZ) <= spacing_d(ze[i + 1]);
  c->xyRegular = xr && yr; c->zRegular = zr;
  c->dX = fX; c->dY = fY; c->dZ = fZ;
  std::vector<double> edges;
  edges.insert(edges.end(), c->xe.begin(), c->xe.end());
  edges.insert(edges.end(), c->ye.begin(), c->ye.end());
  edges.insert(edges.end(), c->ze.begin(), c->ze.end());
  if (upload(c, &c->dEdges, edges.data(), edges.size())) return 1;
  // relative column areas, computeRadiativeTransfer :334-340 (real(8) expression stored to default real)
  std::vector<float> rel((size_t)nx * ny);
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i)
      rel[(size_t)i + (size_t)nx * j] = (float)(((ye[j + 1] - ye[j]) * (xe[i + 1] - xe[i])) / ((xe[nx] - xe[0]) * (ye[ny] - ye[0])));
  if (upload(c, &c->dRelArea, rel.data(), rel.size())) return 1;
  c->haveGrid = true; c->haveOptics = false; c->haveSource = false; c->haveLast = false; c->tuned = false;
  // everything sized by the old grid goes: the moment arrays, the last batch's results (the finish kernels write
  // moments_len() elements of it), and the per-lane buffers are re-sized by the next call (their capacities are
  // kept in elements and compared with what that call needs)
  if (c->dMomentsOwned) { (void)hipFree(c->dMomentsOwned); c->dMomentsOwned = nullptr; }
  c->dMoments = nullptr;
  if (c->dLast) { (void)hipFree(c->dLast); c->dLast = nullptr; }
  return 0;
}

int mcbrat_set_optics(mcbrat_ctx *c, int32_t nc, const double *totalExt, const double *cumExt, const double *ssa,
                      const int32_t *pfIndex, double albedo) {
  if (!c) return 1;
  if (!c->haveGrid) return fail(c, "getInfo_Domain: domain hasn't been initialized.");
  if (nc < 1 || nc > MCBRAT_MAX_COMPONENTS) return fail(c, "getOpticalPropertiesByComponent: unsupported number of components.");
  if (!totalExt || !cumExt || !ssa || !pfIndex) return fail(c, "getOpticalPropertiesByComponent: domain contains no optical components.");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  const size_t nvox = (size_t)c->nx * c->ny * c->nz;
  std::vector<float> e(nvox), cu(nvox * nc), s(nvox * nc);
  std::vector<uint16_t> pf(nvox * nc);
  std::vector<int> maxPfi(nc, 0);
  for (size_t v = 0; v < nvox; ++v) {
    if (!(totalExt[v] >= 0.0)) return fail(c, "validateOpticalComponent: extinction must be >= 0.");
    e[v] = (float)totalExt[v];
  }
  for (int k = 0; k < nc; ++k)
    for (size_t v = 0; v < nvox; ++v) {
      const size_t i = (size_t)k * nvox + v;
      if (totalExt[v] > 0.0 && !(ssa[i] >= 0.0 && ssa[i] <= 1.0)) return fail(c, "validateOpticalComponent: singleScatteringAlbedo must be between 0 and 1.");
      cu[i] = (float)cumExt[i];
      s[i] = totalExt[v] > 0.0 ? (float)ssa[i] : 0.0f;  // (nothing collides without extinction; the block walk tells vacuum by this 0)
      int idx = pfIndex[i];
      if (totalExt[v] > 0.0 && (idx < 1 || idx > 65535)) return fail(c, "validateOpticalComponent: phaseFunctionIndex out of range.");
      if (idx < 1) idx = 1;
      pf[i] = (uint16_t)(idx - 1);
      if (totalExt[v] > 0.0) maxPfi[k] = std::max(maxPfi[k], idx - 1);
    }
  if (!(albedo >= 0.0 && albedo <= 1.0)) return fail(c, "new_Domain: surfaceAlbedo must be between 0 and 1.");
  if (upload(c, &c->dExt, e.data(), e.size()) || upload(c, &c->dCum, cu.data(), cu.size()) ||
      upload(c, &c->dSsa, s.data(), s.size()) || upload(c, &c->dPfi, pf.data(), pf.size()))
    return 1;
  if (c->dRec) { (void)hipFree(c->dRec); c->dRec = nullptr; }
  if (nc <= 2) {
    std::vector<uint32_t> rec(4 * nvox);
    for (size_t v = 0; v < nvox; ++v) {
      const float c0 = nc > 1 ? cu[v] : 2.0f;  // uniform >= c0 picks the second component: never with one component
      std::memcpy(&rec[4 * v + 0], &c0, 4);
      std::memcpy(&rec[4 * v + 1], &s[v], 4);
      const float s1 = nc > 1 ? s[nvox + v] : 0.0f;
      std::memcpy(&rec[4 * v + 2], &s1, 4);
      rec[4 * v + 3] = (uint32_t)pf[v] | ((uint32_t)(nc > 1 ? pf[nvox + v] : 0) << 16);
    }
    if (upload(c, &c->dRec, rec.data(), rec.size())) return 1;
  }
  c->bricksBuilt = false;
  if (build_bricks(c, e, cu, s, pf, nc)) return 1;
  if (build_blocks(c, e, cu, s, pf, nc)) return 1;
  c->nc = nc;
  c->albedo = (float)albedo;
  c->maxPfi = maxPfi;
  c->tables.resize(nc); c->tblNSteps.resize(nc, 0); c->tblNEntries.resize(nc, 0);
  c->tablesDirty = true;
  c->haveOptics = true;
  c->tuned = false;
  return 0;
}

int mcbrat_set_inverse_table(mcbrat_ctx *c, int32_t component, int32_t nSteps, int32_t nEntries, const float *table) {
  if (!c) return 1;
  if (!c->haveOptics) return fail(c, "tabulateInversePhaseFunctions: domain has no optical components yet.");
  if (component < 1 || component > c->nc) return fail(c, "tabulateInversePhaseFunctions: failed on component" + std::to_string(component));
  if (nSteps < 2 || nEntries < 1 || !table) return fail(c, "computeInversePhaseFunctionTable: Array for inverse table has the wrong number of entries");
  c->tables[component - 1].assign(table, table + (size_t)nSteps * nEntries);
  c->tblNSteps[component - 1] = nSteps;
  c->tblNEntries[component - 1] = nEntries;
  c->tablesDirty = true;
  return 0;
}

int mcbrat_specify_parameters(mcbrat_ctx *c, int32_t useRayTracing, int32_t useRussianRoulette, float lwFlag) {
  if (!c) return 1;
  if (!useRayTracing)
    return fail(c, "specifyParameters: useRayTracing=.false. (maximum cross-section) is not supported: "
                   "that branch of the reference never refreshes the cell indices (SURVEY.md 8a quirk 4).");
  c->useRR = useRussianRoulette ? 1 : 0;
  c->lwFlag = lwFlag;
  return 0;
}

int mcbrat_set_source_solar(mcbrat_ctx *c, float solarMu, float solarAzimuthDeg) {
  if (!c) return 1;
  if (!c->haveGrid) return fail(c, "setIllumination: domain hasn't been initialized.");
  if (solarAzimuthDeg < 0.f || solarAzimuthDeg > 360.f) return fail(c, "setIllumination: solarAzimuth out of bounds");
  if (std::fabs(solarMu) > 1.f || std::fabs(solarMu) <= FLT_MIN) return fail(c, "setIllumination: solarMu out of bounds");
  // newPhotonStream_Directional, monteCarloIllumination.f95:93-96, then makeDirectionCosines :1876-1894
  const float mu = -std::fabs(solarMu);
  const float phi = (solarAzimuthDeg * std::acos(-1.0f)) / 180.0f;
  const float sinTheta = std::sqrt(1.0f - mu * mu);
  c->dir0[0] = sinTheta * std::cos(phi); c->dir0[1] = sinTheta * std::sin(phi); c->dir0[2] = mu;
  // launch height z = 1 - spacing(1.) of the column (:93), mapped as computeRT :484-493
  const double frac = (double)(1.0f - 1.1920929e-07f);
  const double z0 = c->ze.front(), zMax = c->ze.back();
  if (c->zRegular) {
    c->zLaunch = z0 + frac * (zMax - z0);
    int zi = (int)((c->zLaunch - z0) / c->dZ) + 1;
    zi = std::min(zi, c->nz);
    if (std::fabs(c->ze[zi] - c->zLaunch) < spacing_d(c->zLaunch)) zi++;
    c->izLaunch = std::min(zi, c->nz) - 1;
  } else {
    const double t = (frac - z0) * (double)c->nz;
    const double fl = std::floor(t);
    int zi = std::min((int)fl + 1, c->nz);
    zi = std::max(zi, 1);
    c->izLaunch = zi - 1;
    c->zLaunch = c->ze[zi - 1] + (t - fl) * (c->ze[zi] - c->ze[zi - 1]);
  }
  if (c->srcKind != 0) c->tuned = false;
  c->srcKind = 0;
  c->haveSource = true;
  return 0;
}

int mcbrat_set_source_emission(mcbrat_ctx *c, const double *voxelWeights, double fracAtmsPower) {
  if (!c) return 1;
  if (!c->haveGrid) return fail(c, "setIllumination: domain hasn't been initialized.");
  if (!voxelWeights || !(fracAtmsPower >= 0.0 && fracAtmsPower <= 1.0)) return fail(c, "setIllumination: invalid emission weights.");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  if (upload(c, &c->dVoxelCDF, voxelWeights, (size_t)c->nx * c->ny * c->nz)) return 1;
  c->fracAtms = fracAtmsPower;
  if (c->srcKind != 1) c->tuned = false;
  c->srcKind = 1;
  c->haveSource = true;
  return 0;
}

int64_t mcbrat_moments_length(const mcbrat_ctx *c) { return c && c->haveGrid ? moments_len(c) : 0; }

int mcbrat_bind_moments(mcbrat_ctx *c, double *deviceBuffer) {
  if (!c) return 1;
  if (!c->haveGrid) return fail(c, "bind_moments: set the grid first.");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  if (deviceBuffer) c->dMoments = deviceBuffer;
  else { c->dMoments = c->dMomentsOwned; return ensure_moments(c); }
  return 0;
}

int mcbrat_reset_moments(mcbrat_ctx *c) {
  if (!c) return 1;
  if (!c->haveGrid) return fail(c, "reset_moments: set the grid first.");
  (void)hipSetDevice(c->device);
  if (ensure_moments(c)) return 1;
  // c->cur is the lane of the latest call: its stream already orders this after every earlier finish chain
  if (c->externalPending) { HIP_OK(c, hipStreamWaitEvent(c->L().stream, c->evExternal, 0)); c->externalPending = false; }
  if (c->chainAfter) {
    if (c->chainSnapshot) HIP_OK(c, hipStreamWaitEvent(c->L().stream, c->chainSnapshot, 0));
    c->chainAfter = false; c->chainSnapshot = nullptr;
  }
  HIP_OK(c, hipMemsetAsync(c->dMoments, 0, sizeof(double) * (8 + 2 * (size_t)moments_len(c)), c->L().stream));
  // stream-ordered before whatever the context enqueues next; readers synchronise (get_moments, report_results)
  HIP_OK(c, hipEventRecord(c->L().evDone, c->L().stream));
  c->lastDone = c->L().evDone;
  return 0;
}

int mcbrat_get_moments(mcbrat_ctx *c, double *host) {
  if (!c || !host) return 1;
  (void)hipSetDevice(c->device);
  if (ensure_moments(c)) return 1;
  if (sync_all(c)) return 1;
  HIP_OK(c, hipMemcpy(host, c->dMoments, sizeof(double) * (8 + 2 * (size_t)moments_len(c)), hipMemcpyDeviceToHost));
  return 0;
}

int mcbrat_enable_counters(mcbrat_ctx *c, int32_t enable) {
  if (!c) return 1;
  c->countersOn = enable != 0;
  return 0;
}
int mcbrat_get_counters(mcbrat_ctx *c, mcbrat_counters *out) {
  if (!c || !out) return 1;
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;  // (asynchronous mode: badPhotons is read when the enqueued work has finished)
  *out = c->lastCounters;
  return 0;
}
float mcbrat_last_trace_ms(const mcbrat_ctx *c) { return c ? c->lastTraceMs : 0.f; }

int mcbrat_set_forward_table(mcbrat_ctx *c, int32_t component, int32_t nAngles, int32_t nEntries, const float *table,
                             const float *origTable) {
  if (!c) return 1;
  if (!c->haveOptics) return fail(c, "tabulateForwardPhaseFunctions: domain has no optical components yet.");
  if (component < 1 || component > c->nc) return fail(c, "tabulatePhaseFunctions: failed on component" + std::to_string(component));
  if (nAngles < 2 || nEntries < 1 || !table) return fail(c, "tabulateForwardPhaseFunctions: table has the wrong number of entries");
  c->fwd.resize(c->nc); c->fwdOrig.resize(c->nc); c->fwdNAngles.resize(c->nc, 0); c->fwdNEntries.resize(c->nc, 0);
  c->fwd[component - 1].assign(table, table + (size_t)nAngles * nEntries);
  const float *o = origTable ? origTable : table;
  c->fwdOrig[component - 1].assign(o, o + (size_t)nAngles * nEntries);
  c->fwdNAngles[component - 1] = nAngles;
  c->fwdNEntries[component - 1] = nEntries;
  c->fwdDirty = true;
  return 0;
}

int mcbrat_specify_intensity(mcbrat_ctx *c, int32_t nDirections, const float *mus, const float *phisDeg,
                             int32_t useRussianRouletteForIntensity, float zetaMin,
                             int32_t useHybridPhaseFunsForIntenCalcs, int32_t numOrdersOrigPhaseFunIntenCalcs,
                             int32_t limitIntensityContributions, float maxIntensityContribution) {
  if (!c) return 1;
  if (!c->haveGrid) return fail(c, "specifyParameters: set the grid first.");
  if (nDirections < 0 || nDirections > MCBRAT_MAX_DIRECTIONS || (nDirections > 0 && (!mus || !phisDeg)))
    return fail(c, "specifyParameters: invalid number of intensity directions.");
  for (int i = 0; i < nDirections; ++i) {  // specifyParameters :1140-1145
    if (mus[i] < -1.f || mus[i] > 1.f) return fail(c, "specifyParameters: intensityMus must be between -1 and 1");
    if (std::fabs(mus[i]) < FLT_MIN) return fail(c, "specifyParameters: intensityMus can't be 0 (directly sideways)");
    if (phisDeg[i] < 0.f || phisDeg[i] > 360.f) return fail(c, "specifyParameters: intensityPhis must be between 0 and 360");
  }
  if (limitIntensityContributions && !(maxIntensityContribution > 0.f))
    return fail(c, "specifyParameters: maxIntensityContribution must be > 0");
  if (zetaMin < 0.f) return fail(c, "specifyParameters: zetaMin must be >= 0.");
  if (numOrdersOrigPhaseFunIntenCalcs < 0) return fail(c, "specifyParameters: numOrdersOrigPhaseFunIntenCalcs must be >= 0");
  if (useRussianRouletteForIntensity)
    for (int i = 0; i < nDirections; ++i)
      if (mus[i] < 0.f)
        return fail(c, "specifyParameters: useRussianRouletteForIntensity only works for upward directions "
                       "(the reference restarts the walk below the surface for mu < 0).");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  constexpr float kPi = 3.14159265358979312f;
  c->dirData.assign((size_t)8 * std::max(nDirections, 1), 0.f);
  for (int i = 0; i < nDirections; ++i) {  // makeDirectionCosines(mu, phi * Pi/180) :1270-1272, :1876-1894
    const float mu = mus[i], phi = phisDeg[i] * kPi / 180.0f;
    const float sinTheta = std::sqrt(1.0f - mu * mu);
    float *d = &c->dirData[(size_t)8 * i];
    d[0] = sinTheta * std::cos(phi); d[1] = sinTheta * std::sin(phi); d[2] = mu;
    d[3] = (4.0f * kPi) * std::fabs(d[2]);
    for (int a = 0; a < 3; ++a) d[4 + a] = std::fabs(d[a]) >= 2.0f * FLT_MIN ? 1.0f / d[a] : 0.0f;
  }
  const int limit = (limitIntensityContributions && maxIntensityContribution < FLT_MAX) ? 1 : 0;
  c->limitContrib = limit;  // (the batch slabs change length: the next call compares element counts)
  c->maxContrib = maxIntensityContribution;
  if (nDirections != c->nDir) {  // the moment arrays change length: start them afresh
    if (c->dMomentsOwned) { (void)hipFree(c->dMomentsOwned); c->dMomentsOwned = nullptr; }
    c->dMoments = nullptr;
    if (c->dLast) { (void)hipFree(c->dLast); c->dLast = nullptr; }
    c->haveLast = false;
    c->tuned = false;
  }
  c->nDir = nDirections;
  c->useRRIntensity = useRussianRouletteForIntensity ? 1 : 0;
  c->zetaMin = zetaMin;
  c->useHybrid = useHybridPhaseFunsForIntenCalcs ? 1 : 0;
  c->numOrdersOrig = numOrdersOrigPhaseFunIntenCalcs;
  c->fwdDirty = true;
  return 0;
}

int mcbrat_report_intensity(mcbrat_ctx *c, float *meanIntensity, float *intensity) {
  if (!c) return 1;
  if (c->nDir == 0) return fail(c, "reportResults: intensity information not available");
  if (!c->haveLast) return fail(c, "reportResults: no batch has been traced yet.");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  const size_t ncol = (size_t)c->nx * c->ny, base = 3 + 3 * ncol + c->nz + ncol * c->nz;
  std::vector<float> h(ncol * c->nDir);
  HIP_OK(c, hipMemcpy(h.data(), c->dLast + base, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
  if (intensity) std::memcpy(intensity, h.data(), sizeof(float) * h.size());
  if (meanIntensity)
    for (int d = 0; d < c->nDir; ++d) {  // reportResults :980-992
      float s = 0.f;
      for (size_t i = 0; i < ncol; ++i) s += h[(size_t)d * ncol + i];
      meanIntensity[d] = s / (float)ncol;
    }
  return 0;
}

int mcbrat_get_event_threshold(const mcbrat_ctx *c) { return c ? c->eventThreshold : 0; }

int mcbrat_set_async(mcbrat_ctx *c, int32_t enable) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  c->asyncOn = enable != 0;
  c->cur = 0; c->nextLane = 0;
  if (c->asyncOn)  // (every stream now: the first asynchronous call sizes all of them at once -- allocation synchronises the device)
    for (int i = 0; i < mcbrat_ctx::kLanes; ++i)
      if (init_lane(c, i)) return 1;
  return 0;
}

int mcbrat_synchronize(mcbrat_ctx *c) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  return sync_all(c);
}

int mcbrat_stream_wait_done(mcbrat_ctx *c, void *stream) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  if (c->lastDone) HIP_OK(c, hipStreamWaitEvent((hipStream_t)stream, c->lastDone, 0));
  return 0;
}

int mcbrat_wait_stream(mcbrat_ctx *c, void *stream) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  HIP_OK(c, hipEventRecord(c->evExternal, (hipStream_t)stream));
  c->externalPending = true;
  return 0;
}

int mcbrat_chain_after(mcbrat_ctx *c, mcbrat_ctx *previous) {
  if (!c || !previous) return 1;
  if (c == previous) return fail(c, "chain_after: a context follows its own calls by itself.");
  if (c->device != previous->device) return fail(c, "chain_after: the two contexts are on different devices.");
  (void)hipSetDevice(c->device);
  c->chainAfter = false; c->chainSnapshot = nullptr;
  if (!previous->lastDone) return 0;  // (nothing enqueued there yet: nothing to follow)
  // "so far" is fixed NOW: an event of this context on the stream of `previous`'s latest call, whose finish chain follows all
  // of its earlier ones.  `previous` may be destroyed or re-used afterwards; what is waited for does not change.
  if (!c->evChain) HIP_OK(c, hipEventCreateWithFlags(&c->evChain, hipEventDisableTiming));
  HIP_OK(c, hipEventRecord(c->evChain, previous->L().stream));
  c->chainAfter = true;
  c->chainSnapshot = c->evChain;
  return 0;
}

int mcbrat_set_tuning(mcbrat_ctx *c, int32_t blocksPerCU, int32_t eventThreshold, int32_t maxBatchesInFlight,
                      int32_t privateTallies, int32_t blockSize, int32_t launchThreshold, int32_t surfaceThreshold,
                      int32_t brickLayout) {
  if (!c) return 1;
  if (blocksPerCU >= 0) c->blocksPerCU = blocksPerCU;
  if (eventThreshold > 0) { c->eventThreshold = eventThreshold; c->autoTune = false; }
  if (eventThreshold == 0) { c->autoTune = true; c->tuned = false; }
  if (maxBatchesInFlight >= 0) c->maxBatchesInFlight = maxBatchesInFlight;
  // privateTallies: 0 global atomics; 1 the library's plan; 2 private tallies without the optical grid in LDS; 3 as 1 without the
  // wide plan (a slab too large for a shared compute unit then tallies with global atomics); 4 the wide plan (one workgroup
  // of 1024 lanes per compute unit) even where the shared plan would do; 5 as 4 with the per-cell optics left in global memory;
  // 6 as 4 without the per-cell optics in LDS: per block in LDS where every block is uniform in them, else as 5
  if (privateTallies >= 0) {
    if (privateTallies > 6) return fail(c, "set_tuning: privateTallies must be 0 ... 6");
    c->privMode = privateTallies ? 1 : 0;
    c->gridLdsMode = privateTallies == 2 ? 0 : (privateTallies == 5 ? 2 : (privateTallies == 6 ? 3 : 1));
    c->wideMode = privateTallies == 3 ? 0 : (privateTallies >= 4 ? 2 : c->wideDefault);
    c->tuned = false;
  }
  if (blockSize == 0 || blockSize == 256 || blockSize == 512 || blockSize == 768) c->blockSize = blockSize;
  else if (blockSize > 0) return fail(c, "set_tuning: blockSize must be 0, 256, 512 or 768");
  if (launchThreshold > 0) { c->launchThreshold = launchThreshold; c->launchThresholdSet = true; }
  if (surfaceThreshold > 0) { c->surfaceThreshold = surfaceThreshold; c->surfaceThresholdSet = true; }
  if (brickLayout >= 0 && brickLayout <= 2) { if (brickLayout != c->brickMode) c->tuned = false; c->brickMode = brickLayout; }
  return 0;
}

const char *mcbrat_first_drop(mcbrat_ctx *c) {
  if (!c) return "";
  (void)hipSetDevice(c->device);
  c->dropText.clear();
  unsigned long long h[kBadWords] = {0};
  if (sync_all(c) || hipMemcpy(h, c->dBad, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return c->dropText.c_str();
  if (h[0] == 0ull) return c->dropText.c_str();
  static const char *kinds[] = {"?", "leg budget (maxEvents)", "leg budget of a photon with a NaN direction at full weight (maxEventsNaN)", "wave watchdog (no lane made progress)",
                                "block crossings of one leg", "horizontal leg through vacuum in a block that spans both periodic axes", "view ray longer than the geometry allows"};
  char buf[768];
  int at = snprintf(buf, sizeof(buf), "%llu dropped; bounds that fired:", h[0]);
  for (unsigned k = 1; k < 7; ++k)
    if ((h[7] >> (k - 1)) & 1ull) at += snprintf(buf + at, sizeof(buf) - (size_t)at, " [%s]", kinds[k]);
  if (h[1] != 0ull) {  // (the instrumented instantiations also record the first one)
    const unsigned kind = (unsigned)(h[2] & 0xffu), kernel = (unsigned)((h[2] >> 8) & 0xffu), state = (unsigned)((h[2] >> 16) & 0xffffu);
    snprintf(buf + at, sizeof(buf) - (size_t)at, "; first recorded: %s in %s, photon id %llu, %s %u, leg %u", kind < 7 ? kinds[kind] : "?",
             kernel == 2 ? "trace_block_kernel" : "trace_kernel", h[3], kind == DROP_RAY ? "view direction" : "state", state, (unsigned)(h[2] >> 32));
  } else {
    snprintf(buf + at, sizeof(buf) - (size_t)at, "; no photon record (production kernels keep none: trace the same photons with mcbrat_trace_fates)");
  }
  c->dropText = buf;
  return c->dropText.c_str();
}

int mcbrat_set_option(mcbrat_ctx *c, const char *name, int32_t value) {
  if (!c) return 1;
  if (!name) return fail(c, "set_option: no option name");
  const std::string n(name);
  if (n == "jumpThreshold") c->jumpThreshold = std::max(1, std::min(64, (int)value));
  else if (n == "crossThreshold") c->crossThreshold = std::max(1, std::min(64, (int)value));
  else return fail(c, "set_option: unknown option '" + n + "'");
  return 0;
}

int mcbrat_set_surface_description(mcbrat_ctx *c, int32_t numX, int32_t numY, const double *xPosition, const double *yPosition,
                                   const float *reflectance) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  if (numX <= 0 || numY <= 0) { c->surfNumX = c->surfNumY = 0; return 0; }  // back to the domain's albedo
  // the checks of newSurfaceDescriptionXY, src/surfaceProperties.f95:66-82
  if (numX < 2 || numY < 2 || !xPosition || !yPosition || !reflectance)
    return fail(c, "new_SurfaceDescription: position vector(s) are incorrect length.");
  for (int i = 1; i < numX; ++i) if (!(xPosition[i] - xPosition[i - 1] > 0.)) return fail(c, "new_SurfaceDescription: positions must be unique, increasing.");
  for (int i = 1; i < numY; ++i) if (!(yPosition[i] - yPosition[i - 1] > 0.)) return fail(c, "new_SurfaceDescription: positions must be unique, increasing.");
  for (size_t i = 0; i < (size_t)(numX - 1) * (numY - 1); ++i)
    if (!(reflectance[i] >= 0.f && reflectance[i] <= 1.f)) return fail(c, "new_SurfaceDescription: surface reflectance must be between 0 and 1");
  if (upload(c, &c->dSurfX, xPosition, (size_t)numX) || upload(c, &c->dSurfY, yPosition, (size_t)numY) ||
      upload(c, &c->dSurfRefl, reflectance, (size_t)(numX - 1) * (numY - 1)))
    return 1;
  c->surfNumX = numX; c->surfNumY = numY;
  return 0;
}

int mcbrat_set_walk_options(mcbrat_ctx *c, int32_t layerSkip, int32_t blockWalk) {
  if (!c) return 1;
  if (layerSkip >= 0) { if (layerSkip != c->layerSkip) c->tuned = false; c->layerSkip = layerSkip > 3 ? 1 : layerSkip; }
  if (blockWalk >= 0) { if (blockWalk != c->blockWalk) c->tuned = false; c->blockWalk = std::min(blockWalk, 2); }
  return 0;
}

double *mcbrat_moments_device_pointer(mcbrat_ctx *c) {
  if (!c || !c->haveGrid) return nullptr;
  (void)hipSetDevice(c->device);
  if (ensure_moments(c)) return nullptr;
  return c->dMoments;
}

int mcbrat_frequency_distribution(mcbrat_ctx *c, uint64_t seed, uint64_t firstDraw, int32_t numLambda, const double *cdf,
                                  int64_t totalPhotons, int64_t *distribution) {
  if (!c) return 1;
  if (numLambda < 1 || !cdf || !distribution || totalPhotons < 0) return fail(c, "getFrequencyDistr: invalid arguments.");
  for (int i = 1; i < numLambda; ++i)
    if (!(cdf[i] >= cdf[i - 1])) return fail(c, "getFrequencyDistr: the power CDF must not decrease.");
  (void)hipSetDevice(c->device);
  if (init_lane(c, 0)) return 1;
  hipStream_t st = c->lane[0].stream;
  if (c->freqCapacity < numLambda) {  // the context's own buffers, grown when a longer CDF arrives (freed with the context)
    HIP_OK(c, hipStreamSynchronize(st));
    if (c->dFreqCdf) { (void)hipFree(c->dFreqCdf); c->dFreqCdf = nullptr; }
    if (c->dFreqCounts) { (void)hipFree(c->dFreqCounts); c->dFreqCounts = nullptr; }
    c->freqCapacity = 0;
    HIP_OK(c, dev_malloc((void **)&c->dFreqCdf, sizeof(double) * numLambda));
    HIP_OK(c, dev_malloc((void **)&c->dFreqCounts, sizeof(unsigned long long) * numLambda));
    c->freqCapacity = numLambda;
  }
  double *dCdf = c->dFreqCdf;
  unsigned long long *dCounts = c->dFreqCounts;
  HIP_OK(c, hipMemcpyAsync(dCdf, cdf, sizeof(double) * numLambda, hipMemcpyHostToDevice, st));
  HIP_OK(c, hipMemsetAsync(dCounts, 0, sizeof(unsigned long long) * numLambda, st));
  if (totalPhotons > 0) {
    const unsigned long long blocks4 = ((unsigned long long)totalPhotons + 3) / 4 + 1;
    const unsigned grid = (unsigned)std::min<unsigned long long>((blocks4 + 255) / 256, (unsigned long long)c->numCUs * 8);
    const size_t lds = numLambda <= 8192 ? sizeof(unsigned) * (size_t)numLambda : 0;
    hipLaunchKernelGGL(frequency_distribution_kernel, dim3(std::max(1u, grid)), dim3(256), lds, st, dCdf, numLambda,
                       (unsigned long long)totalPhotons, (unsigned long long)firstDraw, (uint32_t)seed, (uint32_t)(seed >> 32), dCounts);
    HIP_OK(c, hipGetLastError());
  }
  std::vector<unsigned long long> h(numLambda);
  HIP_OK(c, hipMemcpyAsync(h.data(), dCounts, sizeof(unsigned long long) * numLambda, hipMemcpyDeviceToHost, st));
  HIP_OK(c, hipStreamSynchronize(st));
  for (int i = 0; i < numLambda; ++i) distribution[i] = (int64_t)h[i];
  return 0;
}

int mcbrat_get_walk_mode(const mcbrat_ctx *c) {
  if (!c) return 0;
  int m = (c->layerSkip ? 1 : 0) | (c->blockWalk ? 2 : 0);
  if (c->haveGrid && c->haveOptics) {  // what a flux launch of the loaded domain would do (the plan decides, as launch_trace does)
    const size_t ncol = (size_t)c->nx * c->ny;
    LaunchPlan L = plan_launch(c, 2 * ncol + ncol * c->nz);
    if (L.priv && L.brick) { L.priv = false; L.gridLds = false; }
    m = (c->layerSkip ? 1 : 0) | (block_walk_applies(c, L) ? 2 : 0) | (L.fly ? 4 : 0) | (c->blockWalk ? 8 : 0) |
        (L.wide ? 16 : 0) | ((L.blockLite && L.optics == 1) ? 32 : 0) | (L.priv ? 64 : 0) | ((L.blockLite && L.optics == 2) ? 128 : 0) |
        (L.cdfTop ? 256 : 0);
  }
  return m;
}

int mcbrat_compute_radiative_transfer(mcbrat_ctx *c, uint64_t seed, uint64_t firstPhotonId, int64_t ppb, int32_t nBatches,
                                      int64_t *numPhotonsProcessed) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  if (check_ready(c)) return 1;
  if (ppb < 1 || nBatches < 1) return fail(c, "computeRadiativeTransfer: Didn't process any photons.");
  if (ensure_moments(c)) return 1;
  // the trial launches that choose the event threshold run synchronously, once, on the first call large enough
  const bool tuneNow = c->autoTune && !c->tuned && (unsigned long long)ppb * (unsigned long long)nBatches >= c->tuneTrialPhotons;
  const bool async = c->asyncOn && !c->countersOn;
  if ((!async || tuneNow) && sync_all(c)) return 1;
  if (c->asyncOn) {  // rotate over the lanes
    c->cur = c->nextLane;
    c->nextLane = (c->nextLane + 1) % mcbrat_ctx::kLanes;
    if (init_lane(c, c->cur)) return 1;
  } else {
    c->cur = 0;
  }
  const size_t ncol = (size_t)c->nx * c->ny, nvox = ncol * c->nz;
  // [fluxUp | fluxDown | volume | intensity per direction | (limitIntensityContributions:) intensity by component, excess]
  const size_t slabStride = 2 * ncol + nvox + (size_t)c->nDir * ncol +
                            (c->limitContrib ? (size_t)(c->nc + 1) * c->nDir * (ncol + 1) : 0);
  // batches in flight: bounded by a memory budget (slabs are 8 B per tally bin per batch)
  size_t inFlight = std::max<size_t>(1, (size_t)(4ull << 30) / (slabStride * sizeof(long long)));
  if (c->maxBatchesInFlight > 0) inFlight = std::min<size_t>(inFlight, (size_t)c->maxBatchesInFlight);
  inFlight = std::min<size_t>(inFlight, (size_t)nBatches);
  // asynchronous mode sizes every lane at once: allocation synchronises the device, so it must not recur
  for (int li = 0; li < mcbrat_ctx::kLanes; ++li) {
    mcbrat_ctx::Lane &L = c->lane[li];
    if (li != c->cur && !(c->asyncOn && L.stream)) continue;
    const size_t needSlab = slabStride * inFlight, needCol = 3 * ncol * inFlight, needScal = (size_t)(3 + c->nz) * inFlight;
    if (L.slabCapacity >= needSlab && L.colCapacity >= needCol && L.scalCapacity >= needScal) continue;
    HIP_OK(c, hipStreamSynchronize(L.stream));
    if (L.slabCapacity < needSlab) {
      if (L.dSlabs) (void)hipFree(L.dSlabs);
      L.dSlabs = nullptr; L.slabCapacity = 0;
      HIP_OK(c, dev_malloc((void **)&L.dSlabs, sizeof(long long) * (needSlab + 1)));  // (+1: the launch's photon cursor sits behind the slabs in use, zeroed with them)
      L.slabCapacity = needSlab;
    }
    if (L.colCapacity < needCol) {
      if (L.dColVals) (void)hipFree(L.dColVals);
      L.dColVals = nullptr; L.colCapacity = 0;
      HIP_OK(c, dev_malloc((void **)&L.dColVals, sizeof(float) * needCol));
      L.colCapacity = needCol;
    }
    if (L.scalCapacity < needScal) {
      if (L.dScalVals) (void)hipFree(L.dScalVals);
      L.dScalVals = nullptr; L.scalCapacity = 0;
      HIP_OK(c, dev_malloc((void **)&L.dScalVals, sizeof(float) * needScal));
      L.scalCapacity = needScal;
    }
  }
  if (!c->dLast) HIP_OK(c, dev_malloc((void **)&c->dLast, sizeof(float) * (size_t)moments_len(c)));

  DevParams p;
  fill_params(c, p);
  p.seedLo = (uint32_t)seed; p.seedHi = (uint32_t)(seed >> 32);
  p.slabs = c->L().dSlabs; p.slabStride = slabStride;
  p.ppb = (unsigned long long)ppb;
  p.fates = nullptr;
  p.counters = c->countersOn ? c->dEventCounters : nullptr;
  if (c->countersOn) HIP_OK(c, hipMemsetAsync(c->dEventCounters, 0, 32 * sizeof(unsigned long long), c->L().stream));

  if (c->autoTune && !c->tuned) {
    if (autotune(c, p, (unsigned long long)ppb, (int)inFlight)) return 1;
    p.eventThreshold = c->eventThreshold;
  }

  float traceMs = 0.f;
  for (int b0 = 0; b0 < nBatches; b0 += (int)inFlight) {
    const int nb = std::min<int>((int)inFlight, nBatches - b0);
    p.total = (unsigned long long)ppb * (unsigned long long)nb;
    p.firstPhoton = firstPhotonId + (unsigned long long)b0 * (unsigned long long)ppb;
    HIP_OK(c, hipMemsetAsync(c->L().dSlabs, 0, sizeof(long long) * (slabStride * nb + 1), c->L().stream));  // zero tallies :248-252, and the cursor
    p.counter = reinterpret_cast<unsigned long long *>(c->L().dSlabs + slabStride * nb);
    std::pair<hipEvent_t, hipEvent_t> bracket(c->L().ev0, c->L().ev1);
    if (async && event_pair(c, bracket)) return 1;
    HIP_OK(c, hipEventRecord(bracket.first, c->L().stream));
    if (launch_trace(c, p, c->countersOn, nb)) return 1;
    HIP_OK(c, hipEventRecord(bracket.second, c->L().stream));
    if (async) c->timing.push_back(bracket);
    if (c->timing.size() > 1024) { if (sync_all(c)) return 1; }  // bound the number of live events
    // the finish kernels accumulate into one moment array: they run in call order across lanes, and after
    // whatever a caller's stream did to that array (mcbrat_wait_stream)
    if (c->lastDone && c->lastDone != c->L().evDone) HIP_OK(c, hipStreamWaitEvent(c->L().stream, c->lastDone, 0));
    if (c->externalPending) { HIP_OK(c, hipStreamWaitEvent(c->L().stream, c->evExternal, 0)); c->externalPending = false; }
    if (c->chainAfter) {  // contexts that share one moment array (one per wavelength): this finish chain after the other context's
      if (c->chainSnapshot) HIP_OK(c, hipStreamWaitEvent(c->L().stream, c->chainSnapshot, 0));
      c->chainAfter = false; c->chainSnapshot = nullptr;
    }
    FinishParams f;
    f.nx = c->nx; f.ny = c->ny; f.nz = c->nz; f.nBatches = nb; f.xyRegular = c->xyRegular; f.nDir = c->nDir; f.nc = c->nc; f.limitContrib = c->limitContrib;
    f.ppb = p.ppb; f.total = p.total; f.slabStride = slabStride;
    f.slabs = c->L().dSlabs; f.relArea = c->dRelArea; f.ze = c->dEdges + (c->nx + 1) + (c->ny + 1);
    f.colVals = c->L().dColVals; f.scalVals = c->L().dScalVals; f.moments = c->dMoments; f.last = c->dLast;
    f.bad = c->dBad; f.badHost = c->hBadDev;
    f.gatherColumns = (unsigned)((ncol * (size_t)nb + kFinishBlock - 1) / kFinishBlock);
    f.gatherVolume = (unsigned)((nvox + kVolVox - 1) / kVolVox);
    f.gatherReduce = (unsigned)(3 + c->nz) * (unsigned)nb;
    f.gatherIntensity = (unsigned)((ncol * (size_t)c->nDir + kFinishBlock - 1) / kFinishBlock);
    f.foldColumns = (unsigned)((3 * ncol + kFinishBlock - 1) / kFinishBlock);
    f.foldScalars = (unsigned)((3 + c->nz + kFinishBlock - 1) / kFinishBlock);
    if (c->nDir > 0 && c->limitContrib)
      hipLaunchKernelGGL(finish_excess, dim3(c->nDir, nb), dim3(256), 0, c->L().stream, f);
    hipLaunchKernelGGL(finish_gather, dim3(f.gatherColumns + f.gatherVolume + f.gatherReduce + f.gatherIntensity), dim3(kFinishBlock), 0,
                       c->L().stream, f);
    hipLaunchKernelGGL(finish_fold, dim3(f.foldColumns + f.foldScalars), dim3(kFinishBlock), 0, c->L().stream, f);  // (also hands the dropped-photon count to the host)
    HIP_OK(c, hipGetLastError());
    HIP_OK(c, hipEventRecord(c->L().evDone, c->L().stream));
    c->lastDone = c->L().evDone;
    if (async) continue;
    HIP_OK(c, hipStreamSynchronize(c->L().stream));
    float ms = 0.f;
    HIP_OK(c, hipEventElapsedTime(&ms, c->L().ev0, c->L().ev1));
    traceMs += ms;
  }
  if (!async) { c->lastTraceMs = traceMs; c->lastCounters.badPhotons = (int64_t)*c->hBad; }
  if (c->countersOn) {
    unsigned long long h[16];
    HIP_OK(c, hipMemcpy(h, c->dEventCounters, sizeof(h), hipMemcpyDeviceToHost));
#ifdef MCBRAT_STAMPS  // development aid: wave cycles per section of the tracing loop
    {
      unsigned long long st[16];
      HIP_OK(c, hipMemcpy(st, c->dEventCounters + 16, sizeof(st), hipMemcpyDeviceToHost));
      const char *names[9] = {"launch", "block crossings (block walk)", "collideA(pos,optics,absorb,roulette)", "collideB(angle,cos)", "collideC(next_direct)",
                              "exits(top,surface)", "leg", "walk", "phase-head(jump,column look-up)"};
      double tot = 0;
      for (int i = 0; i < 9; i++) tot += (double)st[i];
      for (int i = 0; i < 9; i++) fprintf(stderr, "stamp %-40s %14llu  %5.1f %%\n", names[i], st[i], 100.0 * (double)st[i] / tot);
    }
#endif
    c->lastCounters = mcbrat_counters{(int64_t)h[0], (int64_t)h[1], (int64_t)h[2], (int64_t)h[3],
                                      (int64_t)h[4], (int64_t)h[5], (int64_t)h[6], (int64_t)h[7],
                                      (int64_t)h[8], (int64_t)h[9], (int64_t)h[10], (int64_t)h[11], (int64_t)h[12], (int64_t)h[13],
                                      (int64_t)*c->hBad};
  }
  c->haveLast = true;
  if (numPhotonsProcessed) *numPhotonsProcessed = ppb * (int64_t)nBatches;
  c->err = "computeRadiativeTransfer: finished with photons";
  return 0;
}

int mcbrat_report_results(mcbrat_ctx *c, float *meanUp, float *meanDown, float *meanAbs, float *fluxUp, float *fluxDown,
                          float *fluxAbs, float *absorbedProfile, float *volumeAbsorption) {
  if (!c) return 1;
  if (!c->haveLast) return fail(c, "reportResults: no batch has been traced yet.");
  (void)hipSetDevice(c->device);
  if (sync_all(c)) return 1;
  const size_t ncol = (size_t)c->nx * c->ny, nvox = ncol * c->nz;
  std::vector<float> h((size_t)moments_len(c));
  HIP_OK(c, hipMemcpy(h.data(), c->dLast, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
  if (meanUp) *meanUp = h[0];
  if (meanDown) *meanDown = h[1];
  if (meanAbs) *meanAbs = h[2];
  if (fluxUp) std::memcpy(fluxUp, &h[3], sizeof(float) * ncol);
  if (fluxDown) std::memcpy(fluxDown, &h[3 + ncol], sizeof(float) * ncol);
  if (fluxAbs) std::memcpy(fluxAbs, &h[3 + 2 * ncol], sizeof(float) * ncol);
  if (absorbedProfile) std::memcpy(absorbedProfile, &h[3 + 3 * ncol], sizeof(float) * c->nz);
  if (volumeAbsorption) std::memcpy(volumeAbsorption, &h[3 + 3 * ncol + c->nz], sizeof(float) * nvox);
  return 0;
}

int mcbrat_trace_fates(mcbrat_ctx *c, uint64_t seed, uint64_t firstPhotonId, int64_t n, mcbrat_fate *fates) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  if (check_ready(c)) return 1;
  if (n < 1 || !fates) return fail(c, "trace_fates: nothing to trace.");
  if (sync_all(c)) return 1;
  if (c->nDir > 0) return fail(c, "trace_fates: not available together with intensity directions.");
  const size_t ncol = (size_t)c->nx * c->ny, nvox = ncol * c->nz;
  const size_t slabStride = 2 * ncol + nvox;
  long long *scratch = nullptr;
  mcbrat_fate *dF = nullptr;
  HIP_OK(c, dev_malloc((void **)&scratch, sizeof(long long) * slabStride));
  HIP_OK(c, dev_malloc((void **)&dF, sizeof(mcbrat_fate) * (size_t)n));
  HIP_OK(c, hipMemsetAsync(scratch, 0, sizeof(long long) * slabStride, c->L().stream));
  HIP_OK(c, hipMemsetAsync(dF, 0xff, sizeof(mcbrat_fate) * (size_t)n, c->L().stream));
  HIP_OK(c, hipMemsetAsync(c->L().dCounter, 0, sizeof(unsigned long long), c->L().stream));
  HIP_OK(c, hipMemsetAsync(c->dEventCounters, 0, 32 * sizeof(unsigned long long), c->L().stream));
  DevParams p;
  fill_params(c, p);
  p.seedLo = (uint32_t)seed; p.seedHi = (uint32_t)(seed >> 32);
  p.slabs = scratch; p.slabStride = slabStride;
  p.ppb = (unsigned long long)n; p.total = (unsigned long long)n; p.firstPhoton = firstPhotonId;
  p.fates = dF; p.counters = c->dEventCounters;
  double *dTrace = nullptr;
  const int traceCap = 4096;
  if (const char *tp = getenv("MCBRAT_TRACE_PHOTON")) {  // development aid: per-collision records of one photon
    HIP_OK(c, dev_malloc((void **)&dTrace, sizeof(double) * 12 * traceCap));
    HIP_OK(c, hipMemsetAsync(dTrace, 0, sizeof(double) * 12 * traceCap, c->L().stream));
    p.traceBuf = dTrace; p.traceIndex = strtoull(tp, nullptr, 10); p.traceCap = traceCap;
  }
  int rc = launch_trace(c, p, true, 1);
  if (!rc && dTrace) {
    std::vector<double> h(12 * traceCap);
    (void)hipStreamSynchronize(c->L().stream);
    (void)hipMemcpy(h.data(), dTrace, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    if (h[0] == 777) {
      for (int r = 0; r < 64 && h[20 * r] == 777; ++r) {  // (-DMCBRAT_STUCK_PROBE: state tcur tnx tny tnz rx ry rz px py pz dx dy dz acc tau ext spans id)
        fprintf(stderr, "STUCK %d:", r);
        for (int i = 1; i < 20; ++i) fprintf(stderr, " %.9g", h[20 * r + i]);
        fprintf(stderr, "\n");
      }
    }
    for (int i = 0; i < traceCap && h[0] != 777; ++i)  // (indexed by scattering order: surface reflections leave gaps)
      if (h[12 * i] > 0) fprintf(stderr, "GPUTRACE %d ev %.0f cell %.0f %.0f %.0f pos %.17g %.17g %.17g dir %.9g %.9g %.9g tau %.9g w %.9g\n", i, h[12 * i],
              h[12 * i + 1], h[12 * i + 2], h[12 * i + 3], h[12 * i + 4], h[12 * i + 5], h[12 * i + 6], h[12 * i + 7], h[12 * i + 8],
              h[12 * i + 9], h[12 * i + 10], h[12 * i + 11]);
    (void)hipFree(dTrace);
  }
  if (!rc) {
    hipError_t e = hipStreamSynchronize(c->L().stream);
    if (e == hipSuccess) e = hipMemcpy(fates, dF, sizeof(mcbrat_fate) * (size_t)n, hipMemcpyDeviceToHost);
    unsigned long long h[16];
    if (e == hipSuccess) e = hipMemcpy(h, c->dEventCounters, sizeof(h), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(c->hBad, c->dBad, sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, std::string("trace_fates: ") + hipGetErrorString(e));
    else c->lastCounters = mcbrat_counters{(int64_t)h[0], (int64_t)h[1], (int64_t)h[2], (int64_t)h[3],
                                           (int64_t)h[4], (int64_t)h[5], (int64_t)h[6], (int64_t)h[7],
                                           (int64_t)h[8], (int64_t)h[9], (int64_t)h[10], (int64_t)h[11], (int64_t)h[12], (int64_t)h[13],
                                           (int64_t)*c->hBad};
  }
  (void)hipFree(scratch);
  (void)hipFree(dF);
  return rc;
}

}  // extern "C"
