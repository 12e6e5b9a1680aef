// mcbrat_kernels.hip -- gfx950 (CDNA4) kernels of the photon-tracing path.
//
// What is replaced: computeRT, Integrators/monteCarloRadiativeTransfer.f95:393-841, and
// the routines it calls per photon (getNextPhoton src/monteCarloIllumination.f95:561,
// accumulateExtinctionAlongPath src/opticalProperties.f95:1656-1815, computeScatteringAngle
// :1594, next_direct :1921, makeDirectionCosines :1876), plus the per-batch epilogue
// (computeRadiativeTransfer :328-364, reportResults :845-1042, driver moments
// Drivers/monteCarloDriver.f95:1023-1050).
//
// Design (MI355X first, not a translation):
//  * one photon per work-item in persistent 64-wide waves.  Idle lanes are refilled in place
//    (regeneration) with ballot/popcount ranks: from a chunk of photon ids the wave took from
//    one global counter, or -- PRIV mode -- from the workgroup's own range via an LDS counter.
//  * the loop is split in two dense halves.  A tight walk loop in which every walking lane
//    crosses exactly one voxel face per iteration, branch-free, runs until fewer than
//    `eventThreshold` lanes of the wave are still walking; then one event phase serves all
//    waiting lanes at once (launch, collision, surface reflection, next leg set-up).
//  * cell-authoritative voxel walk: the distance along the leg to the next x/y/z face is kept
//    per axis in registers (float), only the crossed axis is updated from the edge table in
//    LDS; periodic wrap shifts the leg origin; the position (double, as in the reference) is
//    materialised once per leg.  Statistically, not bitwise, identical to the reference's
//    position-stepping walk with spacing() snaps.
//  * counter-based Philox4x32-10: key = seed, counter = (event, block, photon id).  Roles have
//    fixed slots so that ONE block serves a whole leg in the common case:
//       event 0 (launch)   block 0 = [x, y, -, -]                        (solar source)
//                          block 0 = [select, r1, r2, r3], block 1..   (emission source)
//       event e >= 1       block 0 = [tau, X, Y, Z], block 1 = [component, roulette, -, -]
//            collision: angle = X, next_direct round 0 = (Y, Z), round k>=1 -> block 2+(k-1)/2
//            surface:   mu = sqrt(X) (retry Z, then block 2..), phi = 2 pi Y
//    block 1 is only generated when it is used (nc > 1, or weight < 1/2 with roulette on).
//    The oracle's Philox mode uses the same table (oracle/mcbrat_oracle.c).
//  * cell edges and, when they fit, the inverse phase-function tables are staged in LDS by
//    coalesced loads; extinction / ssa / phase index are float / float / u16 grids in HBM that
//    stay resident in L1/L2/Infinity Cache.
//  * tallies are signed 64-bit fixed point (2^-32): integer sums are order independent, so
//    results are bitwise reproducible and independent of the number of GPUs.  Small domains
//    (PRIV): each workgroup traces photons of ONE batch and tallies into an LDS slab that is
//    flushed once with global atomics; large domains: global atomics into the batch's slab.
//    fluxAbsorbed is the column sum of the volume tally (same deposits, :766-769).
#include <float.h>

#include "mcbrat_device.h"

// Development aid: -DMCBRAT_STAMPS makes the DEBUG instantiation report wave cycles per section of the
// loop in place of the event counters 0..7 (launch, top, collide A/B/C, surface, leg set-up, walk).
#ifdef MCBRAT_STAMPS
#define STAMP(i) do { if (DEBUG && p.counters) { const unsigned long long t_ = clock64(); \
    if (lane == __ffsll((long long)__ballot(1)) - 1) { s_stamp[threadIdx.x >> 6][i] += t_ - s_tprev[threadIdx.x >> 6]; s_tprev[threadIdx.x >> 6] = t_; } } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace mcbrat {

constexpr unsigned kChunk = 256;  // photon ids a wave takes per global atomic (global mode)

// Register budget: minimum waves per SIMD the compiler must leave room for (the second
// argument of __launch_bounds__ on AMD).  Measured choice, see DESIGN.md section 5.
#ifndef MCBRAT_MIN_WAVES_PER_SIMD
#define MCBRAT_MIN_WAVES_PER_SIMD 4
#endif

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;  // v_mad_u64_u32
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    c1 = (uint32_t)p1;
    c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// getRandomReal semantics (src/RandomNumbersForMC.f95:277-301): a float on the closed interval
// [0,1] with 32-bit resolution.  The reference forms u/(2^32-1) in double and rounds to float;
// here u is rounded to float and scaled by 2^-32 (both exact operations, so the CPU oracle
// reproduces it bit for bit); the two maps differ by at most one float ulp, in 0.4 % of draws.
__device__ __forceinline__ float u01(uint32_t u) { return (float)u * 2.3283064365386963e-10f; }

// 1/x and a/b with the hardware reciprocal (1 ulp): geometry set-up and direction algebra
#ifdef MCBRAT_PRECISE_MATH  // A/B switch: correctly rounded division / library log, cos, sqrt everywhere
__device__ __forceinline__ float rcp_fast(float x) { return 1.0f / x; }
__device__ __forceinline__ float div_fast(float a, float b) { return a / b; }
#else
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float div_fast(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#endif

// cos(x) for a scattering angle x in [0, pi]: one quadrant reduction (q = 0, 1, 2; Cody-Waite pi/2)
// and the classic single-precision kernels on |r| <= pi/4 (coefficients of the FreeBSD/msun
// k_cosf / k_sinf minimax polynomials).  About 1 ulp, like the library cosf it replaces, at a
// third of the instructions: the generic routine must handle any argument, this one need not.
__device__ __forceinline__ float cos_0_pi(float x) {
  const float q = rintf(x * 0.636619772f);
  float r = __fmaf_rn(q, -1.57079637050628662109375f, x);  // exact (Sterbenz)
  r = __fmaf_rn(q, 4.371138828673793e-8f, r);
  const float z = r * r;
  const float c = __fmaf_rn(z, __fmaf_rn(z, __fmaf_rn(z, __fmaf_rn(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f),
                                                      4.16666233237390631894e-2f), -0.499999997251031003120f), 1.0f);
  const float s = __fmaf_rn(r * z, __fmaf_rn(z, __fmaf_rn(z, __fmaf_rn(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f),
                                                          8.3333293858894631756e-3f), -0.166666666416265235595f), r);
  return q == 0.0f ? c : (q == 1.0f ? -s : -c);
}

// element i (0..3) of a Philox block without dynamic register indexing
__device__ __forceinline__ uint32_t pick4(const uint32_t (&r)[4], uint32_t i) {
  const uint32_t a = (i & 1u) ? r[1] : r[0], b = (i & 1u) ? r[3] : r[2];
  return (i & 2u) ? b : a;
}

__device__ __forceinline__ unsigned long long to_fixed(double v) {
  return (unsigned long long)__double2ll_rn(v * kTallyScale);
}
// photon weights and deposits lie in [0, 1]: a float times 2^32 is exact, so the conversion needs
// no double arithmetic (1.0 itself does not fit 32 bits and is handled apart)
__device__ __forceinline__ unsigned long long weight_to_fixed(float v) {
  return v >= 1.0f ? (1ull << 32) : (unsigned long long)__float2uint_rn(v * 4294967296.0f);
}

// findIndex(value, table) for cell edges: largest i with e[i] <= v, clamped to [0, n-1]
// (src/numericUtilities.f95:207-260, no first guess).
__device__ __forceinline__ int find_cell(const double *e, int n, double v) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (v >= e[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

// findCDFIndex (src/numericUtilities.f95:317-348) on a strided slice of the running CDF:
// smallest 1-based i with value <= t(i).
__device__ __forceinline__ int find_cdf(const double *t, int n, long long stride, float vf) {
  const double v = (double)vf;
  int lo = 0, hi = n;
  while (!(lo == n || hi <= lo + 1)) {
    const int mid = (lo + hi) >> 1;
    if (v > t[(long long)(mid - 1) * stride]) lo = mid; else hi = mid;
  }
  return hi;
}

// Optical-property lookup.  Dense: one gather from the [nz][ny][nx] grid.  BRICK: the grid is cut
// into 4x4x4 bricks; a brick whose 64 cells all equal their layer's background value (clear air:
// most of a cloud scene) has no storage -- its cells read the per-layer background from LDS --
// and only the other bricks are stored, 64 cells each.  The brick table (4 B per 64 cells) and the
// stored bricks are a fraction of the dense grid, so the walk's working set fits the XCD's 4 MiB
// L2 instead of thrashing it (L2 hit rate 72 % -> measured there).
struct CellRef {
  int dense;   // dense cell index (tallies, and optics when !BRICK)
  int stored;  // BRICK: index into the stored-brick arrays, or -1 for a background cell
};

template <bool BRICK>
__device__ __forceinline__ CellRef locate_cell(const DevParams &p, int ix, int iy, int iz) {
  CellRef r;
  r.dense = ix + p.nx * (iy + p.ny * iz);
  r.stored = -1;
  if (BRICK) {
    const uint32_t base = p.brickTable[(ix >> 2) + p.nbx * ((iy >> 2) + p.nby * (iz >> 2))];
    if (base != 0xffffffffu) r.stored = (int)(base + (uint32_t)((ix & 3) + 4 * ((iy & 3) + 4 * (iz & 3))));
  }
  return r;
}

template <bool BRICK>
__device__ __forceinline__ float load_ext(const DevParams &p, const float *s_bgExt, const CellRef &r, int iz) {
  if (!BRICK) return p.ext[r.dense];
  return r.stored >= 0 ? p.ext[r.stored] : s_bgExt[iz];
}

// PRIV: 0 tallies by global atomics, 1 LDS-private tally slab, 2 private slab and the optical grid in LDS too
// INTEN: radiance by local estimation (computeIntensityContribution :1623-1832): every launch of an emitted
// photon, surface reflection and scattering event sends a contribution along each view direction.
template <int BLOCK, bool TBL_LDS, int PRIV, bool BRICK, bool DEBUG, bool INTEN = false>
__global__ void __launch_bounds__(BLOCK, MCBRAT_MIN_WAVES_PER_SIMD) trace_kernel(const DevParams p) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // LDS map: [edges x|y|z (double)] [private tally slab (i64), PRIV] [unit cursor, PRIV]
  //          [background extinction per layer, BRICK] [optical grid, PRIV == 2]
  //          [tables (float), TBL_LDS]
  double *s_edge = reinterpret_cast<double *>(smem_raw);
  const int nEdges = p.nx + p.ny + p.nz + 3;
  const int ncol = p.nx * p.ny;
  const int slabLen = PRIV ? (int)p.slabStride : 0;
  long long *s_slab = reinterpret_cast<long long *>(s_edge + nEdges);
  unsigned *s_cursor = reinterpret_cast<unsigned *>(s_slab + slabLen);
  float *s_bgExt = reinterpret_cast<float *>(s_cursor + (PRIV ? 4 : 0));  // BRICK: background extinction per layer
  // PRIV (small domains): the whole optical grid is staged in LDS too -- extinction, ssa, phase index
  constexpr bool gridLds = PRIV == 2;
  const int nvoxS = gridLds ? ncol * p.nz : 0;
  float *s_ext = s_bgExt + (BRICK ? ((p.nz + 3) & ~3) : 0);
  float *s_ssa = s_ext + nvoxS;                       // [nc][nvox]
  float *s_cum = s_ssa + (size_t)p.nc * nvoxS;        // [nc][nvox]
  uint16_t *s_pfi = reinterpret_cast<uint16_t *>(s_cum + (size_t)p.nc * nvoxS);  // [nc][nvox]
  float *s_tbl = reinterpret_cast<float *>(s_pfi + (((size_t)p.nc * nvoxS + 1) & ~(size_t)1));
  // per-component table descriptors: indexed by a per-lane component number, so they must not stay in the
  // kernel-argument segment (a vector load from there costs a trip to memory on every collision)
  __shared__ int s_tblOffset[MCBRAT_MAX_COMPONENTS], s_tblNSteps[MCBRAT_MAX_COMPONENTS];
  __shared__ float s_tblInvN[MCBRAT_MAX_COMPONENTS];
  __shared__ int s_fwdOffset[MCBRAT_MAX_COMPONENTS], s_fwdNAngles[MCBRAT_MAX_COMPONENTS];
  if (INTEN && threadIdx.x < MCBRAT_MAX_COMPONENTS) {
    s_fwdOffset[threadIdx.x] = p.fwdOffset[threadIdx.x];
    s_fwdNAngles[threadIdx.x] = p.fwdNAngles[threadIdx.x];
  }
  if (threadIdx.x < MCBRAT_MAX_COMPONENTS) {
    s_tblOffset[threadIdx.x] = p.tblOffset[threadIdx.x];
    s_tblNSteps[threadIdx.x] = p.tblNSteps[threadIdx.x];
    s_tblInvN[threadIdx.x] = p.tblInvN[threadIdx.x];
  }
  for (int i = threadIdx.x; i < nEdges; i += BLOCK) s_edge[i] = p.edges[i];
  if (TBL_LDS)
    for (int i = threadIdx.x; i < p.tblTotalFloats; i += BLOCK) s_tbl[i] = p.tables[i];
  if (PRIV) {
    for (int i = threadIdx.x; i < slabLen; i += BLOCK) s_slab[i] = 0;
    if (threadIdx.x == 0) s_cursor[0] = 0;
  }
  if (BRICK)
    for (int i = threadIdx.x; i < p.nz; i += BLOCK) s_bgExt[i] = p.bgExt[i];
  if (gridLds) {  // coalesced HBM reads, once per workgroup
    for (int i = threadIdx.x; i < nvoxS; i += BLOCK) s_ext[i] = p.ext[i];
    for (int i = threadIdx.x; i < p.nc * nvoxS; i += BLOCK) { s_ssa[i] = p.ssa[i]; s_cum[i] = p.cum[i]; s_pfi[i] = p.pfi[i]; }
  }
  __syncthreads();
  const float *__restrict__ tbl = TBL_LDS ? s_tbl : p.tables;
  const int offY = p.nx + 1, offZ = p.nx + p.ny + 2;  // edge table offsets

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long laneBelow = (1ull << lane) - 1ull;

  // PRIV: this workgroup's unit = photons [unitFirst, unitFirst + unitCount) of ONE batch
  unsigned long long unitFirst = 0;
  unsigned unitCount = 0;
  long long *unitSlab = nullptr;

  // lane state --------------------------------------------------------------------------
  int state = ST_DEAD;
  bool more = true;
  uint32_t idLo = 0, idHi = 0, event = 0, batch = 0;
  double px = 0, py = 0, pz = 0;      // leg origin (in the periodic image the walk is currently in)
  float dx = 0, dy = 0, dz = 1;       // direction cosines
  float ivx = 0, ivy = 0, ivz = 0;    // 1/direction
  float tnx = 0, tny = 0, tnz = 0, tcur = 0;  // distance along the leg to the next x/y/z face
  float acc = 0, tau = 0, w = 0, extCur = 0, uX = 0, uY = 0, uZ = 0;
  int ex = 0, ey = 0, ez = 0;  // index in the LDS edge table of the next x / y / z face of the leg
  int cell = 0;                // linear index of the current cell
  int nScat = 0, nLegs = 0;
  // INTEN: a pending local estimate of this lane: 0 none, 1 Lambertian surface (component 0 of the reference),
  // 2 isotropic emission (component -1), 3 + c scattering by component c; its weight, incoming direction and
  // the offset of the phase-function entry in the forward tables
  int iFlag = 0, iEntry = 0;
  float iW = 0, iDx = 0, iDy = 0, iDz = 0;
  unsigned long long chunkNext = 0, chunkEnd = 0;  // global mode, wave-uniform
  unsigned int cLegs = 0, cCross = 0, cColl = 0, cAbs = 0, cTop = 0, cSurf = 0, cKill = 0, cSurv = 0;
  // DEBUG, wave level (lane 0): loop iterations and how many lanes each kind of phase served
#ifdef MCBRAT_STAMPS
  __shared__ unsigned long long s_tprev[BLOCK / 64], s_stamp[BLOCK / 64][9];
  if (lane == 0) { s_tprev[threadIdx.x >> 6] = clock64(); for (int i = 0; i < 9; i++) s_stamp[threadIdx.x >> 6][i] = 0; }
#endif
  unsigned long long wWalkIters = 0, wWalkLanes = 0, wEventPhases = 0, wEventLanes = 0, wLaunchPhases = 0, wSurfPhases = 0;

  for (unsigned long long unit = blockIdx.x;; unit += gridDim.x) {
    if (PRIV) {  // workgroup-uniform
      if (unit >= p.nUnits) break;
      const unsigned long long b = unit / p.unitsPerBatch, s = unit % p.unitsPerBatch;
      const unsigned long long bp = (p.total - b * p.ppb) < p.ppb ? (p.total - b * p.ppb) : p.ppb;
      const unsigned long long lo = (bp * s) / p.unitsPerBatch, hi = (bp * (s + 1)) / p.unitsPerBatch;
      unitFirst = b * p.ppb + lo;
      unitCount = (unsigned)(hi - lo);
      unitSlab = p.slabs + b * p.slabStride;
      batch = (uint32_t)b;
      more = true;
    }

    for (;;) {
      // ================= event phase: every lane that is not walking =====================
      bool needLeg = false;
      // 0-based cell of a waiting lane: the face ahead minus one where the leg runs forward
      int ix = ex - (dx >= 0.0f ? 1 : 0), iy = ey - offY - (dy >= 0.0f ? 1 : 0), iz = ez - offZ - (dz >= 0.0f ? 1 : 0);
      // Launches and surface reflections are rare next to collisions (1 and ~0.7 per ~17 legs in
      // the I3RC step cloud).  Serving them whenever a single lane asks would run their code at a
      // few percent lane occupancy in almost every event phase, so they wait until enough lanes
      // have queued up -- or until the wave has nothing else to do.
      const unsigned long long mDead = __ballot(state == ST_DEAD && more);
      const unsigned long long mSurf = __ballot(state == ST_SURFACE);
      const int nBusy = __popcll(__ballot(state == ST_WALK || state == ST_COLLIDE || state == ST_TOP));
      const bool doLaunch = __popcll(mDead) >= p.launchThreshold || nBusy < p.eventThreshold;
      const bool doSurface = __popcll(mSurf) >= p.surfaceThreshold || nBusy < p.eventThreshold;
      const unsigned long long want = doLaunch ? mDead : 0ull;
      if (DEBUG) {
        wEventPhases++;
        wEventLanes += __popcll(__ballot(state == ST_COLLIDE || state == ST_TOP));
        if (want) wLaunchPhases++;
        if (doSurface && mSurf) wSurfPhases++;
      }
      STAMP(8);
      if (want != 0ull) {  // wave-uniform
        const int nWant = __popcll(want);
        const int rank = __popcll(want & laneBelow);
        unsigned long long myIdx;  // photon index inside this launch
        bool valid;
        if (PRIV) {
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&s_cursor[0], (unsigned)nWant);
          base = (unsigned)__shfl((int)base, 0);
          const unsigned k = base + (unsigned)rank;
          valid = k < unitCount;
          myIdx = unitFirst + k;
        } else {
          myIdx = chunkNext + (unsigned long long)rank;
          const unsigned long long avail = chunkEnd - chunkNext;
          if (avail < (unsigned long long)nWant) {  // wave-uniform: take a new chunk
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(p.counter, (unsigned long long)kChunk);
            const uint32_t bl = __shfl((int)(uint32_t)base, 0), bh = __shfl((int)(uint32_t)(base >> 32), 0);
            base = ((unsigned long long)bh << 32) | bl;
            if ((unsigned long long)rank >= avail) myIdx = base + ((unsigned long long)rank - avail);
            chunkNext = base + ((unsigned long long)nWant - avail);
            chunkEnd = base + kChunk;
          } else {
            chunkNext += (unsigned long long)nWant;
          }
          valid = myIdx < p.total;
        }
        if (state == ST_DEAD && more) {
          if (valid) {
            // ---- launch: getNextPhoton + computeRT :466-508 --------------------------------
            if (!PRIV) batch = (uint32_t)(myIdx / p.ppb);
            const unsigned long long id = p.firstPhoton + myIdx;
            idLo = (uint32_t)id; idHi = (uint32_t)(id >> 32);
            event = 0; nScat = 0; nLegs = 0;
            uint32_t r[4];
            philox4x32_10(0u, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
            double fx, fy, fz;  // fractional launch position in [0,1]
            if (p.srcKind == 0) {  // newPhotonStream_Directional, monteCarloIllumination.f95:88-96
              fx = (double)u01(r[0]);
              fy = (double)u01(r[1]);
              fz = 0.0;
              dx = p.dir0[0]; dy = p.dir0[1]; dz = p.dir0[2];
            } else {  // newPhotonStream_BBEmission :481-516
              float mu = 0.f, phi = 0.f;
              const float sel = u01(r[0]);
              if ((double)sel > p.fracAtms) {  // surface emission :484-493
                fx = (double)u01(r[1]);
                fy = (double)u01(r[2]);
                fz = 0.0;
                uint32_t r1[4];
                for (uint32_t j = 0;; j++) {
                  if ((j & 3u) == 0) philox4x32_10(0u, 1u + (j >> 2), idLo, idHi, p.seedLo, p.seedHi, r1);
                  mu = sqrtf(u01(pick4(r1, j & 3u)));
                  if (fabsf(mu) > 2.0f * FLT_MIN) break;
                }
                phi = (u01(r[3]) * 2.0f) * 3.14159274f;  // * 2. * acos(-1.)
              } else {  // atmosphere :495-510: one uniform picks level, row and voxel from one running CDF
                const float rn = u01(r[1]);
                const long long nxy = (long long)p.nx * p.ny;
                const int ik = find_cdf(p.voxelCDF + ((long long)p.nx - 1) + (long long)p.nx * (p.ny - 1), p.nz, nxy, rn);
                const int ij = find_cdf(p.voxelCDF + ((long long)p.nx - 1) + nxy * (ik - 1), p.ny, p.nx, rn);
                const int ii = find_cdf(p.voxelCDF + (long long)p.nx * ((ij - 1) + (long long)p.ny * (ik - 1)), p.nx, 1, rn);
                uint32_t r1[4];
                philox4x32_10(0u, 1u, idLo, idHi, p.seedLo, p.seedHi, r1);
                fz = ((double)(ik - 1) * 1.0 / (double)p.nz) + (double)(u01(r[2]) / (float)p.nz);
                if (ik == 1 && fz == 0.0) fz = 2.220446049250313e-16;  // spacing(1.0_8)
                if (ik == p.nz && fz > 1.0 - 2.0 * 2.220446049250313e-16) fz = fz - 2.0 * 2.220446049250313e-16;
                fx = ((double)(ii - 1) * 1.0 / (double)p.nx) + (double)(u01(r[3]) * (1.0f / (float)p.nx));
                fy = ((double)(ij - 1) * 1.0 / (double)p.ny) + (double)(u01(r1[0]) * (1.0f / (float)p.ny));
                uint32_t r2[4];
                for (uint32_t j = 0;; j++) {
                  uint32_t uu;
                  if (j < 2) uu = j ? r1[3] : r1[2];
                  else {
                    if (((j - 2) & 3u) == 0) philox4x32_10(0u, 2u + ((j - 2) >> 2), idLo, idHi, p.seedLo, p.seedHi, r2);
                    uu = pick4(r2, (j - 2) & 3u);
                  }
                  mu = 1.0f - (2.0f * u01(uu));
                  if (fabsf(mu) > 2.0f * FLT_MIN) break;
                }
                phi = (u01(r1[1]) * 2.0f) * 3.14159274f;
              }
              const float sinTheta = sqrtf(1.0f - mu * mu);  // makeDirectionCosines :1876-1894
              dx = sinTheta * cosf(phi); dy = sinTheta * sinf(phi); dz = mu;
            }
            w = 1.0f;
            px = p.x0 + fx * (p.xMax - p.x0);  // :480-482
            py = p.y0 + fy * (p.yMax - p.y0);
            if (p.xyRegular) {  // findXYIndicies :1558-1562
              ix = min((int)((px - p.x0) * p.invDX), p.nx - 1);
              iy = min((int)((py - p.y0) * p.invDY), p.ny - 1);
            } else {
              ix = find_cell(s_edge, p.nx, px);
              iy = find_cell(s_edge + offY, p.ny, py);
            }
            if (p.srcKind == 0) {
              pz = p.zLaunch; iz = p.izLaunch;
            } else if (p.zRegular) {  // :485-486
              pz = p.z0 + fz * (p.zMax - p.z0);
              iz = min((int)((pz - p.z0) / ((p.zMax - p.z0) / (double)p.nz)), p.nz - 1);
            } else {  // :491-493 layer-index fraction
              const double t = (fz - p.z0) * (double)p.nz;
              const double fl = floor(t);
              iz = min((int)fl, p.nz - 1);
              pz = s_edge[offZ + iz] + (t - fl) * (s_edge[offZ + iz + 1] - s_edge[offZ + iz]);
            }
            cell = ix + p.nx * (iy + p.ny * iz);
            extCur = gridLds ? s_ext[cell] : load_ext<BRICK>(p, s_bgExt, locate_cell<BRICK>(p, ix, iy, iz), iz);
            if (p.lwFlag && pz > 0.0) {  // :504-508 emission counts as negative absorption
              if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + 2 * ncol + cell), to_fixed(-1.0));
              else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + 2 * ncol + cell), to_fixed(-1.0));
            }
            if (INTEN && p.lwFlag) { iFlag = pz == 0.0 ? 1 : 2; iW = w; }  // :510-541 emission seen directly
            needLeg = true;
          } else {
            more = false;
          }
        }
      }
      STAMP(0);
      // ---- deferred events -------------------------------------------------------------
      if (state == ST_TOP) {
        // out the top, computeRT :573-617: tally and free the lane
        const unsigned long long dep = weight_to_fixed(w);
        if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + (ix + p.nx * iy)), dep);
        else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + (ix + p.nx * iy)), dep);
        if (DEBUG) {
          cTop++;
          if (p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{0, ix + 1, iy + 1, p.nz + 1, nScat, nLegs, w};
        }
        state = ST_DEAD;
      }
      STAMP(1);
      if (state == ST_COLLIDE) {
        // scattering event, computeRT :703-821 (the zero-extinction back-step :728-754 cannot
        // arise: a collision is only declared inside a cell with extinction > 0)
        {  // opticalProperties.f95:1729-1738: the point inside the cell where tau is used up
          const double s = (double)(tcur + div_fast(tau - acc, extCur));
          px = px + s * (double)dx;
          py = py + s * (double)dy;
          pz = pz + s * (double)dz;
        }
        if (DEBUG && p.traceBuf && ((((unsigned long long)idHi << 32) | idLo) - p.firstPhoton) == p.traceIndex && nScat < p.traceCap) {
          double *t = p.traceBuf + 12 * nScat;  // one record per collision of the traced photon
          t[0] = event; t[1] = ix + 1; t[2] = iy + 1; t[3] = iz + 1; t[4] = px; t[5] = py; t[6] = pz;
          t[7] = dx; t[8] = dy; t[9] = dz; t[10] = tau; t[11] = w;
        }
        nScat++;
        if (DEBUG) cColl++;
        CellRef cr;
        cr.dense = cell;
        cr.stored = -1;
        if (BRICK) cr = locate_cell<BRICK>(p, ix, iy, iz);
        // optics of this cell: dense grid, stored brick, or the layer's background record
        const long long nvox = BRICK ? (cr.stored >= 0 ? p.nStored : (long long)p.nz) : (long long)ncol * p.nz;
        const int oc = BRICK ? (cr.stored >= 0 ? cr.stored : iz) : cell;
        const float *cumA = gridLds ? s_cum : ((BRICK && cr.stored < 0) ? p.bgCum : p.cum);
        const float *ssaA = gridLds ? s_ssa : ((BRICK && cr.stored < 0) ? p.bgSsa : p.ssa);
        const uint16_t *pfiA = gridLds ? s_pfi : ((BRICK && cr.stored < 0) ? p.bgPfi : p.pfi);
        uint32_t r1[4] = {0u, 0u, 0u, 0u};
        bool haveR1 = false;
        int c = 0;  // component pick :759-760 (findIndex over [0, cumExt(:)])
        if (p.nc > 1) {
          philox4x32_10(event, 1u, idLo, idHi, p.seedLo, p.seedHi, r1);
          haveR1 = true;
          const float uA = u01(r1[0]);
          for (int k = 0; k < p.nc - 1; k++)
            if (uA >= cumA[(long long)k * nvox + oc]) c = k + 1;
        }
        const float ssa = ssaA[(long long)c * nvox + oc];
        if (ssa < 1.0f) {  // absorption :765-771
          const unsigned long long dep = weight_to_fixed(w * (1.0f - ssa));
          if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + 2 * ncol + cell), dep);
          else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + 2 * ncol + cell), dep);
          w = w * ssa;
          if (DEBUG) cAbs++;
        }
        if (INTEN) {  // :776-790: weight after the single scattering albedo, before roulette; incoming direction
          iFlag = 3 + c; iW = w; iDx = dx; iDy = dy; iDz = dz;
          iEntry = s_fwdOffset[c] + (int)pfiA[(long long)c * nvox + oc] * s_fwdNAngles[c];
        }
        if (p.useRR && w < 0.5f) {  // Russian roulette :805-811, RussianRouletteW = 1
          if (!haveR1) philox4x32_10(event, 1u, idLo, idHi, p.seedLo, p.seedHi, r1);
          if (u01(r1[1]) >= w) { w = 0.0f; if (DEBUG) cKill++; }
          else { w = 1.0f; if (DEBUG) cSurv++; }
        }
        STAMP(2);
        if (w <= FLT_MIN) {  // :812
          if (DEBUG && p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{2, ix + 1, iy + 1, iz + 1, nScat, nLegs, 0.0f};
          state = ST_DEAD;
        } else {
          // computeScatteringAngle :1594-1621 (table point count N, floor-type lookup as written)
          const int pf = pfiA[(long long)c * nvox + oc];
          const int n = s_tblNSteps[c];
          const float *t = tbl + s_tblOffset[c] + (long long)pf * n;
          const int ai = (int)(uX * (float)n) + 1;
          float ang;
          if (ai < n) {
            #ifdef MCBRAT_PRECISE_MATH
            const float left = uX - (float)(ai - 1) / (float)n;
#else
            const float left = uX - (float)(ai - 1) * s_tblInvN[c];
#endif
            ang = (1.0f - left) * t[ai - 1] + left * t[ai];
          } else {
            ang = t[n - 1];
          }
#ifdef MCBRAT_PRECISE_MATH
          const float cs = cosf(ang);
#else
          const float cs = cos_0_pi(ang);
#endif
          STAMP(3);
          // next_direct :1921-1948
          float AX = 1.0f - 2.0f * uY, AY = 1.0f - 2.0f * uZ;
          float D = AX * AX + AY * AY;
          if (D > 1.0f) {
            uint32_t r[4];
            for (uint32_t k = 0; D > 1.0f; k++) {
              if ((k & 1u) == 0) philox4x32_10(event, 2u + (k >> 1), idLo, idHi, p.seedLo, p.seedHi, r);
              AX = 1.0f - 2.0f * u01((k & 1u) ? r[2] : r[0]);
              AY = 1.0f - 2.0f * u01((k & 1u) ? r[3] : r[1]);
              D = AX * AX + AY * AY;
            }
          }
#ifdef MCBRAT_PRECISE_MATH
          float B = sqrtf(div_fast(1.0f - cs * cs, D));
#else
          float B = __builtin_amdgcn_sqrtf(div_fast(1.0f - cs * cs, D));
#endif
          AX = AX * B;
          AY = AY * B;
          B = dx * AX - dy * AY;
          D = cs - div_fast(B, 1.0f + fabsf(dz));
          dx = dx * D + AX;
          dy = dy * D - AY;
          dz = dz * cs - copysignf(fabsf(B), dz * B);
          needLeg = true;
          STAMP(4);
        }
      }
      STAMP(4);
      if (state == ST_SURFACE && doSurface) {
        // surface, computeRT :619-676 (Lambertian); fluxDown gets the incident weight :634
        px = px + (double)tcur * (double)dx;  // where the leg met z0 (:1809-1812)
        py = py + (double)tcur * (double)dy;
        pz = p.zSurf;
        iz = 0;
        cell = ix + p.nx * iy;
        const unsigned long long dep = weight_to_fixed(w);
        if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + ncol + (ix + p.nx * iy)), dep);
        else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + ncol + (ix + p.nx * iy)), dep);
        nScat++;
        if (DEBUG) cSurf++;
        float mu = sqrtf(uX);
        if (!(fabsf(mu) > 2.0f * FLT_MIN)) {
          mu = sqrtf(uZ);
          uint32_t r[4];
          for (uint32_t j = 0; !(fabsf(mu) > 2.0f * FLT_MIN); j++) {
            if ((j & 3u) == 0) philox4x32_10(event, 2u + (j >> 2), idLo, idHi, p.seedLo, p.seedHi, r);
            mu = sqrtf(u01(pick4(r, j & 3u)));
          }
        }
        const float phi = (2.0f * 3.14159274f) * uY;
        const float wIn = w;
        w = (float)((double)w * (double)p.albedo);  // :673
        if (w <= FLT_MIN) {
          if (DEBUG && p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{1, ix + 1, iy + 1, 1, nScat, nLegs, wIn};
          state = ST_DEAD;
        } else {
          const float sinTheta = sqrtf(1.0f - mu * mu);
          dx = sinTheta * cosf(phi); dy = sinTheta * sinf(phi); dz = mu;
          needLeg = true;
          if (INTEN) { iFlag = 1; iW = w; }  // :680-702 reflected weight, Lambertian
        }
      }
      // ---- local estimates: one ray per view direction from every lane with a pending contribution ----
      // All rays of a direction are parallel, so the lanes of the wave walk in lockstep over similar path
      // lengths; 1/direction and the step signs are wave-uniform scalars.
      if (INTEN) {
        if (__ballot(iFlag != 0) != 0ull) {
          constexpr float kPi = 3.14159265358979312f;
          for (int d = 0; d < p.nDir; ++d) {
            const float *dd = p.dirData + 8 * d;  // uniform address: scalar loads
            const float ddx = dd[0], ddy = dd[1], ddz = dd[2], fourPiMu = dd[3], jvx = dd[4], jvy = dd[5], jvz = dd[6];
            const bool fX = ddx >= 0.0f, fY = ddy >= 0.0f, fZ = ddz >= 0.0f;
            bool act = iFlag != 0;
            float npf = 0.0f;  // normalizedPhaseFunc :1685-1727
            if (iFlag == 1) npf = 1.0f / kPi;
            else if (iFlag == 2) npf = 1.0f / fourPiMu;
            else if (iFlag >= 3) {
              float proj = iDx * ddx + iDy * ddy + iDz * ddz;
              if (fabsf(proj) > 1.0f) proj = copysignf(1.0f, proj);
              const float ang = acosf(proj);
              const int nA = s_fwdNAngles[iFlag - 3];
              const float *t = ((p.useHybrid && nScat <= p.numOrdersOrig) ? p.fwdOrig : p.fwdTables) + iEntry;
              const float deltaTheta = kPi / (float)(nA - 1);  // lookUpPhaseFuncValsFromTable :1835-1870
              const int ai = (int)(ang / deltaTheta) + 1;
              float val;
              if (ai < nA) {
                const float wt = 1.0f - (ang - (float)(ai - 1) * deltaTheta) / deltaTheta;
                val = wt * t[ai - 1] + (1.0f - wt) * t[ai];
              } else {
                val = t[nA - 1];
              }
              npf = val / fourPiMu;
            }
            // Iwabuchi (2006) roulette :1753-1813: phase 2 = free path tauFree, phase 1 = tauMax first
            float limit = FLT_MAX, tauFree = 0.0f, u2 = 0.0f;
            int phase = 0;
            if (p.useRRIntensity && act) {
              uint32_t r[4];
              philox4x32_10(event, 0x100u + (uint32_t)d, idLo, idHi, p.seedLo, p.seedHi, r);
              tauFree = -logf(fmaxf(FLT_MIN, u01(r[0])));
              u2 = u01(r[1]);
              if (kPi * npf <= p.zetaMin) { phase = 2; limit = tauFree; }
              else { phase = 1; limit = -logf(p.zetaMin / fmaxf(FLT_MIN, kPi * npf)); }
            }
            // ray state (the photon's own walk state is left alone)
            double qx = px, qy = py;
            int rex = ix + (fX ? 1 : 0), rey = offY + iy + (fY ? 1 : 0), rez = offZ + iz + (fZ ? 1 : 0), rcell = cell;
            float rtx = jvx != 0.0f ? (float)(s_edge[act ? rex : 0] - qx) * jvx : FLT_MAX;
            float rty = jvy != 0.0f ? (float)(s_edge[act ? rey : 0] - qy) * jvy : FLT_MAX;
            float rtz = jvz != 0.0f ? (float)(s_edge[act ? rez : 0] - pz) * jvz : FLT_MAX;
            float rcur = 0.0f, racc = 0.0f;
            float rext = 0.0f;
            if (act) rext = gridLds ? s_ext[rcell] : p.ext[rcell];
            bool outTop = false, outBottom = false;
            while (__ballot(act) != 0ull) {
              if (act) {
                const bool yLtX = rty < rtx;
                const float m2 = yLtX ? rty : rtx;
                const bool isZ = rtz < m2;
                const float tmin = isZ ? rtz : m2;
                const float accNew = racc + (tmin - rcur) * rext;
                if (accNew > limit) {  // the optical depth limit is reached inside this cell
                  if (phase == 1) {    // :1790-1796 continue from the stop point with a fresh free path
                    rcur = rcur + (limit - racc) / rext;
                    racc = 0.0f; limit = tauFree; phase = 2;
                  } else {
                    act = false;
                  }
                } else {
                  racc = accNew;
                  rcur = tmin;
                  if (isZ) {
                    rez += fZ ? 1 : -1;
                    if (rez > offZ + p.nz) { act = false; outTop = true; }
                    else if (rez < offZ) { act = false; outBottom = true; }
                    else rcell += fZ ? ncol : -ncol;
                  } else if (yLtX) {
                    rey += fY ? 1 : -1;
                    rcell += fY ? p.nx : -p.nx;
                    if (rey > offY + p.ny) { rey = offY + 1; rcell -= ncol; qy -= p.Ly; }
                    else if (rey < offY) { rey = offY + p.ny - 1; rcell += ncol; qy += p.Ly; }
                  } else {
                    rex += fX ? 1 : -1;
                    rcell += fX ? 1 : -1;
                    if (rex > p.nx) { rex = 1; rcell -= p.nx; qx -= p.Lx; }
                    else if (rex < 0) { rex = p.nx - 1; rcell += p.nx; qx += p.Lx; }
                  }
                  if (act) {
                    rext = gridLds ? s_ext[rcell] : p.ext[rcell];
                    const int eSel = isZ ? rez : (yLtX ? rey : rex);
                    const double origin = isZ ? pz : (yLtX ? qy : qx);
                    const float jv = isZ ? jvz : (yLtX ? jvy : jvx);
                    const float tNew = (float)(s_edge[eSel] - origin) * jv;
                    rtz = isZ ? tNew : rtz;
                    rty = (!isZ && yLtX) ? tNew : rty;
                    rtx = (!isZ && !yLtX) ? tNew : rtx;
                  }
                }
              }
            }
            if (iFlag != 0) {
              float contrib = 0.0f;
              if (!p.useRRIntensity) {  // :1745-1752 transmission to the boundary the ray leaves through
                contrib = (iW * npf) * expf(-racc);
              } else if (kPi * npf <= p.zetaMin) {  // :1760-1775
                if (u2 <= kPi * npf / p.zetaMin && outTop) contrib = iW * p.zetaMin / kPi;
              } else if (outTop) {  // :1782-1806
                contrib = phase == 1 ? (iW * npf) * expf(-racc) : iW * p.zetaMin / kPi;
              }
              if (contrib != 0.0f && (outTop || outBottom)) {  // (a truncated Legendre series can go negative: tallied as is)
                const int col = outTop ? rcell - ncol * (p.nz - 1) : rcell;
                const unsigned long long dep = (unsigned long long)__double2ll_rn((double)contrib * kTallyScale);
                const long long bin = 2LL * ncol + (long long)ncol * p.nz + (long long)d * ncol + col;
                if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + bin), dep);
                else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + bin), dep);
              }
            }
          }
          iFlag = 0;
        }
      }
      STAMP(5);
      // ---- start the next leg: tau and the per-axis face distances -------------------------
      if (needLeg) {
        event++;
        nLegs++;
        if (DEBUG) cLegs++;
        uint32_t r[4];
        philox4x32_10(event, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
#ifdef MCBRAT_PRECISE_MATH
        tau = -logf(fmaxf(FLT_MIN, u01(r[0])));
#else
        tau = -0.693147182f * __builtin_amdgcn_logf(fmaxf(FLT_MIN, u01(r[0])));
#endif
         // :554 (hardware log2 * ln 2; u >= 2^-32, no denormals)
        uX = u01(r[1]); uY = u01(r[2]); uZ = u01(r[3]);
        acc = 0.0f; tcur = 0.0f;
        // opticalProperties.f95:1690-1712: side 1 where direction >= 0; huge step for a zero cosine
        ex = ix + (dx >= 0.0f ? 1 : 0);
        ey = offY + iy + (dy >= 0.0f ? 1 : 0);
        ez = offZ + iz + (dz >= 0.0f ? 1 : 0);
        if (fabsf(dx) >= 2.0f * FLT_MIN) { ivx = rcp_fast(dx); tnx = (float)(s_edge[ex] - px) * ivx; }
        else { ivx = 0.0f; tnx = FLT_MAX; }
        if (fabsf(dy) >= 2.0f * FLT_MIN) { ivy = rcp_fast(dy); tny = (float)(s_edge[ey] - py) * ivy; }
        else { ivy = 0.0f; tny = FLT_MAX; }
        if (fabsf(dz) >= 2.0f * FLT_MIN) { ivz = rcp_fast(dz); tnz = (float)(s_edge[ez] - pz) * ivz; }
        else { ivz = 0.0f; tnz = FLT_MAX; }
        state = ST_WALK;
      }
      // wave-uniform exit: nothing alive and every lane has already been refused a new photon.
      // (A lane that died in THIS phase still has `more` set: it gets its refill attempt next time.)
      STAMP(6);
      if (__ballot(state != ST_DEAD || more) == 0ull) break;

      // ================= walk phase: one voxel face per iteration ===========================
      // accumulateExtinctionAlongPath :1697-1814.  Per axis the lane holds the index of the NEXT face
      // in the LDS edge table (ex, ey, ez) and the distance to it; `cell` is the linear cell index.
      // The crossed axis is served by its own short predicated block (add, bounds test, one LDS read),
      // so no three-way selects over the axes are needed.
      int nWalk;
      do {
        if (DEBUG) { wWalkIters++; wWalkLanes += __popcll(__ballot(state == ST_WALK)); }
        if (state == ST_WALK) {
          const bool yLtX = tny < tnx;
          const float m2 = yLtX ? tny : tnx;
          const bool isZ = tnz < m2;
          const float tmin = isZ ? tnz : m2;
          const float accNew = acc + (tmin - tcur) * extCur;  // :1743
          if (accNew > tau) {
            state = ST_COLLIDE;  // :1729-1738: the stop point inside this cell is resolved in the event phase
          } else {
            acc = accNew;
            tcur = tmin;
            if (DEBUG) cCross++;
            // integer state of the crossed axis (short predicated blocks) ...
            if (isZ) {
              ez += dz >= 0.0f ? 1 : -1;
              if (ez > offZ + p.nz) state = ST_TOP;        // :1801-1812; tallied in the event phase
              else if (ez < offZ) state = ST_SURFACE;
              else cell += dz >= 0.0f ? ncol : -ncol;
            } else if (yLtX) {
              ey += dy >= 0.0f ? 1 : -1;
              cell += dy >= 0.0f ? p.nx : -p.nx;
              if (ey > offY + p.ny) { ey = offY + 1; cell -= ncol; py -= p.Ly; }          // periodic y :1782-1796:
              else if (ey < offY) { ey = offY + p.ny - 1; cell += ncol; py += p.Ly; }     // continue in the next image
            } else {
              ex += dx >= 0.0f ? 1 : -1;
              cell += dx >= 0.0f ? 1 : -1;
              if (ex > p.nx) { ex = 1; cell -= p.nx; px -= p.Lx; }
              else if (ex < 0) { ex = p.nx - 1; cell += p.nx; px += p.Lx; }
            }
            // ... then ONE extinction read and ONE edge read for whichever axis it was (a lane that just
            // left through the top or the surface reads a valid, unused entry)
            if (gridLds) extCur = s_ext[cell];
            else if (!BRICK) extCur = p.ext[cell];
            else {
              const int bz = min(max(ez - offZ - (dz >= 0.0f ? 1 : 0), 0), p.nz - 1);
              extCur = load_ext<BRICK>(p, s_bgExt, locate_cell<BRICK>(p, ex - (dx >= 0.0f ? 1 : 0), ey - offY - (dy >= 0.0f ? 1 : 0), bz), bz);
            }
            const int eSel = isZ ? min(max(ez, offZ), offZ + p.nz) : (yLtX ? ey : ex);
            const double edge = s_edge[eSel];
            const double origin = isZ ? pz : (yLtX ? py : px);
            const float iv = isZ ? ivz : (yLtX ? ivy : ivx);
            const float tNew = (float)(edge - origin) * iv;
            tnz = isZ ? tNew : tnz;
            tny = (!isZ && yLtX) ? tNew : tny;
            tnx = (!isZ && !yLtX) ? tNew : tnx;
          }
        }
        nWalk = __popcll(__ballot(state == ST_WALK));
      } while (nWalk >= p.eventThreshold);
      STAMP(7);
    }

    if (!PRIV) break;
    // PRIV: flush this unit's private tallies into the batch slab, once
    __syncthreads();
    for (int i = threadIdx.x; i < slabLen; i += BLOCK) {
      const long long v = s_slab[i];
      if (v != 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(unitSlab + i), (unsigned long long)v);
        s_slab[i] = 0;
      }
    }
    if (threadIdx.x == 0) s_cursor[0] = 0;
    __syncthreads();
  }

#ifdef MCBRAT_STAMPS
  if (DEBUG && p.counters && lane == 0) for (int i = 0; i < 9; i++) atomicAdd(p.counters + 16 + i, s_stamp[threadIdx.x >> 6][i]);
#endif
  if (DEBUG && p.counters) {
    atomicAdd(p.counters + 0, (unsigned long long)cLegs);
    atomicAdd(p.counters + 1, (unsigned long long)cCross);
    atomicAdd(p.counters + 2, (unsigned long long)cColl);
    atomicAdd(p.counters + 3, (unsigned long long)cAbs);
    atomicAdd(p.counters + 4, (unsigned long long)cTop);
    atomicAdd(p.counters + 5, (unsigned long long)cSurf);
    atomicAdd(p.counters + 6, (unsigned long long)cKill);
    atomicAdd(p.counters + 7, (unsigned long long)cSurv);
    if (lane == 0) {
      atomicAdd(p.counters + 8, wWalkIters); atomicAdd(p.counters + 9, wWalkLanes);
      atomicAdd(p.counters + 10, wEventPhases); atomicAdd(p.counters + 11, wEventLanes);
      atomicAdd(p.counters + 12, wLaunchPhases); atomicAdd(p.counters + 13, wSurfPhases);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Per-batch epilogue on the device: normalisation (computeRadiativeTransfer :328-364),
// reportResults (:877-884, :966) and the driver's moments (monteCarloDriver.f95:1023-1050).
// Batches are folded in index order by one owner thread per output element, so the
// double sums are reproducible.
// ---------------------------------------------------------------------------------------
struct FinishParams {
  int nx, ny, nz, nBatches, xyRegular, nDir;
  unsigned long long ppb, total, slabStride;
  const long long *slabs;
  const float *relArea;   // [ncol] relative column area (irregular xy)
  const double *ze;       // device edges z
  float *colVals;         // [nBatches][3][ncol] scratch: normalised fluxUp/Down/Absorbed per batch
  float *scalVals;        // [nBatches][3+nz] scratch: means and absorption profile per batch
  double *moments;        // header(8) + S1[M] + S2[M]
  float *last;            // normalised results of the last batch: [3 + 3 ncol + nz + nvox]
};

__device__ __forceinline__ float photons_per_column(const FinishParams &f, int col, unsigned long long n) {
  const int ncol = f.nx * f.ny;
  return f.xyRegular ? (float)(long long)n / (float)ncol : f.relArea[col] * (float)(long long)n;  // :331, :342
}
__device__ __forceinline__ unsigned long long batch_photons(const FinishParams &f, int b) {
  const unsigned long long start = (unsigned long long)b * f.ppb;
  return (f.total - start) < f.ppb ? (f.total - start) : f.ppb;
}

// one thread per (batch, column): normalised column fluxes of every batch (:348-350)
__global__ void finish_columns(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)ncol * f.nBatches) return;
  const int col = (int)(i % ncol), b = (int)(i / ncol);
  const unsigned long long n = batch_photons(f, b);
  const float nppc = photons_per_column(f, col, n);
  const long long *slab = f.slabs + (unsigned long long)b * f.slabStride;
  long long rawAbs = 0;
  for (int k = 0; k < f.nz; k++) rawAbs += slab[2 * ncol + col + (long long)ncol * k];
  const long long raw[3] = {slab[col], slab[ncol + col], rawAbs};
  for (int q = 0; q < 3; q++)
    f.colVals[((long long)b * 3 + q) * ncol + col] = (float)((double)raw[q] * kTallyInv) / nppc;
}

// one thread per column flux element: fold batches in order into S1/S2
__global__ void finish_column_moments(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long M = 3 + 3LL * ncol + f.nz + (long long)ncol * f.nz + (long long)f.nDir * ncol;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * ncol) return;
  const int q = i / ncol, col = i % ncol;
  double s1 = 0, s2 = 0;
  float lastv = 0;
  for (int b = 0; b < f.nBatches; b++) {
    const double n = (double)(long long)batch_photons(f, b);
    const float v = f.colVals[((long long)b * 3 + q) * ncol + col];
    s1 += (double)v * n;
    s2 += n * ((double)v * (double)v);
    lastv = v;
  }
  f.moments[8 + 3 + i] += s1;
  f.moments[8 + M + 3 + i] += s2;
  f.last[3 + i] = lastv;
}

__global__ void finish_volume(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long nvox = (long long)ncol * f.nz;
  const long long M = 3 + 3LL * ncol + f.nz + nvox + (long long)f.nDir * ncol;
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const int col = (int)(v % ncol), k = (int)(v / ncol);
  const double dz = f.ze[k + 1] - f.ze[k];
  double s1 = 0, s2 = 0;
  float lastv = 0;
  for (int b = 0; b < f.nBatches; b++) {
    const unsigned long long n = batch_photons(f, b);
    const float nppc = photons_per_column(f, col, n);
    const long long raw = f.slabs[(unsigned long long)b * f.slabStride + 2 * ncol + v];
    const float x = (float)(((double)raw * kTallyInv) / (((double)nppc * dz) * 1000.0));  // :361-364
    s1 += (double)x * (double)(long long)n;
    s2 += (double)(long long)n * ((double)x * (double)x);
    lastv = x;
  }
  f.moments[8 + 3 + 3LL * ncol + f.nz + v] += s1;
  f.moments[8 + M + 3 + 3LL * ncol + f.nz + v] += s2;
  f.last[3 + 3LL * ncol + f.nz + v] = lastv;
}

// intensity(:, :, d) / numPhotonsPerColumn (:369-371), then the driver's RadianceStats moments
__global__ void finish_intensity(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long nvox = (long long)ncol * f.nz;
  const long long M = 3 + 3LL * ncol + f.nz + nvox + (long long)f.nDir * ncol;
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (long long)ncol * f.nDir) return;
  const int col = (int)(v % ncol);
  double s1 = 0, s2 = 0;
  float lastv = 0;
  for (int b = 0; b < f.nBatches; b++) {
    const unsigned long long n = batch_photons(f, b);
    const float nppc = photons_per_column(f, col, n);
    const long long raw = f.slabs[(unsigned long long)b * f.slabStride + 2 * ncol + nvox + v];
    const float x = (float)((double)raw * kTallyInv) / nppc;
    s1 += (double)x * (double)(long long)n;
    s2 += (double)(long long)n * ((double)x * (double)x);
    lastv = x;
  }
  const long long off = 3 + 3LL * ncol + f.nz + nvox + v;
  f.moments[8 + off] += s1;
  f.moments[8 + M + off] += s2;
  f.last[off] = lastv;
}

// One block per (batch, quantity): quantity 0..2 = domain-mean fluxes (reportResults :881-884),
// 3.. = absorbedProfile(k) (:966).  Fixed-shape tree reduction in float.
__global__ void finish_reduce(const FinishParams f) {
  __shared__ float red[256];
  const int ncol = f.nx * f.ny;
  const int b = blockIdx.y, q = blockIdx.x;
  float s = 0.0f;
  if (q < 3) {
    const float *v = f.colVals + ((long long)b * 3 + q) * ncol;
    for (int c = threadIdx.x; c < ncol; c += blockDim.x) s += v[c];
  } else {
    const int k = q - 3;
    const unsigned long long n = batch_photons(f, b);
    const double dz = f.ze[k + 1] - f.ze[k];
    const long long *slab = f.slabs + (unsigned long long)b * f.slabStride + 2 * ncol + (long long)ncol * k;
    for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
      const float nppc = photons_per_column(f, c, n);
      s += (float)(((double)slab[c] * kTallyInv) / (((double)nppc * dz) * 1000.0));
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) f.scalVals[(long long)b * (3 + f.nz) + q] = red[0] / (float)ncol;
}

__global__ void finish_scalars(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long M = 3 + 3LL * ncol + f.nz + (long long)ncol * f.nz + (long long)f.nDir * ncol;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0) {  // header: photons and batches done
    f.moments[0] += (double)(long long)f.total;
    f.moments[1] += (double)f.nBatches;
  }
  if (q >= 3 + f.nz) return;
  double s1 = 0, s2 = 0;
  float lastv = 0;
  for (int b = 0; b < f.nBatches; b++) {
    const unsigned long long n = batch_photons(f, b);
    const float x = f.scalVals[(long long)b * (3 + f.nz) + q];
    s1 += (double)x * (double)(long long)n;
    s2 += (double)(long long)n * ((double)x * (double)x);
    lastv = x;
  }
  const long long off = q < 3 ? q : 3 + 3LL * ncol + (q - 3);
  f.moments[8 + off] += s1;
  f.moments[8 + M + off] += s2;
  f.last[off] = lastv;
}

}  // namespace mcbrat
