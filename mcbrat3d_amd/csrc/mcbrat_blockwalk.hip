// mcbrat_blockwalk.hip -- the photon loop for domains whose optical grid is resident in LDS, walking from BLOCK
// face to block face.
//
// What is replaced: the same computeRT loop as mcbrat_kernels.hip (Integrators/monteCarloRadiativeTransfer.f95:393-841,
// accumulateExtinctionAlongPath src/opticalProperties.f95:1656-1815).  The reference's walk stops at every cell face
// to add `segment x extinction of the cell` (:1743).  Where neighbouring cells carry the SAME extinction those stops
// add the same product in pieces: the I3RC step cloud is two homogeneous slabs side by side, a plane-parallel
// medium is one, and a leg of either crosses 3-4 cell faces on average before it collides.  Here the host cuts the
// grid into axis-aligned blocks of cells with one extinction value (mcbrat_api.hip: build_blocks) and a leg goes from
// block face to block face; the cell of a collision (tallies, single-scattering albedo, phase function entry) and
// the column of an exit come from the POSITION.  Same physics, the optical depth of a leg summed per block instead
// of per cell (float rounding, as with the layer-skipping walk of the large-domain kernel).
//
// Why it is a kernel of its own and not an option of trace_kernel: with 3-4 cell faces per leg trace_kernel
// alternates between a walk loop and an event phase, each at about half lane occupancy.  With blocks nearly every
// leg of the step cloud ends in the block it starts in, so there is no walk loop left: ONE loop in which every
// lane that has a collision pending is served each iteration (collision -> new direction -> next leg's free path
// -> distance to the block faces -> collide again or not), and the rare kinds of work -- launches, surface
// reflections, block crossings -- wait until enough lanes have queued up, as launches and reflections already did.
//
// Shared with mcbrat_kernels.hip: Philox slots (same table: the oracle's Philox mode replays these photons), fixed
// point tallies, private LDS tally slab per workgroup unit, launch / surface / scattering arithmetic (bit for bit).
#include <float.h>

#include "mcbrat_device.h"

namespace mcbrat {

// lane states of this kernel
enum : int { BW_DEAD = 0, BW_MOVE = 1, BW_COLLIDE = 2, BW_SURFACE = 3, BW_TOP = 4, BW_CROSS = 5 };

// 0-based layer of height z (no periodicity in z): by division where the reference calls z regular, bisection otherwise
__device__ __forceinline__ int locate_z(const double *ze, int nz, bool regular, double z0, double invDz, double z) {
  if (!regular) return find_cell(ze, nz, z);
  return min(max((int)((z - z0) * invDz), 0), nz - 1);  // the reference's own map on a regular axis (findZIndex :1580-1592)
}
// find_cell's answer -- the largest i with e[i] <= v, clamped to [0, n-1] -- on an axis whose edges are equally spaced to 1e-6 of
// a cell: the division is right to within one cell, the edge table decides (no bisection)
__device__ __forceinline__ int locate_near_uniform(const double *e, int n, double x0, double invCell, double v) {
  int i = min(max((int)((v - x0) * invCell), 0), n - 1);
  const double lo = e[i], hi = e[i + 1];
  i += (v >= hi && i < n - 1) ? 1 : 0;
  i -= (v < lo && i > 0) ? 1 : 0;
  return i;
}

// DEBUG: cell index along a periodic axis counted through the periodic images (faces crossed = |difference|)
__device__ __forceinline__ long long unwrapped_index(const double *e, int n, double x0, double L, double invL, double x) {
  const double k = floor((x - x0) * invL);
  return (long long)k * n + find_cell(e, n, x - k * L);
}

// LDS layout of trace_block_kernel (byte offsets), computed the same way by the host (plan_launch) and the kernel
struct BlockLds {
  size_t slab, cursor, rec, cdf, ext, ssa, cum, pfi, blockOf, tbl, total;
};
// optics: where a collision finds single-scattering albedo, cumulative fractions and phase-function index of its cell (OPT below)
//   0  per cell in LDS, beside the extinction per cell (the small domains: step cloud, plane parallel);
//   1  per cell in global memory (L2: a few tens of KB), LDS holds the extinction per BLOCK -- 2 bytes per cell (its block)
//      instead of 12, for domains whose tally slab takes most of a compute unit's 160 KB (broadband 20 x 20 x 20: 70 KB of
//      tallies + 36 KB of table + 16 KB of block numbers);
//   2  per BLOCK in LDS, where every block is uniform in them too (a homogeneous medium, slabs): no gather at a collision.
// cdfTop: the thermal source's level and row sums of the emission CDF (nz + ny * nz doubles) staged in LDS: ten of the fifteen
// dependent reads of a launch's three bisections stay on chip.
__host__ __device__ inline BlockLds block_lds_layout(int nx, int ny, int nz, int nc, size_t slabLen, int nBlocks, size_t tblFloats,
                                                      int optics = 0, bool cdfTop = false) {
  BlockLds L;
  const size_t nvox = (size_t)nx * ny * nz;
  const size_t nB4 = ((size_t)nBlocks + 3) & ~(size_t)3;
  size_t o = sizeof(double) * (size_t)(nx + ny + nz + 3);
  L.slab = o; o += sizeof(long long) * slabLen;
  L.cursor = o; o += 16;
  o = (o + 15) & ~(size_t)15;
  L.rec = o; o += 16 * (size_t)nBlocks;
  L.cdf = o; o += cdfTop ? sizeof(double) * ((size_t)nz + (size_t)ny * nz) : 0;
  L.ext = o; o += optics == 0 ? 4 * nvox : 4 * nB4;
  L.ssa = o; o += optics == 0 ? 4 * nvox * nc : (optics == 2 ? 4 * nB4 * nc : 0);
  L.cum = o; o += nc > 1 ? (optics == 0 ? 4 * nvox * nc : (optics == 2 ? 4 * nB4 * nc : 0)) : 0;  // (read only when there is more than one component)
  L.pfi = o; o += optics == 0 ? 2 * ((nvox * nc + 1) & ~(size_t)1) : (optics == 2 ? 2 * nB4 * nc : 0);
  L.blockOf = o; o += 2 * ((nvox + 1) & ~(size_t)1);
  o = (o + 3) & ~(size_t)3;
  L.tbl = o; o += 4 * tblFloats;
  L.total = o;
  return L;
}

// EMIT: the thermal-emission source (newPhotonStream_BBEmission); its launch code and parameters are compiled out of the
// solar instantiations (registers and instruction cache for the loop that matters).
// SIMPLE: equally spaced axes (as the reference's own test says, new_Integrator :163-181), one component, the domain's
// Lambertian albedo -- the I3RC step cloud, plane-parallel and homogeneous domains.  What the general kernel decides
// at run time from wave-uniform parameters (bisection or division, how many components, surface description or albedo)
// is decided at compile time here: the branches and the scalar registers they keep alive leave the loop.
// SIMPLE = 3 (round 4): one component and the domain's albedo as 1, on axes that the reference's test calls NOT equally spaced but
// that are equally spaced to 1e-6 of a cell -- its test compares with a spacing held in single precision (new_Integrator :140,
// :163-181), so a grid of 0.1 km cells, config 4's, fails it.  Cells are then what the edge table says (find_cell's answer, bit for
// bit the general instantiation's), found by division + table check instead of bisection, with nothing left to decide at run
// time: the general instantiation spills 116 scalar registers into vector lanes on that domain.
// SIMPLE = 2: as 1, and the domain is one cell wide in y (the I3RC step cloud and the plane-parallel cases are x-z
// problems): with a uniform surface nothing depends on the y position, so the y part of the leg origin, of the face
// distances and of the cell look-ups is compiled out (the direction keeps its y component; the instrumented
// instantiation keeps y, it counts the periodic y faces a leg crosses as the reference does).
// OPT: where a collision finds the optics of its cell (0 per cell in LDS, 1 per cell in global memory, 2 per block in LDS), see
// block_lds_layout.  Workgroups of 1024 lanes are ONE per compute unit (they own most of its 160 KB of LDS): 4 waves per SIMD.
template <int BLOCK, bool TBL_LDS, bool DEBUG, bool EMIT, int SIMPLE, int OPT = 0>
__global__ void __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : (BLOCK > 512 ? BLOCK / 128 : MCBRAT_MIN_WAVES_PER_SIMD))
trace_block_kernel(const DevParams p) {
  constexpr bool NOY = SIMPLE == 2 && !DEBUG;
  constexpr bool NEAR = SIMPLE == 3;  // every axis: not regular by the reference's test, equally spaced to 1e-6 of a cell
  const bool xyRegular = NEAR ? false : (SIMPLE != 0 ? true : p.xyRegular != 0);
  const bool zRegular = NEAR ? false : (SIMPLE != 0 ? true : p.zRegular != 0);
  const int nc = SIMPLE != 0 ? 1 : p.nc;
  extern __shared__ __align__(16) unsigned char smem_raw[];
#ifdef MCBRAT_POISON  // audit build: whatever the kernel reads from LDS before it wrote it reads 0xff..
  for (unsigned i = threadIdx.x; i < p.ldsBytes / 4u; i += BLOCK) reinterpret_cast<unsigned *>(smem_raw)[i] = 0xffffffffu;
  __syncthreads();
#endif
  // LDS map: [edges x|y|z (double)] [private tally slab (i64)] [unit cursor] [block records (uint4)] [extinction]
  //          [ssa] [cum] [phase index (u16)] [block of each cell (u16)] [tables (float), TBL_LDS]
  double *s_edge = reinterpret_cast<double *>(smem_raw);
  const int nEdges = p.nx + p.ny + p.nz + 3;
  const int ncol = p.nx * p.ny;
  const int nvox = ncol * p.nz;
  const int slabLen = (int)p.slabStride;
  const bool cdfTop = EMIT && p.cdfTopLds != 0;  // (wave-uniform)
  const BlockLds lay = block_lds_layout(p.nx, p.ny, p.nz, nc, (size_t)slabLen, p.nBlocks, TBL_LDS ? (size_t)p.tblTotalFloats : 0, OPT, cdfTop);
  const int nB4 = (p.nBlocks + 3) & ~3;
  long long *s_slab = reinterpret_cast<long long *>(smem_raw + lay.slab);
  unsigned *s_cursor = reinterpret_cast<unsigned *>(smem_raw + lay.cursor);
  uint4 *s_blockRec = reinterpret_cast<uint4 *>(smem_raw + lay.rec);
  float *s_ext = reinterpret_cast<float *>(smem_raw + lay.ext);         // [nvox]; OPT != 0: [nBlocks], the blocks' extinction
  double *s_cdfLevel = reinterpret_cast<double *>(smem_raw + lay.cdf);  // [nz] level sums of the emission CDF, then [nz][ny] row sums (cdfTop)
  double *s_cdfRow = s_cdfLevel + p.nz;
  float *s_ssa = reinterpret_cast<float *>(smem_raw + lay.ssa);         // [nc][nvox]
  float *s_cum = reinterpret_cast<float *>(smem_raw + lay.cum);         // [nc][nvox]
  uint16_t *s_pfi = reinterpret_cast<uint16_t *>(smem_raw + lay.pfi);   // [nc][nvox]
  uint16_t *s_blockOf = reinterpret_cast<uint16_t *>(smem_raw + lay.blockOf);  // [nvox]
  float *s_tbl = reinterpret_cast<float *>(smem_raw + lay.tbl);
  __shared__ int s_tblOffset[MCBRAT_MAX_COMPONENTS], s_tblNSteps[MCBRAT_MAX_COMPONENTS];
  __shared__ float s_tblInvN[MCBRAT_MAX_COMPONENTS];
  if (threadIdx.x < MCBRAT_MAX_COMPONENTS) {
    s_tblOffset[threadIdx.x] = p.tblOffset[threadIdx.x];
    s_tblNSteps[threadIdx.x] = p.tblNSteps[threadIdx.x];
    s_tblInvN[threadIdx.x] = p.tblInvN[threadIdx.x];
  }
  for (int i = threadIdx.x; i < nEdges; i += BLOCK) s_edge[i] = p.edges[i];
  if (TBL_LDS)
    for (int i = threadIdx.x; i < p.tblTotalFloats; i += BLOCK) s_tbl[i] = p.tables[i];
  for (int i = threadIdx.x; i < slabLen; i += BLOCK) s_slab[i] = 0;
  if (threadIdx.x == 0) s_cursor[0] = 0;
  for (int i = threadIdx.x; i < p.nBlocks; i += BLOCK) s_blockRec[i] = p.blockRec[i];
  for (int i = threadIdx.x; i < nvox; i += BLOCK) s_blockOf[i] = p.blockOf[i];
  if (OPT == 0) {
    for (int i = threadIdx.x; i < nvox; i += BLOCK) s_ext[i] = p.ext[i];
    for (int i = threadIdx.x; i < nc * nvox; i += BLOCK) { s_ssa[i] = p.ssa[i]; s_pfi[i] = p.pfi[i]; }
    if (nc > 1)
      for (int i = threadIdx.x; i < nc * nvox; i += BLOCK) s_cum[i] = p.cum[i];
  } else {
    for (int i = threadIdx.x; i < p.nBlocks; i += BLOCK) s_ext[i] = p.blockExt[i];
    if (OPT == 2) {  // per-block optics, [component][nB4]
      for (int i = threadIdx.x; i < nc * nB4; i += BLOCK) {
        const int k = i / nB4, b = i - k * nB4;
        if (b < p.nBlocks) { s_ssa[i] = p.blockSsa[k * p.nBlocks + b]; s_pfi[i] = p.blockPfi[k * p.nBlocks + b]; if (nc > 1) s_cum[i] = p.blockCum[k * p.nBlocks + b]; }
      }
    }
  }
  if (cdfTop) {  // newPhotonStream_BBEmission's levelWeights(k) = voxelWeights(nx, ny, k) and colWeights(j, k) = voxelWeights(nx, j, k) (monteCarloIllumination.f95:56-57)
    const long long nxy = (long long)p.nx * p.ny;
    for (int i = threadIdx.x; i < p.nz; i += BLOCK) s_cdfLevel[i] = p.voxelCDF[((long long)p.nx - 1) + (long long)p.nx * (p.ny - 1) + nxy * i];
    for (int i = threadIdx.x; i < p.ny * p.nz; i += BLOCK) { const int k = i / p.ny, j = i - k * p.ny; s_cdfRow[i] = p.voxelCDF[((long long)p.nx - 1) + (long long)p.nx * j + nxy * k]; }
  }
  __syncthreads();
  // optics of a cell (collisions): component k of cell `cell`
  auto cumAt = [&](int k, int cell) -> float { if constexpr (OPT == 0) return s_cum[k * nvox + cell]; else if constexpr (OPT == 1) return p.cum[k * nvox + cell]; else return s_cum[k * nB4 + s_blockOf[cell]]; };
  auto ssaAt = [&](int k, int cell) -> float { if constexpr (OPT == 0) return s_ssa[k * nvox + cell]; else if constexpr (OPT == 1) return p.ssa[k * nvox + cell]; else return s_ssa[k * nB4 + s_blockOf[cell]]; };
  auto pfiAt = [&](int k, int cell) -> int { if constexpr (OPT == 0) return (int)s_pfi[k * nvox + cell]; else if constexpr (OPT == 1) return (int)p.pfi[k * nvox + cell]; else return (int)s_pfi[k * nB4 + s_blockOf[cell]]; };
  auto extOfCell = [&](int cell) -> float { if constexpr (OPT == 0) return s_ext[cell]; else return s_ext[s_blockOf[cell]]; };
  const float *__restrict__ tbl = TBL_LDS ? s_tbl : p.tables;
  const int offY = p.nx + 1, offZ = p.nx + p.ny + 2;  // edge table offsets
  const double invDz = (double)p.nz / (p.zMax - p.z0);

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long laneBelow = (1ull << lane) - 1ull;

  // this workgroup's unit = photons [unitFirst, unitFirst + unitCount) of ONE batch
  unsigned long long unitFirst = 0;
  unsigned unitCount = 0;
  long long *unitSlab = nullptr;

  // lane state --------------------------------------------------------------------------
  int state = BW_DEAD;
  bool more = true;
  uint32_t idLo = 0, idHi = 0, event = 0;
  double px = 0, py = 0, pz = 0;      // leg origin (in the periodic image the leg is currently in)
  float dx = 0, dy = 0, dz = 1;       // direction cosines
  float ivx = 0, ivy = 0, ivz = 0;    // 1/direction
  float tnx = 0, tny = 0, tnz = 0, tcur = 0;  // distance along the leg to the x/y/z face of the current BLOCK
  float acc = 0, tau = 0, w = 0, extCur = 0, uX = 0, uY = 0, uZ = 0;
  // cell range of the current block per axis, lo | hi << 16 (the face ahead of the lane is hi or lo by its direction).
  // Kept whole, not just the face ahead: the cell a lane enters on the far side of a face comes from its POSITION on the
  // two other axes, and a position that lies on a second face of the block (a leg through an edge of the block: float
  // ties do occur, equal spacing and a diagonal sun make them common) may round to either side.  Clamped to the range of
  // the block the lane is leaving it cannot fall behind a face the leg has already passed -- unclamped, two position
  // look-ups that both round backwards hand the lane to and fro between two blocks for ever (found by the soak run of
  // test_random_box_media_against_face_by_face_kernel).
  unsigned rx = 0, ry = 0, rz = 0;
  unsigned spans = 0;                  // bit 0 / 1: the current block spans the whole periodic x / y axis
  int nScat = 0, nLegs = 0;
  long long dbgX = 0, dbgY = 0;        // DEBUG: cell of the leg's start counted through the periodic images
  int dbgZ = 0;
  unsigned int cLegs = 0, cCross = 0, cColl = 0, cAbs = 0, cTop = 0, cSurf = 0, cKill = 0, cSurv = 0;
  unsigned long long wIters = 0, wCrossPhases = 0, wCrossLanes = 0, wEventLanes = 0, wLaunchPhases = 0, wSurfPhases = 0;

  // faces crossed by the leg that ends at (x, y) / layer index kz (DEBUG: the reference's crossing count)
  auto countCrossings = [&](double x, double y, int kzEnd) {
    const long long ex = unwrapped_index(s_edge, p.nx, p.x0, p.Lx, p.invLx, x);
    const long long ey = unwrapped_index(s_edge + offY, p.ny, p.y0, p.Ly, p.invLy, y);
    cCross += (unsigned)(llabs(ex - dbgX) + llabs(ey - dbgY) + abs(kzEnd - dbgZ));
  };

  // Cell of position xw / yw on a periodic axis; folds the leg origin into the image that position lies in (DEBUG:
  // the leg's start index follows the origin through the images, so that index differences stay face counts).
  // `canLeave`: the position may lie outside the principal image (the lane comes out of a block that spans the whole
  // axis); otherwise it is inside the domain and no fold is needed.  On axes the reference itself calls equally
  // spaced the cell is its own formula, int((x - x0) / delta) (findXYIndicies :1558-1562); else the edge table decides.
  auto locX = [&](double xw, bool canLeave, bool force = false) {
    if (p.nx == 1 && !force) return 0;  // (one column: nothing to find; the origin is folded where the position itself is used)
    if (canLeave) {
      const double shift = floor((xw - p.x0) * p.invLx) * p.Lx;
      px -= shift; xw -= shift;
      if (DEBUG) dbgX -= (long long)rint(shift * p.invLx) * p.nx;
    }
    if (xyRegular) return min(max((int)((xw - p.x0) * p.invDX), 0), p.nx - 1);
    double o = 0.0;
    return locate_periodic(s_edge, p.nx, p.x0, p.Lx, p.invLx, p.invCellX, NEAR || p.xyNearUniform != 0, o, xw);
  };
  auto locY = [&](double yw, bool canLeave, bool force = false) {
    if (NOY) return 0;
    if (p.ny == 1 && !force) return 0;
    if (canLeave) {
      const double shift = floor((yw - p.y0) * p.invLy) * p.Ly;
      py -= shift; yw -= shift;
      if (DEBUG) dbgY -= (long long)rint(shift * p.invLy) * p.ny;
    }
    if (xyRegular) return min(max((int)((yw - p.y0) * p.invDY), 0), p.ny - 1);
    double o = 0.0;
    return locate_periodic(s_edge + offY, p.ny, p.y0, p.Ly, p.invLy, p.invCellY, NEAR || p.xyNearUniform != 0, o, yw);
  };
  auto inRange = [](int j, unsigned r) { return min(max(j, (int)(r & 0xffffu)), (int)(r >> 16) - 1); };
  // TEST ONLY (DevParams::legacyTies): the tie handling from before the three fixes the soak runs led to, to show that the
  // bounds below end the kernel without them (tests/test_gpu_edge_cases.py).  Wave-uniform, read in the rare phases only.
  // (compiled into the instrumented instantiation only -- the one mcbrat_trace_fates runs: the production kernels carry none of it)
  const bool legacyNoClamp = (DEBUG || kTestBounds) && (p.legacyTies & 1) != 0, legacyMoveNaN = (DEBUG || kTestBounds) && (p.legacyTies & 2) != 0,
             legacyKeepSpans = (DEBUG || kTestBounds) && (p.legacyTies & 4) != 0;
  auto inRangeX = [&](int j, unsigned r) { return legacyNoClamp ? j : inRange(j, r); };  // clamp of a block crossing
  // Distances along the leg to the faces of the block that holds cell (ix, iy, iz); its extinction.
  auto enterBlock = [&](int ix, int iy, int iz) {
    const int cell = ix + p.nx * (iy + p.ny * iz);
    const unsigned blk = s_blockOf[cell];
    const uint4 rec = s_blockRec[blk];
    if constexpr (OPT == 0) extCur = s_ext[cell]; else extCur = s_ext[blk];
    rx = rec.x; rz = rec.z;
    if (!NOY) ry = rec.y;
    const int fx = dx >= 0.0f ? (int)(rx >> 16) : (int)(rx & 0xffffu);
    const int fy = NOY ? 0 : (dy >= 0.0f ? (int)(ry >> 16) : (int)(ry & 0xffffu));
    const int fz = dz >= 0.0f ? (int)(rz >> 16) : (int)(rz & 0xffffu);
    // a block that spans a whole periodic axis has no face on it (the lane's position runs through the images)
    spans |= rec.w;  // (in such a block the lane may leave the principal image: folded when it leaves the block)
    tnx = ((rec.w & 1u) || ivx == 0.0f) ? FLT_MAX : (float)(s_edge[fx] - px) * ivx;
    tny = (NOY || (rec.w & 2u) || ivy == 0.0f) ? FLT_MAX : (float)(s_edge[offY + fy] - py) * ivy;
    tnz = ivz == 0.0f ? FLT_MAX : (float)(s_edge[offZ + fz] - pz) * ivz;
  };

  // Why the loop ends.  In one iteration a lane in BW_MOVE always leaves that state (collision, block face, exit or
  // drop) and a lane in BW_COLLIDE dies or starts a leg; lanes waiting for a crossing or an exit are served as soon as
  // fewer than eventThreshold lanes collide.  So every iteration starts a leg, serves a crossing or ends a photon, and
  // two counters bound the lot: legs per photon (maxEvents) and block crossings per leg (the `watchdog` figure, here a
  // per-lane count -- the kernel has VGPRs to spare and no SGPR).  Neither costs the iteration anything: the first is
  // one compare where a leg starts, the second lives in the crossing's own branch.
  unsigned nCrossLeg = 0;
  unsigned nBadLane = 0;  // photons this lane dropped at a loop bound; added to *p.bad when the kernel ends
  const unsigned maxEvents = (DEBUG || kTestBounds) ? p.maxEvents : kMaxEvents, maxEventsNaN = (DEBUG || kTestBounds) ? p.maxEventsNaN : kMaxEventsNaN,
                 watchdog = (DEBUG || kTestBounds) ? p.watchdog : kWatchdog;
#define MCBRAT_BW_DROP(kind_) do { \
    if constexpr (DEBUG || kTestBounds) record_first_drop(p.bad, (kind_), 2u, idLo, idHi, state, event);  /* (which bound, which photon) */ \
    if (DEBUG && p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{3, 0, 0, 0, nScat, nLegs, w}; \
    state = BW_DEAD; \
    nBadLane = count_drop(nBadLane, (kind_));  /* (added to *p.bad when the kernel ends: no atomic, no pointer in the loop) */ \
  } while (0)
#ifdef MCBRAT_STAMPS  // development aid (-DMCBRAT_STAMPS): wave cycles per section of the loop, reported with the event counters
  __shared__ unsigned long long s_tprev[BLOCK / 64], s_stamp[BLOCK / 64][9];
  if (lane == 0) { s_tprev[threadIdx.x >> 6] = clock64(); for (int i = 0; i < 9; i++) s_stamp[threadIdx.x >> 6][i] = 0; }
#endif
  for (unsigned long long unit = blockIdx.x;; unit += gridDim.x) {
    if (unit >= p.nUnits) break;  // workgroup-uniform
    uint32_t batch;
    {
      const unsigned long long b = unit / p.unitsPerBatch, s = unit % p.unitsPerBatch;
      const unsigned long long bp = (p.total - b * p.ppb) < p.ppb ? (p.total - b * p.ppb) : p.ppb;
      const unsigned long long lo = (bp * s) / p.unitsPerBatch, hi = (bp * (s + 1)) / p.unitsPerBatch;
      unitFirst = b * p.ppb + lo;
      unitCount = (unsigned)(hi - lo);
      unitSlab = p.slabs + b * p.slabStride;
      batch = (uint32_t)b;
      more = true;
    }
    (void)batch;

    for (;;) {
      bool needLeg = false;
      int ix = 0, iy = 0, iz = 0;  // cell of a lane that starts a leg in this iteration

      // Rare kinds of work -- exits, launches, block crossings -- wait until enough lanes ask for them, or until the
      // wave has little else to do.  Exits come first and launches right after them: under a black surface every exit
      // frees its lane, so the lanes that have just left are refilled in the same iteration instead of queueing again.
      const unsigned long long mSurf = __ballot(state == BW_SURFACE || state == BW_TOP);  // exits, served together
      const unsigned long long mCross = __ballot(state == BW_CROSS);
      const int nBusy = __popcll(__ballot(state == BW_COLLIDE));
      const bool idle = nBusy < p.eventThreshold;
      const bool doSurface = __popcll(mSurf) >= p.surfaceThreshold || idle;
      const bool doCross = __popcll(mCross) >= p.crossThreshold || idle;
      // ---- exits: out the top (computeRT :573-617) or down to the surface (:619-676, Lambertian) ----
      // Both need the column the leg left through, from the position; fluxDown gets the incident weight (:634).
      if ((state == BW_SURFACE || state == BW_TOP) && doSurface) {
        const bool top = state == BW_TOP;
        {
          const double xw = px + (double)tcur * (double)dx, yw = py + (double)tcur * (double)dy;  // where the leg met the face (:1801-1812)
          if (DEBUG) countCrossings(xw, yw, top ? p.nz : -1);
          px = xw;
          if (!NOY) py = yw;
          // (always folded here: the surface description takes the position itself.  A position on the domain boundary
          // folds to either side of it by rounding: where the clamp then keeps a cell at the other end of the axis, the
          // origin takes the image next to that cell -- cell and position must agree when the reflected leg starts.)
          const int jx = locX(xw, true, true);
          ix = inRange(jx, rx);
          if (ix != jx) {
            const double lo = s_edge[rx & 0xffffu], hi = s_edge[rx >> 16];
            if (px < lo && lo - px > px + p.Lx - hi) px += p.Lx;
            else if (px > hi && px - hi > lo - (px - p.Lx)) px -= p.Lx;
          }
          iy = 0;
          if (!NOY) {
            const int jy = locY(yw, true, true);
            iy = inRange(jy, ry);
            if (iy != jy) {
              const double lo = s_edge[offY + (ry & 0xffffu)], hi = s_edge[offY + (ry >> 16)];
              if (py < lo && lo - py > py + p.Ly - hi) py += p.Ly;
              else if (py > hi && py - hi > lo - (py - p.Ly)) py -= p.Ly;
            }
          }
        }
        atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + (top ? 0 : ncol) + (ix + p.nx * iy)), weight_to_fixed(w));
        if (top) {
          if (DEBUG) {
            cTop++;
            if (p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{0, ix + 1, iy + 1, p.nz + 1, nScat, nLegs, w};
          }
          state = BW_DEAD;
        } else {
          pz = p.zSurf;
          iz = 0;
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
          const float wIn = w;
          if (SIMPLE == 0 && p.surfNumX > 0) w = w * surface_reflectance(p, px, py);  // useSurfaceBDRF :667-670
          else w = (float)((double)w * (double)p.albedo);              // :673
          if (w <= FLT_MIN) {
            if (DEBUG && p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{1, ix + 1, iy + 1, 1, nScat, nLegs, wIn};
            state = BW_DEAD;
          } else {
            const float sinTheta = sqrtf(1.0f - mu * mu);  // makeDirectionCosines(mu, 2 pi Y) :1876-1894
            float cphi, sphi;
            sincos_2pi(uY, cphi, sphi);
            dx = sinTheta * cphi; dy = sinTheta * sphi; dz = mu;
            needLeg = true;
          }
        }
      }
      STAMP(5);
      const unsigned long long mDead = __ballot(state == BW_DEAD && more);
      const bool doLaunch = __popcll(mDead) >= p.launchThreshold || idle;
      const unsigned long long want = doLaunch ? mDead : 0ull;
      if (DEBUG) {
        wIters++;
        wEventLanes += nBusy;
        if (want) wLaunchPhases++;
        if (doSurface && mSurf) wSurfPhases++;
        if (doCross && mCross) { wCrossPhases++; wCrossLanes += __popcll(mCross); }
      }
      if (want != 0ull) {  // wave-uniform
        const int nWant = __popcll(want);
        const int rank = __popcll(want & laneBelow);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&s_cursor[0], (unsigned)nWant);
        base = (unsigned)__shfl((int)base, 0);
        const unsigned k = base + (unsigned)rank;
        const bool valid = k < unitCount;
        const unsigned long long myIdx = unitFirst + k;
        if (state == BW_DEAD && more) {
          if (valid) {
            // ---- launch: getNextPhoton + computeRT :466-508 (as in trace_kernel, bit for bit) ----
            const unsigned long long id = p.firstPhoton + myIdx;
            idLo = (uint32_t)id; idHi = (uint32_t)(id >> 32);
            event = 0; nScat = 0; nLegs = 0;
            uint32_t r[4];
            philox4x32_10(0u, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
            double lx, ly, lz;  // fractional launch position in [0,1]
            if (!EMIT) {  // newPhotonStream_Directional, monteCarloIllumination.f95:88-96
              lx = (double)u01(r[0]);
              ly = (double)u01(r[1]);
              lz = 0.0;
              dx = p.dir0[0]; dy = p.dir0[1]; dz = p.dir0[2];
            } else {  // newPhotonStream_BBEmission :481-516
              float mu = 0.f, phi = 0.f;
              const float sel = u01(r[0]);
              if ((double)sel > p.fracAtms) {  // surface emission :484-493
                lx = (double)u01(r[1]);
                ly = (double)u01(r[2]);
                lz = 0.0;
                uint32_t r1[4];
                for (uint32_t j = 0;; j++) {
                  if ((j & 3u) == 0) philox4x32_10(0u, 1u + (j >> 2), idLo, idHi, p.seedLo, p.seedHi, r1);
                  mu = sqrtf(u01(pick4(r1, j & 3u)));
                  if (fabsf(mu) > 2.0f * FLT_MIN) break;
                }
                phi = (u01(r[3]) * 2.0f) * 3.14159274f;
              } else {  // atmosphere :495-510
                const float rn = u01(r[1]);
                const long long nxy = (long long)p.nx * p.ny;
                // (level and row from the sums staged in LDS where they are: the same doubles, the same comparisons)
                const int ik = cdfTop ? find_cdf(s_cdfLevel, p.nz, 1, rn) : find_cdf(p.voxelCDF + ((long long)p.nx - 1) + (long long)p.nx * (p.ny - 1), p.nz, nxy, rn);
                const int ij = cdfTop ? find_cdf(s_cdfRow + p.ny * (ik - 1), p.ny, 1, rn) : find_cdf(p.voxelCDF + ((long long)p.nx - 1) + nxy * (ik - 1), p.ny, p.nx, rn);
                const int ii = find_cdf(p.voxelCDF + (long long)p.nx * ((ij - 1) + (long long)p.ny * (ik - 1)), p.nx, 1, rn);
                uint32_t r1[4];
                philox4x32_10(0u, 1u, idLo, idHi, p.seedLo, p.seedHi, r1);
                lz = ((double)(ik - 1) * 1.0 / (double)p.nz) + (double)(u01(r[2]) / (float)p.nz);
                if (ik == 1 && lz == 0.0) lz = 2.220446049250313e-16;
                if (ik == p.nz && lz > 1.0 - 2.0 * 2.220446049250313e-16) lz = lz - 2.0 * 2.220446049250313e-16;
                lx = ((double)(ii - 1) * 1.0 / (double)p.nx) + (double)(u01(r[3]) * (1.0f / (float)p.nx));
                ly = ((double)(ij - 1) * 1.0 / (double)p.ny) + (double)(u01(r1[0]) * (1.0f / (float)p.ny));
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
            px = p.x0 + lx * (p.xMax - p.x0);  // :480-482
            if (!NOY) py = p.y0 + ly * (p.yMax - p.y0);
            if (xyRegular) {  // findXYIndicies :1558-1562
              ix = min((int)((px - p.x0) * p.invDX), p.nx - 1);
              iy = NOY ? 0 : min((int)((py - p.y0) * p.invDY), p.ny - 1);
            } else {
              ix = NEAR ? locate_near_uniform(s_edge, p.nx, p.x0, p.invCellX, px) : find_cell(s_edge, p.nx, px);
              iy = NOY ? 0 : (NEAR ? locate_near_uniform(s_edge + offY, p.ny, p.y0, p.invCellY, py) : find_cell(s_edge + offY, p.ny, py));
            }
            if (!EMIT) {
              pz = p.zLaunch; iz = p.izLaunch;
            } else if (zRegular) {  // :485-486
              pz = p.z0 + lz * (p.zMax - p.z0);
              iz = min((int)((pz - p.z0) / ((p.zMax - p.z0) / (double)p.nz)), p.nz - 1);
            } else {  // :491-493 layer-index fraction
              const double t = (lz - p.z0) * (double)p.nz;
              const double fl = floor(t);
              iz = min((int)fl, p.nz - 1);
              pz = s_edge[offZ + iz] + (t - fl) * (s_edge[offZ + iz + 1] - s_edge[offZ + iz]);
            }
            if (EMIT && p.lwFlag && pz > 0.0)  // :504-508 emission counts as negative absorption
              atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + 2 * ncol + (ix + p.nx * (iy + p.ny * iz))), to_fixed(-1.0));
            needLeg = true;
          } else {
            more = false;
          }
        }
      }
      STAMP(0);
      // ---- scattering event, computeRT :703-821 ----
      if (state == BW_COLLIDE) {
        // opticalProperties.f95:1729-1738: the point inside the block where tau is used up.  (Not for a leg whose direction
        // is NaN: the reference's inverse phase-function tables can hold a NaN -- a negative discriminant in the closed-form
        // inversion, inversePhaseFunctions.f95:148-166; one entry in 9001 for a 64-term HG series with g = 0.5 -- and a photon
        // that draws it keeps a NaN direction from then on.  In the reference, the oracle and the face-by-face kernel such a
        // photon goes on colliding in the CELL it is in until its weight is gone; here the cell comes from the position, so
        // the position must stay where it is -- moved along a NaN it would never die: found by the soak run against the oracle.)
        if (dz == dz || legacyMoveNaN) {
          const double s = (double)(tcur + div_fast(tau - acc, extCur));
          px = px + s * (double)dx;
          if (!NOY) py = py + s * (double)dy;
          pz = pz + s * (double)dz;
        }
        // its cell, from the position (the periodic fold moves the position into the domain)
        {
          const double xw = px, yw = py;
          iz = NEAR ? locate_near_uniform(s_edge + offZ, p.nz, p.z0, invDz, pz) : locate_z(s_edge + offZ, p.nz, zRegular, p.z0, invDz, pz);
          ix = locX(xw, (spans & 1u) != 0);
          iy = locY(yw, (spans & 2u) != 0);
        }
        int cell = ix + p.nx * (iy + p.ny * iz);
        nScat++;
        if (DEBUG) cColl++;
        int c = 0;  // component pick :759-760 (findIndex over [0, cumExt(:)]), uniform = slot Z of the leg's block
        if (nc > 1) {
          for (int k = 0; k < nc - 1; k++)
            if (uZ >= cumAt(k, cell)) c = k + 1;
        }
        float ssa = ssaAt(c, cell);
        int pfEntry = pfiAt(c, cell);
        // A collision point within an ulp of a face of its block can be located next door -- harmless (the deposit and the
        // phase function of a cell an ulp away) unless next door is vacuum, where nothing collides: booked there the photon
        // would lose its whole weight (the host stores a single-scattering albedo of 0 where there is no extinction; the
        // soak run of the random box media caught one such photon in 8 million).  Then, and only then (clamping every
        // collision cost the step cloud 3 %), the cell is clamped to the block the lane is in.
        if (ssa <= 0.0f && extOfCell(cell) != extCur) {
          ix = inRange(ix, rx); iz = inRange(iz, rz);
          if (!NOY) iy = inRange(iy, ry);
          cell = ix + p.nx * (iy + p.ny * iz);
          c = 0;
          if (nc > 1) {
            for (int k = 0; k < nc - 1; k++)
              if (uZ >= cumAt(k, cell)) c = k + 1;
          }
          ssa = ssaAt(c, cell);
          pfEntry = pfiAt(c, cell);
        }
        if (DEBUG) countCrossings(px, py, iz);  // (after the fold of the look-ups above: px, py are the collision point)
        if (ssa < 1.0f) {  // absorption :765-771
          atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + 2 * ncol + cell), weight_to_fixed(w * (1.0f - ssa)));
          w = w * ssa;
          if (DEBUG) cAbs++;
        }
        if (p.useRR && w < 0.5f) {  // Russian roulette :805-811, RussianRouletteW = 1
          float uR = uZ;  // one component: Z decides nothing at the component pick and serves here (no second block; mcbrat_kernels.hip)
          if (nc != 1) {  // (wave-uniform; compile time in the SIMPLE instantiations)
            uint32_t r1[4];
            philox4x32_10(event, 1u, idLo, idHi, p.seedLo, p.seedHi, r1);
            uR = u01(r1[1]);
          }
          if (uR >= w) { w = 0.0f; if (DEBUG) cKill++; }
          else { w = 1.0f; if (DEBUG) cSurv++; }
        }
        STAMP(2);
        if (w <= FLT_MIN) {  // :812
          if (DEBUG && p.fates) p.fates[(((unsigned long long)idHi << 32) | idLo) - p.firstPhoton] = mcbrat_fate{2, ix + 1, iy + 1, iz + 1, nScat, nLegs, 0.0f};
          state = BW_DEAD;
        } else {
          // computeScatteringAngle :1594-1621 (table point count N, floor-type lookup as written)
          const int n = s_tblNSteps[c];
          const float *t = tbl + s_tblOffset[c] + (long long)pfEntry * n;
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
          // next_direct :1921-1948 with (AX, AY)/sqrt(D) = (cos, sin) of a uniform azimuth
          float AX, AY;
          sincos_2pi(uY, AX, AY);
#ifdef MCBRAT_PRECISE_MATH
          float B = sqrtf(1.0f - cs * cs);
#else
          float B = __builtin_amdgcn_sqrtf(1.0f - cs * cs);
#endif
          AX = AX * B;
          AY = AY * B;
          B = dx * AX - dy * AY;
          float D = cs - div_fast(B, 1.0f + fabsf(dz));
          dx = dx * D + AX;
          dy = dy * D - AY;
          dz = dz * cs - copysignf(fabsf(B), dz * B);
          needLeg = true;
          STAMP(4);
        }
      }
      STAMP(4);
      // ---- start the next leg: tau, 1/direction, the block the leg starts in ----
      // (a photon is allowed maxEvents legs; one with a NaN direction -- see the collision above -- maxEventsNaN: where
      // omega0 = 1 its weight never falls and neither the reference nor the roulette would ever end it)
      if (needLeg && event >= maxEventsNaN) {  // (one compare on the hot path: maxEventsNaN <= maxEvents; the rule itself in the rare branch)
        if (event >= maxEvents || !(dz == dz || w < 1.0f)) { needLeg = false; MCBRAT_BW_DROP(event >= maxEvents ? DROP_LEGS : DROP_NAN_LEGS); }
      }
      if (needLeg) {
        event++;
        nLegs++;
        if (DEBUG) {
          cLegs++;
          dbgX = unwrapped_index(s_edge, p.nx, p.x0, p.Lx, p.invLx, px);
          dbgY = unwrapped_index(s_edge + offY, p.ny, p.y0, p.Ly, p.invLy, py);
          dbgZ = iz;
        }
        uint32_t r[4];
        philox4x32_10(event, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
#ifdef MCBRAT_PRECISE_MATH
        tau = -logf(fmaxf(FLT_MIN, u01(r[0])));
#else
        tau = -0.693147182f * __builtin_amdgcn_logf(fmaxf(FLT_MIN, u01(r[0])));  // :554
#endif
        uX = u01(r[1]); uY = u01(r[2]); uZ = u01(r[3]);
        acc = 0.0f; tcur = 0.0f;
        // opticalProperties.f95:1705-1712: huge step for a zero cosine
        ivx = fabsf(dx) >= 2.0f * FLT_MIN ? rcp_fast(dx) : 0.0f;
        ivy = (!NOY && fabsf(dy) >= 2.0f * FLT_MIN) ? rcp_fast(dy) : 0.0f;
        ivz = fabsf(dz) >= 2.0f * FLT_MIN ? rcp_fast(dz) : 0.0f;
        spans = 0;  // (the leg starts inside the domain: its origin was folded where it was located)
        nCrossLeg = 0u;
        enterBlock(ix, iy, iz);
        state = BW_MOVE;
      }
      STAMP(6);
      // ---- block crossings: the cell on the other side of the face, from the position; its block ----
      if (state == BW_CROSS && doCross) {
        const bool yLtX = !NOY && tny < tnx;
        const float m2 = yLtX ? tny : tnx;
        const bool isZ = tnz < m2;  // (the axis whose face was reached: the same comparison the move made)
        const double xw = px + (double)tcur * (double)dx, yw = py + (double)tcur * (double)dy, zw = pz + (double)tcur * (double)dz;
        // (the face crossed gives its axis' index exactly; the two others come from the position, clamped to the
        // range of the block the lane is leaving -- see rx, ry, rz above)
        int jx, jy, jz;
        if (isZ) {
          jz = dz >= 0.0f ? (int)(rz >> 16) : (int)(rz & 0xffffu) - 1;  // (0 <= jz < nz: leaving the domain was decided when the face was reached)
          jx = inRangeX(locX(xw, (spans & 1u) != 0), rx);
          jy = NOY ? 0 : inRangeX(locY(yw, (spans & 2u) != 0), ry);
        } else if (yLtX) {
          jy = dy >= 0.0f ? (int)(ry >> 16) : (int)(ry & 0xffffu) - 1;
          if (jy >= p.ny) { jy = 0; py -= p.Ly; if (DEBUG) dbgY -= p.ny; }            // periodic y :1790-1796: continue in the next image
          else if (jy < 0) { jy = p.ny - 1; py += p.Ly; if (DEBUG) dbgY += p.ny; }
          jx = inRangeX(locX(xw, (spans & 1u) != 0), rx);
          jz = inRangeX(NEAR ? locate_near_uniform(s_edge + offZ, p.nz, p.z0, invDz, zw) : locate_z(s_edge + offZ, p.nz, zRegular, p.z0, invDz, zw), rz);
        } else {
          jx = dx >= 0.0f ? (int)(rx >> 16) : (int)(rx & 0xffffu) - 1;
          if (jx >= p.nx) { jx = 0; px -= p.Lx; if (DEBUG) dbgX -= p.nx; }            // periodic x :1782-1788
          else if (jx < 0) { jx = p.nx - 1; px += p.Lx; if (DEBUG) dbgX += p.nx; }
          jy = NOY ? 0 : inRangeX(locY(yw, (spans & 2u) != 0), ry);
          jz = inRangeX(NEAR ? locate_near_uniform(s_edge + offZ, p.nz, p.z0, invDz, zw) : locate_z(s_edge + offZ, p.nz, zRegular, p.z0, invDz, zw), rz);
        }
        // (every axis the block left spans has just been folded -- it cannot be the axis crossed, a spanning block has no
        // face there.  The bits must not outlive the fold: a lane that has just WRAPPED stands on the domain boundary, and
        // a second fold by floor() may take the image on the far side of it while the clamp keeps the cell on this side;
        // the next face then lies behind the lane, tcur steps back, and the lane goes round a corner of four blocks for
        // ever -- the third hang the soak runs found, test_random_domains_against_the_oracle seed 763, a grazing sun.)
        if (!legacyKeepSpans) spans = 0;
        enterBlock(jx, jy, jz);
        state = BW_MOVE;
        if (++nCrossLeg > watchdog) MCBRAT_BW_DROP(DROP_CROSSINGS);  // (a leg that has crossed 2^20 blocks is going round in circles)
      }
      STAMP(1);
      // wave-uniform exit: nothing alive and every lane has already been refused a new photon
      if (__ballot(state != BW_DEAD || more) == 0ull) break;

#ifdef MCBRAT_STUCK_PROBE  // development aid: a wave still looping after 3e6 iterations records 64 iterations of its first live lane, then drops it
      if (DEBUG && p.traceBuf && wIters > 3000000ull) {
        const unsigned long long live = __ballot(state != BW_DEAD);
        if (live != 0ull && lane == __ffsll((long long)live) - 1) {
          const unsigned long long r = wIters - 3000001ull;
          if (r < 64ull) {
            double *t = p.traceBuf + 20 * r;
            t[0] = 777; t[1] = state; t[2] = tcur; t[3] = tnx; t[4] = tny; t[5] = tnz; t[6] = (double)rx; t[7] = (double)ry; t[8] = (double)rz;
            t[9] = px; t[10] = py; t[11] = pz; t[12] = dx; t[13] = dy; t[14] = dz; t[15] = acc; t[16] = tau; t[17] = extCur; t[18] = spans; t[19] = idLo;
          } else {
            state = BW_DEAD;
          }
        }
      }
#endif
      // ---- move: to the collision point inside this block, or to the block face ahead (:1718-1744) ----
      if (state == BW_MOVE) {
        const bool yLtX = !NOY && tny < tnx;
        const float m2 = yLtX ? tny : tnx;
        const bool isZ = tnz < m2;
        const float tmin = isZ ? tnz : m2;
        const float accNew = acc + (tmin - tcur) * extCur;  // :1743
        if (accNew > tau) {
          state = BW_COLLIDE;  // :1729-1738: the stop point is resolved at the head of the next iteration
        } else if (!(tmin < FLT_MAX)) {
          MCBRAT_BW_DROP(DROP_VACUUM);    // no face ahead and nothing to collide with (a horizontal leg through vacuum): the reference walks for ever
        } else {
          acc = accNew;
          tcur = tmin;
          state = BW_CROSS;
          if (isZ && (dz >= 0.0f ? (int)(rz >> 16) == p.nz : (rz & 0xffffu) == 0u)) state = dz >= 0.0f ? BW_TOP : BW_SURFACE;  // :1801-1812
        }
      }
      STAMP(7);
    }

    // flush this unit's private tallies into the batch slab, once
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

  publish_drops(p.bad, nBadLane);
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
      // slots shared with trace_kernel's report: "walk" = block-crossing phases, "event" = loop iterations
      atomicAdd(p.counters + 8, wCrossPhases); atomicAdd(p.counters + 9, wCrossLanes);
      atomicAdd(p.counters + 10, wIters); atomicAdd(p.counters + 11, wEventLanes);
      atomicAdd(p.counters + 12, wLaunchPhases); atomicAdd(p.counters + 13, wSurfPhases);
    }
  }
}

}  // namespace mcbrat
